"""Make `impop_amd` importable when a drop-in script is run by path, as the reference's bash
drivers do (`python3 "$SCRIPT_PATH" ...`, run_pica2_impg.sh:175)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
