import base64
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


def fh(x):
    """float.fromhex with None passthrough."""
    return None if x is None else float.fromhex(x)


def golden_bits(rec):
    return np.frombuffer(base64.b64decode(rec["bits_u64_b64"]), dtype=np.uint64).reshape(rec["n"], -1).copy()


def golden_counts(rec):
    return np.frombuffer(base64.b64decode(rec["I_b64"]), dtype=np.int64).reshape(rec["n"], rec["n"]).copy()


def rel_close(a, b, rel=1e-9, abs_=0.0):
    # the reference's None (pi_per_site without -l) is NaN on the C side
    a = float("nan") if a is None else a
    b = float("nan") if b is None else b
    if np.isnan(a) or np.isnan(b):
        return bool(np.isnan(a) and np.isnan(b))
    return abs(a - b) <= max(rel * max(abs(a), abs(b)), abs_)


def stat_close(key, got, want, dxy, rel=1e-9):
    """The tolerance policy of INTEGRATION.md §4: pi / pi_a / pi_b / pi_xy / Dxy / Tajima's D to 1e-9 relative.  Fst and Da are
    DIFFERENCES (Da = Dxy - pi_xy, Fst = Da / Dxy): where Dxy and pi_xy cancel, what is left is the rounding error of the
    sums, which in the reference itself moves with PYTHONHASHSEED (h-fst.py:141-171 sums over Python sets) — so they are
    compared to 1e-9 relative OR an absolute floor of 1e-12 (Fst) / 1e-12 * Dxy (Da), whichever is larger."""
    floor = 1e-12 if key == "fst" else 1e-12 * abs(dxy) if key == "da" else 0.0
    return rel_close(got, want, rel, floor)


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as o
    o.build()
    return o
