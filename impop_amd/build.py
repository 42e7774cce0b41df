"""Build libimpop_hip.so (hand-written HIP for gfx950) in-tree with hipcc.

    python -m impop_amd.build [--force]

The .so is git-ignored but travels to the GPU box with the gpurun snapshot.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
SO = os.path.join(HERE, "libimpop_hip.so")
SOURCES = ["context.hip", "layout.hip", "scan.hip", "stats.hip", "pairwise.hip", "simparse.hip", "ehh.hip"]
FLAGS = [
    "-O3", "-std=c++17", "-fPIC", "-shared", "--offload-arch=gfx950",
    "-ffp-contract=off",          # fp64 epilogues follow the reference's operation order exactly
    "-fvisibility=hidden", "-Wall", "-Wno-unused-function", "-Wno-unused-value", "-Wno-unused-result",
]


def _stale() -> bool:
    if not os.path.exists(SO):
        return True
    t = os.path.getmtime(SO)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "..", "include", "impop_hip.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not _stale():
        return SO
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: cannot build libimpop_hip.so")
    cmd = [hipcc] + FLAGS + ["-o", SO + ".tmp"] + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + r.stdout + r.stderr)
    os.replace(SO + ".tmp", SO)
    return SO


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
