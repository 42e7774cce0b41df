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
SOURCES = ["context.hip", "layout.hip", "scan.hip", "stats.hip", "stats_small.hip", "pairwise.hip", "simparse.hip", "ehh.hip", "multigpu.hip", "gfaparse.hip"]
FLAGS = [
    "-O3", "-std=c++17", "-fPIC", "-shared", "--offload-arch=gfx950",
    "-ffp-contract=off",          # fp64 epilogues follow the reference's operation order exactly
    "-fvisibility=hidden", "-Wall", "-Wno-unused-function", "-Wno-unused-value", "-Wno-unused-result",
]


OBJ_DIR = os.path.join(HERE, "_obj")  # per-source objects (git-ignored *.o); only stale ones are recompiled
LINK_LIBS = ["-ldl"]  # RCCL itself is dlopen()ed by multigpu.hip when a communicator is created


def _headers_mtime() -> float:
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    deps.append(os.path.join(HERE, "..", "include", "impop_hip.h"))
    deps.append(os.path.abspath(__file__))
    return max(os.path.getmtime(d) for d in deps)


def _compile_one(args):
    hipcc, src, obj, verbose = args
    cmd = [hipcc] + [f for f in FLAGS if f != "-shared"] + ["-c", "-o", obj + ".tmp", src]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        return src, r.stdout + r.stderr
    os.replace(obj + ".tmp", obj)
    return src, None


def build(force: bool = False, verbose: bool = False) -> str:
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    os.makedirs(OBJ_DIR, exist_ok=True)
    hdr_t = _headers_mtime()
    jobs, objs = [], []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(OBJ_DIR, os.path.splitext(s)[0] + ".o")
        objs.append(obj)
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), hdr_t):
            jobs.append((hipcc, src, obj, verbose))
    if not jobs and os.path.exists(SO) and os.path.getmtime(SO) >= max(os.path.getmtime(o) for o in objs):
        return SO
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: cannot build libimpop_hip.so")
    if jobs:
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=min(len(jobs), max((os.cpu_count() or 2) - 1, 1))) as ex:
            for src, err in ex.map(_compile_one, jobs):
                if err is not None:
                    raise RuntimeError(f"hipcc failed on {src}:\n{err}")
    cmd = [hipcc, "-shared", "-fPIC", "--offload-arch=gfx950", "-fvisibility=hidden", "-o", SO + ".tmp"] + objs + LINK_LIBS
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n" + r.stdout + r.stderr)
    os.replace(SO + ".tmp", SO)
    return SO


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
