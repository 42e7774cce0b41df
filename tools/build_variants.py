#!/usr/bin/env python3
"""Timing / A-B builds of libimpop_hip.so: recompile ONE source with extra flags and link it with the product's other objects.

    python tools/build_variants.py pairwise.hip ab1:-DIMPOP_GRAM_ABLATE=1 ab2:-DIMPOP_GRAM_ABLATE=2 ab3:-DIMPOP_GRAM_ABLATE=3

-> impop_amd/_variants/libimpop_<tag>.so (git-ignored, travels with gpurun); select with IMPOP_HIP_LIBRARY=<path>."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from impop_amd import build as B  # noqa: E402


def main():
    src_name, specs = sys.argv[1], sys.argv[2:]
    B.build()  # the product objects are current
    vdir = os.path.join(ROOT, "impop_amd", "_variants")
    os.makedirs(vdir, exist_ok=True)
    stem = os.path.splitext(src_name)[0]
    procs = []
    for spec in specs:
        tag, _, flags = spec.partition(":")
        obj = os.path.join(vdir, f"{stem}_{tag}.o")
        cmd = ["hipcc"] + [f for f in B.FLAGS if f != "-shared"] + flags.split(",") + ["-c", "-o", obj, os.path.join(B.CSRC, src_name)]
        procs.append((tag, obj, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for tag, obj, p in procs:
        out, _ = p.communicate()
        if p.returncode:
            print(out)
            sys.exit(1)
        objs = [obj if os.path.basename(o) == stem + ".o" else o for o in
                (os.path.join(B.OBJ_DIR, os.path.splitext(s)[0] + ".o") for s in B.SOURCES)]
        so = os.path.join(vdir, f"libimpop_{tag}.so")
        subprocess.run(["hipcc", "-shared", "-fPIC", "--offload-arch=gfx950", "-fvisibility=hidden", "-o", so] + objs + B.LINK_LIBS, check=True)
        print(so)


if __name__ == "__main__":
    main()
