"""The drop-in boundary is a plain C ABI: include/impop_hip.h compiles as C and examples/scan_from_c.c
links against libimpop_hip.so with gcc alone (CPU check); on a GPU it runs and its numbers agree with the
oracle."""
import os
import re
import subprocess

import numpy as np
import pytest

from conftest import ROOT

SRC = os.path.join(ROOT, "examples", "scan_from_c.c")
SRC_SHARDED = os.path.join(ROOT, "examples", "scan_sharded_from_c.c")


def _build(out, src=SRC):
    subprocess.check_call(["gcc", "-O2", "-Wall", "-Werror", "-std=c99", "-I" + os.path.join(ROOT, "include"), src, "-o", out,
                           "-L" + os.path.join(ROOT, "impop_amd"), "-limpop_hip", "-Wl,-rpath," + os.path.join(ROOT, "impop_amd")])


def test_c_example_compiles_and_links(tmp_path):
    from impop_amd import build
    build.build(force=False, verbose=False)
    _build(str(tmp_path / "scan_from_c"))
    _build(str(tmp_path / "scan_sharded_from_c"), SRC_SHARDED)


@pytest.mark.gpu
@pytest.mark.parametrize("n_shards", [1, 2, 3, 5])
def test_c_sharded_example_runs(tmp_path, n_shards):
    """The multi-GPU entries from plain C: impop_scan_sharded over n contexts (sharing the one device of the test
    box) returns byte for byte the records of one context, and impop_gather_records over a one-rank RCCL
    communicator hands them back intact."""
    exe = str(tmp_path / "scan_sharded_from_c")
    _build(exe, SRC_SHARDED)
    r = subprocess.run([exe, str(n_shards)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert f"over {n_shards} shards: records byte-identical to one context's" in r.stdout
    assert f"all-pairs mode over {n_shards} shards: records byte-identical to one context's" in r.stdout
    assert "RCCL all-gather (1 rank): records intact" in r.stdout


@pytest.mark.gpu
def test_c_example_runs(tmp_path, oracle):
    exe = str(tmp_path / "scan_from_c")
    _build(exe)
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    lines = r.stdout.strip().splitlines()
    assert len(lines) == 4 and all(ln.startswith("window [") for ln in lines[:3])
    D, _ = oracle.tajimas_d(446, 20.0, 0.59146123)
    assert lines[3] == f"tajimas_d(446, 20, 0.59146123) = {D:.10f}"
    # window 3 has no seq_len: pi_site is NaN, everything else finite
    assert "pi_site=nan" in lines[2].lower() and "pi_site=nan" not in lines[0].lower()
    vals = [float(v) for v in re.findall(r"pi=([-0-9.e+]+)", lines[0])]
    assert vals and 0 < vals[0] < 1
