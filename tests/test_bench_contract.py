"""GPU: bench.py keeps the driver's contract — one JSON line with the required keys, `roofline` and
(when not disabled) `cpu_baseline` — at N = 1, and the N > 1 code path (process group, sharded windows,
all-gather of records, MAX over ranks) runs with two ranks sharing the one GPU over gloo (rehearsal mode;
the measured configuration is one rank per GPU over RCCL)."""
import json
import os
import socket
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu
REQUIRED = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline")


def _last_json(text):
    lines = [ln for ln in text.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, text  # exactly ONE JSON line on stdout
    return json.loads(lines[0])


def test_bench_single_gpu_line():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--n-windows", "96",
                        "--no-cpu-baseline"], capture_output=True, text=True, cwd=ROOT)
    assert r.returncode == 0, r.stderr
    d = _last_json(r.stdout)
    for k in REQUIRED:
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["unit"] == "windows/s" and d["scaling"] == "weak" and d["data"] == "synthetic" and d["vs_baseline"] is None
    assert "workload" in d["config"] and d["value"] > 0
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12


def test_bench_two_ranks_gloo_rehearsal():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, IMPOP_BENCH_BACKEND="gloo")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3",
                        "--warmup", "1", "--n-windows", "96"], capture_output=True, text=True, cwd=ROOT, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _last_json(r.stdout)
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["cpu_baseline"] is None
    assert d["config"]["windows_per_gpu"] == 96
