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
            "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "ranks")
RANK_KEYS = ("backend", "kernel_ms_avg_min", "kernel_ms_avg_max", "kernel_ms_avg_per_rank", "gather_exposed_ms_per_step_max",
             "gather_issue_host_ms_per_step_max", "elapsed_s_min", "elapsed_s_max", "gathered_records_checked")


def _last_json(text):
    lines = [ln for ln in text.strip().splitlines() if ln.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), text  # stdout is exactly ONE JSON line (RCCL's banner goes to stderr)
    return json.loads(lines[0])


def test_bench_single_gpu_line():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--n-windows", "96",
                        "--no-cpu-baseline", "--all-pairs-windows", "96"], capture_output=True, text=True, cwd=ROOT)
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
    for k in RANK_KEYS:  # the fields a bad N-GPU number is diagnosed from are present at N = 1 too
        assert k in d["ranks"], k
    assert len(d["ranks"]["kernel_ms_avg_per_rank"]) == 1
    # the default command also measures the reference's DEFAULT chain (run_tajd.sh: pica2 -t 0.999 -r 5 -> %.8f -> tj_d) on
    # the all-pairs path, parity-gated against the oracle inside bench.py, with its own roofline block
    ap = d["secondary"]["all_pairs_mode"]
    assert "error" not in ap, ap
    assert ap["windows"] == 96 and ap["windows_per_s"] > 0 and ap["roofline"]["bound"] == "mfma" and ap["roofline"]["unit"] == "TFLOP/s"
    assert abs(ap["roofline"]["frac"] - ap["roofline"]["achieved"] / ap["roofline"]["peak"]) < 1e-12
    # ... and the dominant kernel alone from HIP events on its stream: shorter than the call, hence a higher fraction
    assert ap["roofline"]["gram_kernel_launches"] == 5 and 0 < ap["roofline"]["gram_kernel_ms_avg"] < ap["s_per_call"] * 1e3
    assert ap["roofline"]["gram_kernel_frac"] > ap["roofline"]["frac"]


def test_bench_one_rank_rccl_gather():
    """The RCCL path itself (communicator, all_gather_into_tensor on device buffers, stream ordering between the
    scan and the collective) with the one rank a one-GPU box allows; bench.py asserts inside that the buffers
    gathered by the first AND the timed steps equal the records the rank computed.  2000 windows = a scan of
    about a millisecond, so an unordered gather would read a buffer that is still being written."""
    env = dict(os.environ, IMPOP_BENCH_FORCE_DIST="1")
    env.pop("MASTER_PORT", None)  # no launcher: bench.py must pick a free port itself, not a fixed one
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "6", "--warmup", "2", "--n-windows", "2000",
                        "--no-cpu-baseline", "--no-secondary"], capture_output=True, text=True, cwd=ROOT, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _last_json(r.stdout)
    assert d["n_gpus"] == 1 and d["ranks"]["backend"] == "rccl" and d["ranks"]["gathered_records_checked"] is True
    assert d["ranks"]["gather_bytes_per_rank"] == 2000 * 128
    assert d["ranks"]["kernel_ms_avg_min"] > 0


def test_bench_two_ranks_gloo_rehearsal():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, IMPOP_BENCH_BACKEND="gloo")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4",
                        "--warmup", "1", "--n-windows", "1500"], capture_output=True, text=True, cwd=ROOT, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _last_json(r.stdout)
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["cpu_baseline"] is None
    assert d["config"]["windows_per_gpu"] == 1500  # a scan of ~0.7 ms per rank: ordering bugs would show
    assert d["ranks"]["backend"] == "gloo-rehearsal" and d["ranks"]["gathered_records_checked"] is True
    assert len(d["ranks"]["kernel_ms_avg_per_rank"]) == 2


def test_bench_config4_sharded_sliding_windows():
    """BASELINE configs[3] through the rank-aware path: ONE global list of 10 kb windows at a 5 kb step over a (here: small) genome,
    impop_shard_windows ranges + halo, every rank generating only its slab, one gather per step.  bench.py asserts inside that the
    records gathered by the first and the timed steps are the ranks' own and that the windows around every shard boundary equal
    stand-alone scans byte for byte.  Here: two ranks over gloo sharing the box's one GPU, then the RCCL path with the one rank a
    one-GPU box allows; both must report the same windows as a plain one-rank run."""
    base = [os.path.join(ROOT, "bench.py"), "--config4", "--genome-sites", "20000123", "--steps", "3", "--warmup", "1"]
    one = subprocess.run([sys.executable] + base, capture_output=True, text=True, cwd=ROOT, timeout=900)
    assert one.returncode == 0, one.stderr[-3000:]
    d1 = _last_json(one.stdout)
    assert d1["scaling"] == "strong" and d1["n_gpus"] == 1 and "configs[3]" in d1["config"]["workload"]
    assert d1["config"]["windows_total"] == 4001 and d1["config"]["window_sites"] == 10000 and d1["config"]["step_sites"] == 5000  # the last one clipped
    # every site is read once although every site lies in two windows
    assert abs(d1["config"]["sites_read_per_step"] - 20000123) < 64 * 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, IMPOP_BENCH_BACKEND="gloo")
    two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", str(port)] + base + ["--gpus", "2"], capture_output=True, text=True, cwd=ROOT, env=env, timeout=900)
    assert two.returncode == 0, two.stderr[-3000:]
    d2 = _last_json(two.stdout)
    assert d2["n_gpus"] == 2 and d2["config"]["windows_total"] == 4001 and d2["config"]["windows_per_rank"] == [2001, 2000]
    assert d2["ranks"]["gathered_records_checked"] is True and d2["ranks"]["boundary_windows_checked_against_standalone_scans"] is True
    # the halo: the two slabs together hold (window - step) sites more than the genome
    assert 0 < d2["config"]["sites_read_per_step"] - 20000123 <= 5000 + 2 * 64
    assert "byte-identical to stand-alone scans" in two.stderr
    env = dict(os.environ, IMPOP_BENCH_FORCE_DIST="1")
    env.pop("MASTER_PORT", None)
    rc = subprocess.run([sys.executable] + base, capture_output=True, text=True, cwd=ROOT, env=env, timeout=900)
    assert rc.returncode == 0, rc.stderr[-3000:]
    d3 = _last_json(rc.stdout)
    assert d3["ranks"]["backend"] == "rccl" and d3["ranks"]["gathered_records_checked"] is True
