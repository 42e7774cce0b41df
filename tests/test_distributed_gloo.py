"""CPU, world_size 2, gloo: the sharding + single all-gather path.  The per-shard scan is the
CPU oracle here (tests may use it); on the GPU the same code path calls BitMatrix.scan."""
import os
import socket

import numpy as np
import pytest

from conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n, W, seed, size, step, q):
    import sys
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import impop_amd
    from impop_amd.distributed import scan_sharded
    from oracle import oracle as orc
    from synth_ref import synth_matrix
    windows = impop_amd.fixed_windows(W, size, step)
    inA = np.zeros(n, np.uint8); inA[: n // 3] = 1
    inB = np.zeros(n, np.uint8); inB[n // 3: n // 2] = 1
    ones = orc.pack_mask(np.ones(n, np.uint8))

    def local_scan(loc, s0, s1):
        # this rank only materialises its own slab [s0, s1) of the (counter-based) matrix
        bits = orc.pack_hap_major(synth_matrix(n, s0, s1, seed=seed))
        out = np.zeros(len(loc), dtype=impop_amd.STATS_DTYPE)
        for i, w in enumerate(loc):
            r = orc.window_sitecount(bits, n, int(w["site_begin"]), int(w["site_end"]), ones, orc.pack_mask(inA),
                                     orc.pack_mask(inB), int(w["seq_len"]))
            for k, v in r.items():
                out[i][k] = v
        return out

    rec = scan_sharded(windows, local_scan, world, rank)
    q.put((rank, rec.tobytes()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("size,step", [(1000, None), (1000, 500)])
def test_two_rank_gather_equals_single_process(size, step):
    import torch.multiprocessing as mp
    from impop_amd.distributed import shard_range, shard_windows
    import impop_amd
    n, W, seed = 40, 7300, 9
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n, W, seed, size, step, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got[0] == got[1]  # every rank holds the full, identical result
    # single-process reference
    q1 = ctx.Queue()
    p1 = ctx.Process(target=_worker, args=(0, 1, _free_port(), n, W, seed, size, step, q1))
    p1.start()
    _, single = q1.get(timeout=120)
    p1.join(timeout=60)
    a = np.frombuffer(got[0], dtype=impop_amd.STATS_DTYPE)
    b = np.frombuffer(single, dtype=impop_amd.STATS_DTYPE)
    assert len(a) == len(b) == len(impop_amd.fixed_windows(W, size, step))
    assert a.tobytes() == b.tobytes()  # bit-identical: 1 vs 2 shards
    # sharding helpers
    assert [shard_range(10, 3, r) for r in range(3)] == [(0, 4), (4, 7), (7, 10)]
    w = impop_amd.fixed_windows(W, size, step)
    loc, s0, s1, (lo, hi) = shard_windows(w, 2, 1)
    assert int(loc["site_begin"].min()) == 0 and s0 == int(w[lo]["site_begin"]) and s1 == W


def _gram_worker(rank, world, port, n, W, seed, q):
    import sys
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from impop_amd.distributed import pairwise_counts_sharded
    from oracle import oracle as orc
    from synth_ref import synth_matrix

    def local_counts(lo, hi):  # this rank only materialises its own site range
        bits = orc.pack_hap_major(synth_matrix(n, lo, hi, seed=seed))
        return orc.pairwise_counts(bits, n, 0, hi - lo)

    I = pairwise_counts_sharded(13, W - 7, local_counts, world, rank)
    q.put((rank, I.tobytes()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gram_allreduce_equals_single_process():
    """K-split of one long window over ranks + one all-reduce(sum) of the integer Gram matrix."""
    import torch.multiprocessing as mp
    from oracle import oracle as orc
    from synth_ref import synth_matrix
    n, W, seed = 23, 5001, 4
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gram_worker, args=(r, 2, port, n, W, seed, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got[0] == got[1]
    want = orc.pairwise_counts(orc.pack_hap_major(synth_matrix(n, 0, W, seed=seed)), n, 13, W - 7)
    assert (np.frombuffer(got[0], dtype=np.int64).reshape(n, n) == want).all()
