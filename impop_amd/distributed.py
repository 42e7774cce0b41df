"""Multi-GPU: windows shard embarrassingly over ranks (one process per GPU); the only
exchange is ONE all-gather of fixed-size per-window records (RCCL over xGMI with the
"nccl" backend; gloo on CPU for tests).  No statistic spans windows, so there is no
reduction and no data-path collective."""
from __future__ import annotations

from typing import Callable, Tuple

import numpy as np

from .engine import WINDOW_DTYPE


def shard_range(n_items: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous near-equal split: the first (n_items % world) ranks get one extra item."""
    base, extra = divmod(n_items, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_windows(windows: np.ndarray, world: int, rank: int):
    """-> (local windows rebased to the slab, slab_begin, slab_end, (lo, hi) window index range).

    Each rank needs only the sites its windows touch: sliding windows simply make
    neighbouring slabs overlap by (window - step) sites (the halo)."""
    w = np.ascontiguousarray(windows, dtype=WINDOW_DTYPE)
    lo, hi = shard_range(len(w), world, rank)
    loc = w[lo:hi].copy()
    if len(loc) == 0:
        return loc, 0, 0, (lo, hi)
    s0 = int(loc["site_begin"].min())
    s1 = int(loc["site_end"].max())
    loc["site_begin"] -= np.uint64(s0)
    loc["site_end"] -= np.uint64(s0)
    return loc, s0, s1, (lo, hi)


def gather_records(local_records, n_total: int, world: int, rank: int, device=None):
    """One all_gather of the per-window records.  `local_records`: numpy structured array (any
    fixed itemsize) or a flat uint8 torch tensor already on the right device.  Every rank
    returns the full array in global window order."""
    import torch
    import torch.distributed as dist

    if isinstance(local_records, np.ndarray):
        itemsize = local_records.dtype.itemsize
        dtype = local_records.dtype
        t = torch.from_numpy(np.frombuffer(local_records.tobytes(), dtype=np.uint8).copy())
        if device is not None:
            t = t.to(device)
    else:
        raise TypeError("gather_records expects a numpy structured array")
    if world == 1:
        return local_records.copy()
    # ranks may own different counts (n_total % world != 0): pad to the largest shard
    counts = [shard_range(n_total, world, r)[1] - shard_range(n_total, world, r)[0] for r in range(world)]
    cap = max(counts) * itemsize
    padded = torch.zeros(cap, dtype=torch.uint8, device=t.device)
    padded[: t.numel()] = t
    out = torch.empty(world * cap, dtype=torch.uint8, device=t.device)
    dist.all_gather_into_tensor(out, padded)
    flat = out.cpu().numpy()
    parts = [flat[r * cap: r * cap + counts[r] * itemsize] for r in range(world)]
    return np.frombuffer(np.concatenate(parts).tobytes(), dtype=dtype)


def scan_sharded(windows: np.ndarray, local_scan: Callable[[np.ndarray, int, int], np.ndarray], world: int, rank: int,
                 device=None) -> np.ndarray:
    """Shard `windows` over ranks, run `local_scan(local_windows, slab_begin, slab_end)` on this
    rank's shard (product: BitMatrix.scan on the slab resident on this GPU) and all-gather."""
    loc, s0, s1, _ = shard_windows(windows, world, rank)
    rec = local_scan(loc, s0, s1)
    return gather_records(rec, len(windows), world, rank, device)


def pairwise_counts_sharded(site_begin: int, site_end: int, local_counts: Callable[[int, int], np.ndarray], world: int, rank: int,
                            device=None) -> np.ndarray:
    """The one place where the all-pairs path has a real exchange step (BASELINE config 5:
    one giant window): the site axis of [site_begin, site_end) is cut into `world` contiguous ranges, every
    rank computes the integer Gram matrix I_ij of ITS range (`local_counts(lo, hi)`, product:
    BitMatrix.pairwise_counts on the slab resident on this GPU) and ONE all-reduce(sum) adds them up —
    RCCL over xGMI with the "nccl" backend, gloo on the CPU for tests.  Integer sums, so the result is
    bit-identical for any number of ranks.  Every rank returns the full n x n int64 matrix."""
    lo, hi = shard_range(site_end - site_begin, world, rank)
    part = np.ascontiguousarray(local_counts(site_begin + lo, site_begin + hi), dtype=np.int64)
    if world == 1:
        return part
    import torch
    import torch.distributed as dist

    t = torch.from_numpy(part)
    if device is not None:
        t = t.to(device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.cpu().numpy()
