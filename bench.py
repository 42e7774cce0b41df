#!/usr/bin/env python3
"""bench.py — windows/sec of pi + Hudson Fst + Tajima's D, 465 haplotypes, 50 kb windows.

One step = one pass of the hot path (impop_scan_plan_launch) over one chromosome-scale batch of
synthetic windows that is already resident in HBM: BASELINE.json configs[1]/[2] — chr2-sized
(4 854 windows x 50 000 sites), 465 haplotypes, populations A = rows 0-139, B = rows 140-239 —
computing pi, pi_A, pi_B, Dxy, Fst, S and Tajima's D for every window in the same pass.

    python bench.py --gpus N --steps K --warmup W

N > 1 is launched by torch.distributed.run, one rank per GPU; every rank holds its own
chromosome shard (weak scaling, no data-path collective) and the per-window records are
all-gathered once per step over RCCL.  Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6.3 TB/s is what a copy achieves


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def pmc_traffic(n_hap, window, n_windows):
    """HBM bytes per launch of the streaming kernel from the committed rocprofv3 PMC passes
    (profiles/r03f_pmc_hbm_traffic.json, written by tools/summarise_pmc_traffic.py: FETCH_SIZE x1024x2 +
    WRITE_SIZE x1024, collected in their own runs as the microarch guide prescribes).  PMC cannot be read live inside this process;
    the figure is reported only when it was collected on this exact workload, else null."""
    path = os.path.join(ROOT, "profiles", "r03f_pmc_hbm_traffic.json")
    try:
        with open(path) as f:
            d = json.load(f)
    except OSError:
        return None, None
    wl = d.get("workload", {})
    if (wl.get("n_hap"), wl.get("window_sites"), wl.get("windows_per_gpu")) != (n_hap, window, n_windows):
        return None, None
    for k, e in d["kernels"].items():
        if "scan_tiles_kernel" in k:
            return e["hbm_bytes_per_launch"], os.path.relpath(path, ROOT)
    return None, None


def host_cores():
    """Threads for the all-core CPU port: the affinity mask, clipped by the cgroup CPU quota and by the
    16-core share a one-GPU box of this pool grants (IMPOP_BENCH_CPU_THREADS overrides)."""
    if os.environ.get("IMPOP_BENCH_CPU_THREADS"):
        return max(int(os.environ["IMPOP_BENCH_CPU_THREADS"]), 1)
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(int(int(quota) / int(period) + 0.999), 1))
    except (OSError, ValueError):
        pass
    return max(min(n, 16), 1)


FP4_DENSE_PEAK_TFLOPS = 10000.0  # MI355X FP4 / FP6 dense MFMA peak (MI355X_MICROARCH.md: ~10 PF dense); 1 MAC = 2 FLOP


def all_pairs_point(ctx, n, W, in_a, in_b, n_windows=4096):
    """The all-pairs (Gram) mode at the reference's DEFAULT settings — run_tajd.sh:9-10,166-180: pica2 -t 0.999 -r 5 per window,
    its "%.8f" pi into Tajima's D; h-fst on the same identities — on the headline window shape (n haplotypes x W sites).
    This path is MFMA-bound: one impop_pairwise_scan call over `n_windows` resident windows, host clock around the
    whole call (launches, epilogue kernels, copies of the 96-byte records included).  Reported NEXT TO the headline under
    `secondary.all_pairs_mode`, with its own roofline block; never folded into `value`."""
    import impop_amd
    t0 = time.perf_counter()
    pm = ctx.synthetic(n, W * n_windows, seed=20251031, keep_hap_major=True)
    ctx.synchronize()
    t_gen = time.perf_counter() - t0
    pw = impop_amd.fixed_windows(W * n_windows, W)
    kw = dict(kind="match", threshold=0.999, round_digits=5)
    pm.pairwise_scan(pw[:64], None, in_a, in_b, **kw)  # scratch, code objects, the matrix's site bitmap
    for _ in range(3):  # warm-up calls: the chip settles its clock under FP4 MFMA load over the first launches (profiles/r03_gram_experiments.txt)
        first = pm.pairwise_scan(pw, None, in_a, in_b, **kw)
    ctx.synchronize()
    reps = 5
    ctx.gram_timing(True)  # HIP events on the launch stream around the Gram kernel of every call
    t0 = time.perf_counter()
    for _ in range(reps):
        res = pm.pairwise_scan(pw, None, in_a, in_b, **kw)
    dt = (time.perf_counter() - t0) / reps
    gram_ms, gram_launches = ctx.gram_elapsed()
    ctx.gram_timing(False)
    gram_s = gram_ms / 1e3 / max(gram_launches, 1)
    assert res.tobytes() == first.tobytes()  # integer Gram + ordered fp64 epilogues: byte-reproducible
    # parity gate of this mode: two windows against the CPU oracle's dense restatement of the same chain
    import numpy as np
    from oracle import oracle as orc
    orc.build()
    ones = orc.pack_mask(np.ones(n, np.uint8))
    for wi in (0, n_windows - 1):
        s0, s1 = int(pw[wi]["site_begin"]), int(pw[wi]["site_end"])
        bits = pm.download(s0, s1)
        sim = orc.identity(orc.pairwise_counts(bits, n, 0, s1 - s0), s1 - s0, 0)
        pi, ps, _, G = orc.pica2(sim, 0.999, W, 5)
        S = orc.window_sitecount(bits, n, 0, s1 - s0, ones, orc.pack_mask(in_a), orc.pack_mask(in_b), W)["s_all"]
        D = orc.tajimas_d(n, float(S), orc.py_round(ps, 8))[0]
        h, _ = orc.hfst(sim, in_a, in_b, W, 5)
        got = res[wi]
        assert int(got["n_groups"]) == G and int(got["s_all"]) == S, ("all-pairs parity gate", wi, int(got["n_groups"]), G)
        for a_, b_ in ((float(got["pi"]), pi), (float(got["pi_site"]), ps), (float(got["tajima_d"]), D), (float(got["fst"]), h["fst"]),
                       (float(got["dxy"]), h["dxy"])):
            assert abs(a_ - b_) <= 1e-9 * max(abs(a_), abs(b_)), ("all-pairs parity gate", wi, a_, b_)
    macs = n * (n + 1) // 2 * W  # algorithmic MACs per window (upper triangle incl. diagonal)
    achieved = 2.0 * macs * n_windows / dt / 1e12
    out = {"windows_per_s": n_windows / dt, "windows": n_windows, "s_per_call": dt, "matrix_generation_s": t_gen,
           "what": f"impop_pairwise_scan, {n} haplotypes x {W} sites per window: FP4-MFMA Gram + pica2 (-t 0.999 -r 5) + h-fst + S "
                   "(cached site bitmap) + Tajima's D from the %.8f pi, per window, incl. launches and record copies",
           "settings": "run_tajd.sh defaults: pica2 -t 0.999 -r 5 -> %.8f -> tj_d",
           "mean_groups_per_window": float(res["n_groups"].mean()),
           "roofline": {"bound": "mfma", "achieved": achieved, "peak": FP4_DENSE_PEAK_TFLOPS, "unit": "TFLOP/s",
                        "frac": achieved / FP4_DENSE_PEAK_TFLOPS, "traffic": None,
                        "algorithmic_macs_per_window": macs, "timed": "host clock around the whole call (end to end)",
                        "kernel": "gram_fp4_kernel + epilogue kernels + launches + record copies",
                        # the dominant kernel alone, HIP events on its stream (what rocprofv3 --kernel-trace reports for it)
                        "gram_kernel_ms_avg": gram_s * 1e3, "gram_kernel_launches": int(gram_launches),
                        "gram_kernel_achieved": 2.0 * macs * n_windows / gram_s / 1e12 if gram_s > 0 else None,
                        "gram_kernel_frac": 2.0 * macs * n_windows / gram_s / 1e12 / FP4_DENSE_PEAK_TFLOPS if gram_s > 0 else None}}
    return out, pm, pw, res


def secondary_points(ctx, bm, windows, in_a, in_b, ref_records, extended, n_pairs_windows=4096):
    """Secondary measurements reported NEXT TO the headline (never folded into `value`).  Always: the all-pairs (Gram,
    MFMA-bound) mode at the reference's default settings.  With --secondary also SURVEY §8(d)'s "W = S-only" point (the same
    windows scanned from the matrix compacted to its variable sites, records byte-identical) and the all-pairs mode on the
    compacted matrix."""
    out = {}
    n, W = bm.n_hap, int(windows[0]["site_end"]) - int(windows[0]["site_begin"])
    pm = None
    try:
        out["all_pairs_mode"], pm, pw, ref_pw = all_pairs_point(ctx, n, W, in_a, in_b, n_pairs_windows)
    except Exception as e:  # a secondary point must not take the headline line down; the error is reported
        out["all_pairs_mode"] = {"error": repr(e)}
    if extended and pm is not None:
        try:
            kw = dict(kind="match", threshold=0.999, round_digits=5)
            cmp_ = pm.compact()  # the all-pairs path on the variable sites only (+ the dropped all-ones count): identical records
            got_pw = cmp_.pairwise_scan(pw, None, in_a, in_b, **kw)
            ctx.synchronize()
            t0 = time.perf_counter()
            for _ in range(3):
                cmp_.pairwise_scan(pw, None, in_a, in_b, **kw)
            dtc = (time.perf_counter() - t0) / 3
            out["all_pairs_mode_variable_sites_only"] = {"windows_per_s": len(pw) / dtc, "windows": len(pw), "kept_sites": cmp_.n_site,
                                                         "of_sites": pm.n_site,
                                                         "records_identical_to_full_matrix": bool(got_pw.tobytes() == ref_pw.tobytes())}
            cmp_.free()
        except Exception as e:
            out["all_pairs_mode_variable_sites_only"] = {"error": repr(e)}
        try:  # BASELINE configs[3]'s window shape (10 kb windows at a 5 kb step) at the reference's default chain: overlapping windows
            # share 5 kb segment Gram matrices, a window's statistics sum two of them
            import impop_amd
            kw = dict(kind="match", threshold=0.999, round_digits=5)
            n_sl = min(len(pw), 4096)
            sw = impop_amd.fixed_windows(10000 * (n_sl // 2 + 1), 10000, 5000)[:n_sl]
            pm.pairwise_scan(sw, None, in_a, in_b, **kw)
            ctx.synchronize()
            t0 = time.perf_counter()
            for _ in range(3):
                got_sl = pm.pairwise_scan(sw, None, in_a, in_b, **kw)
            dts = (time.perf_counter() - t0) / 3
            one = pm.pairwise_scan(sw[7:8], None, in_a, in_b, **kw)  # the same window asked for alone: no shared segments
            out["all_pairs_mode_sliding_10kb_5kb"] = {"windows_per_s": len(sw) / dts, "windows": len(sw), "ms_per_call": dts * 1e3,
                                                      "window_alone_identical": bool(one.tobytes() == got_sl[7:8].tobytes())}
        except Exception as e:
            out["all_pairs_mode_sliding_10kb_5kb"] = {"error": repr(e)}
    if pm is not None:
        pm.free()
    if extended:
        try:
            t0 = time.perf_counter()
            cm = bm.compact()
            ctx.synchronize()
            t_c = time.perf_counter() - t0
            plan = cm.plan(windows, None, in_a, in_b)
            plan.launch(); ctx.synchronize()
            plan.timing(True)
            for _ in range(20):
                plan.launch()
            ms, k = plan.elapsed()
            same = plan.fetch().tobytes() == ref_records.tobytes()
            out["variable_sites_only"] = {"windows_per_s_kernel": len(windows) / (ms / k / 1e3), "kernel_ms": ms / k,
                                          "variable_sites": cm.n_site, "of_sites": bm.n_site, "compaction_s": t_c,
                                          "layout_GBps": plan.bytes_streamed / (ms / k / 1e3) / 1e9,
                                          "records_identical_to_full_matrix": bool(same)}
            plan.destroy(); cm.free()
        except Exception as e:
            out["variable_sites_only"] = {"error": repr(e)}
    return out


def cpu_baseline(bm, windows, in_a, in_b, budget_s=12.0):
    """Time the CPU oracle (oracle/impop_oracle.c, the restated reference algorithm: all-pairs
    Hamming -> identity -> pica2/h-fst/tj_d) on a bounded sample of the SAME windows, one host
    core.  Also time the oracle's site-count formulation (the algorithm the GPU kernel uses)."""
    import numpy as np

    from oracle import oracle as orc
    orc.build()
    n = bm.n_hap
    ones = orc.pack_mask(np.ones(n, np.uint8))
    ma, mb = orc.pack_mask(in_a), orc.pack_mask(in_b)
    t_all, n_all = 0.0, 0
    t_sc, n_sc = 0.0, 0
    first = None
    i = 0
    while t_all < budget_s and i < len(windows):
        w = windows[i]
        s0, s1 = int(w["site_begin"]), int(w["site_end"])
        bits = bm.download(s0, s1)
        t0 = time.perf_counter()
        rec = orc.window_allpairs(bits, n, 0, s1 - s0, ones, ma, mb, int(w["seq_len"]))
        t_all += time.perf_counter() - t0
        n_all += 1
        if first is None:
            first = rec
        i += 1
    # site-count port on site-major words (conversion not timed: the GPU's layout is not timed either)
    j = 0
    while t_sc < budget_s / 2 and j < len(windows):
        w = windows[j]
        s0, s1 = int(w["site_begin"]), int(w["site_end"])
        bits = bm.download(s0, s1)
        sm = orc.to_sitemajor(bits, n, s1 - s0)
        t0 = time.perf_counter()
        orc.site_scan_sitemajor(sm, n, 0, s1 - s0, ones, ma, mb)
        t_sc += time.perf_counter() - t0
        n_sc += 1
        j += 1
    # the same port on every host core (OpenMP over windows) on a 64-window slab, repeated for ~3 s
    cores = host_cores()
    Wn = int(windows[0]["site_end"]) - int(windows[0]["site_begin"])
    n_slab = min(64, len(windows))
    uniform = all(int(w["site_end"]) - int(w["site_begin"]) == Wn and int(w["site_begin"]) == k * Wn
                  for k, w in enumerate(windows[:n_slab]))
    mt = None
    if uniform and n_slab:
        sm = orc.to_sitemajor(bm.download(0, n_slab * Wn), n, n_slab * Wn)
        ints1, sums1 = orc.site_scan_sitemajor_windows(sm, n, Wn, n_slab, ones, ma, mb, 1)
        t_mt, reps = 0.0, 0
        while t_mt < 3.0:
            t0 = time.perf_counter()
            ints_m, sums_m = orc.site_scan_sitemajor_windows(sm, n, Wn, n_slab, ones, ma, mb, cores)
            t_mt += time.perf_counter() - t0
            reps += 1
        assert (ints_m == ints1).all() and (sums_m == sums1).all()
        mt = {"value": reps * n_slab / t_mt, "unit": "windows/s", "cores": cores,
              "sample": f"{n_slab}-window slab x {reps} passes, OpenMP over windows, {t_mt:.1f} s"}
        del sm
    # reference-STYLE chain (TSV text -> csv.DictReader -> dict algorithms), pure Python, 2 windows
    from oracle import ref_style
    t_py, n_py, py_rec = 0.0, 0, None
    names = [f"H{i // 2:05d}#{i % 2 + 1}#chr2:0-{int(windows[0]['seq_len'])}" for i in range(n)]
    for wi in range(min(2, len(windows))):
        w = windows[wi]
        s0, s1 = int(w["site_begin"]), int(w["site_end"])
        bits = bm.download(s0, s1)
        sim = orc.identity(orc.pairwise_counts(bits, n, 0, s1 - s0), s1 - s0, 0)  # what `impg similarity` would hand over (not timed)
        S = orc.window_sitecount(bits, n, 0, s1 - s0, ones, ma, mb, int(w["seq_len"]))["s_all"]
        t0 = time.perf_counter()
        py_rec = ref_style.window_chain(names, sim, in_a, in_b, int(w["seq_len"]), S)
        t_py += time.perf_counter() - t0
        n_py += 1
    # ... and the same chain the way the reference scales today: a process per window over all host cores
    py_mp = None
    try:
        import subprocess
        import tempfile
        with tempfile.TemporaryDirectory() as td:
            k = min(4, len(windows))
            stack = np.stack([bm.download(int(windows[i]["site_begin"]), int(windows[i]["site_end"])) for i in range(k)])
            np.savez(os.path.join(td, "w.npz"), bits=stack, n=np.int64(n), W=np.int64(Wn), in_a=in_a, in_b=in_b)
            r = subprocess.run([sys.executable, "-m", "oracle.ref_style_mp", os.path.join(td, "w.npz"), str(cores)], cwd=ROOT,
                               capture_output=True, text=True, timeout=300)
            if r.returncode == 0:
                py_mp = json.loads(r.stdout.strip().splitlines()[-1])
            else:
                py_mp = {"error": r.stderr[-300:]}
    except Exception as e:  # a secondary figure must not take the bench line down
        py_mp = {"error": repr(e)}
    return {
        "value": n_all / t_all if t_all > 0 else None, "unit": "windows/s", "cores": 1, "kind": "port",
        "reference_style_python": {"value": n_py / t_py if t_py > 0 else None, "unit": "windows/s", "cores": 1,
                                   "sample": f"{n_py} windows: identity table -> .sim text -> csv.DictReader -> dict algorithms "
                                             f"(oracle/ref_style.py, CPython, {t_py:.1f} s)"},
        "sample": f"first {n_all} windows of the timed workload through oracle_window_allpairs "
                  f"(all-pairs Hamming + pica2/h-fst/tj_d restatement, gcc -O2, 1 thread, {t_all:.1f} s)",
        "sitecount_port": {"value": n_sc / t_sc if t_sc > 0 else None, "unit": "windows/s", "cores": 1,
                           "sample": f"{n_sc} windows, oracle_site_scan_sitemajor (integer sums only), {t_sc:.1f} s"},
        "sitecount_port_allcores": mt,
        "reference_style_python_allcores": py_mp,
        "reference_python_measured_in_build_container": "0.67-0.91 s/window for pica2.py alone (BASELINE.md §2)",
    }, first


def run_config4(args, ctx, stream, dev, rank, world, backend, use_dist):
    """BASELINE configs[3]: whole-genome pi + Fst + D scan in 10 kb windows at a 5 kb step, sharded over the GPUs of a node with
    ONE gather of records (the shape of doc/how_pi.md:40-42; the loops of run_tajd.sh:103 / run_h-fst.sh:155 over a genome BED).

    One global window list.  impop_shard_windows gives rank r a contiguous window range and the site slab it touches — its share of
    the genome plus a (window - step) halo towards the next rank — and the rank generates ONLY that slab (the synthetic generator is
    counter-based on the global site index, impop_matrix_synthetic_slab).  A step = every rank scans its windows (each site is read
    once: overlapping windows share elementary segments) and the 128-byte records are all-gathered (uneven shards padded).  Strong
    scaling: the total work is fixed, `value` = windows of the whole list / time of a step.  Checked inside the run: the records
    gathered by the first and by the timed steps are this rank's own; on rank 0, windows on both sides of every shard boundary
    (the ones that need the halo), the first and the last are re-scanned from a small slab generated on its own and must come
    back byte-identical from whichever rank computed them."""
    import numpy as np
    import torch
    import torch.distributed as dist

    import impop_amd
    from impop_amd import engine

    n, G, Wn, step = args.n_hap, args.genome_sites, args.c4_window, args.c4_step
    seed = 20251031
    all_wins = impop_amd.fixed_windows(G, Wn, step)
    NWt = len(all_wins)
    first, cnt, s0, s1 = engine.shard_windows_c(all_wins, world, rank)
    a0 = s0 // 64 * 64                      # slabs start on a 64-site block
    t0 = time.perf_counter()
    bm = ctx.synthetic(n, max(s1 - a0, 64), seed=seed, site_begin=a0)
    ctx.synchronize()
    t_gen = time.perf_counter() - t0
    loc = all_wins[first:first + cnt].copy()
    loc["site_begin"] -= np.uint64(a0)
    loc["site_end"] -= np.uint64(a0)
    in_a = np.zeros(n, np.uint8); in_a[: min(140, n)] = 1
    in_b = np.zeros(n, np.uint8); in_b[min(140, n): min(240, n)] = 1
    plan = bm.plan(loc, None, in_a, in_b, tile_blocks=args.tile_blocks)
    counts = [engine.shard_windows_c(all_wins, world, r)[1] for r in range(world)]
    firsts = [engine.shard_windows_c(all_wins, world, r)[0] for r in range(world)]
    cap = max(max(counts), 1)
    if rank == 0:
        log(f"[bench config4] {NWt} windows of {Wn} sites at step {step} over {G} sites; rank 0 holds sites [{a0}, {s1}) = "
            f"{bm.device_bytes / 1e9:.2f} GB, generated in {t_gen:.2f} s; shards {counts}")
    with torch.cuda.stream(stream):
        bufs = [torch.zeros(cap * 128, dtype=torch.uint8, device=dev) for _ in range(2)]
        gdev = dev if backend == "nccl" else torch.device("cpu")
        gathered = [torch.empty(world * cap * 128, dtype=torch.uint8, device=gdev) for _ in range(2)] if use_dist else None
    works = [None, None]
    counter = [0]

    def step_once():
        b = counter[0] & 1
        counter[0] += 1
        with torch.cuda.stream(stream):
            if works[b] is not None:
                works[b].wait(); works[b] = None
            plan.launch(bufs[b].data_ptr())
            if use_dist:
                src = bufs[b] if backend == "nccl" else bufs[b].cpu()
                works[b] = dist.all_gather_into_tensor(gathered[b], src, async_op=True)

    def drain():
        with torch.cuda.stream(stream):
            for b in (0, 1):
                if works[b] is not None:
                    works[b].wait(); works[b] = None
        torch.cuda.synchronize(dev)

    def all_records(b):
        """the gathered buffer -> records of the whole list in global window order (padding dropped)"""
        if not use_dist:
            return np.frombuffer(bufs[b].cpu().numpy().tobytes(), dtype=impop_amd.STATS_DTYPE)[:cnt].copy()
        raw = gathered[b].cpu().numpy().tobytes()
        parts = [np.frombuffer(raw[r * cap * 128: r * cap * 128 + counts[r] * 128], dtype=impop_amd.STATS_DTYPE) for r in range(world)]
        return np.concatenate(parts)

    step_once(); drain()
    mine = np.frombuffer(bufs[0].cpu().numpy().tobytes(), dtype=impop_amd.STATS_DTYPE)[:cnt].copy()
    ref_all = all_records(0)
    assert ref_all[first:first + cnt].tobytes() == mine.tobytes(), f"rank {rank}: gathered records differ from local (first step)"
    assert len(ref_all) == NWt
    if rank == 0:  # windows that straddle shard boundaries (they need the halo), the first and the last: from slabs of their own
        sample = {0, NWt - 1}
        for r in range(1, world):
            if counts[r]:
                sample |= {max(firsts[r] - 1, 0), firsts[r], min(firsts[r] + 1, NWt - 1)}
        for wi in sorted(sample):
            wb, we = int(all_wins[wi]["site_begin"]), int(all_wins[wi]["site_end"])
            b0 = wb // 64 * 64
            sm = ctx.synthetic(n, we - b0, seed=seed, site_begin=b0)
            one = sm.scan([(wb - b0, we - b0, int(all_wins[wi]["seq_len"]))], None, in_a, in_b)
            sm.free()
            assert one[0].tobytes() == ref_all[wi].tobytes(), f"config4: window {wi} differs from its stand-alone scan"
        log(f"[bench config4] {len(sample)} windows (shard boundaries, first, last) byte-identical to stand-alone scans")
    for _ in range(args.warmup):
        step_once()
    drain()
    plan.timing(True)
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step_once()
    drain()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize(dev)
    elapsed = time.perf_counter() - t0
    kern_ms, launches = plan.elapsed()
    plan.timing(False)
    for b in (0, 1):
        if args.steps > b:
            assert all_records(b).tobytes() == ref_all.tobytes(), f"rank {rank}: records of a timed step differ from the checked first step"
    per = [elapsed, kern_ms / max(launches, 1), float(bm.device_bytes), float(cnt)]
    if use_dist:
        t = torch.tensor(per, dtype=torch.float64, device=gdev)
        allr = torch.empty(world * 4, dtype=torch.float64, device=gdev)
        dist.all_gather_into_tensor(allr, t)
        allr = allr.cpu().numpy().reshape(world, 4)
    else:
        allr = np.array([per])
    elapsed = float(allr[:, 0].max())
    if rank == 0:
        sites_read = float(allr[:, 2].sum()) / bm.bytes_per_site  # every rank reads its slab once per step
        slow = int(allr[:, 1].argmax())
        algo_bytes_slow = n * (allr[slow, 2] / bm.bytes_per_site) / 8.0
        achieved = algo_bytes_slow / (allr[slow, 1] / 1e3) / 1e9
        out = {
            "metric": "windows/sec (pi+Fst+D) for 465-hap HPRC, whole-genome 10 kb sliding windows", "value": NWt * args.steps / elapsed,
            "unit": "windows/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "u32", "data": "synthetic",
            "config": {"workload": f"BASELINE configs[3]: whole-genome scan, {NWt} windows of {Wn} sites at step {step} over {G} sites, {n} "
                                   f"haplotypes, pi + Hudson Fst (A=140 vs B=100) + Tajima's D + S per window; windows sharded over {world} "
                                   f"GPU(s) (contiguous ranges + {Wn - step}-site halo), one all_gather of records per step",
                       "n_hap": n, "window_sites": Wn, "step_sites": step, "genome_sites": G, "windows_total": NWt,
                       "windows_per_rank": [int(c) for c in allr[:, 3]], "slab_GB_per_rank": [float(x) / 1e9 for x in allr[:, 2]],
                       "sites_read_per_step": sites_read, "parallelism": f"windows sharded over {world} GPU(s), one all_gather of records per step"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": None, "kernel": "scan_tiles_kernel (slowest rank)", "kernel_ms_avg": float(allr[slow, 1]),
                         "algorithmic_bytes_per_launch": algo_bytes_slow},
            "cpu_baseline": None,
            "ranks": {"backend": ("rccl" if backend == "nccl" else "gloo-rehearsal") if use_dist else None,
                      "elapsed_s_min": float(allr[:, 0].min()), "elapsed_s_max": float(allr[:, 0].max()),
                      "kernel_ms_avg_per_rank": [float(x) for x in allr[:, 1]], "gathered_records_checked": bool(use_dist),
                      "boundary_windows_checked_against_standalone_scans": True},
            "secondary": None,
        }
        print(json.dumps(out), flush=True)
    plan.destroy()
    bm.free()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--n-hap", type=int, default=465)
    ap.add_argument("--window", type=int, default=50000)
    ap.add_argument("--n-windows", type=int, default=4854, help="per GPU; 4854 = chr2 (242.7 Mb) in 50 kb windows")
    ap.add_argument("--tile-blocks", type=int, default=0)
    ap.add_argument("--config4", action="store_true",
                    help="BASELINE configs[3] instead of the headline: ONE whole-genome list of sliding windows (--c4-window / "
                         "--c4-step over --genome-sites), cut into contiguous ranges over the ranks (impop_shard_windows), every rank "
                         "holding only its slab + halo, one all-gather of the records per step; strong scaling")
    ap.add_argument("--genome-sites", type=int, default=3_100_000_000, help="--config4: sites of the whole genome (3.1 Gb = 186 GB at 465 haplotypes)")
    ap.add_argument("--c4-window", type=int, default=10000)
    ap.add_argument("--c4-step", type=int, default=5000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--secondary", action="store_true",
                    help="also measure the variable-sites-only scan and the all-pairs mode on the compacted matrix (under "
                         "`secondary`, never in `value`).  Off by default: the compacted scan launches the headline kernel's own "
                         "instantiation on another workload, which would pollute a rocprofv3 --stats average of the default command")
    ap.add_argument("--all-pairs-windows", type=int, default=4096, help="windows of the all-pairs point (one impop_pairwise_scan call)")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the all-pairs point (run_tajd.sh's default chain on the MFMA path) the default command measures "
                         "after the timed region")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            log(f"bench.py: --gpus {args.gpus} needs `python -m torch.distributed.run --nproc-per-node {args.gpus} ...`")
            sys.exit(2)

    import numpy as np
    import torch
    import torch.distributed as dist

    import impop_amd

    if not torch.cuda.is_available():
        log("bench.py: no GPU visible; impop_amd has no CPU path to measure")
        sys.exit(2)
    # IMPOP_BENCH_BACKEND=gloo is a REHEARSAL mode only (several ranks on one GPU, records staged
    # through the host); the measured configuration is one rank per GPU over RCCL ("nccl").
    backend = os.environ.get("IMPOP_BENCH_BACKEND", "nccl")
    if backend == "gloo":
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # IMPOP_BENCH_FORCE_DIST=1: build the process group even for one rank, so that the RCCL
    # communicator + all_gather_into_tensor path can be exercised on a one-GPU box
    use_dist = world > 1 or os.environ.get("IMPOP_BENCH_FORCE_DIST") == "1"
    if use_dist:
        # The rendezvous comes from the launcher (torch.distributed.run exports MASTER_ADDR / MASTER_PORT).  Only the
        # one-rank rehearsal without a launcher has to invent one: a port the kernel hands out, never a fixed
        # number two jobs on one node could share.
        if "MASTER_PORT" not in os.environ:
            if world > 1:
                log("bench.py: WORLD_SIZE > 1 but MASTER_PORT is not set; launch with torch.distributed.run")
                sys.exit(2)
            import socket
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # RCCL prints a version banner to STDOUT when the communicator is built; stdout belongs to the ONE JSON line, so the
        # file descriptor points at stderr until the first collective has gone through
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
        try:
            if backend == "nccl":
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
            else:
                dist.init_process_group("gloo", rank=rank, world_size=world)
            dist.barrier()
            if backend == "nccl":
                torch.cuda.synchronize(dev)
        finally:
            sys.stdout.flush()
            os.dup2(saved_stdout, 1)
            os.close(saved_stdout)

    # ONE explicit stream carries everything: the scan kernels (the engine context is built on it), the
    # collective's pre-event (RCCL waits for the work that is on the CURRENT stream when the collective is
    # issued) and work.wait() (makes the CURRENT stream wait for the collective).  torch's default stream has
    # handle 0, which the C ABI reads as "create a private stream" — that would leave scan and gather unordered.
    stream = torch.cuda.Stream(dev)
    assert stream.cuda_stream != 0
    ctx = impop_amd.Context(local_rank, stream=stream.cuda_stream)
    if args.config4:
        run_config4(args, ctx, stream, dev, rank, world, backend, use_dist)
        ctx.close()
        if use_dist:
            dist.destroy_process_group()
        return
    n, W, NW = args.n_hap, args.window, args.n_windows
    n_site = W * NW
    t0 = time.perf_counter()
    bm = ctx.synthetic(n, n_site, seed=20251031 + rank)  # each rank: its own chromosome shard
    ctx.synchronize()
    if rank == 0:
        log(f"[bench] synthetic matrix {n} x {n_site} sites = {bm.device_bytes / 1e9:.2f} GB in HBM "
            f"({bm.bytes_per_site} B/site) generated in {time.perf_counter() - t0:.2f} s")
    windows = impop_amd.fixed_windows(n_site, W)
    in_a = np.zeros(n, np.uint8); in_a[: min(140, n)] = 1
    in_b = np.zeros(n, np.uint8); in_b[min(140, n): min(240, n)] = 1
    plan = bm.plan(windows, None, in_a, in_b, tile_blocks=args.tile_blocks)
    # double-buffered records: the all-gather of step i (RCCL's stream) overlaps the scan of step i+1
    # (our stream); a buffer is rewritten only after its previous gather has been waited for
    with torch.cuda.stream(stream):
        bufs = [torch.empty(NW * 128, dtype=torch.uint8, device=dev) for _ in range(2)]
        gdev = dev if backend == "nccl" else torch.device("cpu")
        gathered = [torch.empty(world * NW * 128, dtype=torch.uint8, device=gdev) for _ in range(2)] if use_dist else None
    works = [None, None]
    counter = [0]
    # exposed gather time: events on OUR stream around work.wait() — the time the scan stream stood still
    # because a gather had not finished (0 when the collective hides behind the next scan)
    wait_events = []
    host_gather_s = [0.0]

    def wait_for(b, timed):
        if works[b] is None:
            return
        if timed and backend == "nccl":
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            works[b].wait()
            e1.record(stream)
            wait_events.append((e0, e1))
        else:
            works[b].wait()
        works[b] = None

    def step(timed=False):
        b = counter[0] & 1
        counter[0] += 1
        with torch.cuda.stream(stream):
            wait_for(b, timed)
            plan.launch(bufs[b].data_ptr())
            if use_dist:
                th = time.perf_counter()
                if backend == "nccl":
                    works[b] = dist.all_gather_into_tensor(gathered[b], bufs[b], async_op=True)
                else:  # rehearsal: .cpu() is ordered on `stream` behind the scan and blocks the host
                    works[b] = dist.all_gather_into_tensor(gathered[b], bufs[b].cpu(), async_op=True)
                host_gather_s[0] += time.perf_counter() - th

    def drain(timed=False):
        with torch.cuda.stream(stream):
            for b in (0, 1):
                wait_for(b, timed)
        torch.cuda.synchronize(dev)

    def check_gathered(b, what):
        """every rank must hold every rank's records after the gather; its own slice is checkable locally"""
        mine = bufs[b].cpu().numpy().tobytes()
        allrec = gathered[b].cpu().numpy().tobytes()
        assert allrec[rank * NW * 128: (rank + 1) * NW * 128] == mine, f"rank {rank}: gathered records differ from local ({what})"
        return mine

    # ---- parity gate before timing (rank 0): GPU records vs the CPU oracle on sampled windows
    step()
    drain()
    recs = np.frombuffer(bufs[0].cpu().numpy().tobytes(), dtype=impop_amd.STATS_DTYPE)
    if use_dist:
        check_gathered(0, "first step")
    cpu, first = None, None
    if rank == 0:
        from oracle import oracle as orc
        orc.build()
        ones = orc.pack_mask(np.ones(n, np.uint8))
        for wi in sorted({0, NW // 2, NW - 1}):
            s0, s1 = int(windows[wi]["site_begin"]), int(windows[wi]["site_end"])
            want = orc.window_sitecount(bm.download(s0, s1), n, 0, s1 - s0, ones, orc.pack_mask(in_a), orc.pack_mask(in_b),
                                        int(windows[wi]["seq_len"]))
            got = recs[wi]
            for k in ("n_sites", "s_all", "s_p", "s_a", "s_b", "sum_p", "sum_a", "sum_b", "sum_ab"):
                assert int(got[k]) == int(want[k]), ("parity gate", wi, k, int(got[k]), int(want[k]))
            for k in ("pi", "pi_site", "pi_a", "pi_b", "pi_xy", "dxy", "da", "fst", "tajima_d"):
                a, b = float(got[k]), float(want[k])
                assert abs(a - b) <= 1e-9 * max(abs(a), abs(b)), ("parity gate", wi, k, a, b)
        log("[bench] parity gate passed (3 windows vs CPU oracle: integers exact, doubles <= 1e-9 rel)")

    # ---- timed region
    for _ in range(args.warmup):
        step()
    drain()
    plan.timing(True)
    host_gather_s[0] = 0.0
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(timed=True)
    drain(timed=True)
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize(dev)
    elapsed = time.perf_counter() - t0
    kern_ms, launches = plan.elapsed()
    plan.timing(False)
    diag = None
    if use_dist:
        # the records gathered by the TIMED steps (both buffers) must be the ones this rank computed
        for b in (0, 1):
            if args.steps > b:
                mine = check_gathered(b, "timed steps")
                assert mine == recs.tobytes(), f"rank {rank}: records of a timed step differ from the gated first step"
        exposed_ms = sum(e0.elapsed_time(e1) for e0, e1 in wait_events)
        mine = torch.tensor([elapsed, kern_ms / max(launches, 1), exposed_ms / max(args.steps, 1),
                             host_gather_s[0] * 1e3 / max(args.steps, 1)], dtype=torch.float64, device=gdev)
        per_rank = torch.empty(world * 4, dtype=torch.float64, device=gdev)
        dist.all_gather_into_tensor(per_rank, mine)
        per_rank = per_rank.cpu().numpy().reshape(world, 4)
        elapsed = float(per_rank[:, 0].max())
        kern_ms = float(per_rank[:, 1].max()) * max(launches, 1)
        diag = {"backend": "rccl" if backend == "nccl" else "gloo-rehearsal",
                "elapsed_s_min": float(per_rank[:, 0].min()), "elapsed_s_max": float(per_rank[:, 0].max()),
                "kernel_ms_avg_min": float(per_rank[:, 1].min()), "kernel_ms_avg_max": float(per_rank[:, 1].max()),
                "kernel_ms_avg_per_rank": [float(x) for x in per_rank[:, 1]],
                "gather_exposed_ms_per_step_max": float(per_rank[:, 2].max()),
                "gather_issue_host_ms_per_step_max": float(per_rank[:, 3].max()),
                "gather_bytes_per_rank": NW * 128, "gathered_records_checked": True}
    else:
        diag = {"backend": None, "elapsed_s_min": elapsed, "elapsed_s_max": elapsed,
                "kernel_ms_avg_min": kern_ms / max(launches, 1), "kernel_ms_avg_max": kern_ms / max(launches, 1),
                "kernel_ms_avg_per_rank": [kern_ms / max(launches, 1)],
                "gather_exposed_ms_per_step_max": 0.0, "gather_issue_host_ms_per_step_max": 0.0,
                "gather_bytes_per_rank": 0, "gathered_records_checked": False}

    secondary = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu, first = cpu_baseline(bm, windows, in_a, in_b)
    if rank == 0 and world == 1 and not args.no_secondary:
        secondary = secondary_points(ctx, bm, windows, in_a, in_b, recs, args.secondary, args.all_pairs_windows)

    if rank == 0:
        total_windows = NW * world * args.steps
        value = total_windows / elapsed
        algo_bytes = n * n_site / 8.0                      # n*W/8 per window x windows per launch
        avg_kern_s = (kern_ms / 1e3) / max(launches, 1)
        achieved = algo_bytes / avg_kern_s / 1e9
        traffic, traffic_src = pmc_traffic(n, W, NW)
        out = {
            "metric": "windows/sec (pi+Fst+D) for 465-hap HPRC, 50 kb windows",
            "value": value, "unit": "windows/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u32", "data": "synthetic",
            "config": {"workload": f"chr2-scale scan: {NW} windows x {W} sites per GPU, {n} haplotypes, pi + Hudson Fst "
                                   f"(A=140 vs B=100) + Tajima's D + S in one pass; BASELINE configs[1]+[2]",
                       "n_hap": n, "window_sites": W, "windows_per_gpu": NW, "bytes_per_site": bm.bytes_per_site,
                       "layout": "SB64 site-blocked wave-interleaved bit matrix", "tiles": plan.n_tiles,
                       "parallelism": f"windows sharded over {world} GPU(s), one all_gather of records per step"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         "kernel": ("scan_tiles_kernel<%d,false>" % ((n + 31) // 32)) if n <= 512 else "scan_tiles_anyn_kernel<false,false>", "kernel_ms_avg": avg_kern_s * 1e3,
                         "algorithmic_bytes_per_launch": algo_bytes, "layout_bytes_per_launch": plan.bytes_streamed,
                         "layout_GBps": plan.bytes_streamed / avg_kern_s / 1e9},
            "cpu_baseline": cpu,
            "ranks": diag,
            "secondary": secondary,
        }
        print(json.dumps(out), flush=True)
    plan.destroy()
    bm.free()
    ctx.close()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
