#!/usr/bin/env python3
"""Batch window driver: replaces the per-window `while read chr start end` loops of
run_pica2_impg.sh:126-190, run_h-fst.sh:155-190 and run_tajd.sh:103-196 with ONE streaming GPU
pass over a resident presence matrix, and prints the same TSV tables (headers
run_pica2_impg.sh:119,122 / run_h-fst.sh:148 / run_tajd.sh:101) so plot_*_trend.R work unchanged.

    impop_scan.py --matrix chr2.npz --bed windows.bed --format tajd -l samples.txt
    impop_scan.py --matrix chr1.npz --bed windows.bed --format hfst -A afr.txt -B eas.txt
    impop_scan.py --matrix chr2.npz --bed windows.bed --format pica2 [-u subset.txt] [-t 0.999 -r 5]

--format pica2 with a threshold < 1 (or -r) needs the all-pairs path (impop_pairwise_scan);
threshold >= 1 without rounding uses the streaming site-count scan.
"""
import argparse
import os
import sys

import numpy as np

import _bootstrap  # noqa: F401
import impop_amd
from impop_amd.matrixio import load_matrix
from impop_amd.popnames import expand_population, read_subset_file


def read_bed(path, contig_filter=None):
    rows = []
    with open(path) as f:
        for line in f:
            p = line.rstrip("\n").split("\t")
            if not p or not p[0] or p[0].startswith("#"):
                continue  # run_tajd.sh:104-106
            if len(p) < 3 or not (p[1].isdigit() and p[2].isdigit()):
                print(f"Warning: Skipping malformed BED entry: {' '.join(p[:3])}", file=sys.stderr)  # run_tajd.sh:108-111
                continue
            s, e = int(p[1]), int(p[2])
            if e - s <= 0:
                print(f"Warning: Skipping non-positive interval length for {p[0]}:{s}-{e}", file=sys.stderr)
                continue
            rows.append((p[0], s, e))
    return rows


def awk_line_count(list_file):
    """SAMPLE_COUNT of run_tajd.sh:83: `awk 'NF && $1 !~ /^#/' list | wc -l` — lines with at least one field whose
    first field does not start with '#'; a name listed twice counts twice."""
    n = 0
    with open(list_file) as f:
        for line in f:
            fields = line.split()
            if fields and not fields[0].startswith("#"):
                n += 1
    return n


def flags_for(list_file, names):
    raw = read_subset_file(list_file)
    members, missing = expand_population(raw, set(names))
    if missing:
        print(f"Warning: {len(missing)} identifiers from {os.path.basename(list_file)} did not match any sequences", file=sys.stderr)
    return np.array([1 if n in members else 0 for n in names], dtype=np.uint8), raw


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--matrix", required=True, help=".npz presence matrix (impop_amd.matrixio)")
    ap.add_argument("--bed", "-b", required=True)
    ap.add_argument("--format", choices=["pica2", "hfst", "tajd", "fst3pi", "all"], default="all",
                    help="fst3pi = the 3 x pi table of run_fst_impg.sh (needs -A and -B, disjoint)")
    ap.add_argument("-A", "--pop-a"); ap.add_argument("-B", "--pop-b")
    ap.add_argument("--panel", nargs="+", metavar="POP.txt", help="hfst: K >= 2 disjoint population lists; every pair "
                    "in ONE pass (replaces run_h_fst_panels.sh); one table per pair, labelled POP_A-vs-POP_B")
    ap.add_argument("-l", "--sample-list", help="tajd: sample list (run_tajd.sh -l); n = its line count")
    ap.add_argument("-u", "--subset", help="pica2: --subset-sequence-list")
    ap.add_argument("-t", "--threshold", type=float, default=None)
    ap.add_argument("-r", "--round-digits", type=int, default=None)
    ap.add_argument("-p", "--region-prefix", default="CHM13#0#")
    ap.add_argument("-o", "--output")
    ap.add_argument("--identity", choices=["match", "dice"], default="match")
    ap.add_argument("--fst-method", choices=["direct", "grouped"], default="direct",
                    help="hfst: grouped = scripts/hudson/hud.py -m grouped at -t (default 0.999), per window on the all-pairs path")
    ap.add_argument("--compact", action="store_true", help="scan from the matrix compacted to its variable sites "
                    "(impop_matrix_compact): identical output, far fewer bytes (and, on the all-pairs path, multiply-adds) per pass")
    ap.add_argument("--device", type=int, default=None, help="default: LOCAL_RANK, else 0")
    ap.add_argument("--devices", type=int, default=1, metavar="N",
                    help="ONE process driving N GPUs through the C ABI (impop_scan_sharded; impop_pairwise_scan_sharded for the "
                         "all-pairs formats): the BED rows are cut into N contiguous ranges, each device holds the slab of sites its "
                         "rows touch, every device works before the first result is fetched; no torch, no launcher.  With fewer "
                         "than N devices the contexts share device 0.  Not with --panel or --compact")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend when launched with WORLD_SIZE > 1 "
                    "(nccl = RCCL over xGMI; gloo for rehearsals)")
    args = ap.parse_args()
    # Multi-GPU: `python -m torch.distributed.run --nproc-per-node N scripts/impop_scan.py ...` — the BED rows
    # are sharded over ranks, each rank uploads only the slab its windows touch, scans it, and ONE
    # all-gather of the fixed-size records brings everything to rank 0, which prints.
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        import torch
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)
    if args.device is None:
        args.device = local_rank if (world == 1 or args.backend == "nccl") else 0

    mf = load_matrix(args.matrix)
    names = mf.names
    bed = read_bed(args.bed)
    wins, regions = [], []
    for chrom, s, e in bed:
        b, en = mf.site_range(s, e)
        wins.append((b, en, e - s))  # seq_len = LENGTH = end - start (run_pica2_impg.sh:133)
        region = f"{chrom}:{s}-{e}" if chrom.startswith(args.region_prefix) else f"{args.region_prefix}{chrom}:{s}-{e}"
        regions.append(region)
    out = (open(args.output, "w") if args.output else sys.stdout) if rank == 0 else open(os.devnull, "w")
    ctx = impop_amd.Context(args.device)
    if args.devices > 1 and (world > 1 or args.panel or args.compact):
        print("Error: --devices N is the one-process form: not under torch.distributed.run, not with --panel / --compact", file=sys.stderr)
        sys.exit(2)
    grouped_fst = args.format == "hfst" and args.fst_method == "grouped"
    need_pairwise = grouped_fst or (args.format == "pica2" and ((args.threshold is not None and args.threshold < 1.0)
                                                                 or args.round_digits is not None or args.identity != "match"))
    multi_dev = args.devices > 1
    all_wins = impop_amd.make_windows(wins)
    if multi_dev:
        bm = None  # every device gets only its slab, further down
    elif world > 1:
        from impop_amd.distributed import shard_windows
        loc, s0, s1, _ = shard_windows(all_wins, world, rank)
        w0, w1 = s0 // 64, (s1 + 63) // 64  # slab = whole 64-bit words of the hap-major rows
        slab = np.ascontiguousarray(mf.bits[:, w0:w1]) if len(loc) else np.zeros((mf.n_hap, 1), np.uint64)
        shift = s0 - 64 * w0
        loc = loc.copy()
        loc["site_begin"] += np.uint64(shift)
        loc["site_end"] += np.uint64(shift)
        n_slab = max(min(mf.n_site, 64 * w1) - 64 * w0, 0)
        bm = ctx.upload(slab, n_slab, keep_hap_major=need_pairwise)
        if mf.site_weight is not None:
            bm.set_site_weights(mf.site_weight[64 * w0: 64 * w0 + n_slab])
        wins = [(int(w["site_begin"]), int(w["site_end"]), int(w["seq_len"])) for w in loc]
    else:
        bm = ctx.upload(mf.bits, mf.n_site, keep_hap_major=need_pairwise)
        if mf.site_weight is not None:
            bm.set_site_weights(mf.site_weight)
    if args.compact:
        full = bm
        bm = full.compact()
        full.free()
    mask_p = mask_a = mask_b = None
    sample_count = mf.n_hap
    if args.sample_list:
        mask_p, raw = flags_for(args.sample_list, names)
        sample_count = awk_line_count(args.sample_list)  # run_tajd.sh:83
        if args.format in ("tajd", "all") and sample_count < 2:
            print(f"Error: Need at least two samples to compute Tajima's D (found {sample_count})", file=sys.stderr)  # :84-87
            sys.exit(1)
    if args.subset:
        mask_p, _ = flags_for(args.subset, names)
    if args.pop_a and args.pop_b:
        mask_a, _ = flags_for(args.pop_a, names)
        mask_b, _ = flags_for(args.pop_b, names)
        if not mask_a.any() or not mask_b.any():
            print("Error: No valid sequences found in one or both populations", file=sys.stderr)  # h-fst.py:319-321
            sys.exit(1)
    if args.panel:
        labels = [os.path.splitext(os.path.basename(f))[0] for f in args.panel]
        pops = [flags_for(f, names)[0] for f in args.panel]
        pr = bm.scan_multi(wins, pops)
        if world > 1:  # one all-gather of [window, pair] records, a window's pairs travelling as one item
            from impop_amd.distributed import gather_records
            import torch
            n_pairs = pr.shape[1]
            item = np.dtype((np.void, n_pairs * pr.dtype.itemsize))
            flat = np.ascontiguousarray(pr).reshape(-1).view(item) if len(pr) else np.zeros(0, dtype=item)
            dev = torch.device("cuda", local_rank) if args.backend == "nccl" else None
            full = gather_records(flat, len(all_wins), world, rank, dev)
            pr = full.view(pr.dtype).reshape(len(all_wins), n_pairs)
            wins = [(int(w["site_begin"]), int(w["site_end"]), int(w["seq_len"])) for w in all_wins]
        p = 0
        for k in range(len(pops)):
            for l in range(k + 1, len(pops)):
                print(f"# {labels[k]}-vs-{labels[l]}", file=out)
                print("REGION\tLENGTH\tFST\tPI_A\tPI_B\tPI_XY\tDXY\tDA", file=out)
                for reg, (b, e, L), r in zip(regions, wins, pr[:, p]):
                    print(f"{reg}\t{L}\t{float(r['fst']):.8f}\t{float(r['pi_a']):.8f}\t{float(r['pi_b']):.8f}\t"
                          f"{float(r['pi_xy']):.8f}\t{float(r['dxy']):.8f}\t{float(r['da']):.8f}", file=out)
                p += 1
        if args.output or rank != 0:
            out.close()
        bm.free()
        ctx.close()
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
            dist.destroy_process_group()
        return
    # CLI defaults of the reference: pica2.py:175 (-t 0.99), hud.py -t 0.999
    thr = (0.999 if grouped_fst else 0.99) if args.threshold is None else args.threshold
    if need_pairwise and not multi_dev:
        # the pica2 / hfst tables print neither S nor D: s_scope 2 skips the site scan of the all-pairs path
        res = bm.pairwise_scan(wins, mask_p, mask_a, mask_b, kind=args.identity, threshold=thr, round_digits=args.round_digits,
                               s_scope=2, fst_method=args.fst_method if grouped_fst else "direct")
    elif multi_dev:
        # one process, several devices: a context + the slab of its BED rows per device (impop_shard_windows says which)
        from impop_amd import engine
        import ctypes as C
        n_dev = C.c_int(0)
        impop_amd._lib.load().impop_device_count(C.byref(n_dev))
        ctxs, slabs, begins = [], [], []
        for k in range(args.devices):
            first, cnt, s0, s1 = engine.shard_windows_c(all_wins, args.devices, k)
            w0, w1 = s0 // 64, max((s1 + 63) // 64, s0 // 64 + 1)
            ck = impop_amd.Context(k if n_dev.value >= args.devices else 0)
            n_slab = max(min(mf.n_site, 64 * w1) - 64 * w0, 0)
            sk = ck.upload(np.ascontiguousarray(mf.bits[:, w0:w1]), n_slab, keep_hap_major=need_pairwise)
            if mf.site_weight is not None:
                sk.set_site_weights(mf.site_weight[64 * w0: 64 * w0 + n_slab])
            ctxs.append(ck); slabs.append(sk); begins.append(64 * w0)
        if need_pairwise:  # impop_pairwise_scan_sharded: every device contracts its own windows, a host thread each
            res = engine.pairwise_scan_sharded(slabs, begins, all_wins, mask_p, mask_a, mask_b, kind=args.identity, threshold=thr,
                                               round_digits=args.round_digits, s_scope=2,
                                               fst_method=args.fst_method if grouped_fst else "direct")
        else:
            res = engine.scan_sharded(slabs, begins, all_wins, mask_p, mask_a, mask_b)
        for sk, ck in zip(slabs, ctxs):
            sk.free(); ck.close()
    else:
        res = bm.scan(wins, mask_p, mask_a, mask_b)
    if world > 1:
        from impop_amd.distributed import gather_records
        import torch
        dev = torch.device("cuda", local_rank) if args.backend == "nccl" else None
        res = gather_records(res, len(all_wins), world, rank, dev)
    wins = [(int(w["site_begin"]), int(w["site_end"]), int(w["seq_len"])) for w in all_wins]
    fmt = args.format
    thr_txt = "1.0" if args.threshold is None and not need_pairwise else str(args.threshold if args.threshold is not None else 0.99)
    r_txt = "" if args.round_digits is None else str(args.round_digits)
    if fmt in ("pica2", "all"):
        if args.subset:
            print("REGION\tSUBSET\tLENGTH\tTHRESHOLD\tR_VALUE\tPICA_OUTPUT", file=out)
        else:
            print("REGION\tLENGTH\tTHRESHOLD\tR_VALUE\tPICA_OUTPUT", file=out)
        for reg, (b, e, L), r in zip(regions, wins, res):
            cell = f"{float(r['pi_site']):.8f} (sequence length: {L})"  # pica2.py:226
            if args.subset:
                print(f"{reg}\t{os.path.basename(args.subset)}\t{L}\t{thr_txt}\t{r_txt}\t{cell}", file=out)
            else:
                print(f"{reg}\t{L}\t{thr_txt}\t{r_txt}\t{cell}", file=out)
    if fmt in ("hfst", "all") and mask_a is not None:
        print("REGION\tLENGTH\tFST\tPI_A\tPI_B\tPI_XY\tDXY\tDA", file=out)
        for reg, (b, e, L), r in zip(regions, wins, res):
            print(f"{reg}\t{L}\t{float(r['fst']):.8f}\t{float(r['pi_a']):.8f}\t{float(r['pi_b']):.8f}\t"
                  f"{float(r['pi_xy']):.8f}\t{float(r['dxy']):.8f}\t{float(r['da']):.8f}", file=out)
    if fmt == "fst3pi":
        from impop_amd.drivers import fst_3pi_fields, pi_union_site
        if mask_a is None:
            print("Error: --format fst3pi needs -A and -B", file=sys.stderr)
            sys.exit(2)
        if (mask_a & mask_b).any():
            print("Error: --format fst3pi needs disjoint populations", file=sys.stderr)
            sys.exit(2)
        nA, nB = int(mask_a.sum()), int(mask_b.sum())
        print("REGION\tLENGTH\tTHRESHOLD\tR_VALUE\tPI_A\tPI_B\tPI_C\tPI_AB_AVG\tFST", file=out)  # run_fst_impg.sh:158
        for reg, (b, e, L), r in zip(regions, wins, res):
            ta, tb, tc, avg, fst = fst_3pi_fields(float(r["pi_a"]), float(r["pi_b"]), pi_union_site(r, nA, nB, L))
            print(f"{reg}\t{L}\t{thr_txt}\t{r_txt}\t{ta}\t{tb}\t{tc}\t{avg}\t{fst}", file=out)
    if fmt in ("tajd", "all"):
        print("REGION\tLENGTH\tSAMPLES\tSEGREGATING_SITES\tPI\tTAJIMAS_D", file=out)
        d_col = res["tajima_d"]
        n_matched = int(mask_p.sum()) if mask_p is not None else mf.n_hap
        if sample_count != n_matched and len(res):
            # run_tajd.sh:180 hands tj_d.py `-n SAMPLE_COUNT`, the list's LINE count, whatever the number of
            # haplotypes those lines select (a bare sample name selects two, an unknown name none, a repeated
            # line counts twice).  The scan evaluated D with n = matched haplotypes: redo D (on the GPU,
            # impop_tajimas_d) with the reference's n, pi through the same "%.8f" text (run_tajd.sh:174) and S.
            print(f"Warning: sample list has {sample_count} lines but selects {n_matched} haplotypes; Tajima's D uses "
                  f"n = {sample_count} like run_tajd.sh", file=sys.stderr)
            pi_txt = np.array([float(f"{float(x):.8f}") for x in res["pi_site"]])
            d_col = ctx.tajimas_d(np.full(len(res), sample_count, dtype=np.int64), res["s_all"].astype(np.float64), pi_txt)
        for reg, (b, e, L), r, D in zip(regions, wins, res, d_col):
            D = float(D)
            taj = "NA" if D != D else repr(D)  # run_tajd.sh:192-194
            print(f"{reg}\t{L}\t{sample_count}\t{int(r['s_all'])}\t{float(r['pi_site']):.8f}\t{taj}", file=out)
    if args.output or rank != 0:
        out.close()
    if bm is not None:
        bm.free()
    ctx.close()
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
