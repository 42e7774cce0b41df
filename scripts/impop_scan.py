#!/usr/bin/env python3
"""Batch window driver: replaces the per-window `while read chr start end` loops of
run_pica2_impg.sh:126-190, run_h-fst.sh:155-190, run_tajd.sh:103-196 and run_fst_impg.sh:160-221 with GPU passes
over resident presence matrices, and prints the same TSV tables (headers run_pica2_impg.sh:119,122 /
run_h-fst.sh:148 / run_tajd.sh:101 / run_fst_impg.sh:158) so plot_*_trend.R work unchanged.

    impop_scan.py --matrix chr2.npz --bed windows.bed --format tajd -l samples.txt            # -t 0.999 -r 5 like run_tajd.sh
    impop_scan.py --matrix chr1.npz --bed windows.bed --format hfst -A afr.txt -B eas.txt [-r 5]
    impop_scan.py --matrix chr2.npz --bed windows.bed --format pica2 -t 0.999 -r 5 [-u subset.txt]
    impop_scan.py --matrix chr1.npz chr2.npz ... --bed genome.bed --format all ...            # one matrix per chromosome

What -t / -r mean, per format (the THRESHOLD / R_VALUE columns always print what was computed):
  tajd    pica2's threshold / rounding behind the PI column and Tajima's D.  Defaults 0.999 and 5 — run_tajd.sh:9-10.
  pica2   pica2.py -t / -r (run_pica2_impg.sh requires both; here the default is 1.0 / no rounding).
  fst3pi  the three pica2 runs of run_fst_impg.sh:73 (same defaults as pica2).
  hfst    -r is h-fst.py -r (run_h-fst.sh:76-78); -t only with --fst-method grouped (hud.py -t, default 0.999).
  all     -t / -r as for tajd (same defaults) for the pica2 and tajd tables; the h-fst table rounds with --fst-round-digits.
  `-r none` switches rounding off where a default would apply.

A threshold >= 1 without rounding on the `match` identity is the streaming site-count scan (every haplotype its own
group: exact integer identities); anything else runs the all-pairs path (impop_pairwise_scan).
BED rows are matched to matrices by chromosome (run_pica2_impg.sh:139-151 builds REGION from each row's own chromosome):
a row whose chromosome no matrix holds is skipped with a warning.
"""
import argparse
import os
import sys

import numpy as np

import _bootstrap  # noqa: F401
import impop_amd
from impop_amd.matrixio import load_matrix
from impop_amd.popnames import expand_population, read_subset_file


def read_bed(path, fmt="tajd"):
    """BED rows -> [(chrom, start, end)].  Comment / empty rows are skipped silently, unusable rows with the warning the
    reference driver of that table prints on stderr: run_tajd.sh:104-117 (tajd, all), run_h-fst.sh:159-181 (hfst),
    run_fst_impg.sh:166-179 (fst3pi), run_pica2_impg.sh:128-136 (pica2; it validates only the length)."""
    rows = []
    with open(path) as f:
        for line_no, line in enumerate(f, 1):
            p = line.rstrip("\n").split("\t")
            if not p or not p[0] or p[0].startswith("#"):
                continue
            chrom = p[0]
            start, end = (p[1] if len(p) > 1 else ""), (p[2] if len(p) > 2 else "")
            numeric = start.isdigit() and end.isdigit()
            if fmt == "hfst":
                if not start or not end:
                    print(f"Warning: Incomplete BED entry at line {line_no}, skipping", file=sys.stderr)
                    continue
                if not numeric:
                    print(f"Warning: Non-integer coordinates at line {line_no}: {chrom}:{start}-{end}, skipping", file=sys.stderr)
                    continue
                if int(start) >= int(end):
                    print(f"Warning: Invalid interval at line {line_no}: {chrom}:{start}-{end}, skipping", file=sys.stderr)
                    continue
            elif fmt == "fst3pi":
                if not start or not end:
                    print(f"Warning: Incomplete BED entry for chromosome {chrom}, skipping", file=sys.stderr)
                    continue
                if not numeric:  # the driver's plain `echo` prints the backslash-t sequences literally
                    print(f"Warning: Non-integer coordinates in BED entry {chrom}\\t{start}\\t{end}, skipping", file=sys.stderr)
                    continue
                if int(end) - int(start) <= 0:
                    print(f"Warning: Non-positive interval length for {chrom}:{start}-{end}, skipping", file=sys.stderr)
                    continue
            else:
                if not numeric:
                    print(f"Warning: Skipping malformed BED entry: {chrom} {start} {end}", file=sys.stderr)  # run_tajd.sh:108-111
                    continue
                if int(end) - int(start) <= 0:
                    if fmt == "pica2":
                        print(f"Warning: Skipping region with non-positive length: {chrom}:{start}-{end}", file=sys.stderr)
                    else:
                        print(f"Warning: Skipping non-positive interval length for {chrom}:{start}-{end}", file=sys.stderr)
                    continue
            rows.append((chrom, int(start), int(end)))
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
    return np.array([1 if n in members else 0 for n in names], dtype=np.uint8)


def round_arg(v):
    if v.lower() in ("none", "off", ""):
        return "none"
    r = int(v)
    if r < 0:
        raise argparse.ArgumentTypeError("round digits must be >= 0 (or `none`)")
    return r


def threshold_arg(v):
    """-t keeps the user's TEXT next to the number: the bash drivers print "${THRESHOLD}" as typed (run_pica2_impg.sh:185-187,
    run_fst_impg.sh:220), so `-t 0.9990` must come back as 0.9990 in the THRESHOLD column"""
    x = float(v)
    if not (x == x):
        raise argparse.ArgumentTypeError("threshold must be a number")
    return (x, v)


class Runner:
    """One matrix on the device(s) + the three ways a row set can be scanned: one device, one process driving several
    contexts (--devices N, impop_*_sharded) or one rank of a torch.distributed job (records all-gathered once per scan)."""

    def __init__(self, args, mf, windows, need_pairs, rank, world, local_rank):
        self.args, self.mf, self.rank, self.world, self.local_rank = args, mf, rank, world, local_rank
        self.all_wins = impop_amd.make_windows(windows)
        self.n_total = len(self.all_wins)
        self.ctx = impop_amd.Context(args.device)
        self.bm, self.slabs, self.ctxs, self.begins = None, [], [], []
        self.local_wins = self.all_wins
        if args.devices > 1:
            from impop_amd import engine
            import ctypes as C
            n_dev = C.c_int(0)
            impop_amd._lib.load().impop_device_count(C.byref(n_dev))
            for k in range(args.devices):
                first, cnt, s0, s1 = engine.shard_windows_c(self.all_wins, args.devices, k)
                w0, w1 = s0 // 64, max((s1 + 63) // 64, s0 // 64 + 1)
                ck = impop_amd.Context(k if n_dev.value >= args.devices else 0)
                n_slab = max(min(mf.n_site, 64 * w1) - 64 * w0, 0)
                sk = ck.upload(np.ascontiguousarray(mf.bits[:, w0:w1]), n_slab, keep_hap_major=need_pairs)
                if mf.site_weight is not None:
                    sk.set_site_weights(mf.site_weight[64 * w0: 64 * w0 + n_slab])
                self.ctxs.append(ck); self.slabs.append(sk); self.begins.append(64 * w0)
        elif world > 1:
            from impop_amd.distributed import shard_windows
            loc, s0, s1, _ = shard_windows(self.all_wins, world, rank)
            w0, w1 = s0 // 64, (s1 + 63) // 64  # slab = whole 64-bit words of the hap-major rows
            slab = np.ascontiguousarray(mf.bits[:, w0:w1]) if len(loc) else np.zeros((mf.n_hap, 1), np.uint64)
            shift = s0 - 64 * w0
            loc = loc.copy()
            loc["site_begin"] += np.uint64(shift)
            loc["site_end"] += np.uint64(shift)
            n_slab = max(min(mf.n_site, 64 * w1) - 64 * w0, 0)
            self.bm = self.ctx.upload(slab, n_slab, keep_hap_major=need_pairs)
            if mf.site_weight is not None:
                self.bm.set_site_weights(mf.site_weight[64 * w0: 64 * w0 + n_slab])
            self.local_wins = loc
        else:
            self.bm = self.ctx.upload(mf.bits, mf.n_site, keep_hap_major=need_pairs)
            if mf.site_weight is not None:
                self.bm.set_site_weights(mf.site_weight)
        if args.compact:
            full = self.bm
            self.bm = full.compact()
            full.free()

    def _gather(self, res):
        if self.world > 1:
            from impop_amd.distributed import gather_records
            import torch
            dev = torch.device("cuda", self.local_rank) if self.args.backend == "nccl" else None
            res = gather_records(res, self.n_total, self.world, self.rank, dev)
        return res

    def stream(self, mask_p, mask_a, mask_b):
        """the streaming site-count scan: pica2 at threshold >= 1 unrounded, h-fst unrounded, S, D (STATS records)"""
        if self.slabs:
            from impop_amd import engine
            return engine.scan_sharded(self.slabs, self.begins, self.all_wins, mask_p, mask_a, mask_b)
        return self._gather(self.bm.scan(self.local_wins, mask_p, mask_a, mask_b))

    def pairs(self, mask_p, mask_a, mask_b, threshold, round_digits, want_s, fst_method="direct"):
        """the all-pairs path: thresholded / rounded pica2, rounded or grouped Fst (PAIRWISE records).  want_s False skips
        the S / D part (s_scope 2)."""
        kw = dict(kind=self.args.identity, threshold=threshold, round_digits=round_digits, s_scope=0 if want_s else 2,
                  fst_method=fst_method)
        if self.slabs:
            from impop_amd import engine
            return engine.pairwise_scan_sharded(self.slabs, self.begins, self.all_wins, mask_p, mask_a, mask_b, **kw)
        return self._gather(self.bm.pairwise_scan(self.local_wins, mask_p, mask_a, mask_b, **kw))

    def panel(self, pops):
        pr = self.bm.scan_multi(self.local_wins, pops)
        if self.world > 1:  # one all-gather of [window, pair] records, a window's pairs travelling as one item
            from impop_amd.distributed import gather_records
            import torch
            n_pairs = pr.shape[1]
            item = np.dtype((np.void, n_pairs * pr.dtype.itemsize))
            flat = np.ascontiguousarray(pr).reshape(-1).view(item) if len(pr) else np.zeros(0, dtype=item)
            dev = torch.device("cuda", self.local_rank) if self.args.backend == "nccl" else None
            full = gather_records(flat, self.n_total, self.world, self.rank, dev)
            pr = full.view(pr.dtype).reshape(self.n_total, n_pairs)
        return pr

    def close(self):
        for sk, ck in zip(self.slabs, self.ctxs):
            sk.free(); ck.close()
        if self.bm is not None:
            self.bm.free()
        self.ctx.close()


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--matrix", required=True, nargs="+", help=".npz presence matrices (impop_amd.matrixio), one per chromosome")
    ap.add_argument("--bed", "-b", required=True)
    ap.add_argument("--format", choices=["pica2", "hfst", "tajd", "fst3pi", "all"], default="all",
                    help="fst3pi = the 3 x pi table of run_fst_impg.sh (needs -A and -B, disjoint)")
    ap.add_argument("-A", "--pop-a"); ap.add_argument("-B", "--pop-b")
    ap.add_argument("--panel", nargs="+", metavar="POP.txt", help="hfst: K >= 2 disjoint population lists; every pair "
                    "in ONE pass (replaces run_h_fst_panels.sh); one table per pair, labelled POP_A-vs-POP_B")
    ap.add_argument("-l", "--sample-list", help="tajd: sample list (run_tajd.sh -l); n = its line count")
    ap.add_argument("-u", "--subset", help="pica2: --subset-sequence-list")
    ap.add_argument("--sequence-length", type=int, default=None, metavar="L",
                    help="pica2: run_pica2_impg.sh -l — the length handed to pica2 AND printed in the LENGTH column, instead of end - start")
    ap.add_argument("-t", "--threshold", type=threshold_arg, default=None, help="see the table above")
    ap.add_argument("-r", "--round-digits", type=round_arg, default=None, help="an integer, or `none`; see the table above")
    ap.add_argument("--fst-round-digits", type=round_arg, default=None, help="--format all: h-fst.py -r for the h-fst table")
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
    threshold_text = None
    if args.threshold is not None:
        args.threshold, threshold_text = args.threshold
    # Multi-GPU: `python -m torch.distributed.run --nproc-per-node N scripts/impop_scan.py ...` — the BED rows
    # are sharded over ranks, each rank uploads only the slab its windows touch, scans it, and ONE
    # all-gather of the fixed-size records per scan brings everything to rank 0, which prints.
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
    if args.devices > 1 and (world > 1 or args.panel or args.compact):
        print("Error: --devices N is the one-process form: not under torch.distributed.run, not with --panel / --compact", file=sys.stderr)
        sys.exit(2)
    fmt = args.format
    grouped_fst = fmt == "hfst" and args.fst_method == "grouped"

    # ---- what is computed: (threshold, round digits) behind the pica2-derived columns, round digits of the h-fst table
    given_r = args.round_digits
    if fmt in ("tajd", "all"):  # run_tajd.sh:9-10
        pica_t = 0.999 if args.threshold is None else args.threshold
        pica_r = 5 if given_r is None else (None if given_r == "none" else given_r)
    else:
        pica_t = 1.0 if args.threshold is None else args.threshold
        pica_r = None if given_r in (None, "none") else given_r
    if fmt == "hfst":
        fst_r = None if given_r in (None, "none") else given_r
        if args.threshold is not None and not grouped_fst:
            print("Error: --format hfst takes -t only with --fst-method grouped (h-fst.py has no threshold)", file=sys.stderr)
            sys.exit(2)
    else:
        fst_r = None if args.fst_round_digits in (None, "none") else args.fst_round_digits
    if args.fst_round_digits is not None and fmt not in ("all",):
        print("Error: --fst-round-digits belongs to --format all (use -r with --format hfst)", file=sys.stderr)
        sys.exit(2)
    grouped_t = 0.999 if args.threshold is None else args.threshold  # hud.py -t
    pica_pairs = not (pica_t >= 1.0 and pica_r is None and args.identity == "match")
    fst_pairs = grouped_fst or fst_r is not None or args.identity != "match"
    want_pica = fmt in ("pica2", "tajd", "all", "fst3pi")
    want_fst = fmt in ("hfst", "all") and not args.panel
    need_pairs = (want_pica and pica_pairs) or (want_fst and fst_pairs)
    if args.panel and (fmt != "hfst" or fst_pairs):
        print("Error: --panel is the streaming K-population scan of --format hfst (direct method, match identity, no rounding)", file=sys.stderr)
        sys.exit(2)

    # ---- matrices by chromosome, BED rows to their matrix
    mats = [load_matrix(p) for p in args.matrix]
    def full_name(chrom):
        return chrom if chrom.startswith(args.region_prefix) else args.region_prefix + chrom
    by_contig = {}
    for p, mf in zip(args.matrix, mats):
        key = full_name(mf.contig) if mf.contig else ""
        if key in by_contig:
            print(f"Error: two matrices for contig '{mf.contig}' ({p})", file=sys.stderr)
            sys.exit(2)
        by_contig[key] = mf
    if "" in by_contig and len(mats) > 1:
        print("Error: several --matrix files need a contig name each (matrixio `contig`)", file=sys.stderr)
        sys.exit(2)
    bed = read_bed(args.bed, fmt)
    rows, per_mat = [], {}
    for chrom, s, e in bed:
        region = f"{full_name(chrom)}:{s}-{e}"
        key = "" if "" in by_contig else full_name(chrom)
        if key not in by_contig:
            print(f"Warning: Skipping region {region}: no matrix holds chromosome {chrom}", file=sys.stderr)
            continue
        per_mat.setdefault(key, []).append(len(rows))
        rows.append((region, key, s, e))
    n_rows = len(rows)
    L_col = np.array([e - s for _, _, s, e in rows], dtype=np.int64)  # LENGTH = end - start (run_pica2_impg.sh:133)
    if args.sequence_length is not None:
        if fmt != "pica2" or args.sequence_length <= 0:
            print("Error: --sequence-length (a positive integer) belongs to --format pica2 (run_pica2_impg.sh -l)", file=sys.stderr)
            sys.exit(2)
        L_col[:] = args.sequence_length  # EFFECTIVE_LENGTH, run_pica2_impg.sh:153-157
    out = (open(args.output, "w") if args.output else sys.stdout) if rank == 0 else open(os.devnull, "w")

    sample_count = None
    if args.sample_list:
        sample_count = awk_line_count(args.sample_list)  # run_tajd.sh:83
        if fmt in ("tajd", "all") and sample_count < 2:
            print(f"Error: Need at least two samples to compute Tajima's D (found {sample_count})", file=sys.stderr)  # :84-87
            sys.exit(1)

    f64 = lambda: np.full(n_rows, np.nan)  # noqa: E731
    col = {k: f64() for k in ("pi_site", "tajima_d", "fst", "pi_a", "pi_b", "pi_xy", "dxy", "da", "pi3_a", "pi3_b", "pi3_c")}
    s_all = np.zeros(n_rows, dtype=np.int64)
    panel_tables, panel_labels = None, None
    samples_col = 0
    for key, idx in per_mat.items():
        mf = by_contig[key]
        names = mf.names
        idx = np.array(idx)
        wins = []
        for i in idx:
            _, _, s, e = rows[i]
            b, en = mf.site_range(s, e)
            wins.append((b, en, int(L_col[i])))
        run = Runner(args, mf, wins, need_pairs, rank, world, local_rank)
        mask_p = mask_a = mask_b = None
        n_matched = mf.n_hap
        if args.sample_list:
            mask_p = flags_for(args.sample_list, names)
        if args.subset:
            mask_p = flags_for(args.subset, names)
        if mask_p is not None:
            n_matched = int(mask_p.sum())
        if args.pop_a and args.pop_b:
            mask_a, mask_b = flags_for(args.pop_a, names), flags_for(args.pop_b, names)
            if not mask_a.any() or not mask_b.any():
                print("Error: No valid sequences found in one or both populations", file=sys.stderr)  # h-fst.py:319-321
                sys.exit(1)
        if args.panel:
            panel_labels = [os.path.splitext(os.path.basename(f))[0] for f in args.panel]
            pr = run.panel([flags_for(f, names) for f in args.panel])
            if panel_tables is None:
                panel_tables = np.zeros((n_rows, pr.shape[1]), dtype=pr.dtype)
            panel_tables[idx] = pr
            run.close()
            continue
        if fmt == "fst3pi":
            if mask_a is None:
                print("Error: --format fst3pi needs -A and -B", file=sys.stderr)
                sys.exit(2)
            if (mask_a & mask_b).any():
                print("Error: --format fst3pi needs disjoint populations", file=sys.stderr)
                sys.exit(2)
            if pica_pairs:  # run_fst_impg.sh:73: pica2.py -t T -r R on the lists A, B and A u B
                for name, sel in (("pi3_a", mask_a), ("pi3_b", mask_b), ("pi3_c", mask_a | mask_b)):
                    col[name][idx] = run.pairs(sel, None, None, pica_t, pica_r, False)["pi_site"]
            else:
                from impop_amd.drivers import pi_union_site
                res = run.stream(None, mask_a, mask_b)
                nA, nB = int(mask_a.sum()), int(mask_b.sum())
                col["pi3_a"][idx], col["pi3_b"][idx] = res["pi_a"], res["pi_b"]
                col["pi3_c"][idx] = [pi_union_site(r, nA, nB, int(L)) for r, L in zip(res, L_col[idx])]
            run.close()
            continue
        want_s = fmt in ("tajd", "all")
        one_call = want_pica and want_fst and pica_pairs and fst_pairs and pica_r == fst_r and not grouped_fst
        pica_rec = fst_rec = None
        if want_pica:
            pica_rec = run.pairs(mask_p, mask_a if one_call else None, mask_b if one_call else None, pica_t, pica_r, want_s) \
                if pica_pairs else run.stream(mask_p, mask_a, mask_b)
            if one_call or (want_fst and not pica_pairs and not fst_pairs):
                fst_rec = pica_rec
        if want_fst and fst_rec is None and mask_a is not None:
            fst_rec = run.pairs(None, mask_a, mask_b, grouped_t, fst_r, False, args.fst_method if grouped_fst else "direct") \
                if fst_pairs else run.stream(None, mask_a, mask_b)
        if pica_rec is not None:
            col["pi_site"][idx] = pica_rec["pi_site"]
            if want_s:
                s_all[idx] = pica_rec["s_all"]
                D = pica_rec["tajima_d"]
                samples_col = sample_count if sample_count is not None else mf.n_hap
                if sample_count is not None and sample_count != n_matched and len(idx):
                    # run_tajd.sh:180 hands tj_d.py `-n SAMPLE_COUNT`, the list's LINE count, whatever the number of
                    # haplotypes those lines select (a bare sample name selects two, an unknown name none, a repeated
                    # line counts twice).  The scan evaluated D with n = matched haplotypes: redo D (on the GPU,
                    # impop_tajimas_d) with the reference's n, pi through the same "%.8f" text (run_tajd.sh:174) and S.
                    print(f"Warning: sample list has {sample_count} lines but selects {n_matched} haplotypes; Tajima's D uses "
                          f"n = {sample_count} like run_tajd.sh", file=sys.stderr)
                    pi_txt = np.array([float(f"{float(x):.8f}") for x in pica_rec["pi_site"]])
                    D = run.ctx.tajimas_d(np.full(len(idx), sample_count, dtype=np.int64), pica_rec["s_all"].astype(np.float64), pi_txt)
                col["tajima_d"][idx] = D
        if fst_rec is not None:
            for k in ("fst", "pi_a", "pi_b", "pi_xy", "dxy", "da"):
                col[k][idx] = fst_rec[k]
        run.close()

    regions = [r[0] for r in rows]
    thr_txt = threshold_text if threshold_text is not None else repr(float(pica_t))  # as typed, like "${THRESHOLD}" in the drivers
    r_txt = "" if pica_r is None else str(pica_r)
    if args.panel:
        p = 0
        K = len(args.panel)
        for k in range(K):
            for l in range(k + 1, K):
                print(f"# {panel_labels[k]}-vs-{panel_labels[l]}", file=out)
                print("REGION\tLENGTH\tFST\tPI_A\tPI_B\tPI_XY\tDXY\tDA", file=out)
                for i, reg in enumerate(regions):
                    r = panel_tables[i, p]
                    print(f"{reg}\t{L_col[i]}\t{float(r['fst']):.8f}\t{float(r['pi_a']):.8f}\t{float(r['pi_b']):.8f}\t"
                          f"{float(r['pi_xy']):.8f}\t{float(r['dxy']):.8f}\t{float(r['da']):.8f}", file=out)
                p += 1
    if fmt in ("pica2", "all"):
        if args.subset:
            print("REGION\tSUBSET\tLENGTH\tTHRESHOLD\tR_VALUE\tPICA_OUTPUT", file=out)
        else:
            print("REGION\tLENGTH\tTHRESHOLD\tR_VALUE\tPICA_OUTPUT", file=out)
        for i, reg in enumerate(regions):
            cell = f"{col['pi_site'][i]:.8f} (sequence length: {L_col[i]})"  # pica2.py:226
            if args.subset:
                print(f"{reg}\t{os.path.basename(args.subset)}\t{L_col[i]}\t{thr_txt}\t{r_txt}\t{cell}", file=out)
            else:
                print(f"{reg}\t{L_col[i]}\t{thr_txt}\t{r_txt}\t{cell}", file=out)
    if fmt in ("hfst", "all") and args.pop_a and args.pop_b and not args.panel:
        print("REGION\tLENGTH\tFST\tPI_A\tPI_B\tPI_XY\tDXY\tDA", file=out)
        for i, reg in enumerate(regions):
            print(f"{reg}\t{L_col[i]}\t{col['fst'][i]:.8f}\t{col['pi_a'][i]:.8f}\t{col['pi_b'][i]:.8f}\t"
                  f"{col['pi_xy'][i]:.8f}\t{col['dxy'][i]:.8f}\t{col['da'][i]:.8f}", file=out)
    if fmt == "fst3pi":
        from impop_amd.drivers import fst_3pi_fields
        print("REGION\tLENGTH\tTHRESHOLD\tR_VALUE\tPI_A\tPI_B\tPI_C\tPI_AB_AVG\tFST", file=out)  # run_fst_impg.sh:158
        for i, reg in enumerate(regions):
            ta, tb, tc, avg, fst = fst_3pi_fields(float(col["pi3_a"][i]), float(col["pi3_b"][i]), float(col["pi3_c"][i]))
            print(f"{reg}\t{L_col[i]}\t{thr_txt}\t{r_txt}\t{ta}\t{tb}\t{tc}\t{avg}\t{fst}", file=out)
    if fmt in ("tajd", "all"):
        print("REGION\tLENGTH\tSAMPLES\tSEGREGATING_SITES\tPI\tTAJIMAS_D", file=out)
        for i, reg in enumerate(regions):
            D = float(col["tajima_d"][i])
            taj = "NA" if D != D else repr(D)  # run_tajd.sh:192-194
            print(f"{reg}\t{L_col[i]}\t{samples_col}\t{int(s_all[i])}\t{col['pi_site'][i]:.8f}\t{taj}", file=out)
    if args.output or rank != 0:
        out.close()
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
