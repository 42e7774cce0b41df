"""GPU-backed mirror of the reference's scripts/hudson/hud.py (hud.py:173-308): method='direct' is
h-fst's pairwise mean, method='grouped' groups similar sequences inside each population first
(hud.py:64-128, 235-263).  Same arguments, return dict, stderr warning and log text; the sums run in
libimpop_hip.so (impop_fst_from_identity / impop_fst_grouped_from_identity)."""
from __future__ import annotations

import sys


from .hfst import _flags
from .runtime import default_context
from .simfile import densify, read_dense  # noqa: F401
from .simfile import read_similarity_file_hfst as read_similarity_file  # noqa: F401  (hud.py:18-53, same reader)


def read_subset_file(filename):
    """hud.read_subset_file (hud.py:55-62): exact identifiers, '#' comment lines skipped."""
    try:
        with open(filename) as f:
            return set(line.strip() for line in f if line.strip() and not line.startswith('#'))
    except FileNotFoundError:
        print(f"Error: Subset file not found: {filename}", file=sys.stderr)
        sys.exit(1)


def calculate_fst(similarities, pop_a, pop_b, sequence_length=None, round_digits=None, log_file=None, method="direct",
                  threshold=0.999, ctx=None):
    names = sorted(set(pop_a) | set(pop_b) | {k for pair in similarities for k in pair})
    return calculate_fst_dense(names, densify(similarities, names), pop_a, pop_b, sequence_length, round_digits, log_file,
                               method, threshold, ctx)


def calculate_fst_dense(names, dense, pop_a, pop_b, sequence_length=None, round_digits=None, log_file=None, method="direct",
                        threshold=0.999, ctx=None):
    """calculate_fst on a densified table whose `names` (sorted) cover both populations."""
    def log_print(msg):
        if log_file:
            print(msg, file=log_file)

    overlap = pop_a & pop_b
    if overlap:  # hud.py:186-190
        print(f"Warning: {len(overlap)} sequences appear in both populations", file=sys.stderr)
        pop_a = pop_a - overlap
        pop_b = pop_b - overlap
    ctx = ctx or default_context()
    missing = (set(pop_a) | set(pop_b)) - set(names)
    if missing:
        raise KeyError(f"{len(missing)} population members absent from the identity table's name list")
    fa, fb = _flags(names, pop_a), _flags(names, pop_b)
    if method == "grouped":
        out, cnt = ctx.fst_grouped_from_identity(dense, fa, fb, threshold, None, round_digits)
    else:
        out, cnt = ctx.fst_from_identity(dense, fa, fb, None, round_digits)
    fst, pi_a, pi_b, pi_xy, dxy = (float(v) for v in out[:5])
    cnt = [int(c) for c in cnt]

    log_print("FST Calculation")
    log_print("=" * 50)
    log_print(f"Population A: {len(pop_a)} sequences")
    log_print(f"Population B: {len(pop_b)} sequences")
    log_print(f"Method: {method}")
    if method == 'grouped':
        log_print(f"Grouping threshold: {threshold}")
    if round_digits is not None:
        log_print(f"Rounding similarities to {round_digits} decimal places")
    log_print("")
    if method == 'grouped':
        log_print("Within-population diversity (π) using grouped method:")
        log_print(f"  πA = {pi_a:.6f} ({cnt[0]} groups from {len(pop_a)} sequences, {cnt[1]} missing pairs)")
        log_print(f"  πB = {pi_b:.6f} ({cnt[2]} groups from {len(pop_b)} sequences, {cnt[3]} missing pairs)")
    else:
        log_print("Within-population diversity (π) using direct method:")
        log_print(f"  πA = {pi_a:.6f} (from {cnt[0]} pairs, {cnt[1]} missing)")
        log_print(f"  πB = {pi_b:.6f} (from {cnt[2]} pairs, {cnt[3]} missing)")
    log_print(f"  πXY = {pi_xy:.6f} (average of πA and πB)")
    log_print("")
    log_print("Between-population diversity (Dxy):")
    if method == 'grouped':
        log_print(f"  Dxy = {dxy:.6f} (from {cnt[0]} x {cnt[2]} group pairs, {cnt[5]} missing)")
    else:
        log_print(f"  Dxy = {dxy:.6f} (from {cnt[4]} pairs, {cnt[5]} missing)")
    log_print("")
    if dxy > 0:
        log_print("FST calculation:")
        log_print("  FST = (Dxy - πXY) / Dxy")
        log_print(f"      = ({dxy:.6f} - {pi_xy:.6f}) / {dxy:.6f}")
        log_print(f"      = {fst:.6f}")
    else:
        log_print("FST = 0 (Dxy = 0)")
    if sequence_length and sequence_length > 0:  # hud.py:283-299
        log_print("")
        log_print(f"Per-site values (sequence length = {sequence_length:,}):")
        log_print(f"  πA per site = {pi_a/sequence_length:.8f}")
        log_print(f"  πB per site = {pi_b/sequence_length:.8f}")
        log_print(f"  πXY per site = {pi_xy/sequence_length:.8f}")
        log_print(f"  Dxy per site = {dxy/sequence_length:.8f}")
        return {'fst': fst, 'pi_a': pi_a / sequence_length, 'pi_b': pi_b / sequence_length,
                'pi_xy': pi_xy / sequence_length, 'dxy': dxy / sequence_length, 'da': (dxy - pi_xy) / sequence_length}
    return {'fst': fst, 'pi_a': pi_a, 'pi_b': pi_b, 'pi_xy': pi_xy, 'dxy': dxy, 'da': dxy - pi_xy}
