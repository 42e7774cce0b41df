"""GPU-backed mirror of the reference's scripts/hudson/hud.py (hud.py:173-308): method='direct' is
h-fst's pairwise mean, method='grouped' groups similar sequences inside each population first
(hud.py:64-128, 235-263).  Same arguments, return dict, stderr warning and log text; the sums run in
libimpop_hip.so (impop_fst_from_identity / impop_fst_grouped_from_identity)."""
from __future__ import annotations

import sys

import numpy as np

from .hfst import _flags
from .popnames import read_subset_file  # noqa: F401  (hud.py:55-62 reads its lists the same way)
from .runtime import default_context
from .simfile import densify, read_dense  # noqa: F401
from .simfile import read_similarity_file_hfst as read_similarity_file  # noqa: F401  (hud.py:18-53, same reader)


def calculate_fst(similarities, pop_a, pop_b, sequence_length=None, round_digits=None, log_file=None, method="direct",
                  threshold=0.999, ctx=None):
    names = sorted(set(pop_a) | set(pop_b) | {k for pair in similarities for k in pair})
    return calculate_fst_dense(names, densify(similarities, names), pop_a, pop_b, sequence_length, round_digits, log_file,
                               method, threshold, ctx)


def _log_text(method, threshold, round_digits, size_a, size_b, v, cnt, sequence_length) -> str:
    """The `<basename>_fst.log` text of hud.py:194-289 — and, with method=None, of h-fst.py:187-231, which is the
    same text without the method lines — for the values v = (fst, pi_a, pi_b, pi_xy, dxy) and the kernel's
    counters; a log file is part of what the drop-in replaces, so the wording is the reference's."""
    fst, pi_a, pi_b, pi_xy, dxy = v
    grouped = method == "grouped"
    out = ["FST Calculation", "=" * 50, f"Population A: {size_a} sequences", f"Population B: {size_b} sequences"]
    if method is not None:
        out.append(f"Method: {method}")
    if grouped:
        out.append(f"Grouping threshold: {threshold}")
    if round_digits is not None:
        out.append(f"Rounding similarities to {round_digits} decimal places")
    out += ["", f"Within-population diversity (π) using {method} method:" if method is not None
            else "Within-population diversity (π):"]
    for tag, pi, size, k in (("A", pi_a, size_a, 0), ("B", pi_b, size_b, 2)):
        if grouped:
            out.append(f"  π{tag} = {pi:.6f} ({cnt[k]} groups from {size} sequences, {cnt[k + 1]} missing pairs)")
        else:
            out.append(f"  π{tag} = {pi:.6f} (from {cnt[k]} pairs, {cnt[k + 1]} missing)")
    out += [f"  πXY = {pi_xy:.6f} (average of πA and πB)", "", "Between-population diversity (Dxy):"]
    if grouped:
        out.append(f"  Dxy = {dxy:.6f} (from {cnt[0]} x {cnt[2]} group pairs, {cnt[5]} missing)")
    else:
        out.append(f"  Dxy = {dxy:.6f} (from {cnt[4]} pairs, {cnt[5]} missing)")
    out.append("")
    if dxy > 0:
        out += ["FST calculation:", "  FST = (Dxy - πXY) / Dxy", f"      = ({dxy:.6f} - {pi_xy:.6f}) / {dxy:.6f}", f"      = {fst:.6f}"]
    else:
        out.append("FST = 0 (Dxy = 0)")
    if sequence_length and sequence_length > 0:
        out += ["", f"Per-site values (sequence length = {sequence_length:,}):"]
        out += [f"  {label} per site = {value / sequence_length:.8f}"
                for label, value in (("πA", pi_a), ("πB", pi_b), ("πXY", pi_xy), ("Dxy", dxy))]
    return "\n".join(out) + "\n"


def calculate_fst_dense(names, dense, pop_a, pop_b, sequence_length=None, round_digits=None, log_file=None, method="direct",
                        threshold=0.999, ctx=None):
    """calculate_fst on a densified table whose `names` (sorted) cover both populations."""
    shared = pop_a & pop_b
    if shared:  # members of both populations leave both (hud.py:186-190)
        print(f"Warning: {len(shared)} sequences appear in both populations", file=sys.stderr)
        pop_a, pop_b = pop_a - shared, pop_b - shared
    ctx = ctx or default_context()
    strangers = (set(pop_a) | set(pop_b)) - set(names)
    if strangers:
        raise KeyError(f"{len(strangers)} population members absent from the identity table's name list")
    fa, fb = _flags(names, pop_a), _flags(names, pop_b)
    if method == "grouped":
        # hud.py:67-71 seeds its greedy groups with set.pop() from `set(sequences)`: hand the kernel that
        # iteration order for each population (same expression, same object, same interpreter: same order)
        at = {nm: i for i, nm in enumerate(names)}
        rank = np.zeros(len(names), dtype=np.uint32)
        for pop in (pop_a, pop_b):
            for k, nm in enumerate(set(pop)):
                rank[at[nm]] = k
        out, cnt = ctx.fst_grouped_from_identity(dense, fa, fb, threshold, None, round_digits, seed_rank=rank)
    else:
        out, cnt = ctx.fst_from_identity(dense, fa, fb, None, round_digits)
    fst, pi_a, pi_b, pi_xy, dxy = (float(x) for x in out[:5])
    if log_file:
        log_file.write(_log_text(method, threshold, round_digits, len(pop_a), len(pop_b), (fst, pi_a, pi_b, pi_xy, dxy),
                                 [int(c) for c in cnt], sequence_length))
    scale = sequence_length if (sequence_length and sequence_length > 0) else 1  # Fst itself is never divided
    return {"fst": fst, "pi_a": pi_a / scale, "pi_b": pi_b / scale, "pi_xy": pi_xy / scale, "dxy": dxy / scale,
            "da": (dxy - pi_xy) / scale}
