"""GPU-backed mirror of the reference's scripts/hudson/hud.py calculate_fst (hud.py:173-300):
method='direct' is h-fst's calculation, method='grouped' groups similar sequences inside each
population first (hud.py:64-128, 235-263)."""
from __future__ import annotations

import sys

from .hfst import _flags, calculate_fst as _direct
from .runtime import default_context
from .simfile import densify


def calculate_fst(similarities, pop_a, pop_b, sequence_length=None, round_digits=None, log_file=None, method="direct",
                  threshold=0.999, ctx=None):
    if method != "grouped":
        return _direct(similarities, pop_a, pop_b, sequence_length, round_digits, log_file, ctx=ctx)

    def log_print(msg):
        if log_file:
            print(msg, file=log_file)

    overlap = pop_a & pop_b
    if overlap:  # hud.py:186-190
        print(f"Warning: {len(overlap)} sequences appear in both populations", file=sys.stderr)
        pop_a = pop_a - overlap
        pop_b = pop_b - overlap
    ctx = ctx or default_context()
    names = sorted(set(pop_a) | set(pop_b) | {k for pair in similarities for k in pair})
    dense = densify(similarities, names)
    L = sequence_length if (sequence_length and sequence_length > 0) else None
    out, cnt = ctx.fst_grouped_from_identity(dense, _flags(names, pop_a), _flags(names, pop_b), threshold, L, round_digits)
    log_print("FST Calculation")
    log_print("=" * 50)
    log_print(f"Population A: {len(pop_a)} sequences")
    log_print(f"Population B: {len(pop_b)} sequences")
    log_print("Method: grouped")
    log_print(f"Grouping threshold: {threshold}")
    log_print(f"  groups A = {int(cnt[0])} ({int(cnt[1])} missing pairs), groups B = {int(cnt[2])} ({int(cnt[3])} missing pairs), "
              f"group pairs between = {int(cnt[4])} ({int(cnt[5])} missing)")
    keys = ("fst", "pi_a", "pi_b", "pi_xy", "dxy", "da")
    return {k: float(v) for k, v in zip(keys, out)}
