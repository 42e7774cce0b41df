"""GPU-backed mirror of the reference's scripts/h-fst.py function API (h-fst.py:18-249)."""
from __future__ import annotations

import sys

import numpy as np

from .popnames import canonicalize_identifier, expand_population, read_subset_file  # noqa: F401
from .runtime import _line_writer, default_context
from .simfile import densify, read_dense  # noqa: F401
from .simfile import read_similarity_file_hfst as read_similarity_file  # noqa: F401  (h-fst.py:84)


def _flags(names, members):
    s = set(members)
    return np.fromiter((1 if x in s else 0 for x in names), dtype=np.uint8, count=len(names))


def calculate_diversity(similarities, seq_set1, seq_set2=None, round_digits=None, ctx=None):
    """h-fst.calculate_diversity (h-fst.py:130-171) -> (mean, n_pairs, n_missing)."""
    ctx = ctx or default_context()
    names = sorted(set(seq_set1) | set(seq_set2 or ()) | {k for pair in similarities for k in pair})
    dense = densify(similarities, names)
    if seq_set2 is None:
        out, cnt = ctx.fst_from_identity(dense, _flags(names, seq_set1), np.zeros(len(names), np.uint8), None, round_digits)
        return float(out[1]), int(cnt[0]), int(cnt[1])
    # between: members of both sets would be dropped by calculate_fst's overlap rule, which
    # calculate_diversity itself does not apply — evaluate on de-duplicated copies
    a, b = set(seq_set1), set(seq_set2)
    if a & b:
        raise ValueError("calculate_diversity(between) with overlapping sets: call calculate_fst instead")
    out, cnt = ctx.fst_from_identity(dense, _flags(names, a), _flags(names, b), None, round_digits)
    return float(out[4]), int(cnt[4]), int(cnt[5])


def calculate_fst(similarities, pop_a, pop_b, sequence_length=None, round_digits=None, log_file=None, ctx=None):
    """h-fst.calculate_fst (h-fst.py:173-249) -> dict(fst, pi_a, pi_b, pi_xy, dxy, da)."""
    names = sorted(set(pop_a) | set(pop_b) | {k for pair in similarities for k in pair})
    dense = densify(similarities, names)
    return calculate_fst_dense(names, dense, pop_a, pop_b, sequence_length, round_digits, log_file, ctx)


def calculate_fst_dense(names, dense, pop_a, pop_b, sequence_length=None, round_digits=None, log_file=None, ctx=None):
    """calculate_fst on an already densified table (what the drop-in CLI calls after the native ingest)."""
    overlap = pop_a & pop_b
    if overlap:  # h-fst.py:181-185
        print(f"Warning: {len(overlap)} sequences appear in both populations", file=sys.stderr)
        pop_a = pop_a - overlap
        pop_b = pop_b - overlap
    ctx = ctx or default_context()
    known = set(names)
    extra = sorted((set(pop_a) | set(pop_b)) - known)
    if extra:  # population members absent from the table: rows/cols of NaN (all their pairs are "missing")
        n0 = len(names)
        names = list(names) + extra
        order = sorted(range(len(names)), key=lambda i: names[i])
        big = np.full((len(names), len(names)), np.nan)
        big[:n0, :n0] = dense
        dense = big[np.ix_(order, order)]
        names = [names[i] for i in order]
    out, cnt = ctx.fst_from_identity(dense, _flags(names, pop_a), _flags(names, pop_b), None, round_digits)
    fst, pi_a, pi_b, pi_xy, dxy = (float(x) for x in out[:5])
    if log_file:  # h-fst.py:187-231: hud.py's direct-method text without the method lines
        from .hud import _log_text
        log_file.write(_log_text(None, None, round_digits, len(pop_a), len(pop_b), (fst, pi_a, pi_b, pi_xy, dxy),
                                 [int(c) for c in cnt], sequence_length))
    if sequence_length and sequence_length > 0:  # h-fst.py:233-240 (Fst itself is never divided)
        return {"fst": fst, "pi_a": pi_a / sequence_length, "pi_b": pi_b / sequence_length, "pi_xy": pi_xy / sequence_length,
                "dxy": dxy / sequence_length, "da": (dxy - pi_xy) / sequence_length}
    return {"fst": fst, "pi_a": pi_a, "pi_b": pi_b, "pi_xy": pi_xy, "dxy": dxy, "da": dxy - pi_xy}
