"""GPU-backed mirror of the reference's scripts/pica2.py function API (same names, argument
meaning, return values and error behaviour; pica2.py:6-169)."""
from __future__ import annotations

import math

from .runtime import _line_writer, default_context
from .simfile import densify
from .simfile import read_dense  # noqa: F401
from .simfile import read_similarity_file_pica2 as read_similarity_file  # noqa: F401  (pica2.py:6)


def analyze_similarity_matrix(similarity_dict, elements, pair_count, threshold=1.0, sequence_length=None,
                              log_file=None, round_digits=None, ctx=None):
    """pica2.analyze_similarity_matrix (pica2.py:60-169) -> (pi, pi_per_site).

    Grouping, the group-pair sum and the normalisation run on the GPU
    (impop_pi_from_identity).  Like the reference this rounds `similarity_dict` in
    place when round_digits is given (pica2.py:81-83).  The one documented difference:
    each greedy group is seeded with the lexicographically smallest remaining element,
    where the reference pops an arbitrary set member (pica2.py:100)."""
    if round_digits is not None:  # caller-visible side effect of the reference
        for key in list(similarity_dict.keys()):
            similarity_dict[key] = round(similarity_dict[key], round_digits)
    names = sorted(elements)
    dense = densify(similarity_dict, names)
    # values are already rounded in the dict; the device rounds again, which is idempotent
    return analyze_dense(names, dense, pair_count, threshold, sequence_length, log_file, round_digits, ctx)


def analyze_dense(names, dense, pair_count, threshold=1.0, sequence_length=None, log_file=None, round_digits=None,
                  ctx=None):
    """Same analysis on an already densified table (sorted names, [n,n] identity, NaN = absent):
    what the drop-in CLI calls after the native .sim ingest (simfile.read_dense)."""
    log_print = _line_writer(log_file)
    log_print(f"Loaded {pair_count} pairwise similarities")
    log_print(f"Found {len(names)} unique elements")
    if round_digits is not None:
        log_print(f"Rounded similarities to {round_digits} decimal places")
    ctx = ctx or default_context()
    pi, pi_site, group_of, n_groups = ctx.pi_from_identity(dense, threshold, round_digits, sequence_length)
    groups = [[] for _ in range(n_groups)]
    for name, g in zip(names, group_of):
        groups[int(g)].append(name)
    log_print(f"\nStep 1: Grouping elements (threshold > {threshold})")
    log_print(f"Found {len(groups)} groups:")
    for i, group in enumerate(groups, 1):
        log_print(f"  G{i}: {group} (size: {len(group)})")
    log_print("\nStep 3: Calculating pi")
    log_print(f"  n (total elements) = {len(names)}")
    log_print(f"  pi = {pi:.6f}")
    if not sequence_length:
        # pica2.py:150-152 returns (0.0, 0.0) when nothing was summed, else pi_per_site None
        return (pi, None) if (pi != 0.0 or _has_pairs(dense, group_of, n_groups)) else (0.0, 0.0)
    if math.isnan(pi_site):
        pi_site = None
    else:
        log_print("\nNormalization:")
        log_print(f"  Sequence length = {sequence_length}")
        log_print(f"  pi per site = {pi:.6f} / {sequence_length} = {pi_site:.8f}")
    return pi, pi_site


def _has_pairs(dense, group_of, n_groups) -> bool:
    """True iff some pair of group representatives is present (pica2.py:150): decides
    between the (0.0, 0.0) early return and (0.0, None) when pi is exactly 0 without -l."""
    if n_groups < 2:
        return False
    reps = {}
    for i, g in enumerate(group_of):
        reps.setdefault(int(g), i)
    r = sorted(reps.values())
    sub = dense[r][:, r]
    import numpy as np
    iu = np.triu_indices(len(r), 1)
    return bool((~np.isnan(sub[iu])).any())
