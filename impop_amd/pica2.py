"""GPU-backed mirror of the reference's scripts/pica2.py function API (same names, argument
meaning, return values, log text and error behaviour; pica2.py:6-169)."""
from __future__ import annotations

import math

import numpy as np

from .runtime import _line_writer, default_context
from .simfile import densify
from .simfile import read_dense  # noqa: F401
from .simfile import read_similarity_file_pica2 as read_similarity_file  # noqa: F401  (pica2.py:6)


def seed_rank_of(elements, names):
    """Position of every name of `names` in the order in which the reference's greedy grouping would take
    its seeds.  pica2.py:96-100 copies the caller's set (`remaining = set(elements)`) and pop()s from it;
    set.pop() walks the hash table from slot 0 and removals never rehash, so the seeds come in the iteration
    order of that copy, restricted to what is left.  Evaluating the same expression on the same object in the
    same interpreter gives the same order (it depends on the string hashes, i.e. on PYTHONHASHSEED, and on the
    set's insertion history), which is all the GPU grouping needs to reproduce the reference's groups on
    tables where "> threshold" is not transitive."""
    pos = {name: k for k, name in enumerate(set(elements))}
    return np.fromiter((pos[nm] for nm in names), dtype=np.uint32, count=len(names))


def analyze_similarity_matrix(similarity_dict, elements, pair_count, threshold=1.0, sequence_length=None,
                              log_file=None, round_digits=None, ctx=None):
    """pica2.analyze_similarity_matrix (pica2.py:60-169) -> (pi, pi_per_site).

    Grouping, the group-pair sum and the normalisation run on the GPU (impop_pi_from_identity), with the
    seeds of the greedy grouping taken in the order the reference's `set(elements).pop()` yields them
    (seed_rank_of).  Like the reference this rounds `similarity_dict` in place when round_digits is given
    (pica2.py:81-83)."""
    if round_digits is not None:  # caller-visible side effect of the reference
        for key in list(similarity_dict.keys()):
            similarity_dict[key] = round(similarity_dict[key], round_digits)
    names = sorted(elements)
    dense = densify(similarity_dict, names)
    # values are already rounded in the dict; the device rounds again, which is idempotent
    return analyze_dense(names, dense, pair_count, threshold, sequence_length, log_file, round_digits, ctx,
                         elements=elements)


def analyze_dense(names, dense, pair_count, threshold=1.0, sequence_length=None, log_file=None, round_digits=None,
                  ctx=None, elements=None):
    """Same analysis on an already densified table (sorted names, [n,n] identity, NaN = absent):
    what the drop-in CLI calls after the native .sim ingest (simfile.read_dense).  `elements`: the set the
    reference's reader would have built (same insertion order); None = seed with the smallest remaining
    name (deterministic, and one of the orders the reference can take)."""
    log_print = _line_writer(log_file)
    n = len(names)
    log_print(f"Loaded {pair_count} pairwise similarities")
    log_print(f"Found {n} unique elements")
    if round_digits is not None:
        log_print(f"Rounded similarities to {round_digits} decimal places")
    ctx = ctx or default_context()
    rank = seed_rank_of(elements, names) if elements is not None else None
    pi, pi_site, group_of, n_groups, (sum2, n_pairs) = ctx.pi_from_identity(dense, threshold, round_digits, sequence_length,
                                                                           seed_rank=rank, detail=True)
    groups = [[] for _ in range(n_groups)]
    for name, g in zip(names, group_of):
        groups[int(g)].append(name)
    log_print(f"\nStep 1: Grouping elements (threshold > {threshold})")
    log_print(f"Found {len(groups)} groups:")
    for i, group in enumerate(groups, 1):
        log_print(f"  G{i}: {group} (size: {len(group)})")
    log_print("\nStep 2: Calculating group pairs")
    if n == 0:  # pica2.py:122-124
        log_print("Warning: No elements available to compute group pairs")
        return 0.0, 0.0
    if log_file and n_groups > 1:  # the per-pair table exists for the log only
        first = {}
        for i, g in enumerate(group_of):
            first.setdefault(int(g), i)
        rep = [first[g] for g in range(n_groups)]
        sims, vals = ctx.pica2_pair_terms(dense, round_digits, rep, [len(g) for g in groups])
        k = 0
        for i in range(n_groups):
            for j in range(i + 1, n_groups):
                if math.isnan(sims[k]):
                    log_print(f"Warning: No similarity data found between groups G{i+1} and G{j+1}, skipping...")
                else:
                    log_print(f"  G{i+1}G{j+1}: (1 - {sims[k]:.6f}) * ({len(groups[i])}/{n}) * ({len(groups[j])}/{n}) = {vals[k]:.6f}")
                k += 1
    log_print("\nStep 3: Calculating pi")
    if n_pairs == 0:  # pica2.py:150-152
        log_print("Warning: No group pairs found with similarity data!")
        return 0.0, 0.0
    log_print(f"  n (total elements) = {n}")
    log_print(f"  Number of group pairs with data = {n_pairs}")
    log_print(f"  Sum of 2 * group_pairs = {sum2:.6f}")
    log_print(f"  pi = {n}/{n-1} * {sum2:.6f} = {pi:.6f}")
    if not sequence_length:
        return pi, None
    log_print("\nNormalization:")
    log_print(f"  Sequence length = {sequence_length}")
    log_print(f"  pi per site = {pi:.6f} / {sequence_length} = {pi_site:.8f}")
    return pi, pi_site
