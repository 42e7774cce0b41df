"""CPU: the reformulation greedy_groups_bits (impop_amd/csrc/stats.hip) relies on, as a plain-Python model.

The reference's grouping (pica2.py:94-112 with seeds in position order) takes the smallest free element as seed and moves
every FREE element that joins the seed into its group.  The kernel instead (1) tests the join relation of a BLOCK of
candidate seeds up front — it does not depend on the state — and (2) numbers runs of candidates that are still free and
whose rows miss the free set together, resolving only candidates whose rows take elements with them one by one.  This test
checks that model against the literal loop on random NON-transitive relations (the GPU parity tests check the kernel
against the oracle; this one pins the argument the kernel's comments make)."""
import random


def literal(m, joins):
    free, grp, G = set(range(m)), [-1] * m, 0
    for seed in range(m):
        if seed not in free:
            continue
        for o in [seed] + [o for o in range(seed + 1, m) if o in free and joins(seed, o)]:
            grp[o] = G
            free.discard(o)
        G += 1
    return grp, G


def blocked(m, joins, B):
    free, grp, G = set(range(m)), [-1] * m, 0
    while free:
        cand = sorted(free)[:B]
        rows = [{o for o in range(c + 1, m) if joins(c, o)} for c in cand]   # tested up front, state-independent
        pos, n = 0, len(cand)
        while pos < n:
            alive = [b >= pos and cand[b] in free for b in range(n)]
            busy = [alive[b] and bool(rows[b] & free) for b in range(n)]
            first = next((b for b in range(n) if busy[b]), n)
            lone = [b for b in range(n) if alive[b] and b < first]             # singleton groups, numbered together
            for r, b in enumerate(lone):
                grp[cand[b]] = G + r
                free.discard(cand[b])
            G += len(lone)
            if first >= n:
                break
            for o in [cand[first]] + sorted(rows[first] & free):
                grp[o] = G
                free.discard(o)
            G += 1
            pos = first + 1
    return grp, G


def test_blocked_bulk_grouping_equals_the_literal_loop():
    rng = random.Random(7)
    for _ in range(600):
        m = rng.randint(1, 70)
        p = rng.choice([0.0, 0.01, 0.05, 0.2, 0.6, 1.0])
        rel = {(i, j) for i in range(m) for j in range(i + 1, m) if rng.random() < p}
        joins = lambda a, b: (a, b) in rel
        assert blocked(m, joins, rng.choice([1, 4, 8, 16, 64])) == literal(m, joins)
