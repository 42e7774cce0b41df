"""Reference-STYLE CPU path (test/bench infrastructure only, see oracle/impop_oracle.c header).

A pure-Python restatement of how the reference processes ONE window: the n x n identity table is
serialised to a `.sim` TSV (what `impg similarity` hands over, run_pica2_impg.sh:162-168), parsed
with csv.DictReader into a dict keyed by name pairs (pica2.py:6-58, h-fst.py:84-119) and reduced by
the dict algorithms of pica2.py:94-169 / h-fst.py:130-249 / tj_d.py:47-69.  bench.py times it next
to the GPU run as the closest thing to "the reference's CPU Python path" that can travel to the
GPU box (the reference's own files cannot).  Not a copy: same algorithm, written from SURVEY.md
Appendix A; checked against the C oracle in tests/test_oracle_golden.py."""
from __future__ import annotations

import csv
import io
import math


def write_sim(names, sim) -> str:
    out = io.StringIO()
    out.write("group.a\tgroup.b\testimated.identity\n")
    n = len(names)
    for i in range(n):
        row = sim[i]
        for j in range(n):
            out.write(f"{names[i]}\t{names[j]}\t{float(row[j])!r}\n")
    return out.getvalue()


def parse_sim(text):
    table, elements, rows = {}, set(), 0
    for rec in csv.DictReader(io.StringIO(text), delimiter="\t"):
        rows += 1
        a, b = rec["group.a"], rec["group.b"]
        table[(a, b) if a <= b else (b, a)] = float(rec["estimated.identity"])
        elements.add(a)
        elements.add(b)
    return table, elements, rows


def pica2_pi(table, elements, threshold, seq_len, round_digits=None):
    if round_digits is not None:
        for k in list(table):
            table[k] = round(table[k], round_digits)
    get = lambda a, b: table.get((a, b) if a <= b else (b, a))
    remaining = sorted(elements)  # deterministic seed order (smallest name first)
    groups = []
    while remaining:
        seed = remaining.pop(0)
        grp, rest = [seed], []
        for o in remaining:
            v = get(seed, o)
            (grp if (v is not None and v > threshold) else rest).append(o)
        remaining = rest
        groups.append(sorted(grp))
    groups.sort()
    total = sum(len(g) for g in groups)
    if total == 0:
        return 0.0, 0.0
    pairs = []
    for i in range(len(groups)):
        for j in range(i + 1, len(groups)):
            s = get(groups[i][0], groups[j][0])
            if s is None:
                continue
            pairs.append((1 - s) * (len(groups[i]) / total) * (len(groups[j]) / total))
    if not pairs:
        return 0.0, 0.0
    pi = (total / (total - 1)) * sum(2 * p for p in pairs)
    return pi, (pi / seq_len if seq_len else None)


def hfst(table, pop_a, pop_b, seq_len):
    ov = pop_a & pop_b
    pop_a, pop_b = pop_a - ov, pop_b - ov

    def mean(xs, ys=None):
        vals = []
        if ys is None:
            xs = list(xs)
            for i in range(len(xs)):
                for j in range(i + 1, len(xs)):
                    k = (xs[i], xs[j]) if xs[i] <= xs[j] else (xs[j], xs[i])
                    if k in table:
                        vals.append(1 - table[k])
        else:
            for a in xs:
                for b in ys:
                    k = (a, b) if a <= b else (b, a)
                    if k in table:
                        vals.append(1 - table[k])
        return sum(vals) / len(vals) if vals else 0.0
    pi_a, pi_b, dxy = mean(pop_a), mean(pop_b), mean(pop_a, pop_b)
    pi_xy = 0.5 * (pi_a + pi_b)
    fst = (dxy - pi_xy) / dxy if dxy > 0 else 0.0
    L = seq_len if seq_len and seq_len > 0 else 1
    return {"fst": fst, "pi_a": pi_a / L, "pi_b": pi_b / L, "pi_xy": pi_xy / L, "dxy": dxy / L, "da": (dxy - pi_xy) / L}


def tajimas_d(n, S, pi):
    a1 = sum(1.0 / i for i in range(1, n))
    a2 = sum(1.0 / (i * i) for i in range(1, n))
    b1 = (n + 1.0) / (3.0 * (n - 1.0))
    b2 = 2.0 * (n * n + n + 3.0) / (9.0 * n * (n - 1.0))
    c1 = b1 - 1.0 / a1
    c2 = b2 - (n + 2.0) / (a1 * n) + a2 / (a1 * a1)
    e1, e2 = c1 / a1, c2 / (a1 * a1 + a2)
    den = math.sqrt(e1 * S + e2 * S * (S - 1.0)) if S > 0 else float("nan")
    return (pi - S / a1) / den if den and not math.isclose(den, 0.0) else float("nan")


def window_chain(names, sim, in_a, in_b, seq_len, S):
    """One window the way the three driver scripts do it: three .sim parses (pica2, h-fst, and
    pica2 again inside run_tajd.sh) are collapsed into one text round trip + three reductions."""
    text = write_sim(names, sim)
    table, elements, _ = parse_sim(text)
    pi, pi_site = pica2_pi(dict(table), elements, 1.0, seq_len)
    A = {n for n, f in zip(names, in_a) if f}
    B = {n for n, f in zip(names, in_b) if f}
    h = hfst(table, A, B, seq_len)
    D = tajimas_d(len(names), float(S), float(f"{pi_site:.8f}"))
    return {"pi": pi, "pi_site": pi_site, "tajima_d": D, **h}
