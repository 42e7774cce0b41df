"""Row formatting shared by the batch driver: the text conventions of the reference's bash drivers
(8-decimal pi text, NA handling).  Host-side driver logic, mirrors inline Python of the drivers."""
from __future__ import annotations


def pica_cell(pi_site: float, length: int) -> str:
    """pica2.py:226 stdout, squeezed as run_pica2_impg.sh:182 does."""
    return f"{pi_site:.8f} (sequence length: {length})"


def fst_3pi_fields(pi_a: float, pi_b: float, pi_c: float):
    """run_fst_impg.sh:184-218: PI_A/PI_B/PI_C are pica2's 8-decimal TEXT (first token of its stdout,
    :80); the average and Fst are computed from the parsed text values; Fst is 'NA' when pi_C == 0."""
    ta, tb, tc = f"{pi_a:.8f}", f"{pi_b:.8f}", f"{pi_c:.8f}"
    fa, fb, fc = float(ta), float(tb), float(tc)
    avg = 0.5 * (fa + fb)
    fst = "NA" if fc == 0 else f"{(fc - avg) / fc:.8f}"
    return ta, tb, tc, f"{avg:.8f}", fst


def pi_union_site(rec, n_a: int, n_b: int, length: int) -> float:
    """Per-site pica2 pi (threshold >= 1) of the union C = A u B of two DISJOINT populations from one
    scan record: sum_{i<j in C} H_ij = sum_a + sum_b + sum_ab exactly (integers)."""
    n_c = n_a + n_b
    W = int(rec["n_sites"])
    if n_c < 2 or W == 0 or not length:
        return 0.0
    total = int(rec["sum_a"]) + int(rec["sum_b"]) + int(rec["sum_ab"])
    return total / ((n_c * (n_c - 1) / 2.0) * W) / length
