#!/usr/bin/env python3
"""Drop-in for the reference's scripts/tj_d.py command line (tj_d.py:71-88): prints `Tajima's D: <D>`
(third token = D, run_tajd.sh:186) and, on request, the constants."""
import _bootstrap  # noqa: F401
from _cli import make_parser
from impop_amd.tj_d import tajimas_d

FLAGS = (
    ("-n", "--sample-size", dict(type=int, required=True, help="sequences in the sample, at least 2")),
    ("-S", "--segregating-sites", dict(type=float, required=True, help="segregating sites in the window")),
    ("-p", "--pi", dict(type=float, required=True, help="mean pairwise differences")),
    ("--show-components", dict(action="store_true", help="also print a1 a2 b1 b2 c1 c2 e1 e2 and the two halves of D")),
)
COMPONENT_LINES = (("a1", "a2"), ("b1", "b2"), ("c1", "c2"), ("e1", "e2"), ("numerator", "denominator"))


def main():
    opt = make_parser("Tajima's D from sample size, segregating sites and pi.", FLAGS).parse_args()
    D, parts = tajimas_d(opt.sample_size, opt.segregating_sites, opt.pi, return_components=True)
    report = [f"Tajima's D: {D}"]
    if opt.show_components:
        report.append("--- Components ---")
        report += [" ".join(f"{name}={getattr(parts, name)}" for name in pair) for pair in COMPONENT_LINES]
    print("\n".join(report))


if __name__ == "__main__":
    main()
