#!/usr/bin/env python3
"""Drop-in for the reference's scripts/tj_d.py CLI (tj_d.py:71-88): `Tajima's D: {D}`."""
import argparse

import _bootstrap  # noqa: F401
from impop_amd.tj_d import tajimas_d


def main():
    parser = argparse.ArgumentParser(description="Compute Tajima's D from n, S, and pi.")
    parser.add_argument("-n", "--sample-size", type=int, required=True, help="Number of sequences (n >= 2)")
    parser.add_argument("-S", "--segregating-sites", type=float, required=True, help="Number of segregating sites S (>= 0)")
    parser.add_argument("-p", "--pi", type=float, required=True, help="Mean pairwise differences pi (>= 0)")
    parser.add_argument("--show-components", action="store_true", help="Print intermediate constants (a1, a2, e1, e2, etc.)")
    args = parser.parse_args()
    D, comps = tajimas_d(args.sample_size, args.segregating_sites, args.pi, return_components=True)
    print(f"Tajima's D: {D}")
    if args.show_components:
        print("--- Components ---")
        print(f"a1={comps.a1} a2={comps.a2}")
        print(f"b1={comps.b1} b2={comps.b2}")
        print(f"c1={comps.c1} c2={comps.c2}")
        print(f"e1={comps.e1} e2={comps.e2}")
        print(f"numerator={comps.numerator} denominator={comps.denominator}")


if __name__ == "__main__":
    main()
