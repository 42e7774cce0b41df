#!/usr/bin/env python3
"""Drop-in for the reference's scripts/wip/ehhgfa.py CLI (ehhgfa.py:24-71): same flags, same rows
`window colstart colend allele REF|ALT area` written to -o."""
import argparse

import numpy as np

import _bootstrap  # noqa: F401
from impop_amd.ehh import scan_windows


def main():
    parser = argparse.ArgumentParser()
    parser.add_argument("-i", help="Path to the input file, matrix of haplotypes, no header")
    parser.add_argument("-p", type=int, help="Position of the test SNP in the haplotype window")
    parser.add_argument("-w", type=int, help="Window size")
    parser.add_argument("-refpos", type=int, help="reference position ")
    parser.add_argument("-o", type=str, help="outputfile  ")
    args = parser.parse_args()
    with open(args.o, "w") as out:
        whole = np.loadtxt(args.i)
        for name, colstart, colend, al, typeal, area in scan_windows(whole, args.p, args.w, args.refpos):
            print(name, colstart, colend, al, typeal, area, file=out, flush=True)


if __name__ == "__main__":
    main()
