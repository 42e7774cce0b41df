#!/usr/bin/env python3
"""Drop-in for the reference's scripts/wip/ehhgfa.py command line (ehhgfa.py:24-71): same flags; one
row `window colstart colend allele REF|ALT area` per window and test-SNP allele, written to -o."""
import numpy as np

import _bootstrap  # noqa: F401
from _cli import make_parser
from impop_amd.ehh import scan_windows

FLAGS = (
    ("-i", dict(help="haplotype matrix as whitespace-separated numbers, one haplotype per row, no header")),
    ("-p", dict(type=int, help="1-based column of the test SNP inside each window")),
    ("-w", dict(type=int, help="window width in columns")),
    ("-refpos", dict(type=int, help="1-based row of the haplotype whose allele is called REF")),
    ("-o", dict(type=str, help="output file")),
)


def main():
    opt = make_parser("Integrated EHH around a test SNP, window by window.", FLAGS).parse_args()
    with open(opt.o, "w") as sink:
        matrix = np.loadtxt(opt.i)
        for row in scan_windows(matrix, opt.p, opt.w, opt.refpos):
            print(*row, file=sink, flush=True)


if __name__ == "__main__":
    main()
