#!/usr/bin/env python3
"""GFA / `odgi paths -H` table -> presence-matrix container (.npz) for impop_scan.py.

    impop_extract.py --gfa window.gfa --ref-prefix 'CHM13#0#' -o window.npz
    impop_extract.py --paths-table paths.tsv -o window.npz
"""
import argparse

import _bootstrap  # noqa: F401
from impop_amd import extract, matrixio


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    src = ap.add_mutually_exclusive_group(required=True)
    src.add_argument("--gfa")
    src.add_argument("--paths-table")
    ap.add_argument("--ref-prefix", default=None, help="path-name prefix of the reference (gives bp coordinates per site)")
    ap.add_argument("--no-expand-bp", action="store_true", help="one site per node instead of one per bp")
    ap.add_argument("-o", "--output", required=True)
    a = ap.parse_args()
    mf = extract.from_gfa(a.gfa, a.ref_prefix, not a.no_expand_bp) if a.gfa else extract.from_paths_table(a.paths_table)
    matrixio.save_matrix(a.output, mf)
    print(f"{a.output}: {mf.n_hap} haplotypes x {mf.n_site} sites")


if __name__ == "__main__":
    main()
