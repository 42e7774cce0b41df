#!/usr/bin/env python3
"""Drop-in for the reference's scripts/af.py command line (af.py:70-92): haplotype clusters at an identity
threshold, summary TSV on stdout (or --output) and optional per-sample assignments."""
import sys

import _bootstrap  # noqa: F401
from _cli import make_parser
from impop_amd.af import build_summary, cluster, load_pairs, write_details, write_summary

FLAGS = (
    ("--input", dict(default="loc.sim", help="identity table to cluster (loc.sim)")),
    ("--threshold", dict(type=float, default=1.0, help="samples at least this identical are linked (1.0)")),
    ("--output", dict(help="write the cluster summary here instead of stdout")),
    ("--details", dict(help="also write sample_id / cluster_id / threshold rows here")),
)


def main():
    opt = make_parser("Cluster the samples of an identity table and report cluster frequencies.", FLAGS).parse_args()
    pairs, samples = load_pairs(opt.input)
    summary = build_summary(cluster(pairs, samples, opt.threshold))
    if opt.output:
        with open(opt.output, "w", newline="") as sink:
            write_summary(summary, sink)
    else:
        write_summary(summary, sys.stdout)
    if opt.details:
        write_details(summary, opt.threshold, opt.details)


if __name__ == "__main__":
    main()
