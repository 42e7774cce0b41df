#!/usr/bin/env python3
"""Drop-in for the reference's scripts/af.py CLI (af.py:70-92)."""
import argparse
import sys

import _bootstrap  # noqa: F401
from impop_amd.af import build_summary, cluster, load_pairs, write_details, write_summary


def main():
    parser = argparse.ArgumentParser(description='Cluster samples in loc.sim-style tables by identity threshold.')
    parser.add_argument('--input', default='loc.sim', help='Path to the similarity table (default: loc.sim)')
    parser.add_argument('--threshold', type=float, default=1.0, help='Minimum estimated.identity to link samples (default: 1.0)')
    parser.add_argument('--output', help='Optional output TSV path for cluster summary; stdout if omitted')
    parser.add_argument('--details', help='Optional path to write detailed sample assignments')
    args = parser.parse_args()
    rows, samples = load_pairs(args.input)
    clusters = cluster(rows, samples, args.threshold)
    summary = build_summary(clusters)
    if args.output:
        with open(args.output, 'w', newline='') as fh:
            write_summary(summary, fh)
    else:
        write_summary(summary, sys.stdout)
    if args.details:
        write_details(summary, args.threshold, args.details)


if __name__ == '__main__':
    main()
