#!/usr/bin/env python3
"""Drop-in for the reference's scripts/hudson/hud.py CLI (hud.py:310-415): same flags (incl.
-m direct|grouped, -t), stderr diagnostics, `<base>_fst.log` and the 6 tab-separated `.8f` fields."""
import argparse
import os
import sys

import _bootstrap  # noqa: F401
from impop_amd.hud import calculate_fst_dense, read_dense, read_subset_file


def main():
    parser = argparse.ArgumentParser(description='Calculate FST from pairwise sequence similarities')
    parser.add_argument('similarity_file', help='TSV file with columns: group.a, group.b, estimated.identity')
    parser.add_argument('-a', '--pop-a', required=True, help='File listing sequence IDs for population A')
    parser.add_argument('-b', '--pop-b', required=True, help='File listing sequence IDs for population B')
    parser.add_argument('-l', '--length', type=int, default=None, help='Sequence length for per-site calculations')
    parser.add_argument('-r', '--round', type=int, default=None, help='Round similarities to N decimal places')
    parser.add_argument('-m', '--method', choices=['direct', 'grouped'], default='direct',
                        help='Calculation method: direct or grouped (default: direct)')
    parser.add_argument('-t', '--threshold', type=float, default=0.999,
                        help='Similarity threshold for grouping (default: 0.999, used only with -m grouped)')
    parser.add_argument('-d', '--log-dir', default='.', help='Directory for log file (default: current directory)')
    parser.add_argument('-v', '--verbose', action='store_true', help='Print detailed progress to stderr')
    args = parser.parse_args()

    if args.verbose:
        print(f"Reading similarity file: {args.similarity_file}", file=sys.stderr)
    names, dense, _ = read_dense(args.similarity_file, "hfst")
    all_sequences = set(names)
    if args.verbose:
        print("Reading population files...", file=sys.stderr)
    pop_a = read_subset_file(args.pop_a)
    pop_b = read_subset_file(args.pop_b)
    if args.verbose:
        print(f"Population A: {len(pop_a)} sequences", file=sys.stderr)
        print(f"Population B: {len(pop_b)} sequences", file=sys.stderr)
        print(f"Method: {args.method}", file=sys.stderr)
        if args.method == 'grouped':
            print(f"Grouping threshold: {args.threshold}", file=sys.stderr)
    missing_a = pop_a - all_sequences
    missing_b = pop_b - all_sequences
    if missing_a:
        print(f"Warning: {len(missing_a)} sequences from population A not found in similarity file", file=sys.stderr)
    if missing_b:
        print(f"Warning: {len(missing_b)} sequences from population B not found in similarity file", file=sys.stderr)
    pop_a = pop_a & all_sequences
    pop_b = pop_b & all_sequences
    if not pop_a or not pop_b:
        print("Error: No valid sequences found in one or both populations", file=sys.stderr)
        sys.exit(1)
    base_name = os.path.splitext(os.path.basename(args.similarity_file))[0]
    log_path = os.path.join(args.log_dir, f"{base_name}_fst.log")
    os.makedirs(args.log_dir, exist_ok=True)
    with open(log_path, 'w') as log_file:
        results = calculate_fst_dense(names, dense, pop_a, pop_b, sequence_length=args.length, round_digits=args.round,
                                      log_file=log_file, method=args.method, threshold=args.threshold)
    print(f"{results['fst']:.8f}\t{results['pi_a']:.8f}\t{results['pi_b']:.8f}\t"
          f"{results['pi_xy']:.8f}\t{results['dxy']:.8f}\t{results['da']:.8f}")
    if args.verbose:
        print(f"Detailed log saved to: {log_path}", file=sys.stderr)


if __name__ == "__main__":
    main()
