#!/usr/bin/env python3
"""Drop-in for the reference's scripts/h-fst.py CLI (h-fst.py:251-342): same flags, stderr
diagnostics, log file and the 6 tab-separated `.8f` fields on stdout."""
import argparse
import os
import sys

import _bootstrap  # noqa: F401
from impop_amd.hfst import calculate_fst_dense, expand_population, read_dense, read_subset_file


def main():
    parser = argparse.ArgumentParser(description='Calculate FST from pairwise sequence similarities')
    parser.add_argument('similarity_file', help='TSV file with columns: group.a, group.b, estimated.identity')
    parser.add_argument('-a', '--pop-a', required=True, help='File listing sequence IDs for population A')
    parser.add_argument('-b', '--pop-b', required=True, help='File listing sequence IDs for population B')
    parser.add_argument('-l', '--length', type=int, default=None, help='Sequence length for per-site calculations')
    parser.add_argument('-r', '--round', type=int, default=None, help='Round similarities to N decimal places')
    parser.add_argument('-d', '--log-dir', default='.', help='Directory for log file (default: current directory)')
    parser.add_argument('-v', '--verbose', action='store_true', help='Print detailed progress to stderr')
    args = parser.parse_args()

    if args.verbose:
        print(f"Reading similarity file: {args.similarity_file}", file=sys.stderr)
    names, dense, _ = read_dense(args.similarity_file, "hfst")  # native ingest; reference messages on errors
    all_sequences = set(names)
    if args.verbose:
        print("Reading population files...", file=sys.stderr)
    pop_a_raw = read_subset_file(args.pop_a)
    pop_b_raw = read_subset_file(args.pop_b)
    pop_a, missing_a = expand_population(pop_a_raw, all_sequences)
    pop_b, missing_b = expand_population(pop_b_raw, all_sequences)
    if args.verbose:
        print(f"Population A candidates: {len(pop_a_raw)}", file=sys.stderr)
        print(f"Population B candidates: {len(pop_b_raw)}", file=sys.stderr)
        print(f"Population A sequences matched: {len(pop_a)}", file=sys.stderr)
        print(f"Population B sequences matched: {len(pop_b)}", file=sys.stderr)
    if missing_a:
        print("Warning: {} identifiers from population A did not match any sequences".format(len(missing_a)), file=sys.stderr)
    if missing_b:
        print("Warning: {} identifiers from population B did not match any sequences".format(len(missing_b)), file=sys.stderr)
    if not pop_a or not pop_b:
        print("Error: No valid sequences found in one or both populations", file=sys.stderr)
        sys.exit(1)
    base_name = os.path.splitext(os.path.basename(args.similarity_file))[0]
    log_path = os.path.join(args.log_dir, f"{base_name}_fst.log")
    os.makedirs(args.log_dir, exist_ok=True)
    with open(log_path, 'w') as log_file:
        results = calculate_fst_dense(names, dense, pop_a, pop_b, sequence_length=args.length, round_digits=args.round,
                                      log_file=log_file)
    print(f"{results['fst']:.8f}\t{results['pi_a']:.8f}\t{results['pi_b']:.8f}\t"
          f"{results['pi_xy']:.8f}\t{results['dxy']:.8f}\t{results['da']:.8f}")
    if args.verbose:
        print(f"Detailed log saved to: {log_path}", file=sys.stderr)


if __name__ == "__main__":
    main()
