#!/usr/bin/env python3
"""Drop-in for the reference's scripts/pica2.py CLI (pica2.py:172-228): same flags, same log
file, same stdout — `"{pi_per_site:.8f} (sequence length: L)"` — with the analysis on the GPU."""
import argparse
import os

import _bootstrap  # noqa: F401
from impop_amd.pica2 import analyze_dense, read_dense

if __name__ == "__main__":
    parser = argparse.ArgumentParser(description='Analyze similarity matrix with customizable threshold and sequence length normalization')
    parser.add_argument('input_file', help='Input file with similarity data (TSV format with group.a, group.b, estimated.identity columns)')
    parser.add_argument('--threshold', '-t', type=float, default=0.99,
                        help='Similarity threshold for grouping elements (default: 0.99)')
    parser.add_argument('--sequence-length', '-l', type=int, help='Sequence length for normalizing pi per site')
    parser.add_argument('--log-dir', '-d', type=str, default='.', help='Directory to save log file (default: current directory)')
    parser.add_argument('--round-digits', '-r', type=int, default=None,
                        help='Round similarity values to specified decimal places (default: no rounding)')
    args = parser.parse_args()

    base_name = os.path.splitext(os.path.basename(args.input_file))[0]
    log_filename = os.path.join(args.log_dir, f"{base_name}.log")
    os.makedirs(args.log_dir, exist_ok=True)
    names, dense, pair_count = read_dense(args.input_file, "pica2")  # native ingest; reference messages on errors
    with open(log_filename, 'w') as log_file:
        log_file.write("Nucleotide Diversity Analysis Log\n")
        log_file.write("=================================\n")
        log_file.write(f"Input file: {args.input_file}\n")
        log_file.write(f"Threshold: {args.threshold}\n")
        if args.sequence_length:
            log_file.write(f"Sequence length: {args.sequence_length}\n")
        if args.round_digits is not None:
            log_file.write(f"Similarity rounding: {args.round_digits} decimal places\n")
        log_file.write(f"Log file: {log_filename}\n\n")
        pi, pi_per_site = analyze_dense(names, dense, pair_count, threshold=args.threshold,
                                        sequence_length=args.sequence_length, log_file=log_file,
                                        round_digits=args.round_digits)
        log_file.write("\n" + "=" * 50 + "\n")
        log_file.write("FINAL RESULTS:\n")
        log_file.write(f"pi = {pi:.6f}\n")
        if pi_per_site is not None:
            log_file.write(f"pi per site = {pi_per_site:.8f}\n")
    if args.sequence_length:
        print(f"{pi_per_site:.8f} (sequence length: {args.sequence_length})")
    else:
        print(f"{pi:.6f} (sequence length: {args.sequence_length})")
