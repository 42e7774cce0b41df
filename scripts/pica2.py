#!/usr/bin/env python3
"""Drop-in for the reference's scripts/pica2.py command line (pica2.py:172-228): same flags, same
`<log-dir>/<basename>.log`, same single stdout line `<value> (sequence length: L)` whose first token the
bash drivers read — the analysis itself runs on the GPU."""
import _bootstrap  # noqa: F401
from _cli import log_path_for, make_parser
from impop_amd.pica2 import analyze_dense, read_dense

FLAGS = (
    ("input_file", dict(help=".sim table: TSV with at least group.a, group.b, estimated.identity")),
    ("--threshold", "-t", dict(type=float, default=0.99, help="sequences more similar than this to a group's seed join it (0.99)")),
    ("--sequence-length", "-l", dict(type=int, help="divide pi by this length and print the per-site value")),
    ("--log-dir", "-d", dict(type=str, default=".", help="where <basename>.log goes (current directory)")),
    ("--round-digits", "-r", dict(type=int, default=None, help="round every identity to this many decimals first")),
)


def main():
    opt = make_parser("Nucleotide diversity of one window from its pairwise identity table.", FLAGS).parse_args()
    log_name = log_path_for(opt.input_file, opt.log_dir)
    # native ingest (the reference's messages on errors); `elements` = the reader's name set, rebuilt with the
    # reference's insertion order so that the grouping takes its seeds in the reference's order
    names, dense, n_rows, elements = read_dense(opt.input_file, "pica2", with_elements=True)
    head = ["Nucleotide Diversity Analysis Log", "=================================", f"Input file: {opt.input_file}",
            f"Threshold: {opt.threshold}"]
    if opt.sequence_length:
        head.append(f"Sequence length: {opt.sequence_length}")
    if opt.round_digits is not None:
        head.append(f"Similarity rounding: {opt.round_digits} decimal places")
    head.append(f"Log file: {log_name}\n")
    with open(log_name, "w") as log:
        log.write("\n".join(head) + "\n")
        pi, per_site = analyze_dense(names, dense, n_rows, threshold=opt.threshold, sequence_length=opt.sequence_length,
                                     log_file=log, round_digits=opt.round_digits, elements=elements)
        tail = ["", "=" * 50, "FINAL RESULTS:", f"pi = {pi:.6f}"]
        if per_site is not None:
            tail.append(f"pi per site = {per_site:.8f}")
        log.write("\n".join(tail) + "\n")
    value = f"{per_site:.8f}" if opt.sequence_length else f"{pi:.6f}"
    print(f"{value} (sequence length: {opt.sequence_length})")


if __name__ == "__main__":
    main()
