#!/usr/bin/env python3
"""Drop-in for the reference's scripts/h-fst.py command line (h-fst.py:251-342): same flags, stderr
diagnostics, `<log-dir>/<basename>_fst.log` and the six tab-separated `.8f` fields FST, pi_A, pi_B, pi_XY, Dxy,
Da on stdout (run_h-fst.sh:88 splits them)."""
import sys

import _bootstrap  # noqa: F401
from _cli import log_path_for, make_parser
from impop_amd.hfst import calculate_fst_dense, expand_population, read_dense, read_subset_file

FLAGS = (
    ("similarity_file", dict(help=".sim table: TSV with group.a, group.b, estimated.identity")),
    ("-a", "--pop-a", dict(required=True, help="assembly / sample identifiers of population A, one per line")),
    ("-b", "--pop-b", dict(required=True, help="the same for population B")),
    ("-l", "--length", dict(type=int, default=None, help="window length: the diversities are divided by it")),
    ("-r", "--round", dict(type=int, default=None, help="round identities to this many decimals")),
    ("-d", "--log-dir", dict(default=".", help="where <basename>_fst.log goes (current directory)")),
    ("-v", "--verbose", dict(action="store_true", help="progress notes on stderr")),
)
FIELDS = ("fst", "pi_a", "pi_b", "pi_xy", "dxy", "da")


def note(text):
    print(text, file=sys.stderr)


def main():
    opt = make_parser("Hudson Fst of two populations from a pairwise identity table.", FLAGS).parse_args()
    chatty = note if opt.verbose else (lambda text: None)
    chatty(f"Reading similarity file: {opt.similarity_file}")
    names, dense, _ = read_dense(opt.similarity_file, "hfst")  # native ingest; the reference's messages on errors
    known = set(names)
    chatty("Reading population files...")
    listed = {tag: read_subset_file(path) for tag, path in (("A", opt.pop_a), ("B", opt.pop_b))}
    members, unmatched = {}, {}
    for tag in ("A", "B"):
        members[tag], unmatched[tag] = expand_population(listed[tag], known)
    for tag in ("A", "B"):
        chatty(f"Population {tag} candidates: {len(listed[tag])}")
    for tag in ("A", "B"):
        chatty(f"Population {tag} sequences matched: {len(members[tag])}")
    for tag in ("A", "B"):
        if unmatched[tag]:
            note(f"Warning: {len(unmatched[tag])} identifiers from population {tag} did not match any sequences")
    if not members["A"] or not members["B"]:
        note("Error: No valid sequences found in one or both populations")
        sys.exit(1)
    log_name = log_path_for(opt.similarity_file, opt.log_dir, "_fst")
    with open(log_name, "w") as log:
        res = calculate_fst_dense(names, dense, members["A"], members["B"], sequence_length=opt.length, round_digits=opt.round,
                                  log_file=log)
    print("\t".join(f"{res[k]:.8f}" for k in FIELDS))
    chatty(f"Detailed log saved to: {log_name}")


if __name__ == "__main__":
    main()
