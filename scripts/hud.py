#!/usr/bin/env python3
"""Drop-in for the reference's scripts/hudson/hud.py command line (hud.py:310-415): h-fst's flags plus
`-m direct|grouped` and `-t`; population files hold exact sequence names; same stderr notes, `<basename>_fst.log`
and six `.8f` fields on stdout."""
import sys

import _bootstrap  # noqa: F401
from _cli import log_path_for, make_parser
from impop_amd.hud import calculate_fst_dense, read_dense, read_subset_file

FLAGS = (
    ("similarity_file", dict(help=".sim table: TSV with group.a, group.b, estimated.identity")),
    ("-a", "--pop-a", dict(required=True, help="sequence names of population A, one per line")),
    ("-b", "--pop-b", dict(required=True, help="sequence names of population B")),
    ("-l", "--length", dict(type=int, default=None, help="window length: the diversities are divided by it")),
    ("-r", "--round", dict(type=int, default=None, help="round identities to this many decimals")),
    ("-m", "--method", dict(choices=["direct", "grouped"], default="direct",
                            help="direct: mean over sequence pairs; grouped: frequency-weighted over groups of similar sequences")),
    ("-t", "--threshold", dict(type=float, default=0.999, help="grouping threshold of -m grouped (0.999)")),
    ("-d", "--log-dir", dict(default=".", help="where <basename>_fst.log goes (current directory)")),
    ("-v", "--verbose", dict(action="store_true", help="progress notes on stderr")),
)
FIELDS = ("fst", "pi_a", "pi_b", "pi_xy", "dxy", "da")


def note(text):
    print(text, file=sys.stderr)


def main():
    opt = make_parser("Hudson Fst (direct or grouped) of two populations from a pairwise identity table.", FLAGS).parse_args()
    chatty = note if opt.verbose else (lambda text: None)
    chatty(f"Reading similarity file: {opt.similarity_file}")
    # `known` = the reader's name set with the reference's insertion order: the intersections below, and through
    # them the seed order of the grouped method, then come out as in the reference
    names, dense, _, known = read_dense(opt.similarity_file, "hfst", with_elements=True)
    chatty("Reading population files...")
    pops = {tag: read_subset_file(path) for tag, path in (("A", opt.pop_a), ("B", opt.pop_b))}
    for tag in ("A", "B"):
        chatty(f"Population {tag}: {len(pops[tag])} sequences")
    chatty(f"Method: {opt.method}")
    if opt.method == "grouped":
        chatty(f"Grouping threshold: {opt.threshold}")
    for tag in ("A", "B"):
        absent = pops[tag] - known
        if absent:
            note(f"Warning: {len(absent)} sequences from population {tag} not found in similarity file")
    for tag in ("A", "B"):
        pops[tag] = pops[tag] & known  # hud.py:386-387
    if not pops["A"] or not pops["B"]:
        note("Error: No valid sequences found in one or both populations")
        sys.exit(1)
    log_name = log_path_for(opt.similarity_file, opt.log_dir, "_fst")
    with open(log_name, "w") as log:
        res = calculate_fst_dense(names, dense, pops["A"], pops["B"], sequence_length=opt.length, round_digits=opt.round,
                                  log_file=log, method=opt.method, threshold=opt.threshold)
    print("\t".join(f"{res[k]:.8f}" for k in FIELDS))
    chatty(f"Detailed log saved to: {log_name}")


if __name__ == "__main__":
    main()
