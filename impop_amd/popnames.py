"""Population list -> sequence names (the matching rules of h-fst.py:18-82, 121-128): an assembly
name such as `HG00097_hap1_hprc_r2_v1.0.1` selects the PanSN sequences starting with `HG00097#1#`.
String handling only."""
from __future__ import annotations

import sys

# assembly-name endings and the haplotype digit they stand for, tried in this order
_HAPLOTYPE_OF_ENDING = (("_hap1", "1"), ("_hap2", "2"), ("_mat", "1"), ("_pat", "2"))


def canonicalize_identifier(identifier: str) -> str:
    """The `startswith` prefix an identifier selects; "" for blank lines and `#` comments."""
    name = (identifier or "").strip()
    if name == "" or name[0] == "#":
        return ""
    name = name.partition("_hprc")[0]  # release metadata behind the assembly name
    for ending, digit in _HAPLOTYPE_OF_ENDING:
        if name.endswith(ending):
            return name[: len(name) - len(ending)] + "#" + digit + "#"
    # a name that already ends a PanSN field is kept; anything else (a bare sample: both haplotypes) gets one
    return name if name.endswith("#") else name + "#"


def expand_population(raw_ids, all_sequences):
    """-> (set of sequence names selected by any identifier, identifiers that selected nothing)"""
    pool = tuple(all_sequences)
    selected, unmatched = set(), []
    for rid in raw_ids:
        key = canonicalize_identifier(rid)
        if not key:
            continue
        hits = [seq for seq in pool if seq.startswith(key)]
        if hits:
            selected.update(hits)
        else:
            unmatched.append(rid)
    return selected, unmatched


def read_subset_file(filename):
    """Non-blank lines that do not start with `#`, stripped; a missing file ends the process like the
    reference's reader does (message on stderr, exit code 1)."""
    try:
        handle = open(filename)
    except FileNotFoundError:
        print(f"Error: Subset file not found: {filename}", file=sys.stderr)
        sys.exit(1)
    with handle:
        return {ln.strip() for ln in handle if ln.strip() and ln[0] != "#"}
