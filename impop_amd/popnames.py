"""Population-list -> sequence-name expansion (h-fst.py:18-82, 121-128).  String
handling only; mirrors the reference's matching rules exactly."""
from __future__ import annotations

import sys

_SUFFIX_MAP = (("_hap1", "#1#"), ("_hap2", "#2#"), ("_mat", "#1#"), ("_pat", "#2#"))  # h-fst.py:44-49


def canonicalize_identifier(identifier: str) -> str:
    """h-fst.py:18-61: assembly name -> PanSN prefix usable with str.startswith()."""
    if not identifier:
        return ""
    token = identifier.strip()
    if not token or token.startswith("#"):
        return ""
    if "_hprc" in token:  # trailing metadata, h-fst.py:41-42
        token = token.split("_hprc", 1)[0]
    for suffix, hap_tag in _SUFFIX_MAP:
        if token.endswith(suffix):
            return f"{token[:-len(suffix)]}{hap_tag}"
    if "#" in token:  # already carries a hap delimiter, h-fst.py:57-58
        return token if token.endswith("#") else f"{token}#"
    return f"{token}#"  # both haplotypes of the sample, h-fst.py:61


def expand_population(raw_ids, all_sequences):
    """h-fst.py:64-82 -> (expanded set, ids that matched nothing)"""
    expanded = set()
    missing = []
    for raw_id in raw_ids:
        prefix = canonicalize_identifier(raw_id)
        if not prefix:
            continue
        matches = {seq for seq in all_sequences if seq.startswith(prefix)}
        if matches:
            expanded.update(matches)
        else:
            missing.append(raw_id)
    return expanded, missing


def read_subset_file(filename):
    """h-fst.py:121-128"""
    try:
        with open(filename) as f:
            return set(line.strip() for line in f if line.strip() and not line.startswith("#"))
    except FileNotFoundError:
        print(f"Error: Subset file not found: {filename}", file=sys.stderr)
        sys.exit(1)
