"""GPU-backed mirror of the reference's EHH scripts (scripts/wip/ehhgfa.py, ehh2.py): calc_EHH and
the fixed-width window loop of ehhgfa.main.  The pair work runs in libimpop_hip.so (impop_ehh)."""
from __future__ import annotations

import numpy as np

from .runtime import default_context


def _bit_planes(h: np.ndarray):
    """calc_EHH compares VALUES (ehh2.py's examples hold digits 0-9).  Equality of the values of a
    column == equality of the bit planes of the column's value ranks, so a general matrix becomes a
    0/1 matrix with a few columns per original column; `last[c]` is the last bit column of column c."""
    cols, last = [], []
    for c in range(h.shape[1]):
        _, rank = np.unique(h[:, c], return_inverse=True)
        nb = max(int(rank.max()).bit_length(), 1) if rank.size else 1
        for b in range(nb):
            cols.append(((rank >> b) & 1).astype(np.uint8))
        last.append(len(cols) - 1)
    return (np.stack(cols, axis=1) if cols else np.zeros((h.shape[0], 0), np.uint8)), np.array(last, dtype=np.int64)


def calc_EHH(haplotypes, ctx=None) -> np.ndarray:
    """ehhgfa.calc_EHH (ehhgfa.py:6-21): EHH[i] = round(pairs identical on columns 0..i / C(m,2), 3);
    fewer than two rows -> 500 everywhere."""
    h = np.asarray(haplotypes)
    m, W = h.shape
    if W == 0:
        return np.zeros(0)
    if m < 2:
        return np.full(W, fill_value=500)
    ctx = ctx or default_context()
    binary = bool(np.isin(h, (0, 1)).all())
    if binary:
        bits, last = (h != 0).astype(np.uint8), None
    else:
        bits, last = _bit_planes(h)
    bm = ctx.upload_dense(bits, keep_hap_major=False)
    try:
        out = bm.ehh(0, bits.shape[1])
    finally:
        bm.free()
    return out if last is None else out[last]


def ehh_pair(bm, site_begin: int, site_end: int, mask) -> np.ndarray:
    """ehhgfa.py:58-61 on a resident matrix: flip(calc_EHH(flipped b)) ++ calc_EHH(b), b = the window
    columns to the right of the test SNP, rows = `mask` (the reference builds both halves from b)."""
    return np.concatenate((np.flip(bm.ehh(site_begin, site_end, mask, reverse=True)), bm.ehh(site_begin, site_end, mask)))


def scan_windows(whole, test_snp: int, window_size: int, refpos: int, ctx=None):
    """ehhgfa.main's loop (ehhgfa.py:40-69): consecutive windows of `window_size` columns; for each
    allele at the test SNP (1-based `test_snp`) yield (window_name, colstart, colend, allele, 'REF'|'ALT',
    area) with area = cumsum(ehhvec)[-1].  `test_snp` < 1 (a negative numpy index in the reference) is
    rejected."""
    whole = np.asarray(whole, dtype=np.float64)
    if whole.ndim != 2:
        raise ValueError("haplotype matrix must be 2-D")
    if test_snp < 1:
        raise ValueError("-p is the 1-based position of the test SNP in the window (>= 1)")
    ctx = ctx or default_context()
    t = test_snp - 1
    bm = ctx.upload_dense((whole != 0).astype(np.uint8), keep_hap_major=False)  # ehhgfa.py:50: non-zero -> 1
    try:
        name, colstart, colend = 1, 0, window_size
        n_col = whole.shape[1]
        while colstart < n_col:
            hi = min(colend, n_col)
            if t >= hi - colstart:
                raise IndexError(f"index {t} is out of bounds for axis 1 with size {hi - colstart}")
            col = (whole[:, colstart + t] != 0).astype(np.float64)
            refall = col[refpos - 1]
            for al in np.unique(col):
                members = col == al
                if int(members.sum()) < 2:  # calc_EHH's np.full(W, 500) is an INTEGER vector (prints 12000, not 12000.0)
                    vec = np.full(2 * (hi - (colstart + t + 1)), 500)
                else:
                    vec = ehh_pair(bm, colstart + t + 1, hi, members)
                area = np.cumsum(vec)[-1]  # IndexError on an empty vector, like the reference
                yield name, colstart, colend, al, ('REF' if al == refall else 'ALT'), area
            colstart = colend
            colend = colstart + window_size
            name += 1
    finally:
        bm.free()
