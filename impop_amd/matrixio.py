"""On-disk container of a presence matrix (numpy .npz): what the batch driver reads.

keys  bits      uint64 [n_hap, ceil(n_site/64)]   hap-major bit rows (bit s&63 of word s>>6)
      n_site    int
      names     str [n_hap]                       haplotype / path names (PanSN where available)
      site_pos  int64 [n_site]  (optional)        reference bp coordinate of every site, non-decreasing;
                                                  absent => site index == bp - origin
      origin    int             (optional)        bp coordinate of site 0 when site_pos is absent
      contig    str             (optional)        reference contig the coordinates refer to
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional

import numpy as np

from .engine import pack_hap_major


@dataclass
class MatrixFile:
    bits: np.ndarray
    n_site: int
    names: List[str]
    site_pos: Optional[np.ndarray] = None
    site_weight: Optional[np.ndarray] = None  # bp per column (node-level matrices); None = 1 each
    origin: int = 0
    contig: str = ""

    @property
    def n_hap(self) -> int:
        return self.bits.shape[0]

    def site_range(self, start: int, end: int):
        """BED interval [start, end) in bp on `contig` -> [site_begin, site_end)."""
        if self.site_pos is None:
            b = min(max(start - self.origin, 0), self.n_site)
            e = min(max(end - self.origin, 0), self.n_site)
            return b, max(e, b)
        b = int(np.searchsorted(self.site_pos, start, side="left"))
        e = int(np.searchsorted(self.site_pos, end, side="left"))
        return b, max(e, b)


def save_matrix(path: str, m: MatrixFile) -> None:
    d = {"bits": np.ascontiguousarray(m.bits, dtype=np.uint64), "n_site": np.int64(m.n_site),
         "names": np.array(m.names, dtype=str), "origin": np.int64(m.origin), "contig": np.array(m.contig)}
    if m.site_pos is not None:
        d["site_pos"] = np.ascontiguousarray(m.site_pos, dtype=np.int64)
    if m.site_weight is not None:
        d["site_weight"] = np.ascontiguousarray(m.site_weight, dtype=np.uint32)
    np.savez_compressed(path, **d)


def load_matrix(path: str) -> MatrixFile:
    with np.load(path, allow_pickle=False) as z:
        return MatrixFile(bits=z["bits"], n_site=int(z["n_site"]), names=[str(x) for x in z["names"]],
                          site_pos=z["site_pos"] if "site_pos" in z.files else None,
                          site_weight=z["site_weight"] if "site_weight" in z.files else None,
                          origin=int(z["origin"]) if "origin" in z.files else 0,
                          contig=str(z["contig"]) if "contig" in z.files else "")


def from_dense(mat01, names, **kw) -> MatrixFile:
    m = np.asarray(mat01)
    return MatrixFile(bits=pack_hap_major(m), n_site=m.shape[1], names=list(names), **kw)
