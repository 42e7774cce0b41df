"""Host-side objects over the C ABI: Context, BitMatrix, ScanPlan.

Everything here is marshalling — numpy arrays in, numpy structured records out;
all arithmetic runs in libimpop_hip.so on the GPU.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np

from . import _lib
from ._lib import (IDENTITY_DICE, IDENTITY_MATCH, KEEP_HAP_MAJOR, KEEP_SITE_BLOCKED, ImpopError, PairwiseParams,
                   PairwiseStats, ScanParams, SynthParams, Window, WindowStats, check)

STATS_DTYPE = np.dtype([
    ("n_sites", "<u4"), ("s_all", "<u4"), ("s_p", "<u4"), ("s_a", "<u4"), ("s_b", "<u4"), ("flags", "<u4"),
    ("sum_p", "<u8"), ("sum_a", "<u8"), ("sum_b", "<u8"), ("sum_ab", "<u8"),
    ("pi", "<f8"), ("pi_site", "<f8"), ("pi_a", "<f8"), ("pi_b", "<f8"), ("pi_xy", "<f8"), ("dxy", "<f8"),
    ("da", "<f8"), ("fst", "<f8"), ("tajima_d", "<f8")])
assert STATS_DTYPE.itemsize == 128

PAIRWISE_DTYPE = np.dtype([
    ("pi", "<f8"), ("pi_site", "<f8"), ("fst", "<f8"), ("pi_a", "<f8"), ("pi_b", "<f8"), ("pi_xy", "<f8"),
    ("dxy", "<f8"), ("da", "<f8"), ("tajima_d", "<f8"), ("n_groups", "<u4"), ("s_all", "<u4"), ("s_p", "<u4"),
    ("n_sites", "<u4"), ("reserved", "<u8")])
assert PAIRWISE_DTYPE.itemsize == 96

PAIR_DTYPE = np.dtype([("fst", "<f8"), ("pi_a", "<f8"), ("pi_b", "<f8"), ("pi_xy", "<f8"), ("dxy", "<f8"), ("da", "<f8")])

WINDOW_DTYPE = np.dtype([("site_begin", "<u8"), ("site_end", "<u8"), ("seq_len", "<u8")])

IDENTITY_KINDS = {"match": IDENTITY_MATCH, "dice": IDENTITY_DICE}


def pack_hap_major(mat01) -> np.ndarray:
    """0/1 array [n_hap, n_site] -> uint64 [n_hap, ceil(n_site/64)] (bit s&63 of word s>>6)."""
    m = np.ascontiguousarray(mat01, dtype=np.uint8)
    if m.ndim != 2:
        raise ValueError("presence matrix must be 2-D [haplotype, site]")
    n, W = m.shape
    words = max((W + 63) // 64, 1)
    pad = np.zeros((n, words * 64), dtype=np.uint8)
    pad[:, :W] = m
    return np.ascontiguousarray(np.packbits(pad, axis=1, bitorder="little")).view(np.uint64).reshape(n, words)


def unpack_hap_major(bits: np.ndarray, n_site: int) -> np.ndarray:
    b = np.ascontiguousarray(bits, dtype=np.uint64)
    return np.unpackbits(b.view(np.uint8), axis=1, bitorder="little")[:, :n_site]


def pack_mask(flags, n: int) -> np.ndarray:
    """Length-n membership flags (bool / 0-1) -> n-bit uint64 bitset."""
    f = np.ascontiguousarray(flags).astype(np.uint8).ravel()
    if f.size != n:
        raise ValueError(f"mask has {f.size} flags, expected {n}")
    words = max((n + 63) // 64, 1)
    pad = np.zeros(words * 64, dtype=np.uint8)
    pad[:n] = f != 0
    return np.packbits(pad, bitorder="little").view(np.uint64).copy()


def mask_from_indices(indices, n: int) -> np.ndarray:
    f = np.zeros(n, dtype=np.uint8)
    f[np.asarray(list(indices), dtype=np.int64)] = 1
    return pack_mask(f, n)


def _mask_ptr(mask, n):
    if mask is None:
        return None, None
    a = np.asarray(mask)
    if a.dtype != np.uint64:  # flags; uint64 arrays are taken as ready-made bitsets
        a = pack_mask(a, n)
    elif a.size != max((n + 63) // 64, 1):
        raise ValueError("bitset mask has the wrong number of words")
    a = np.ascontiguousarray(a, dtype=np.uint64)
    return a, a.ctypes.data_as(C.POINTER(C.c_uint64))


def make_windows(windows) -> np.ndarray:
    """[(begin, end[, seq_len])] or structured array -> WINDOW_DTYPE array.
    seq_len defaults to end-begin (LENGTH=end-start, run_pica2_impg.sh:133)."""
    if isinstance(windows, np.ndarray) and windows.dtype == WINDOW_DTYPE:
        return np.ascontiguousarray(windows)
    rows = list(windows)
    out = np.zeros(len(rows), dtype=WINDOW_DTYPE)
    if not rows:
        return out
    # column-wise (a BED file's worth of tuples costs a Python loop over fields otherwise: 1.5 us per field)
    begin = np.fromiter((int(r[0]) for r in rows), dtype=np.uint64, count=len(rows))
    end = np.fromiter((int(r[1]) for r in rows), dtype=np.uint64, count=len(rows))
    out["site_begin"], out["site_end"] = begin, end
    out["seq_len"] = np.fromiter((int(r[2]) if len(r) > 2 and r[2] is not None else max(int(r[1]) - int(r[0]), 0) for r in rows),
                                 dtype=np.uint64, count=len(rows))
    return out


def fixed_windows(n_site: int, size: int, step: Optional[int] = None, seq_len: Optional[int] = None) -> np.ndarray:
    """bedtools-makewindows-style tiling (doc/how_pi.md:42); the last window is clipped."""
    step = step or size
    starts = np.arange(0, max(n_site, 1), step, dtype=np.uint64)
    if step < size:
        starts = starts[starts + np.uint64(1) <= np.uint64(n_site)]
    out = np.zeros(len(starts), dtype=WINDOW_DTYPE)
    out["site_begin"] = starts
    out["site_end"] = np.minimum(starts + np.uint64(size), np.uint64(n_site))
    out["seq_len"] = (out["site_end"] - out["site_begin"]) if seq_len is None else seq_len
    return out[out["site_end"] > out["site_begin"]]


class Context:
    """One GPU + one HIP stream (impop_ctx)."""

    def __init__(self, device: int = 0, stream: Optional[int] = None):
        """stream: None = the context creates its own non-blocking HIP stream; otherwise the integer handle
        of an existing hipStream_t whose ordering the caller relies on (e.g. torch.cuda.Stream(dev).cuda_stream).
        Handle 0 — HIP's legacy null stream, which is what torch's DEFAULT stream reports — is refused: the C
        ABI reads NULL as "create a private stream", and silently doing that would leave the caller's
        collectives / copies unordered with the scans."""
        self._lib = _lib.load()
        self._h = C.c_void_p()
        if stream is not None and int(stream) == 0:
            raise ValueError("Context(stream=0): the null stream cannot be adopted; pass stream=None for a private stream "
                             "or the handle of an explicit stream (torch.cuda.Stream(dev).cuda_stream)")
        check(self._lib.impop_ctx_create(int(device), C.c_void_p(int(stream)) if stream is not None else None, C.byref(self._h)))
        self.device = int(device)

    @property
    def handle(self):
        if not self._h:
            raise ImpopError(_lib.E_INVALID, "context is closed")
        return self._h

    def device_name(self) -> str:
        buf = C.create_string_buffer(128)
        check(self._lib.impop_ctx_device_name(self.handle, buf, 128))
        return buf.value.decode()

    def synchronize(self) -> None:
        check(self._lib.impop_ctx_synchronize(self.handle))

    def gram_timing(self, enable: bool = True) -> None:
        """HIP events around every Gram launch of pairwise_scan on this context (impop_ctx_gram_timing)."""
        check(self._lib.impop_ctx_gram_timing(self.handle, 1 if enable else 0))

    def gram_elapsed(self):
        """-> (summed Gram-kernel ms, launches) since gram_timing(True)"""
        t, k = C.c_double(), C.c_uint64()
        check(self._lib.impop_ctx_gram_elapsed(self.handle, C.byref(t), C.byref(k)))
        return t.value, k.value

    def close(self) -> None:
        if self._h:
            self._lib.impop_ctx_destroy(self._h)
            self._h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- matrices -----------------------------------------------------------------
    def upload(self, bits_hap_major: np.ndarray, n_site: int, keep_hap_major: bool = True) -> "BitMatrix":
        b = np.ascontiguousarray(bits_hap_major, dtype=np.uint64)
        if b.ndim != 2:
            raise ValueError("bits must be [n_hap, words]")
        h = C.c_void_p()
        keep = KEEP_SITE_BLOCKED | (KEEP_HAP_MAJOR if keep_hap_major else 0)
        check(self._lib.impop_matrix_upload(self.handle, b.ctypes.data_as(C.POINTER(C.c_uint64)), b.shape[0], int(n_site),
                                            b.shape[1], keep, C.byref(h)))
        return BitMatrix(self, h)

    def upload_dense(self, mat01, keep_hap_major: bool = True) -> "BitMatrix":
        m = np.asarray(mat01)
        return self.upload(pack_hap_major(m), m.shape[1], keep_hap_major)

    def synthetic(self, n_hap: int, n_site: int, seed: int = 20251031, n_founder: int = 8, p_founder: float = 1e-3,
                  p_private_word: float = 3.2e-3, keep_hap_major: bool = False, site_begin: int = 0) -> "BitMatrix":
        """Sites [site_begin, site_begin + n_site) of the synthetic chromosome of `seed` (counter-based generator: a slab is
        a cut of the whole, impop_matrix_synthetic_slab); site 0 of the result is global site `site_begin`."""
        p = SynthParams(int(seed), int(n_founder), float(p_founder), float(p_private_word))
        h = C.c_void_p()
        keep = KEEP_SITE_BLOCKED | (KEEP_HAP_MAJOR if keep_hap_major else 0)
        check(self._lib.impop_matrix_synthetic_slab(self.handle, int(n_hap), int(site_begin), int(n_site), C.byref(p), keep, C.byref(h)))
        return BitMatrix(self, h)

    # ---- statistics on a given identity matrix (the .sim drop-in path) -----------------
    def pi_from_identity(self, ident: np.ndarray, threshold: float, round_digits: Optional[int], seq_len: Optional[int],
                         seed_rank=None, detail: bool = False):
        """-> (pi, pi_site, group_of, n_groups[, (sum_2pairs, n_pairs_with_data)]).  seed_rank: see
        impop_pi_from_identity (position of every element in the reference's set iteration order)."""
        a = np.ascontiguousarray(ident, dtype=np.float64)
        n = a.shape[0] if a.ndim == 2 else 0
        pi, ps, G = C.c_double(), C.c_double(), C.c_uint32()
        grp = np.zeros(max(n, 1), dtype=np.uint32)
        sr = None if seed_rank is None else np.ascontiguousarray(seed_rank, dtype=np.uint32)
        if sr is not None and sr.shape != (n,):
            raise ValueError("seed_rank must have one entry per element")
        det = _lib.Pica2Detail()
        check(self._lib.impop_pi_from_identity(self.handle, a.ctypes.data_as(C.POINTER(C.c_double)), n, float(threshold),
                                               -1 if round_digits is None else int(round_digits), int(seq_len or 0),
                                               sr.ctypes.data_as(C.POINTER(C.c_uint32)) if sr is not None and n else None,
                                               C.byref(pi), C.byref(ps), grp.ctypes.data_as(C.POINTER(C.c_uint32)),
                                               C.byref(G), C.byref(det)))
        out = (pi.value, ps.value, grp[:n].copy(), G.value)
        return out + ((det.sum_2pairs, int(det.n_pairs_with_data)),) if detail else out

    def pica2_pair_terms(self, ident: np.ndarray, round_digits: Optional[int], rep, group_size):
        """Identity of the representatives and (1 - sim) f_g f_h for every group pair g < h, row-major
        (the Step 2 table of pica2's log, pica2.py:125-145)."""
        a = np.ascontiguousarray(ident, dtype=np.float64)
        n = a.shape[0] if a.ndim == 2 else 0
        r = np.ascontiguousarray(rep, dtype=np.uint32)
        z = np.ascontiguousarray(group_size, dtype=np.uint32)
        G = len(r)
        npair = G * (G - 1) // 2
        sims, vals = np.zeros(max(npair, 1)), np.zeros(max(npair, 1))
        check(self._lib.impop_pica2_pair_terms(self.handle, a.ctypes.data_as(C.POINTER(C.c_double)), n,
                                               -1 if round_digits is None else int(round_digits),
                                               r.ctypes.data_as(C.POINTER(C.c_uint32)), z.ctypes.data_as(C.POINTER(C.c_uint32)), G,
                                               sims.ctypes.data_as(C.POINTER(C.c_double)), vals.ctypes.data_as(C.POINTER(C.c_double))))
        return sims[:npair], vals[:npair]

    def fst_from_identity(self, ident: np.ndarray, in_a, in_b, seq_len: Optional[int], round_digits: Optional[int]):
        a = np.ascontiguousarray(ident, dtype=np.float64)
        n = a.shape[0] if a.ndim == 2 else 0
        fa = np.ascontiguousarray(in_a, dtype=np.uint8)
        fb = np.ascontiguousarray(in_b, dtype=np.uint8)
        out = np.zeros(6)
        cnt = np.zeros(6, dtype=np.uint64)
        check(self._lib.impop_fst_from_identity(self.handle, a.ctypes.data_as(C.POINTER(C.c_double)), n,
                                                fa.ctypes.data_as(C.POINTER(C.c_uint8)), fb.ctypes.data_as(C.POINTER(C.c_uint8)),
                                                int(seq_len) if seq_len and seq_len > 0 else 0,
                                                -1 if round_digits is None else int(round_digits),
                                                out.ctypes.data_as(C.POINTER(C.c_double)), cnt.ctypes.data_as(C.POINTER(C.c_uint64))))
        return out, cnt

    def fst_grouped_from_identity(self, ident: np.ndarray, in_a, in_b, threshold: float, seq_len: Optional[int],
                                  round_digits: Optional[int], seed_rank=None):
        a = np.ascontiguousarray(ident, dtype=np.float64)
        n = a.shape[0] if a.ndim == 2 else 0
        fa = np.ascontiguousarray(in_a, dtype=np.uint8)
        fb = np.ascontiguousarray(in_b, dtype=np.uint8)
        out = np.zeros(6)
        cnt = np.zeros(6, dtype=np.uint64)
        sr = None if seed_rank is None else np.ascontiguousarray(seed_rank, dtype=np.uint32)
        if sr is not None and sr.shape != (n,):
            raise ValueError("seed_rank must have one entry per element")
        check(self._lib.impop_fst_grouped_from_identity(self.handle, a.ctypes.data_as(C.POINTER(C.c_double)), n,
                                                        fa.ctypes.data_as(C.POINTER(C.c_uint8)), fb.ctypes.data_as(C.POINTER(C.c_uint8)),
                                                        float(threshold), int(seq_len) if seq_len and seq_len > 0 else 0,
                                                        -1 if round_digits is None else int(round_digits),
                                                        sr.ctypes.data_as(C.POINTER(C.c_uint32)) if sr is not None and n else None,
                                                        out.ctypes.data_as(C.POINTER(C.c_double)), cnt.ctypes.data_as(C.POINTER(C.c_uint64))))
        return out, cnt

    def tajimas_d(self, n, S, pi, components: bool = False):
        n_a = np.ascontiguousarray(np.atleast_1d(n), dtype=np.int64)
        S_a = np.ascontiguousarray(np.atleast_1d(S), dtype=np.float64)
        pi_a = np.ascontiguousarray(np.atleast_1d(pi), dtype=np.float64)
        cnt = n_a.size
        D = np.zeros(cnt)
        comps = np.zeros((cnt, 10)) if components else None
        check(self._lib.impop_tajimas_d(self.handle, n_a.ctypes.data_as(C.POINTER(C.c_int64)),
                                        S_a.ctypes.data_as(C.POINTER(C.c_double)), pi_a.ctypes.data_as(C.POINTER(C.c_double)),
                                        cnt, D.ctypes.data_as(C.POINTER(C.c_double)),
                                        comps.ctypes.data_as(C.POINTER(C.c_double)) if components else None))
        return (D, comps) if components else D

    def cluster_from_identity(self, ident: np.ndarray, threshold: float):
        a = np.ascontiguousarray(ident, dtype=np.float64)
        n = a.shape[0] if a.ndim == 2 else 0
        cl = np.zeros(max(n, 1), dtype=np.uint32)
        sz = np.zeros(max(n, 1), dtype=np.uint32)
        K = C.c_uint32()
        check(self._lib.impop_cluster_from_identity(self.handle, a.ctypes.data_as(C.POINTER(C.c_double)), n, float(threshold),
                                                    cl.ctypes.data_as(C.POINTER(C.c_uint32)), C.byref(K),
                                                    sz.ctypes.data_as(C.POINTER(C.c_uint32))))
        return cl[:n].copy(), K.value, sz[: K.value].copy()

    def py_round(self, x, ndigits: int) -> np.ndarray:
        a = np.ascontiguousarray(np.atleast_1d(x), dtype=np.float64)
        out = np.zeros_like(a)
        check(self._lib.impop_py_round(self.handle, a.ctypes.data_as(C.POINTER(C.c_double)), a.size, int(ndigits),
                                       out.ctypes.data_as(C.POINTER(C.c_double))))
        return out


class BitMatrix:
    """A haplotype x site presence matrix resident in HBM (impop_matrix)."""

    def __init__(self, ctx: Context, handle):
        self.ctx = ctx
        self._h = handle
        n, s, b, bps = C.c_uint32(), C.c_uint64(), C.c_uint64(), C.c_uint32()
        check(ctx._lib.impop_matrix_info(handle, C.byref(n), C.byref(s), C.byref(b), C.byref(bps)))
        self.n_hap, self.n_site, self.device_bytes, self.bytes_per_site = n.value, s.value, b.value, bps.value

    @property
    def handle(self):
        if not self._h:
            raise ImpopError(_lib.E_INVALID, "matrix is freed")
        return self._h

    def free(self) -> None:
        """Release the device memory.  Raises if scan plans created from this matrix are still alive."""
        if self._h:
            check(self.ctx._lib.impop_matrix_free(self.ctx.handle if self.ctx._h else None, self._h))
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass

    def set_site_weights(self, weights) -> None:
        """Per-site weights (bp per column; None removes them): scans return the records of the bp-expanded
        matrix while the segregating-site counts keep counting columns (impop_matrix_set_site_weights)."""
        if weights is None:
            check(self.ctx._lib.impop_matrix_set_site_weights(self.ctx.handle, self.handle, None))
            return
        w = np.ascontiguousarray(weights, dtype=np.uint32)
        if w.shape != (self.n_site,):
            raise ValueError("weights must have one entry per site")
        check(self.ctx._lib.impop_matrix_set_site_weights(self.ctx.handle, self.handle, w.ctypes.data_as(C.POINTER(C.c_uint32))))

    def compact(self) -> "BitMatrix":
        """A new matrix holding only the sites that are variable among all haplotypes.  Scans of it take
        windows in THIS matrix's site coordinates and return the same records (impop_matrix_compact)."""
        out = C.c_void_p()
        check(self.ctx._lib.impop_matrix_compact(self.ctx.handle, self.handle, C.byref(out)))
        bm = BitMatrix(self.ctx, out)
        bm.compacted_from_sites = self.n_site
        return bm

    def positions(self, first: int = 0, count: Optional[int] = None) -> np.ndarray:
        """Original site index of the kept sites of a compacted matrix."""
        count = self.n_site - first if count is None else count
        out = np.zeros(max(count, 0), dtype=np.uint64)
        check(self.ctx._lib.impop_matrix_positions(self.handle, int(first), int(count), out.ctypes.data_as(C.POINTER(C.c_uint64)), None))
        return out

    def download(self, site_begin: int = 0, site_end: Optional[int] = None) -> np.ndarray:
        site_end = self.n_site if site_end is None else site_end
        words = max((site_end - site_begin + 63) // 64, 1)
        out = np.zeros((self.n_hap, words), dtype=np.uint64)
        check(self.ctx._lib.impop_matrix_download(self.ctx.handle, self.handle, int(site_begin), int(site_end),
                                                  out.ctypes.data_as(C.POINTER(C.c_uint64)), words))
        return out

    def plan(self, windows, mask_p=None, mask_a=None, mask_b=None, d_pi_mode: int = 0, s_scope: int = 0,
             tile_blocks: int = 0) -> "ScanPlan":
        return ScanPlan(self, make_windows(windows), mask_p, mask_a, mask_b, d_pi_mode, s_scope, tile_blocks)

    def scan(self, windows, mask_p=None, mask_a=None, mask_b=None, d_pi_mode: int = 0, s_scope: int = 0,
             tile_blocks: int = 0) -> np.ndarray:
        """pi + Hudson Fst + Tajima's D + S for every window in one streaming pass."""
        p = self.plan(windows, mask_p, mask_a, mask_b, d_pi_mode, s_scope, tile_blocks)
        try:
            p.launch()
            return p.fetch()
        finally:
            p.destroy()

    def scan_multi(self, windows, pops) -> np.ndarray:
        """All-pairs Hudson Fst of K disjoint populations in one pass -> array [n_windows, K(K-1)/2] of
        (fst, pi_a, pi_b, pi_xy, dxy, da); pair order (0,1),(0,2),...,(1,2),..."""
        w = make_windows(windows)
        K = len(pops)
        packed = np.concatenate([_mask_ptr(p, self.n_hap)[0] for p in pops]).astype(np.uint64)
        out = np.zeros((len(w), K * (K - 1) // 2), dtype=PAIR_DTYPE)
        check(self.ctx._lib.impop_scan_multi(self.ctx.handle, self.handle, w.ctypes.data_as(C.POINTER(Window)), len(w),
                                             packed.ctypes.data_as(C.POINTER(C.c_uint64)), K,
                                             out.ctypes.data_as(C.POINTER(_lib.PairStats))))
        return out

    def afs(self, windows, mask=None) -> np.ndarray:
        """Allele-frequency spectrum per window: out[w, c] = #sites with c carriers among `mask`."""
        w = make_windows(windows)
        keep, ptr = _mask_ptr(mask, self.n_hap)
        nP = self.n_hap if mask is None else int(np.unpackbits(keep.view(np.uint8), bitorder="little")[: self.n_hap].sum())
        out = np.zeros((len(w), nP + 1), dtype=np.uint32)
        check(self.ctx._lib.impop_afs(self.ctx.handle, self.handle, w.ctypes.data_as(C.POINTER(Window)), len(w), ptr,
                                      out.ctypes.data_as(C.POINTER(C.c_uint32))))
        return out

    def site_counts(self, site_begin: int, site_end: int, mask=None) -> np.ndarray:
        out = np.zeros(max(site_end - site_begin, 0), dtype=np.uint32)
        keep, ptr = _mask_ptr(mask, self.n_hap)
        check(self.ctx._lib.impop_site_counts(self.ctx.handle, self.handle, ptr, int(site_begin), int(site_end),
                                              out.ctypes.data_as(C.POINTER(C.c_uint32))))
        return out

    def ehh(self, site_begin: int, site_end: int, mask=None, reverse: bool = False) -> np.ndarray:
        """calc_EHH (scripts/wip/ehhgfa.py:6-21) of the members of `mask` over [site_begin, site_end)."""
        out = np.zeros(max(site_end - site_begin, 0))
        keep, ptr = _mask_ptr(mask, self.n_hap)
        check(self.ctx._lib.impop_ehh(self.ctx.handle, self.handle, int(site_begin), int(site_end), ptr, 1 if reverse else 0,
                                      out.ctypes.data_as(C.POINTER(C.c_double)), None))
        return out

    def pairwise_counts(self, site_begin: int, site_end: int) -> np.ndarray:
        out = np.zeros((self.n_hap, self.n_hap), dtype=np.int32)
        check(self.ctx._lib.impop_pairwise_counts(self.ctx.handle, self.handle, int(site_begin), int(site_end),
                                                  out.ctypes.data_as(C.POINTER(C.c_int32))))
        return out

    def pairwise_identity(self, site_begin: int, site_end: int, kind: str = "match") -> np.ndarray:
        out = np.zeros((self.n_hap, self.n_hap))
        check(self.ctx._lib.impop_pairwise_identity(self.ctx.handle, self.handle, int(site_begin), int(site_end),
                                                    IDENTITY_KINDS[kind], out.ctypes.data_as(C.POINTER(C.c_double))))
        return out

    def pairwise_scan(self, windows, mask_p=None, mask_a=None, mask_b=None, kind: str = "match", threshold: float = 0.99,
                      round_digits: Optional[int] = None, d_pi_mode: int = 0, s_scope: int = 0,
                      fst_method: str = "direct") -> np.ndarray:
        """Full pica2 / h-fst semantics (threshold grouping, rounding) per window from the bit matrix;
        fst_method='grouped' switches the Fst fields to hud.py's grouped method at the same threshold."""
        w = make_windows(windows)
        out = np.zeros(len(w), dtype=PAIRWISE_DTYPE)
        prm = PairwiseParams(C.sizeof(PairwiseParams), IDENTITY_KINDS[kind], float(threshold),
                             -1 if round_digits is None else int(round_digits), int(d_pi_mode), int(s_scope),
                             {"direct": 0, "grouped": 1}[fst_method])
        kp, pp = _mask_ptr(mask_p, self.n_hap)
        ka, pa = _mask_ptr(mask_a, self.n_hap)
        kb, pb = _mask_ptr(mask_b, self.n_hap)
        check(self.ctx._lib.impop_pairwise_scan(self.ctx.handle, self.handle, w.ctypes.data_as(C.POINTER(Window)), len(w),
                                                pp, pa, pb, C.byref(prm), out.ctypes.data_as(C.POINTER(PairwiseStats))))
        return out


class ScanPlan:
    """Pre-planned windowed scan (impop_scan_plan): tile tables live on the device, so
    launch() is two kernel launches with no host synchronisation."""

    def __init__(self, matrix: BitMatrix, windows: np.ndarray, mask_p, mask_a, mask_b, d_pi_mode, s_scope, tile_blocks):
        self.matrix = matrix
        self.windows = windows
        lib = matrix.ctx._lib
        prm = ScanParams(C.sizeof(ScanParams), int(d_pi_mode), int(s_scope), int(tile_blocks))
        kp, pp = _mask_ptr(mask_p, matrix.n_hap)
        ka, pa = _mask_ptr(mask_a, matrix.n_hap)
        kb, pb = _mask_ptr(mask_b, matrix.n_hap)
        self._h = C.c_void_p()
        check(lib.impop_scan_plan_create(matrix.ctx.handle, matrix.handle, windows.ctypes.data_as(C.POINTER(Window)),
                                         len(windows), pp, pa, pb, C.byref(prm), C.byref(self._h)))
        t, b = C.c_uint64(), C.c_uint64()
        check(lib.impop_scan_plan_info(self._h, C.byref(t), C.byref(b)))
        self.n_tiles, self.bytes_streamed = t.value, b.value
        self.n_windows = len(windows)

    def set_masks(self, mask_p=None, mask_a=None, mask_b=None) -> None:
        """Swap the subset / population masks; the tile tables (windows) are kept."""
        n = self.matrix.n_hap
        kp, pp = _mask_ptr(mask_p, n)
        ka, pa = _mask_ptr(mask_a, n)
        kb, pb = _mask_ptr(mask_b, n)
        check(self.matrix.ctx._lib.impop_scan_plan_set_masks(self._h, pp, pa, pb))

    def launch(self, d_out: Optional[int] = None) -> None:
        check(self.matrix.ctx._lib.impop_scan_plan_launch(self._h, C.c_void_p(d_out) if d_out else None))

    def fetch(self) -> np.ndarray:
        out = np.zeros(self.n_windows, dtype=STATS_DTYPE)
        check(self.matrix.ctx._lib.impop_scan_plan_fetch(self._h, out.ctypes.data_as(C.POINTER(WindowStats))))
        return out

    def timing(self, enable: bool = True) -> None:
        check(self.matrix.ctx._lib.impop_scan_plan_timing(self._h, 1 if enable else 0))

    def elapsed(self):
        """-> (summed streaming-kernel ms, launches) since timing(True)"""
        t, k = C.c_double(), C.c_uint64()
        check(self.matrix.ctx._lib.impop_scan_plan_elapsed(self._h, C.byref(t), C.byref(k)))
        return t.value, k.value

    def destroy(self) -> None:
        if self._h:
            self.matrix.ctx._lib.impop_scan_plan_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


# ---- multi-GPU entries of the C ABI ------------------------------------------------------------------

def shard_windows_c(windows, n_shards: int, shard: int):
    """impop_shard_windows -> (first_window, n_windows, slab_begin, slab_end)"""
    w = make_windows(windows)
    out = [C.c_uint64() for _ in range(4)]
    check(_lib.load().impop_shard_windows(w.ctypes.data_as(C.POINTER(Window)), len(w), int(n_shards), int(shard),
                                          *[C.byref(x) for x in out]))
    return tuple(int(x.value) for x in out)


def scan_sharded(slabs, slab_site_begin, windows, mask_p=None, mask_a=None, mask_b=None, d_pi_mode: int = 0, s_scope: int = 0,
                 tile_blocks: int = 0) -> np.ndarray:
    """One process, several contexts (impop_scan_sharded): slabs[k] is a BitMatrix resident on its own context and
    holds the sites [slab_site_begin[k], ...) of the chromosome; `windows` in chromosome coordinates.  Returns the
    records in the order of `windows`."""
    lib = _lib.load()
    w = make_windows(windows)
    n = len(slabs)
    n_hap = slabs[0].n_hap
    ctxs = (C.c_void_p * n)(*[s.ctx.handle for s in slabs])
    mats = (C.c_void_p * n)(*[s.handle for s in slabs])
    begins = np.ascontiguousarray(slab_site_begin, dtype=np.uint64)
    prm = ScanParams(C.sizeof(ScanParams), int(d_pi_mode), int(s_scope), int(tile_blocks))
    kp, pp = _mask_ptr(mask_p, n_hap)
    ka, pa = _mask_ptr(mask_a, n_hap)
    kb, pb = _mask_ptr(mask_b, n_hap)
    out = np.zeros(len(w), dtype=STATS_DTYPE)
    check(lib.impop_scan_sharded(ctxs, mats, begins.ctypes.data_as(C.POINTER(C.c_uint64)), n,
                                 w.ctypes.data_as(C.POINTER(Window)), len(w), pp, pa, pb, C.byref(prm),
                                 out.ctypes.data_as(C.POINTER(WindowStats))))
    return out


def pairwise_scan_sharded(slabs, slab_site_begin, windows, mask_p=None, mask_a=None, mask_b=None, kind: str = "match",
                          threshold: float = 0.99, round_digits: Optional[int] = None, d_pi_mode: int = 0, s_scope: int = 0,
                          fst_method: str = "direct") -> np.ndarray:
    """The all-pairs mode over several contexts (impop_pairwise_scan_sharded): like scan_sharded, with
    BitMatrix.pairwise_scan's parameters; every slab needs its hap-major operand (keep_hap_major=True)."""
    lib = _lib.load()
    w = make_windows(windows)
    n = len(slabs)
    n_hap = slabs[0].n_hap
    ctxs = (C.c_void_p * n)(*[s.ctx.handle for s in slabs])
    mats = (C.c_void_p * n)(*[s.handle for s in slabs])
    begins = np.ascontiguousarray(slab_site_begin, dtype=np.uint64)
    prm = PairwiseParams(C.sizeof(PairwiseParams), IDENTITY_KINDS[kind], float(threshold),
                         -1 if round_digits is None else int(round_digits), int(d_pi_mode), int(s_scope),
                         {"direct": 0, "grouped": 1}[fst_method])
    kp, pp = _mask_ptr(mask_p, n_hap)
    ka, pa = _mask_ptr(mask_a, n_hap)
    kb, pb = _mask_ptr(mask_b, n_hap)
    out = np.zeros(len(w), dtype=PAIRWISE_DTYPE)
    check(lib.impop_pairwise_scan_sharded(ctxs, mats, begins.ctypes.data_as(C.POINTER(C.c_uint64)), n,
                                          w.ctypes.data_as(C.POINTER(Window)), len(w), pp, pa, pb, C.byref(prm),
                                          out.ctypes.data_as(C.POINTER(PairwiseStats))))
    return out


def _preload_torch_rccl() -> None:
    """One RCCL per process: if PyTorch-ROCm is installed, load ITS librccl (soname librccl.so.1) first, by path,
    so that the library's dlopen("librccl.so.1") and a later `import torch` bind to the same copy."""
    import importlib.util
    import os
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec and spec.submodule_search_locations:
        cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "librccl.so")
        if os.path.exists(cand):
            C.CDLL(cand, mode=C.RTLD_GLOBAL)


class Comm:
    """One process per GPU: an RCCL communicator bound to a Context (impop_comm).  Rank 0 makes the id with
    Comm.unique_id() and ships the 128 bytes to the other ranks out of band."""

    ID_BYTES = 128

    @staticmethod
    def unique_id() -> bytes:
        _preload_torch_rccl()
        buf = C.create_string_buffer(Comm.ID_BYTES)
        check(_lib.load().impop_comm_unique_id(buf))
        return buf.raw

    def __init__(self, ctx: Context, unique_id: bytes, world: int, rank: int):
        if len(unique_id) != Comm.ID_BYTES:
            raise ValueError("unique_id must be 128 bytes")
        _preload_torch_rccl()
        self.ctx, self.world, self.rank = ctx, int(world), int(rank)
        self._h = C.c_void_p()
        check(ctx._lib.impop_comm_create(ctx.handle, unique_id, int(world), int(rank), C.byref(self._h)))

    def gather(self, d_local: int, bytes_per_rank: int, d_all: int) -> None:
        """ncclAllGather on the context's stream (device pointers, no host sync)."""
        check(self.ctx._lib.impop_gather(self._h, C.c_void_p(d_local), int(bytes_per_rank), C.c_void_p(d_all)))

    def gather_records(self, plan: "ScanPlan", n_total_windows: int) -> np.ndarray:
        """All-gather the records of `plan` (this rank's shard of n_total_windows) -> every window, global order."""
        d = C.c_void_p()
        check(self.ctx._lib.impop_scan_plan_device_records(plan._h, C.byref(d)))
        out = np.zeros(int(n_total_windows), dtype=STATS_DTYPE)
        check(self.ctx._lib.impop_gather_records(self._h, d, int(n_total_windows), out.ctypes.data_as(C.POINTER(WindowStats))))
        return out

    def allreduce_i64(self, d_values: int, count: int) -> None:
        check(self.ctx._lib.impop_allreduce_i64(self._h, C.c_void_p(d_values), int(count)))

    def close(self) -> None:
        if self._h:
            self.ctx._lib.impop_comm_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
