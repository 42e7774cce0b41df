"""ctypes front-end of the CPU oracle (oracle/impop_oracle.c).

TEST INFRASTRUCTURE ONLY — see the header of impop_oracle.c.  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module;
the product package (impop_amd/, scripts/) never does.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libimpop_oracle.so")

_u64p = C.POINTER(C.c_uint64)
_u32p = C.POINTER(C.c_uint32)
_u8p = C.POINTER(C.c_uint8)
_i64p = C.POINTER(C.c_int64)
_f64p = C.POINTER(C.c_double)


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "impop_oracle.c")
    if force or not os.path.exists(_SO) or (
        os.path.exists(src) and os.path.getmtime(src) > os.path.getmtime(_SO)
    ):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        L.oracle_py_round.restype = C.c_double
        L.oracle_py_round.argtypes = [C.c_double, C.c_int]
        L.oracle_tajimas_d.argtypes = [C.c_int64, C.c_double, C.c_double, _f64p, _f64p]
        L.oracle_pica2.argtypes = [_f64p, C.c_uint32, C.c_double, C.c_int, C.c_double, _f64p, _f64p, _u32p, _u32p]
        L.oracle_pica2_seeded.argtypes = [_f64p, C.c_uint32, C.c_double, C.c_int, C.c_double, _u32p, _f64p, _f64p, _u32p, _u32p]
        L.oracle_hud_grouped_seeded.argtypes = [_f64p, C.c_uint32, _u8p, _u8p, C.c_double, C.c_int, C.c_double, _u32p, _f64p, _u64p]
        L.oracle_hfst.argtypes = [_f64p, C.c_uint32, _u8p, _u8p, C.c_double, C.c_int, _f64p, _u64p]
        L.oracle_hud_grouped.argtypes = [_f64p, C.c_uint32, _u8p, _u8p, C.c_double, C.c_int, C.c_double, _f64p, _u64p]
        L.oracle_ehh.argtypes = [_u64p, C.c_uint64, C.c_uint32, C.c_uint64, C.c_uint64, _u8p, C.c_int, _f64p]
        L.oracle_af_cluster.argtypes = [_f64p, C.c_uint32, C.c_double, _u32p, _u32p, _u32p]
        L.oracle_pairwise_counts.argtypes = [_u64p, C.c_uint64, C.c_uint32, C.c_uint64, C.c_uint64, _i64p]
        L.oracle_identity.argtypes = [_i64p, C.c_uint32, C.c_uint64, C.c_int, _f64p]
        L.oracle_site_scan.argtypes = [_u64p, C.c_uint64, C.c_uint32, C.c_uint64, C.c_uint64, _u64p, _u64p, _u64p, _u32p, _u64p]
        for f in (L.oracle_window_allpairs, L.oracle_window_sitecount):
            f.argtypes = [_u64p, C.c_uint64, C.c_uint32, C.c_uint64, C.c_uint64, _u64p, _u64p, _u64p,
                          C.c_double, C.c_int, C.c_int, _u32p, _u64p, _f64p]
        L.oracle_site_scan_sitemajor.argtypes = [_u64p, C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint64,
                                                 _u64p, _u64p, _u64p, _u32p, _u64p]
        L.oracle_site_scan_sitemajor_windows.argtypes = [_u64p, C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint64,
                                                         _u64p, _u64p, _u64p, C.c_int, _u32p, _u64p]
        L.oracle_to_sitemajor.argtypes = [_u64p, C.c_uint64, C.c_uint32, C.c_uint64, _u64p, C.c_uint32]
        _lib = L
    return _lib


def _p(a, t):
    return a.ctypes.data_as(t)


def py_round(x: float, nd: int) -> float:
    return lib().oracle_py_round(float(x), int(nd))


def tajimas_d(n: int, S: float, pi: float):
    D = C.c_double()
    comps = np.zeros(10)
    rc = lib().oracle_tajimas_d(int(n), float(S), float(pi), C.byref(D), _p(comps, _f64p))
    if rc:
        raise ValueError("n must be >= 2" if n < 2 else "S and pi must be non-negative")
    return D.value, comps


def _dense(sim):
    a = np.ascontiguousarray(sim, dtype=np.float64)
    assert a.ndim == 2 and a.shape[0] == a.shape[1]
    return a


def pica2(sim, threshold=1.0, seq_len=None, round_digits=None, seed_rank=None):
    """seed_rank[i] = position of element i in the reference's `set(elements)` iteration order; None = index order"""
    a = _dense(sim)
    n = a.shape[0]
    pi, ps = C.c_double(), C.c_double()
    grp = np.zeros(max(n, 1), dtype=np.uint32)
    G = C.c_uint32()
    sr = None if seed_rank is None else np.ascontiguousarray(seed_rank, dtype=np.uint32)
    lib().oracle_pica2_seeded(_p(a, _f64p), n, float(threshold), -1 if round_digits is None else int(round_digits),
                              float(seq_len or 0), _p(sr, _u32p) if sr is not None else None, C.byref(pi), C.byref(ps),
                              _p(grp, _u32p), C.byref(G))
    return pi.value, ps.value, grp[:n].copy(), G.value


def hfst(sim, in_a, in_b, seq_len=None, round_digits=None):
    a = _dense(sim)
    n = a.shape[0]
    fa = np.ascontiguousarray(in_a, dtype=np.uint8)
    fb = np.ascontiguousarray(in_b, dtype=np.uint8)
    out = np.zeros(6)
    cnt = np.zeros(6, dtype=np.uint64)
    lib().oracle_hfst(_p(a, _f64p), n, _p(fa, _u8p), _p(fb, _u8p), float(seq_len or 0),
                      -1 if round_digits is None else int(round_digits), _p(out, _f64p), _p(cnt, _u64p))
    keys = ("fst", "pi_a", "pi_b", "pi_xy", "dxy", "da")
    return dict(zip(keys, out.tolist())), cnt


def hud_grouped(sim, in_a, in_b, threshold=0.999, seq_len=None, round_digits=None, seed_rank=None):
    a = _dense(sim)
    n = a.shape[0]
    fa = np.ascontiguousarray(in_a, dtype=np.uint8)
    fb = np.ascontiguousarray(in_b, dtype=np.uint8)
    out = np.zeros(6)
    cnt = np.zeros(6, dtype=np.uint64)
    sr = None if seed_rank is None else np.ascontiguousarray(seed_rank, dtype=np.uint32)
    lib().oracle_hud_grouped_seeded(_p(a, _f64p), n, _p(fa, _u8p), _p(fb, _u8p), float(threshold),
                                    -1 if round_digits is None else int(round_digits), float(seq_len or 0),
                                    _p(sr, _u32p) if sr is not None else None, _p(out, _f64p), _p(cnt, _u64p))
    keys = ("fst", "pi_a", "pi_b", "pi_xy", "dxy", "da")
    return dict(zip(keys, out.tolist())), cnt


def af_cluster(sim, threshold):
    a = _dense(sim)
    n = a.shape[0]
    cl = np.zeros(max(n, 1), dtype=np.uint32)
    sz = np.zeros(max(n, 1), dtype=np.uint32)
    K = C.c_uint32()
    lib().oracle_af_cluster(_p(a, _f64p), n, float(threshold), _p(cl, _u32p), C.byref(K), _p(sz, _u32p))
    return cl[:n].copy(), K.value, sz[: K.value].copy()


# ---- bit-matrix side ---------------------------------------------------------

def pack_hap_major(mat01) -> np.ndarray:
    """bool/0-1 array [n, W] -> uint64 [n, ceil(W/64)], bit s&63 of word s>>6."""
    m = np.ascontiguousarray(mat01, dtype=np.uint8)
    n, W = m.shape
    words = (W + 63) // 64
    pad = np.zeros((n, words * 64), dtype=np.uint8)
    pad[:, :W] = m
    by = np.packbits(pad, axis=1, bitorder="little")
    return np.ascontiguousarray(by).view(np.uint64).reshape(n, words)


def pack_mask(flags) -> np.ndarray:
    f = np.ascontiguousarray(flags, dtype=np.uint8).ravel()
    n = f.size
    words = max((n + 63) // 64, 1)
    pad = np.zeros(words * 64, dtype=np.uint8)
    pad[:n] = f
    return np.packbits(pad, bitorder="little").view(np.uint64).copy()


def pairwise_counts(bits, n, s0, s1):
    b = np.ascontiguousarray(bits, dtype=np.uint64)
    I = np.zeros((n, n), dtype=np.int64)
    lib().oracle_pairwise_counts(_p(b, _u64p), b.shape[1], n, int(s0), int(s1), _p(I, _i64p))
    return I


def identity(I, W, kind=0):
    I = np.ascontiguousarray(I, dtype=np.int64)
    n = I.shape[0]
    sim = np.zeros((n, n))
    lib().oracle_identity(_p(I, _i64p), n, int(W), int(kind), _p(sim, _f64p))
    return sim


INT_FIELDS = ("n_sites", "s_all", "s_p", "s_a", "s_b")
SUM_FIELDS = ("sum_p", "sum_a", "sum_b", "sum_ab")
DBL_FIELDS = ("pi", "pi_site", "pi_a", "pi_b", "pi_xy", "dxy", "da", "fst", "tajima_d")


def _window(fn, bits, n, s0, s1, mp, ma, mb, seq_len, d_pi_mode, s_scope):
    b = np.ascontiguousarray(bits, dtype=np.uint64)
    ints = np.zeros(8, dtype=np.uint32)
    sums = np.zeros(4, dtype=np.uint64)
    dbl = np.zeros(9)
    mp, ma, mb = (np.ascontiguousarray(m, dtype=np.uint64) for m in (mp, ma, mb))
    fn(_p(b, _u64p), b.shape[1], n, int(s0), int(s1), _p(mp, _u64p), _p(ma, _u64p), _p(mb, _u64p),
       float(seq_len or 0), int(d_pi_mode), int(s_scope), _p(ints, _u32p), _p(sums, _u64p), _p(dbl, _f64p))
    rec = {k: int(v) for k, v in zip(INT_FIELDS, ints)}
    rec.update({k: int(v) for k, v in zip(SUM_FIELDS, sums)})
    rec.update({k: float(v) for k, v in zip(DBL_FIELDS, dbl)})
    return rec


def window_allpairs(bits, n, s0, s1, mp, ma, mb, seq_len, d_pi_mode=0, s_scope=0):
    return _window(lib().oracle_window_allpairs, bits, n, s0, s1, mp, ma, mb, seq_len, d_pi_mode, s_scope)


def window_sitecount(bits, n, s0, s1, mp, ma, mb, seq_len, d_pi_mode=0, s_scope=0):
    return _window(lib().oracle_window_sitecount, bits, n, s0, s1, mp, ma, mb, seq_len, d_pi_mode, s_scope)


def ehh(bits, n, s0, s1, member=None, reverse=False):
    b = np.ascontiguousarray(bits, dtype=np.uint64)
    out = np.zeros(max(int(s1) - int(s0), 0))
    mem = None if member is None else np.ascontiguousarray(member, dtype=np.uint8)
    lib().oracle_ehh(_p(b, _u64p), b.shape[1], n, int(s0), int(s1), None if mem is None else _p(mem, _u8p),
                     1 if reverse else 0, _p(out, _f64p))
    return out


def to_sitemajor(bits, n, n_site):
    b = np.ascontiguousarray(bits, dtype=np.uint64)
    wps = max((n + 63) // 64, 1)
    sm = np.zeros((n_site, wps), dtype=np.uint64)
    lib().oracle_to_sitemajor(_p(b, _u64p), b.shape[1], n, int(n_site), _p(sm, _u64p), wps)
    return sm


def site_scan_sitemajor(sm, n, s0, s1, mp, ma, mb):
    ints = np.zeros(8, dtype=np.uint32)
    sums = np.zeros(4, dtype=np.uint64)
    mp, ma, mb = (np.ascontiguousarray(m, dtype=np.uint64) for m in (mp, ma, mb))
    lib().oracle_site_scan_sitemajor(_p(sm, _u64p), sm.shape[1], n, int(s0), int(s1), _p(mp, _u64p),
                                     _p(ma, _u64p), _p(mb, _u64p), _p(ints, _u32p), _p(sums, _u64p))
    return ints, sums


def site_scan_sitemajor_windows(sm, n, window_sites, n_win, mp, ma, mb, threads):
    ints = np.zeros((n_win, 8), dtype=np.uint32)
    sums = np.zeros((n_win, 4), dtype=np.uint64)
    mp, ma, mb = (np.ascontiguousarray(m, dtype=np.uint64) for m in (mp, ma, mb))
    lib().oracle_site_scan_sitemajor_windows(_p(sm, _u64p), sm.shape[1], n, int(window_sites), int(n_win), _p(mp, _u64p),
                                             _p(ma, _u64p), _p(mb, _u64p), int(threads), _p(ints, _u32p), _p(sums, _u64p))
    return ints, sums
