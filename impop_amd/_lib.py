"""ctypes binding of libimpop_hip.so — the stub a reference maintainer would add
(see INTEGRATION.md).  Declares every entry point of include/impop_hip.h.

There is no CPU fallback: if the library cannot be loaded, or no gfx950 device
is usable, the product path raises.
"""
from __future__ import annotations

import ctypes as C
import importlib.util
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.environ.get("IMPOP_HIP_LIBRARY") or os.path.join(_HERE, "libimpop_hip.so")  # env override: tuning builds


class ImpopError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"[impop {code}] {message}")
        self.code = code
        self.message = message


ABI_VERSION = 2  # IMPOP_ABI_VERSION of include/impop_hip.h
E_INVALID, E_NODEVICE, E_HIP, E_NOMEM, E_UNSUPPORTED, E_INTERNAL = -1, -2, -3, -4, -5, -6
KEEP_SITE_BLOCKED, KEEP_HAP_MAJOR = 1, 2
IDENTITY_MATCH, IDENTITY_DICE = 0, 1


class Window(C.Structure):
    _fields_ = [("site_begin", C.c_uint64), ("site_end", C.c_uint64), ("seq_len", C.c_uint64)]


class WindowStats(C.Structure):
    _fields_ = [("n_sites", C.c_uint32), ("s_all", C.c_uint32), ("s_p", C.c_uint32), ("s_a", C.c_uint32),
                ("s_b", C.c_uint32), ("flags", C.c_uint32),
                ("sum_p", C.c_uint64), ("sum_a", C.c_uint64), ("sum_b", C.c_uint64), ("sum_ab", C.c_uint64),
                ("pi", C.c_double), ("pi_site", C.c_double), ("pi_a", C.c_double), ("pi_b", C.c_double),
                ("pi_xy", C.c_double), ("dxy", C.c_double), ("da", C.c_double), ("fst", C.c_double),
                ("tajima_d", C.c_double)]


class ScanParams(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("d_pi_mode", C.c_int32), ("s_scope", C.c_int32),
                ("tile_blocks", C.c_uint32)]


class SynthParams(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("n_founder", C.c_uint32), ("p_founder", C.c_double),
                ("p_private_word", C.c_double)]


class PairStats(C.Structure):
    _fields_ = [("fst", C.c_double), ("pi_a", C.c_double), ("pi_b", C.c_double), ("pi_xy", C.c_double),
                ("dxy", C.c_double), ("da", C.c_double)]


class PairwiseParams(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("identity_kind", C.c_int32), ("threshold", C.c_double),
                ("round_digits", C.c_int32), ("d_pi_mode", C.c_int32), ("s_scope", C.c_int32),
                ("fst_method", C.c_uint32)]


class PairwiseStats(C.Structure):
    _fields_ = [("pi", C.c_double), ("pi_site", C.c_double), ("fst", C.c_double), ("pi_a", C.c_double),
                ("pi_b", C.c_double), ("pi_xy", C.c_double), ("dxy", C.c_double), ("da", C.c_double),
                ("tajima_d", C.c_double), ("n_groups", C.c_uint32), ("s_all", C.c_uint32), ("s_p", C.c_uint32),
                ("n_sites", C.c_uint32), ("reserved", C.c_uint64)]


class Pica2Detail(C.Structure):
    _fields_ = [("sum_2pairs", C.c_double), ("n_pairs_with_data", C.c_uint64)]


assert C.sizeof(WindowStats) == 128 and C.sizeof(Window) == 24 and C.sizeof(PairwiseStats) == 96

_vp = C.c_void_p
_u64p = C.POINTER(C.c_uint64)
_u32p = C.POINTER(C.c_uint32)
_u8p = C.POINTER(C.c_uint8)
_i32p = C.POINTER(C.c_int32)
_i64p = C.POINTER(C.c_int64)
_f64p = C.POINTER(C.c_double)

# name -> (restype, argtypes): one line per declaration in include/impop_hip.h
SIGNATURES = {
    "impop_version": (C.c_int, []),
    "impop_last_error": (C.c_char_p, []),
    "impop_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "impop_ctx_create": (C.c_int, [C.c_int, _vp, C.POINTER(_vp)]),
    "impop_ctx_destroy": (C.c_int, [_vp]),
    "impop_ctx_synchronize": (C.c_int, [_vp]),
    "impop_ctx_device_name": (C.c_int, [_vp, C.c_char_p, C.c_size_t]),
    "impop_matrix_upload": (C.c_int, [_vp, _u64p, C.c_uint32, C.c_uint64, C.c_uint64, C.c_uint32, C.POINTER(_vp)]),
    "impop_matrix_synthetic": (C.c_int, [_vp, C.c_uint32, C.c_uint64, C.POINTER(SynthParams), C.c_uint32, C.POINTER(_vp)]),
    "impop_debug_raise_device_error": (C.c_int, [_vp, C.c_uint32]),
    "impop_ctx_gram_timing": (C.c_int, [_vp, C.c_int]),
    "impop_ctx_gram_elapsed": (C.c_int, [_vp, C.POINTER(C.c_double), C.POINTER(C.c_uint64)]),
    "impop_matrix_synthetic_slab": (C.c_int, [_vp, C.c_uint32, C.c_uint64, C.c_uint64, C.POINTER(SynthParams), C.c_uint32, C.POINTER(_vp)]),
    "impop_matrix_download": (C.c_int, [_vp, _vp, C.c_uint64, C.c_uint64, _u64p, C.c_uint64]),
    "impop_matrix_info": (C.c_int, [_vp, _u32p, _u64p, _u64p, _u32p]),
    "impop_matrix_free": (C.c_int, [_vp, _vp]),
    "impop_scan_plan_create": (C.c_int, [_vp, _vp, C.POINTER(Window), C.c_uint64, _u64p, _u64p, _u64p,
                                         C.POINTER(ScanParams), C.POINTER(_vp)]),
    "impop_scan_plan_set_masks": (C.c_int, [_vp, _u64p, _u64p, _u64p]),
    "impop_scan_plan_launch": (C.c_int, [_vp, _vp]),
    "impop_scan_plan_fetch": (C.c_int, [_vp, C.POINTER(WindowStats)]),
    "impop_scan_plan_info": (C.c_int, [_vp, _u64p, _u64p]),
    "impop_scan_plan_device_records": (C.c_int, [_vp, C.POINTER(_vp)]),
    "impop_shard_range": (C.c_int, [C.c_uint64, C.c_int, C.c_int, _u64p, _u64p]),
    "impop_shard_windows": (C.c_int, [C.POINTER(Window), C.c_uint64, C.c_int, C.c_int, _u64p, _u64p, _u64p, _u64p]),
    "impop_scan_sharded": (C.c_int, [C.POINTER(_vp), C.POINTER(_vp), _u64p, C.c_int, C.POINTER(Window), C.c_uint64, _u64p, _u64p,
                                     _u64p, C.POINTER(ScanParams), C.POINTER(WindowStats)]),
    "impop_pairwise_scan_sharded": (C.c_int, [C.POINTER(_vp), C.POINTER(_vp), _u64p, C.c_int, C.POINTER(Window), C.c_uint64, _u64p,
                                              _u64p, _u64p, C.POINTER(PairwiseParams), C.POINTER(PairwiseStats)]),
    "impop_comm_unique_id": (C.c_int, [C.c_char_p]),
    "impop_comm_create": (C.c_int, [_vp, C.c_char_p, C.c_int, C.c_int, C.POINTER(_vp)]),
    "impop_comm_destroy": (C.c_int, [_vp]),
    "impop_gather": (C.c_int, [_vp, _vp, C.c_size_t, _vp]),
    "impop_gather_records": (C.c_int, [_vp, _vp, C.c_uint64, C.POINTER(WindowStats)]),
    "impop_allreduce_i64": (C.c_int, [_vp, _vp, C.c_size_t]),
    "impop_scan_plan_timing": (C.c_int, [_vp, C.c_int]),
    "impop_scan_plan_elapsed": (C.c_int, [_vp, _f64p, _u64p]),
    "impop_scan_plan_destroy": (C.c_int, [_vp]),
    "impop_scan": (C.c_int, [_vp, _vp, C.POINTER(Window), C.c_uint64, _u64p, _u64p, _u64p, C.POINTER(ScanParams),
                             C.POINTER(WindowStats)]),
    "impop_scan_multi": (C.c_int, [_vp, _vp, C.POINTER(Window), C.c_uint64, _u64p, C.c_uint32, C.POINTER(PairStats)]),
    "impop_afs": (C.c_int, [_vp, _vp, C.POINTER(Window), C.c_uint64, _u64p, _u32p]),
    "impop_site_counts": (C.c_int, [_vp, _vp, _u64p, C.c_uint64, C.c_uint64, _u32p]),
    "impop_pairwise_counts": (C.c_int, [_vp, _vp, C.c_uint64, C.c_uint64, _i32p]),
    "impop_pairwise_identity": (C.c_int, [_vp, _vp, C.c_uint64, C.c_uint64, C.c_int, _f64p]),
    "impop_pairwise_scan": (C.c_int, [_vp, _vp, C.POINTER(Window), C.c_uint64, _u64p, _u64p, _u64p,
                                      C.POINTER(PairwiseParams), C.POINTER(PairwiseStats)]),
    "impop_pi_from_identity": (C.c_int, [_vp, _f64p, C.c_uint32, C.c_double, C.c_int, C.c_uint64, _u32p, _f64p, _f64p, _u32p, _u32p,
                                         C.POINTER(Pica2Detail)]),
    "impop_pica2_pair_terms": (C.c_int, [_vp, _f64p, C.c_uint32, C.c_int, _u32p, _u32p, C.c_uint32, _f64p, _f64p]),
    "impop_fst_from_identity": (C.c_int, [_vp, _f64p, C.c_uint32, _u8p, _u8p, C.c_uint64, C.c_int, _f64p, _u64p]),
    "impop_matrix_set_site_weights": (C.c_int, [_vp, _vp, _u32p]),
    "impop_matrix_compact": (C.c_int, [_vp, _vp, C.POINTER(_vp)]),
    "impop_matrix_positions": (C.c_int, [_vp, C.c_uint64, C.c_uint64, _u64p, _u64p]),
    "impop_ehh": (C.c_int, [_vp, _vp, C.c_uint64, C.c_uint64, _u64p, C.c_int, _f64p, _u32p]),
    "impop_fst_grouped_from_identity": (C.c_int, [_vp, _f64p, C.c_uint32, _u8p, _u8p, C.c_double, C.c_uint64, C.c_int, _u32p, _f64p,
                                                  _u64p]),
    "impop_tajimas_d": (C.c_int, [_vp, _i64p, _f64p, _f64p, C.c_uint64, _f64p, _f64p]),
    "impop_cluster_from_identity": (C.c_int, [_vp, _f64p, C.c_uint32, C.c_double, _u32p, _u32p, _u32p]),
    "impop_py_round": (C.c_int, [_vp, _f64p, C.c_uint64, C.c_int, _f64p]),
    "impop_sim_parse": (C.c_int, [C.c_char_p, C.c_int, C.POINTER(_vp)]),
    "impop_sim_info": (C.c_int, [_vp, _u32p, _u64p, _u64p, _i64p, _u64p]),
    "impop_sim_names": (C.c_int, [_vp, C.c_char_p]),
    "impop_sim_first_seen": (C.c_int, [_vp, _u32p]),
    "impop_sim_bad_text": (C.c_int, [_vp, C.c_char_p, C.c_size_t]),
    "impop_sim_dense": (C.c_int, [_vp, _f64p]),
    "impop_sim_free": (C.c_int, [_vp]),
    "impop_gfa_parse": (C.c_int, [C.c_char_p, C.c_char_p, C.POINTER(_vp)]),
    "impop_paths_table_parse": (C.c_int, [C.c_char_p, C.POINTER(_vp)]),
    "impop_gfa_info": (C.c_int, [_vp, _u32p, _u64p, _u64p, _i64p]),
    "impop_gfa_names": (C.c_int, [_vp, C.c_char_p]),
    "impop_gfa_bits": (C.c_int, [_vp, _u64p, C.c_uint64]),
    "impop_gfa_lengths": (C.c_int, [_vp, _u32p]),
    "impop_gfa_positions": (C.c_int, [_vp, _i64p]),
    "impop_gfa_free": (C.c_int, [_vp]),
}

_lib = None


def _preload_hip_runtime() -> None:
    """Make sure exactly one HIP runtime lives in the process.

    PyTorch-ROCm wheels bundle their own libamdhip64.so (soname libamdhip64.so.7,
    same as /opt/rocm's).  If torch is installed we load ITS copy first, by path,
    so that libimpop_hip.so (NEEDED libamdhip64.so.7) and a later `import torch`
    both bind to the same runtime; otherwise the RUNPATH (/opt/rocm/lib) copy is
    used.  torch itself is not imported here.
    """
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec and spec.submodule_search_locations:
        cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
        if os.path.exists(cand):
            C.CDLL(cand, mode=C.RTLD_GLOBAL)


def load() -> C.CDLL:
    """Load libimpop_hip.so; raise (never fall back) if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO_PATH):
        raise ImpopError(E_UNSUPPORTED, f"{SO_PATH} not found: build it with `python -m impop_amd.build` "
                                         "(hipcc, gfx950). impop_amd has no CPU fallback.")
    _preload_hip_runtime()
    lib = C.CDLL(SO_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if an export is missing
        fn.restype = res
        fn.argtypes = args
    if lib.impop_version() != ABI_VERSION:
        raise ImpopError(E_UNSUPPORTED, f"ABI version {lib.impop_version()} != {ABI_VERSION}")
    _lib = lib
    return lib


def check(rc: int) -> None:
    if rc != 0:
        msg = load().impop_last_error().decode("utf-8", "replace")
        raise ImpopError(rc, msg)
