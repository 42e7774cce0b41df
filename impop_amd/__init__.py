"""impop_amd — MI355X-native windowed population statistics (pi, Hudson Fst, Tajima's D,
haplotype-cluster frequencies) for the pairwise-diversity hot path of pangenome/impop.

Hand-written HIP for gfx950 behind a C ABI (include/impop_hip.h, libimpop_hip.so); this
package is the thin ctypes host layer that mirrors the reference's Python interfaces.
There is no CPU fallback: without the built library and a gfx950 GPU every compute
entry point raises ImpopError.
"""
from ._lib import ImpopError, SO_PATH  # noqa: F401
from . import distributed  # noqa: F401
from .engine import (Comm, pairwise_scan_sharded, scan_sharded, shard_windows_c,  # noqa: F401
                     BitMatrix, Context, ScanPlan, STATS_DTYPE, PAIRWISE_DTYPE, WINDOW_DTYPE, fixed_windows,  # noqa: F401
                     make_windows, mask_from_indices, pack_hap_major, pack_mask, unpack_hap_major)

__version__ = "0.1.0"
