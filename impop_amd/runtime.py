"""Process-wide default Context for the drop-in function API (one GPU per process)."""
from __future__ import annotations

import os
from typing import Optional

from .engine import Context

_default: Optional[Context] = None


def default_context() -> Context:
    """Context on device IMPOP_DEVICE (default: LOCAL_RANK, else 0).  Raises ImpopError if
    libimpop_hip.so or a gfx950 GPU is missing — there is no CPU fallback."""
    global _default
    if _default is None:
        dev = int(os.environ.get("IMPOP_DEVICE", os.environ.get("LOCAL_RANK", "0")))
        _default = Context(dev)
    return _default


def set_default_context(ctx: Optional[Context]) -> None:
    global _default
    _default = ctx


def _line_writer(sink):
    """print-like callable writing one line per call to `sink`, or doing nothing when there is no log file"""
    if not sink:
        return lambda text: None
    return lambda text: print(text, file=sink)
