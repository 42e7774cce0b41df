"""GPU-backed mirror of the reference's scripts/tj_d.py (tj_d.py:28-69)."""
from __future__ import annotations

from dataclasses import dataclass

from ._lib import E_INVALID, ImpopError
from .runtime import default_context


@dataclass
class TajimaComponents:  # tj_d.py:28-39
    a1: float
    a2: float
    b1: float
    b2: float
    c1: float
    c2: float
    e1: float
    e2: float
    numerator: float
    denominator: float


def tajimas_d(n: int, S: float, pi: float, return_components: bool = False, ctx=None):
    """tj_d.tajimas_d (tj_d.py:47-69).  ValueError (same text) for n < 2 or negative S / pi."""
    ctx = ctx or default_context()
    try:
        D, comps = ctx.tajimas_d([int(n)], [float(S)], [float(pi)], components=True)
    except ImpopError as e:
        if e.code == E_INVALID:
            raise ValueError(e.message) from None
        raise
    if return_components:
        return float(D[0]), TajimaComponents(*[float(v) for v in comps[0]])
    return float(D[0])
