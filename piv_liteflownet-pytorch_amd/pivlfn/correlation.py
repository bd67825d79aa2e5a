"""Drop-in for the reference's `src/correlation.py` public surface (forward only).

`FunctionCorrelation(tensorFirst, tensorSecond, intStride)` -- src/correlation.py:411-412 -- and
`ModuleCorrelation` -- :417-424 -- with the argument names used at the call sites src/models.py:175-183.
The work is one launch of the HIP kernel behind `pivlfn_corr_fwd` (no CuPy, no rearranged copies, no
zero-filled scratch).  Error behaviour follows `_FunctionCorrelation.forward` (:287-344): non-contiguous
inputs trip an assert (:297-298), CPU tensors raise NotImplementedError (:339-340).
"""
from __future__ import annotations

import math

import torch

from . import _lib


def FunctionCorrelation(tensorFirst: torch.Tensor, tensorSecond: torch.Tensor, intStride: int) -> torch.Tensor:
    first, second = tensorFirst, tensorSecond
    assert (first.is_contiguous() == True)    # noqa: E712  (same asserts as src/correlation.py:297-298)
    assert (second.is_contiguous() == True)   # noqa: E712
    if not first.is_cuda or not second.is_cuda:
        raise NotImplementedError()           # src/correlation.py:339-340: the op has no CPU path
    if first.dtype != torch.float32 or second.dtype != torch.float32:
        raise TypeError("FunctionCorrelation: float32 tensors only")
    if first.shape != second.shape or first.dim() != 4:
        raise ValueError(f"FunctionCorrelation: shapes {tuple(first.shape)} vs {tuple(second.shape)}")
    s = int(intStride)
    B, C, H, W = first.shape
    out = first.new_empty([B, 49, int(math.ceil(H / s)), int(math.ceil(W / s))])
    if out.numel() == 0:
        return out
    lib = _lib.load()
    with torch.cuda.device(first.device):
        _lib.check(lib.pivlfn_corr_fwd(first.data_ptr(), second.data_ptr(), out.data_ptr(), B, C, H, W, s,
                                       _lib.stream_ptr(first.device)), "FunctionCorrelation")
    return out


class ModuleCorrelation(torch.nn.Module):
    def __init__(self):
        super(ModuleCorrelation, self).__init__()

    def forward(self, tensorFirst, tensorSecond, intStride):
        return FunctionCorrelation(tensorFirst, tensorSecond, intStride)
