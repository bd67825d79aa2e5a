"""Drop-in for the reference's `src/correlation.py` public surface (forward and backward).

`FunctionCorrelation(tensorFirst, tensorSecond, intStride)` -- src/correlation.py:411-412 -- and
`ModuleCorrelation` -- :417-424 -- with the argument names used at the call sites src/models.py:175-183.
The work is one launch of the HIP kernel behind `pivlfn_corr_fwd` (no CuPy, no rearranged copies, no
zero-filled scratch).  Error behaviour follows `_FunctionCorrelation.forward` (:287-344): non-contiguous
inputs trip an assert (:297-298), CPU tensors raise NotImplementedError (:339-340).
Like the reference's, the op is a `torch.autograd.Function`: its backward (:348-405) is `pivlfn_corr_bwd`, which needs only
`first`, `second` and `gradOutput` (the reference also keeps its two padded scratch copies alive for it, :293).
"""
from __future__ import annotations

import math

import torch

from . import _lib


def _check(first: torch.Tensor, second: torch.Tensor) -> None:
    assert (first.is_contiguous() == True)    # noqa: E712  (same asserts as src/correlation.py:297-298)
    assert (second.is_contiguous() == True)   # noqa: E712
    if not first.is_cuda or not second.is_cuda:
        raise NotImplementedError()           # src/correlation.py:339-340: the op has no CPU path
    if first.dtype != torch.float32 or second.dtype != torch.float32:
        raise TypeError("FunctionCorrelation: float32 tensors only")
    if first.shape != second.shape or first.dim() != 4:
        raise ValueError(f"FunctionCorrelation: shapes {tuple(first.shape)} vs {tuple(second.shape)}")


class _FunctionCorrelation(torch.autograd.Function):
    @staticmethod
    def forward(ctx, first, second, intStride):
        _check(first, second)
        s = int(intStride)
        ctx.save_for_backward(first, second)
        ctx.intStride = s
        B, C, H, W = first.shape
        out = first.new_empty([B, 49, int(math.ceil(H / s)), int(math.ceil(W / s))])
        if out.numel() == 0:
            return out
        lib = _lib.load()
        with torch.cuda.device(first.device):
            _lib.check(lib.pivlfn_corr_fwd(first.data_ptr(), second.data_ptr(), out.data_ptr(), B, C, H, W, s,
                                           _lib.stream_ptr(first.device)), "FunctionCorrelation")
        return out

    @staticmethod
    def backward(ctx, gradOutput):
        first, second = ctx.saved_tensors
        gradOutput = gradOutput.contiguous()  # the reference asserts contiguity (:352); autograd may hand us a view
        if gradOutput.dtype != torch.float32 or not gradOutput.is_cuda:
            raise TypeError("FunctionCorrelation.backward: float32 GPU gradient expected")
        B, C, H, W = first.shape
        gradFirst = torch.empty_like(first) if ctx.needs_input_grad[0] else None       # :353-356 (every element is written)
        gradSecond = torch.empty_like(second) if ctx.needs_input_grad[1] else None
        if first.numel() and (gradFirst is not None or gradSecond is not None):
            lib = _lib.load()
            with torch.cuda.device(first.device):
                _lib.check(lib.pivlfn_corr_bwd(first.data_ptr(), second.data_ptr(), gradOutput.data_ptr(),
                                               gradFirst.data_ptr() if gradFirst is not None else None,
                                               gradSecond.data_ptr() if gradSecond is not None else None,
                                               B, C, H, W, ctx.intStride, _lib.stream_ptr(first.device)),
                           "FunctionCorrelation.backward")
        return gradFirst, gradSecond, None


def FunctionCorrelation(tensorFirst: torch.Tensor, tensorSecond: torch.Tensor, intStride: int) -> torch.Tensor:
    return _FunctionCorrelation.apply(tensorFirst, tensorSecond, intStride)


class ModuleCorrelation(torch.nn.Module):
    def __init__(self):
        super(ModuleCorrelation, self).__init__()

    def forward(self, tensorFirst, tensorSecond, intStride):
        return _FunctionCorrelation.apply(tensorFirst, tensorSecond, intStride)
