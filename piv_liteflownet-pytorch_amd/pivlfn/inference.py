"""Drop-in for `estimate()` of the reference's inference.py:30-67.

Same signature and return convention: `estimate(net, img1, img2, tensor=False)` takes two [B,3,H,W] tensors
in [0,1] on the network's device, adapts H and W up to multiples of 32 with a bilinear
(align_corners=False) resize, runs the network in eval mode, resizes the raw flow back to H x W and rescales
u by W/W' and v by H/H'.  tensor=True returns [B,2,H,W]; tensor=False returns an H x W x 2 numpy array
(batch 1, as in the reference).  Both resizes run as HIP kernels (`pivlfn_resize_bilinear`); when a size is
already a multiple of 32 the input resize is the identity and is skipped.
"""
from __future__ import annotations

import ctypes
import math

import torch

from . import _lib


def _resize(x: torch.Tensor, Ho: int, Wo: int, mul=None) -> torch.Tensor:
    B, C, H, W = x.shape
    out = torch.empty([B, C, Ho, Wo], dtype=torch.float32, device=x.device)
    m = (ctypes.c_float * 2)(*mul) if mul is not None else None
    with torch.cuda.device(x.device):
        _lib.check(_lib.load().pivlfn_resize_bilinear(x.data_ptr(), out.data_ptr(), B, C, H, W, Ho, Wo, m,
                                                      _lib.stream_ptr(x.device)), "estimate: resize")
    return out


def estimate(net: torch.nn.Module, img1: torch.Tensor, img2: torch.Tensor, tensor: bool = False):
    # Ensure that both the first and second images have the same dimension (inference.py:32-33)
    assert (img1.size(2) == img2.size(2))
    assert (img1.size(3) == img2.size(3))
    if not img1.is_cuda:
        raise NotImplementedError("estimate: GPU tensors only")
    W, H = img1.size(3), img1.size(2)
    aw = int(math.floor(math.ceil(W / 32.0) * 32.0))
    ah = int(math.floor(math.ceil(H / 32.0) * 32.0))
    sw, sh = float(W) / float(aw), float(H) / float(ah)
    a = img1.detach().contiguous().float()
    b = img2.detach().contiguous().float()
    if (ah, aw) != (H, W):
        a = _resize(a, ah, aw)
        b = _resize(b, ah, aw)
    with torch.set_grad_enabled(False):
        if net.training:                # the reference calls net.eval() unconditionally (inference.py:52); it walks every submodule
            net.eval()
        raw = net(a, b)
    if raw.shape[2:] == (H, W) and sw == 1.0 and sh == 1.0:
        flow = raw                      # same-size bilinear resize is the identity; scale factors are 1
    else:
        flow = _resize(raw.contiguous(), H, W, mul=(sw, sh))
    if tensor:
        return flow.detach()
    return torch.squeeze(flow).permute(1, 2, 0).detach().cpu().numpy()
