"""ORACLE -- TEST INFRASTRUCTURE ONLY.

CPU restatement of the reference's PIV-LiteFlowNet / LiteFlowNet inference path.  Nothing under
`piv_liteflownet-pytorch_amd/` imports this module; only `tests/`, `__graft_entry__.smoke()` and
`bench.py`'s `cpu_baseline` leg do, and only as the checker / the reported CPU baseline.

What follows what (all citations into /root/reference/):
  correlation_np        src/correlation.py:36-104 (formula; numpy slicing, independent of the C loops)
  correlation_c         src/correlation.py:9-104, 287-344 (loop-for-loop C restatement, oracle/corr_oracle.c)
  backwarp              src/models.py:20-35 (same normalised-grid arithmetic, via torch grid_sample)
  backwarp_np           src/models.py:20-35 restated in pixel units with explicit bilinear taps
  OracleNet.forward     src/models.py:319-370 (+ Features :66-116, FeatureExt :119-131,
                        Matching :134-187, Subpixel :190-217, Regularization :220-303)
  estimate              inference.py:30-67
The dense arithmetic (conv2d / conv_transpose2d / grid_sample / interpolate / unfold) is PyTorch's,
exactly as in the reference (torch==1.4.0 pinned there, requirements.txt:20; torch 2.10 CPU here --
every `align_corners` is explicit in the reference so the semantics are unchanged).

Parity pin: `oracle/gen_golden.py` runs THIS module against the reference's own `src/models.py`
(imported with in-memory shims, in the build container only) and records the agreement plus golden
vectors under tests/golden/.  The CuPy CUDA kernels themselves cannot execute anywhere in this
pipeline (no CUDA): the correlation is pinned by two independent restatements agreeing.
"""
from __future__ import annotations

import ctypes
import math
import os
import subprocess
from typing import Dict, List, Optional

import numpy as np
import torch
import torch.nn.functional as F

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

K_LEVEL = [0, 7, 7, 5, 5, 3, 3]


def build_c(force: bool = False) -> str:
    """Compile oracle/corr_oracle.c -> oracle/liboracle.so (gcc, -O2, no fast-math, no FMA contraction)."""
    so = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "corr_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-fPIC", "-shared", "-o", so, src, "-lm"])
    return so


def _lib():
    global _LIB
    if _LIB is None:
        _LIB = ctypes.CDLL(build_c())
        fp = ctypes.POINTER(ctypes.c_float)
        _LIB.corr_forward.argtypes = [fp, fp, fp] + [ctypes.c_int] * 5
        _LIB.backwarp_forward.argtypes = [fp, fp, fp] + [ctypes.c_int] * 4
        _LIB.corr_backward.argtypes = [fp, fp, fp, fp, fp] + [ctypes.c_int] * 5
    return _LIB


def _fp(a: np.ndarray):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def correlation_c(first: np.ndarray, second: np.ndarray, stride: int) -> np.ndarray:
    f1 = np.ascontiguousarray(first, dtype=np.float32)
    f2 = np.ascontiguousarray(second, dtype=np.float32)
    B, C, H, W = f1.shape
    out = np.zeros((B, 49, -(-H // stride), -(-W // stride)), dtype=np.float32)
    rc = _lib().corr_forward(_fp(f1), _fp(f2), _fp(out), B, C, H, W, stride)
    if rc != 0:
        raise RuntimeError(f"corr_forward failed rc={rc}")
    return out


def correlation_backward_c(first: np.ndarray, second: np.ndarray, grad_out: np.ndarray, stride: int):
    """(gradFirst, gradSecond) of the reference's backward kernels (src/correlation.py:106-234, 348-405), restated in C."""
    f1 = np.ascontiguousarray(first, dtype=np.float32)
    f2 = np.ascontiguousarray(second, dtype=np.float32)
    go = np.ascontiguousarray(grad_out, dtype=np.float32)
    B, C, H, W = f1.shape
    assert go.shape == (B, 49, -(-H // stride), -(-W // stride))
    g1, g2 = np.empty_like(f1), np.empty_like(f2)
    rc = _lib().corr_backward(_fp(f1), _fp(f2), _fp(go), _fp(g1), _fp(g2), B, C, H, W, stride)
    if rc:
        raise MemoryError("corr_backward")
    return g1, g2


def correlation_backward_autograd(first: torch.Tensor, second: torch.Tensor, grad_out: torch.Tensor, stride: int):
    """The same gradients by torch autograd through the (pinned) forward restatement `correlation_torch`."""
    f1 = first.detach().clone().requires_grad_(True)
    f2 = second.detach().clone().requires_grad_(True)
    correlation_torch(f1, f2, stride).backward(grad_out)
    return f1.grad, f2.grad


def backwarp_c(inp: np.ndarray, flow: np.ndarray) -> np.ndarray:
    a = np.ascontiguousarray(inp, dtype=np.float32)
    f = np.ascontiguousarray(flow, dtype=np.float32)
    B, C, H, W = a.shape
    out = np.zeros_like(a)
    rc = _lib().backwarp_forward(_fp(a), _fp(f), _fp(out), B, C, H, W)
    if rc != 0:
        raise RuntimeError(f"backwarp_forward failed rc={rc}")
    return out


def correlation_np(first: np.ndarray, second: np.ndarray, stride: int) -> np.ndarray:
    """out[b, 7(dy+3)+(dx+3), y, x] = mean_c f1[b,c,sy,sx] * f2[b,c,s(y+dy),s(x+dx)], zeros outside."""
    B, C, H, W = first.shape
    s = stride
    Ho, Wo = -(-H // s), -(-W // s)
    pad = 3 * s
    f2p = np.zeros((B, C, H + 2 * pad + s, W + 2 * pad + s), dtype=first.dtype)
    f2p[:, :, pad:pad + H, pad:pad + W] = second
    f1s = first[:, :, ::s, ::s]
    out = np.zeros((B, 49, Ho, Wo), dtype=first.dtype)
    for dy in range(-3, 4):
        for dx in range(-3, 4):
            y0, x0 = pad + dy * s, pad + dx * s
            f2s = f2p[:, :, y0:y0 + s * Ho:s, x0:x0 + s * Wo:s]
            out[:, 7 * (dy + 3) + (dx + 3)] = (f1s * f2s).sum(1) / first.dtype.type(C)
    return out


def correlation_torch(first: torch.Tensor, second: torch.Tensor, stride: int) -> torch.Tensor:
    """Same as correlation_np on torch tensors (any float dtype); used inside OracleNet."""
    B, C, H, W = first.shape
    s = stride
    Ho, Wo = -(-H // s), -(-W // s)
    pad = 3 * s
    f2p = F.pad(second, (pad, pad + s, pad, pad + s))
    f1s = first[:, :, ::s, ::s]
    outs = []
    for dy in range(-3, 4):
        for dx in range(-3, 4):
            y0, x0 = pad + dy * s, pad + dx * s
            outs.append((f1s * f2p[:, :, y0:y0 + s * Ho:s, x0:x0 + s * Wo:s]).sum(1, keepdim=True) / C)
    return torch.cat(outs, 1)


def backwarp(inp: torch.Tensor, flow: torch.Tensor) -> torch.Tensor:
    """src/models.py:20-35 without the module-global grid cache (and without .cuda())."""
    B, _, H, W = flow.shape
    hor = torch.linspace(-1.0, 1.0, W, dtype=flow.dtype).view(1, 1, 1, W).expand(B, -1, H, -1)
    ver = torch.linspace(-1.0, 1.0, H, dtype=flow.dtype).view(1, 1, H, 1).expand(B, -1, -1, W)
    grid = torch.cat([hor, ver], 1)
    fl = torch.cat([flow[:, 0:1] / ((inp.shape[3] - 1.0) / 2.0), flow[:, 1:2] / ((inp.shape[2] - 1.0) / 2.0)], 1)
    return F.grid_sample(input=inp, grid=(grid + fl).permute(0, 2, 3, 1), mode="bilinear",
                         padding_mode="zeros", align_corners=True)


def backwarp_np(inp: np.ndarray, flow: np.ndarray) -> np.ndarray:
    """Pixel-unit restatement: sample at (x+u, y+v), 4 bilinear taps, out-of-range taps contribute 0."""
    B, C, H, W = inp.shape
    yy, xx = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    out = np.zeros_like(inp)
    for b in range(B):
        fx = xx.astype(inp.dtype) + flow[b, 0]
        fy = yy.astype(inp.dtype) + flow[b, 1]
        x0 = np.floor(fx); y0 = np.floor(fy)
        ax = fx - x0; ay = fy - y0
        x0 = x0.astype(np.int64); y0 = y0.astype(np.int64)
        for oy, ox, wgt in ((0, 0, (1 - ax) * (1 - ay)), (0, 1, ax * (1 - ay)), (1, 0, (1 - ax) * ay), (1, 1, ax * ay)):
            xs, ys = x0 + ox, y0 + oy
            ok = (xs >= 0) & (xs < W) & (ys >= 0) & (ys < H)
            v = inp[b][:, np.clip(ys, 0, H - 1), np.clip(xs, 0, W - 1)]
            out[b] += np.where(ok, wgt, 0).astype(inp.dtype)[None] * v
    return out


def _lrelu(x):
    return F.leaky_relu(x, negative_slope=0.1)


class OracleNet:
    """Functional restatement of `LiteFlowNet` (src/models.py:39-370) driven by a state dict.

    corr: 'torch' (slicing, dtype-generic) or 'c' (the loop-for-loop C restatement, fp32 only).
    """

    def __init__(self, weights: Dict[str, torch.Tensor], starting_scale: float = 40.0, lowest_level: int = 2,
                 rgb_mean=(0.411618, 0.434631, 0.454253, 0.410782, 0.433645, 0.452793),
                 dtype=torch.float32, corr: str = "torch"):
        self.w = {k: v.to(dtype) for k, v in weights.items()}
        self.dtype = dtype
        self.lowest_level = int(lowest_level)
        self.mean = [list(rgb_mean[:3]), list(rgb_mean[3:])]
        self.scale = [float(starting_scale) / (2.0 ** L) for L in range(7)]      # src/models.py:61-63
        self.levels = list(range(self.lowest_level, 7))                          # src/models.py:306
        self.corr = corr

    # -- helpers -------------------------------------------------------------------------------
    def _conv(self, name, x, stride=1, pad=0, act=True):
        y = F.conv2d(x, self.w[name + ".weight"], self.w.get(name + ".bias"), stride=stride, padding=pad)
        return _lrelu(y) if act else y

    def _stack(self, prefix, x, k):
        """conv_M / conv_S: 3x3 + LeakyReLU layers at indices 0, 2, ... then the k x k flow head without activation
        (v1: 3 hidden layers, src/models.py:154-163, 197-207; LiteFlowNet2: 5, :487-500, 534-548)."""
        j = 0
        while f"{prefix}{j + 2}.weight" in self.w:
            x = self._conv(f"{prefix}{j}", x, 1, 1)
            j += 2
        return self._conv(f"{prefix}{j}", x, 1, k // 2, act=False)

    def _corr(self, f1, f2, s):
        if self.corr == "c":
            return torch.from_numpy(correlation_c(f1.numpy(), f2.numpy(), s))
        return correlation_torch(f1, f2, s)

    # -- NetC: src/models.py:66-116 ---------------------------------------------------------------
    def features(self, x) -> List[torch.Tensor]:
        c = self._conv
        l1 = c("NetC.conv1.0", x, 1, 3)
        l2 = c("NetC.conv2.4", c("NetC.conv2.2", c("NetC.conv2.0", l1, 2, 1), 1, 1), 1, 1)
        l3 = c("NetC.conv3.2", c("NetC.conv3.0", l2, 2, 1), 1, 1)
        l4 = c("NetC.conv4.2", c("NetC.conv4.0", l3, 2, 1), 1, 1)
        l5 = c("NetC.conv5.0", l4, 2, 1)
        l6 = c("NetC.conv6.0", l5, 2, 1)
        return [l1, l2, l3, l4, l5, l6]

    # -- Matching: src/models.py:165-187 ------------------------------------------------------------
    def matching(self, i, L, f1, f2, xflow):
        p = f"NetE_M.{i}."
        if xflow is not None:
            xflow = F.conv_transpose2d(xflow, self.w[p + "upConv_M.weight"], None, stride=2, padding=1, groups=2)
            f2 = backwarp(f2, xflow * self.scale[L])
        if L >= 4:
            corr = _lrelu(self._corr(f1, f2, 1))
        else:
            corr = F.conv_transpose2d(_lrelu(self._corr(f1, f2, 2)), self.w[p + "upCorr_M.weight"], None,
                                      stride=2, padding=1, groups=49)
        x = self._stack(p + "conv_M.", corr, K_LEVEL[L])
        return x + (xflow if xflow is not None else 0.0)

    # -- Subpixel: src/models.py:209-217 ------------------------------------------------------------
    def subpixel(self, i, L, f1, f2, xflow):
        p = f"NetE_S.{i}."
        f2 = backwarp(f2, xflow * self.scale[L])
        x = self._stack(p + "conv_S.", torch.cat([f1, f2, xflow], 1), K_LEVEL[L])
        return x + xflow

    # -- Regularization: src/models.py:274-303 --------------------------------------------------------
    def regularization(self, i, L, img1, img2, feat1, xflow):
        p = f"NetE_R.{i}."
        k = K_LEVEL[L]
        B = xflow.shape[0]
        rm = xflow - xflow.view(B, 2, -1).mean(2, True).view(B, 2, 1, 1)
        warp = backwarp(img2, xflow * self.scale[L])
        norm = (img1 - warp).pow(2.0).sum(1, True).sqrt()
        feat = self._conv(p + "moduleFeat.0", feat1, 1, 0) if L < 5 else feat1
        x = torch.cat([norm, rm, feat], 1)
        for j in (0, 2, 4, 6, 8, 10):
            x = self._conv(p + f"conv_R.{j}", x, 1, 1)
        if L < 5:
            x = F.conv2d(x, self.w[p + "conv_dist_R.0.weight"], self.w[p + "conv_dist_R.0.bias"], padding=(k // 2, 0))
            x = F.conv2d(x, self.w[p + "conv_dist_R.1.weight"], self.w[p + "conv_dist_R.1.bias"], padding=(0, k // 2))
        else:
            x = self._conv(p + "conv_dist_R.0", x, 1, k // 2, act=False)
        negsq = x.pow(2.0).neg()
        dist = (negsq - negsq.max(1, True)[0]).exp()
        div = dist.sum(1, True).reciprocal()
        ux = F.unfold(xflow[:, 0:1], kernel_size=k, stride=1, padding=(k - 1) // 2).view_as(dist)
        uy = F.unfold(xflow[:, 1:2], kernel_size=k, stride=1, padding=(k - 1) // 2).view_as(dist)
        fx = self._conv(p + "moduleScaleX", dist * ux, act=False) * div
        fy = self._conv(p + "moduleScaleY", dist * uy, act=False) * div
        return torch.cat([fx, fy], 1)

    # -- forward: src/models.py:319-370 ---------------------------------------------------------------
    def forward(self, img1: torch.Tensor, img2: torch.Tensor, return_levels: bool = False):
        img1 = img1.to(self.dtype).clone()
        img2 = img2.to(self.dtype).clone()
        for c in range(img1.shape[1]):
            img1[:, c] = img1[:, c] - self.mean[0][c]
            img2[:, c] = img2[:, c] - self.mean[1][c]
        feat1 = self.features(img1)
        feat2 = self.features(img2)
        im1, im2 = [img1], [img2]
        for lv in range(1, 6):
            size = (feat1[lv].shape[2], feat1[lv].shape[3])
            im1.append(F.interpolate(im1[-1], size=size, mode="bilinear", align_corners=False))
            im2.append(F.interpolate(im2[-1], size=size, mode="bilinear", align_corners=False))
        idx_diff = 6 - len(self.levels)
        xflow = None
        per_level = []
        for i in reversed(range(len(self.levels))):
            idx = i + idx_diff              # 0-based feature index = L - 1
            L = idx + 1
            if idx < 2:
                e = f"NetC_ext.{idx - 1 if idx - 1 >= 0 else len(range(self.lowest_level - 1, 2)) - 1}.conv_ext.0"
                f1 = self._conv(e, feat1[idx], 1, 0)
                f2 = self._conv(e, feat2[idx], 1, 0)
            else:
                f1, f2 = feat1[idx], feat2[idx]
            fm = self.matching(i, L, f1, f2, xflow)
            fs = self.subpixel(i, L, f1, f2, fm)
            xflow = self.regularization(i, L, im1[idx], im2[idx], feat1[idx], fs)
            per_level.append([fm, fs, xflow])
        out = xflow * self.scale[1]
        return (out, per_level) if return_levels else out

    __call__ = forward


def make_net(model: str, weights, dtype=torch.float32, corr: str = "torch") -> OracleNet:
    """Mirrors the factories src/models.py:719-766; 'hui2' / 'piv2' = the LiteFlowNet2 backbones (version=2)."""
    if model == "hui2":
        return OracleNet(weights, 40.0, 3, (0.411618, 0.434631, 0.454253, 0.410782, 0.433645, 0.452793), dtype, corr)
    if model == "piv2":
        return OracleNet(weights, 10.0, 2, (0.194286, 0.190633, 0.191766, 0.194220, 0.190595, 0.191701), dtype, corr)
    if model == "hui":
        return OracleNet(weights, 40.0, 2, (0.411618, 0.434631, 0.454253, 0.410782, 0.433645, 0.452793), dtype, corr)
    if model == "piv":
        return OracleNet(weights, 10.0, 1, (0.173935, 0.180594, 0.192608, 0.172978, 0.179518, 0.191300), dtype, corr)
    raise ValueError(model)


def estimate(net, img1: torch.Tensor, img2: torch.Tensor, tensor: bool = False):
    """inference.py:30-67."""
    assert img1.size(2) == img2.size(2)
    assert img1.size(3) == img2.size(3)
    W, H = img1.size(3), img1.size(2)
    aw = int(math.floor(math.ceil(W / 32.0) * 32.0))
    ah = int(math.floor(math.ceil(H / 32.0) * 32.0))
    sw, sh = float(W) / float(aw), float(H) / float(ah)
    a = F.interpolate(img1, size=(ah, aw), mode="bilinear", align_corners=False)
    b = F.interpolate(img2, size=(ah, aw), mode="bilinear", align_corners=False)
    with torch.no_grad():
        if hasattr(net, "eval"):
            net.eval()
        raw = net(a, b)
    flow = F.interpolate(raw, size=(H, W), mode="bilinear", align_corners=False)
    flow[:, 0] *= sw
    flow[:, 1] *= sh
    if tensor:
        return flow.detach()
    return torch.squeeze(flow).permute(1, 2, 0).detach().cpu().numpy()
