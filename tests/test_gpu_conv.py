"""GPU: the direct-convolution kernels (fp32 matrix cores / VALU flow head) against torch.nn.functional.conv2d on the CPU."""
import ctypes

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from pivlfn import _lib

pytestmark = pytest.mark.gpu


class Conv:
    def __init__(self, w, b):
        self.w, self.b = w.contiguous(), b.contiguous()
        self.h = ctypes.c_void_p()
        co, ci, kh, kw = w.shape
        _lib.check(_lib.load().pivlfn_conv_create(self.w.data_ptr(), self.b.data_ptr(), co, ci, kh, kw, ctypes.byref(self.h)), "conv_create")

    def __del__(self):
        _lib.load().pivlfn_conv_destroy(self.h)


def run(conv, x_nchw, stride, pad, leaky, dev, res=None, x_lanes=None):
    """x_nchw: CPU [B,C,H,W] -> channels-last on the GPU (with optional extra lanes), returns CPU NCHW output."""
    B, C, H, W = x_nchw.shape
    co, ci, kh, kw = conv.w.shape
    xs = x_lanes or -(-C // 4) * 4
    x = torch.zeros(B, H, W, xs)
    x[..., :C] = x_nchw.permute(0, 2, 3, 1)
    x = x.to(dev)
    Ho, Wo = (H + 2 * pad[0] - kh) // stride + 1, (W + 2 * pad[1] - kw) // stride + 1
    ys = -(-co // 4) * 4
    y = torch.full((B, Ho, Wo, ys), float("nan"), device=dev)
    r = None
    if res is not None:
        r = torch.zeros(B, Ho, Wo, ys)
        r[..., :co] = res.permute(0, 2, 3, 1)
        r = r.to(dev)
    _lib.check(_lib.load().pivlfn_conv2d_nhwc(conv.h, x.data_ptr(), xs, y.data_ptr(), ys, r.data_ptr() if r is not None else None, ys,
                                              B, H, W, stride, pad[0], pad[1], int(leaky), torch.cuda.current_stream(dev).cuda_stream), "conv2d")
    y = y.cpu()
    assert torch.all(y[..., co:] == 0)              # padding lanes are exact zeros
    return y[..., :co].permute(0, 3, 1, 2).contiguous()


CASES = [
    # cout, cin, kh, kw, stride, pad, H, W, B  -- one per kernel family of SURVEY.md appendix B
    (32, 3, 7, 7, 1, (3, 3), 40, 72, 2),          # NetC.conv1
    (32, 3, 7, 7, 1, (3, 3), 261, 530, 2),        # NetC.conv1 at >= 512 tiles of 8 x 32 outputs per image: taps packed into K, ragged edges
    (32, 32, 3, 3, 2, (1, 1), 64, 96, 1),         # NetC stride-2
    (64, 32, 3, 3, 2, (1, 1), 33, 47, 1),         # odd sizes
    (32, 32, 3, 3, 2, (1, 1), 262, 530, 1),       # NetC.conv2.0 at >= 256 tiles of 8 x 16 outputs: the whole-line stride-2 kernel, ragged edges
    (64, 32, 3, 3, 2, (1, 1), 261, 529, 2),       # NetC.conv3.0 on the same kernel (two channel blocks per wave), odd input size, batch 2
    (96, 96, 3, 3, 1, (1, 1), 24, 40, 1),
    (128, 49, 3, 3, 1, (1, 1), 32, 64, 1),        # conv_M.0
    (64, 128, 3, 3, 1, (1, 1), 19, 35, 2),
    (192, 128, 3, 3, 2, (1, 1), 16, 16, 1),       # NetC.conv6 (two N blocks)
    (64, 32, 1, 1, 1, (0, 0), 32, 32, 1),         # NetC_ext
    (128, 96, 1, 1, 1, (0, 0), 17, 23, 1),        # moduleFeat
    (49, 32, 7, 1, 1, (3, 0), 32, 40, 1),         # conv_dist_R.0 separable
    (49, 32, 7, 1, 1, (3, 0), 256, 272, 1),       # conv_dist_R.0 at >= 256 x 256: the streaming matrix-core kernel
    (49, 32, 7, 1, 1, (3, 0), 300, 261, 2),       # ragged, batch 2
    (49, 49, 1, 7, 1, (0, 3), 32, 40, 1),         # conv_dist_R.1
    (49, 49, 1, 7, 1, (0, 3), 256, 272, 1),       # conv_dist_R.1 at >= 256 x 256: its streaming matrix-core kernel (columns walk along x)
    (49, 49, 1, 7, 1, (0, 3), 261, 300, 2),       # ragged rows and columns, batch 2
    (25, 25, 1, 5, 1, (0, 2), 16, 32, 1),
    (9, 32, 3, 3, 1, (1, 1), 8, 8, 3),            # conv_dist_R level 5/6
    (2, 32, 7, 7, 1, (3, 3), 32, 48, 1),          # flow head on the matrix-core path
    (32, 64, 3, 3, 1, (1, 1), 2, 2, 1),           # tiny maps (64x64 input, level 6)
    (128, 386, 3, 3, 1, (1, 1), 8, 8, 1),         # conv_S.0 level 6 width
]


@pytest.mark.parametrize("case", CASES)
def test_conv2d_matches_torch(case, dev):
    co, ci, kh, kw, s, pad, H, W, B = case
    g = torch.Generator().manual_seed(co * 1000 + ci + kh)
    w = torch.randn(co, ci, kh, kw, generator=g) / (ci * kh * kw) ** 0.5
    b = torch.randn(co, generator=g) * 0.1
    x = torch.randn(B, ci, H, W, generator=g)
    conv = Conv(w, b)
    for leaky in (False, True):
        want = F.conv2d(x.double(), w.double(), b.double(), stride=s, padding=pad)
        if leaky:
            want = F.leaky_relu(want, 0.1)
        got = run(conv, x, s, pad, leaky, dev)
        err = (got.double() - want).abs().max().item()
        assert err < 2e-5 * max(1.0, want.abs().max().item()), (case, err)


def test_conv2d_residual_and_strided_input(dev):
    g = torch.Generator().manual_seed(3)
    w = torch.randn(32, 32, 3, 3, generator=g) / 17
    b = torch.randn(32, generator=g)
    x = torch.randn(1, 32, 20, 36, generator=g)
    res = torch.randn(1, 32, 20, 36, generator=g)
    conv = Conv(w, b)
    want = F.leaky_relu(F.conv2d(x, w, b, padding=1) + res, 0.1)
    got = run(conv, x, 1, (1, 1), True, dev, res=res, x_lanes=40)      # x lives in a wider (40-lane) tensor
    assert (got - want).abs().max().item() < 2e-5 * want.abs().max().item()


@pytest.mark.parametrize("k", [3, 5, 7])
def test_flow_head_kernel_matches_torch(k, dev):
    g = torch.Generator().manual_seed(k)
    w = torch.randn(2, 32, k, k, generator=g) / (32 * k * k) ** 0.5
    b = torch.randn(2, generator=g) * 0.1
    conv = Conv(w, b)
    for (B, H, W) in [(1, 16, 16), (2, 37, 53), (1, 3, 70)] + ([(1, 256, 300), (2, 260, 257)] if k == 7 else []):      # >= 256 x 256: the matrix-core head
        x = torch.randn(B, 32, H, W, generator=g)
        res = torch.randn(B, 2, H, W, generator=g)
        want = F.conv2d(x.double(), w.double(), b.double(), padding=k // 2) + res.double()
        xd = x.permute(0, 2, 3, 1).contiguous().to(dev)
        r4 = torch.zeros(B, H, W, 4)
        r4[..., :2] = res.permute(0, 2, 3, 1)
        r4 = r4.to(dev)
        out = torch.full((B, H, W, 4), float("nan"), device=dev)
        _lib.check(_lib.load().pivlfn_conv_head_nhwc(conv.h, xd.data_ptr(), r4.data_ptr(), out.data_ptr(), B, H, W,
                                                     torch.cuda.current_stream(dev).cuda_stream), "head")
        out = out.cpu()
        assert torch.all(out[..., 2:] == 0)
        got = out[..., :2].permute(0, 3, 1, 2).double()
        assert (got - want).abs().max().item() < 2e-5 * max(1.0, want.abs().max().item())


@pytest.mark.parametrize("kernel", ["direct", "wino"])
def test_images_of_2_gib_and_more_per_source(kernel, dev):
    """One image of a source may be of any size (a 2048 x 2048 pair's 128-channel level-1 tensors are exactly
    2^31 bytes and round 3's kernels refused them): the per-workgroup buffer descriptors start at the tile's first patch row, so the
    32-bit per-lane offsets only span those rows.  Forced here with a wide pixel stride: 264 x 260 pixels x 8192 floats = 2.25 GB
    for 8 real channels; the last rows lie beyond 2^31 bytes from the image start."""
    H, W, ci, co, xs = 264, 260, 8, 32, 8192
    g = torch.Generator().manual_seed(5)
    w = torch.randn(co, ci, 3, 3, generator=g) / (ci * 9) ** 0.5
    b = torch.randn(co, generator=g) * 0.1
    x = torch.randn(1, ci, H, W, generator=g)
    conv = Conv(w, b)
    xd = torch.zeros(1, H, W, xs, device=dev)
    xd[..., :ci] = x.permute(0, 2, 3, 1).to(dev)
    assert xd.numel() * 4 >= 2 ** 31
    ys = co
    y = torch.full((1, H, W, ys), float("nan"), device=dev)
    st = torch.cuda.current_stream(dev).cuda_stream
    lib = _lib.load()
    if kernel == "direct":
        _lib.check(lib.pivlfn_conv2d_nhwc(conv.h, xd.data_ptr(), xs, y.data_ptr(), ys, None, ys, 1, H, W, 1, 1, 1, 1, st), "conv2d")
    else:
        _lib.check(lib.pivlfn_conv2d_nhwc_wino(conv.h, xd.data_ptr(), xs, y.data_ptr(), ys, 1, H, W, 1, st), "conv2d_wino")
    want = F.leaky_relu(F.conv2d(x.double(), w.double(), b.double(), padding=1), 0.1)
    got = y.cpu().permute(0, 3, 1, 2).double()
    del xd
    assert (got - want).abs().max().item() < 1e-5 * max(1.0, want.abs().max().item())
