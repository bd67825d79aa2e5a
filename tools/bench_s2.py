#!/usr/bin/env python3
"""NetC.conv1 (7x7, 3 -> 32 at 1024^2: taps packed into K, conv_c3k7, against conv_k1 = knob 134217728) and NetC's 3x3 stride-2 layers from 32 channels (conv2.0 32->32 at 1024^2, conv3.0 32->64 at 512^2, both frames = batch 2): the
whole-line kernel (conv_s2c32) against the general direct kernel (knob 4194304), each checked against a float64 convolution."""
import ctypes
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, "tools")
import _toolslib
from bench_ops import _chk, time_it

lib = _toolslib.load()
dev = torch.device("cuda:0")
st = torch.cuda.current_stream(dev).cuda_stream
for (co, n) in [(32, 1024), (64, 512)]:
    g = torch.Generator().manual_seed(1)
    w = (torch.randn(co, 32, 3, 3, generator=g) / (32 * 9) ** 0.5).contiguous()
    b = torch.randn(co, generator=g).contiguous()
    h = ctypes.c_void_p()
    _chk(lib.pivlfn_conv_create(w.data_ptr(), b.data_ptr(), co, 32, 3, 3, ctypes.byref(h)), "create")
    x = torch.randn(2, n, n, 32, generator=g)
    want = F.leaky_relu(F.conv2d(x.permute(0, 3, 1, 2).double(), w.double(), b.double(), stride=2, padding=1), 0.1).permute(0, 2, 3, 1)
    xd = x.to(dev)
    for v in (0, 4194304):
        y = torch.full((2, n // 2, n // 2, co), float("nan"), device=dev)

        def fn():
            lib.pivlfn_tune(1, v)
            _chk(lib.pivlfn_conv2d_nhwc(h, xd.data_ptr(), 32, y.data_ptr(), co, None, 0, 2, n, n, 2, 1, 1, 1, st), "conv")
        tmin, tmed = time_it(fn, n=20, rounds=4)
        err = (y.cpu().double() - want).abs().max().item()
        mb = (x.numel() + y.numel()) * 4 / 1e6
        print(f"{n}x{n} s2 32->{co} B=2 knob {v:8d}: min {tmin:7.1f} med {tmed:7.1f} us   {mb / tmin:6.2f} TB/s of compulsory traffic   max err vs float64 {err:.2e}", flush=True)
    lib.pivlfn_tune(1, 0)
    lib.pivlfn_conv_destroy(h)

# NetC.conv1: 7 x 7, 3 real channels on 4 lanes, both frames
g = torch.Generator().manual_seed(2)
w = (torch.randn(32, 3, 7, 7, generator=g) / (3 * 49) ** 0.5).contiguous()
b = torch.randn(32, generator=g).contiguous()
h = ctypes.c_void_p()
_chk(lib.pivlfn_conv_create(w.data_ptr(), b.data_ptr(), 32, 3, 7, 7, ctypes.byref(h)), "create")
n = 1024
x = torch.zeros(2, n, n, 4)
x[..., :3] = torch.randn(2, n, n, 3, generator=g)
want = F.leaky_relu(F.conv2d(x[..., :3].permute(0, 3, 1, 2).double(), w.double(), b.double(), padding=3), 0.1).permute(0, 2, 3, 1)
xd = x.to(dev)
for v in (0, 134217728):
    y = torch.full((2, n, n, 32), float("nan"), device=dev)

    def fn():
        lib.pivlfn_tune(1, v)
        _chk(lib.pivlfn_conv2d_nhwc(h, xd.data_ptr(), 4, y.data_ptr(), 32, None, 0, 2, n, n, 1, 3, 3, 1, st), "conv")
    tmin, tmed = time_it(fn, n=20, rounds=4)
    err = (y.cpu().double() - want).abs().max().item()
    print(f"{n}x{n} conv1 7x7 3->32 B=2 knob {v:9d}: min {tmin:7.1f} med {tmed:7.1f} us   {2e-6 * 2 * n * n * 32 * 147 / tmin:6.1f} TFLOP/s real   max err vs float64 {err:.2e}", flush=True)
lib.pivlfn_tune(1, 0)
lib.pivlfn_conv_destroy(h)
