#!/usr/bin/env python3
"""Regularization's separable distance convolutions of levels 1 and 2 (7x1 32->49, then 1x7 49->49, no activation): the streaming
matrix-core kernels (conv_col7 / conv_row7) against the general direct kernel, each checked against a float64 convolution.

  python tools/bench_dist.py [--sizes 1024,512] [--knobs 0,65536]     knob 65536: the general direct kernel"""
import argparse
import ctypes
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, "tools")
import _toolslib
from bench_ops import _chk, time_it

ap = argparse.ArgumentParser()
ap.add_argument("--sizes", default="1024,512")
ap.add_argument("--knobs", default="0,65536")
ap.add_argument("--stamps", action="store_true", help="conv_row7: per-wave timeline of one launch (s_memtime ticks, 100 MHz)")
args = ap.parse_args()
lib = _toolslib.load()
dev = torch.device("cuda:0")
st = torch.cuda.current_stream(dev).cuda_stream
for n in [int(v) for v in args.sizes.split(",")]:
    for (co, ci, kh, kw) in [(49, 32, 7, 1), (49, 49, 1, 7)]:
        g = torch.Generator().manual_seed(1)
        w = (torch.randn(co, ci, kh, kw, generator=g) / (ci * kh * kw) ** 0.5).contiguous()
        b = torch.randn(co, generator=g).contiguous()
        h = ctypes.c_void_p()
        _chk(lib.pivlfn_conv_create(w.data_ptr(), b.data_ptr(), co, ci, kh, kw, ctypes.byref(h)), "create")
        xs = -(-ci // 4) * 4
        x = torch.zeros(1, n, n, xs)
        x[..., :ci] = torch.randn(1, n, n, ci, generator=g)
        want = F.conv2d(x[..., :ci].permute(0, 3, 1, 2).double(), w.double(), b.double(), padding=(kh // 2, kw // 2)).permute(0, 2, 3, 1)
        xd = x.to(dev)
        for v in [int(k) for k in args.knobs.split(",")]:
            y = torch.full((1, n, n, 52), float("nan"), device=dev)

            def fn():
                lib.pivlfn_tune(1, v)
                _chk(lib.pivlfn_conv2d_nhwc(h, xd.data_ptr(), xs, y.data_ptr(), 52, None, 0, 1, n, n, 1, kh // 2, kw // 2, 0, st), "conv")
            tmin, tmed = time_it(fn, n=20, rounds=4)
            err = (y.cpu()[..., :co].double() - want).abs().max().item()
            print(f"{n}x{n} {kh}x{kw} {ci}->{co} knob {v:6d}: min {tmin:7.1f} med {tmed:7.1f} us   {2e-6 * n * n * co * ci * 7 / tmin:6.1f} TFLOP/s real"
                  f"   max err vs float64 {err:.2e}  padding lanes zero: {bool(torch.all(y[..., co:] == 0))}", flush=True)
        lib.pivlfn_tune(1, 0)
        if args.stamps and kw == 7:
            import numpy as np
            nwg = -(-(n // 16) * (n // 16) // 4)
            stamps = torch.zeros(nwg * 16 * 8, dtype=torch.int64, device=dev)
            ptr = stamps.data_ptr()
            lib.pivlfn_tune(5, ctypes.c_int32(ptr & 0xFFFFFFFF).value)
            lib.pivlfn_tune(6, ctypes.c_int32((ptr >> 32) & 0xFFFFFFFF).value)
            _chk(lib.pivlfn_conv2d_nhwc(h, xd.data_ptr(), xs, y.data_ptr(), 52, None, 0, 1, n, n, 1, kh // 2, kw // 2, 0, st), "conv")
            torch.cuda.synchronize()
            lib.pivlfn_tune(5, 0)
            lib.pivlfn_tune(6, 0)
            t = stamps.view(-1, 8).cpu().numpy()
            t = t[t[:, 4] > 0]
            t0 = t[:, 0].min()
            print(f"  {len(t)} units, launch span {(t[:, 4].max() - t0) / 100:.1f} us")
            for blk in range(4):
                u = t[t[:, 5] == blk]
                print(f"  block {blk}: n {len(u)}  weights {np.mean(u[:, 1] - u[:, 0]) / 100:6.2f} us  first column {np.mean(u[:, 2] - u[:, 1]) / 100:6.2f}  "
                      f"columns {np.mean(u[:, 3] - u[:, 2]) / 100:6.2f} (min {np.min(u[:, 3] - u[:, 2]) / 100:.2f} max {np.max(u[:, 3] - u[:, 2]) / 100:.2f})  stores {np.mean(u[:, 4] - u[:, 3]) / 100:6.2f}")
            # a CU's timeline: the units of the workgroups that ran on the CU of workgroup 0
            hw = t[:, 6]
            key = hw & 0xFF00 | ((hw >> 13) & 7) << 16
            print("  end times of the units (us from the launch start), percentiles 10/50/90/100: " +
                  " ".join(f"{np.percentile(t[:, 4] - t0, q) / 100:.1f}" for q in (10, 50, 90, 100)))
            starts = np.sort(t[:, 0] - t0) / 100
            print("  unit start times, percentiles 0/25/50/75/100: " + " ".join(f"{np.percentile(starts, q):.1f}" for q in (0, 25, 50, 75, 100)))
        lib.pivlfn_conv_destroy(h)
