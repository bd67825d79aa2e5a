#!/usr/bin/env python3
"""Where a workgroup of the split-operand Winograd kernel (csrc/conv_wino_b3.hip) spends its time: s_memtime ticks of wave 0, summed over
its tiles (tools build: bash piv_liteflownet-pytorch_amd/csrc/build.sh tools).
  python tools/b3_stamps.py [--size 1024] [--layers 128x128,49x128]"""
import argparse
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import _toolslib  # noqa: E402
from pivlfn import _lib  # noqa: E402

NAMES = ["tile start", "full steps", "last step", "epilogue", "tile switch", "whole", "tiles", "steps", "phase 0", "phase 1", "phase 2", "phase 3", "barrier"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=1024)
    ap.add_argument("--layers", default="128x128,49x128,64x64")
    ap.add_argument("--terms", type=int, default=6)
    a = ap.parse_args()
    lib = _toolslib.load()
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream(dev).cuda_stream
    n = a.size
    for ci, co in [tuple(int(v) for v in s.split("x")) for s in a.layers.split(",")]:
        g = torch.Generator().manual_seed(ci * 7 + co)
        w = (torch.randn(co, ci, 3, 3, generator=g) / (ci * 9) ** 0.5).contiguous()
        b = torch.randn(co, generator=g).contiguous()
        h = ctypes.c_void_p()
        _lib.check(lib.pivlfn_conv_create(w.data_ptr(), b.data_ptr(), co, ci, 3, 3, ctypes.byref(h)), "create")
        xs = -(-ci // 4) * 4
        x = torch.randn(1, n, n, xs, device=dev)
        y = torch.empty(1, n, n, co, device=dev)
        for _ in range(3):
            _toolslib.check(lib, lib.pivlfn_conv2d_nhwc_wino_b3(h, x.data_ptr(), xs, y.data_ptr(), co, 1, n, n, 1, a.terms, st), "b3")
        buf = torch.zeros(1024 * 16, dtype=torch.int64, device=dev)
        _toolslib.set_stamp_buffer(lib, buf.data_ptr())
        _toolslib.check(lib, lib.pivlfn_conv2d_nhwc_wino_b3(h, x.data_ptr(), xs, y.data_ptr(), co, 1, n, n, 1, a.terms, st), "b3")
        torch.cuda.synchronize()
        _toolslib.set_stamp_buffer(lib, 0)
        t = buf.view(-1, 16).cpu().numpy().astype(float)
        t = t[t[:, 5] > 0]
        tiles, steps = t[:, 6].mean(), t[:, 7].mean()
        print(f"{ci}->{co} at {n}x{n}: {len(t)} workgroups, {tiles:.1f} tiles and {steps:.1f} full steps each; ticks per workgroup (100 MHz clock: 1 tick = 10 ns), mean:")
        print("   " + "  ".join(f"{NAMES[i]} {t[:, i].mean():.0f}" for i in (5, 0, 1, 2, 3, 4)))
        print(f"   per tile: start {t[:, 0].mean() / tiles:.1f}  last step {t[:, 2].mean() / tiles:.1f}  epilogue {t[:, 3].mean() / tiles:.1f}  switch {t[:, 4].mean() / tiles:.1f}"
              f"   per full step: {t[:, 1].mean() / max(steps, 1):.1f} = phases " + " ".join(f"{t[:, i].mean() / max(steps, 1):.1f}" for i in (8, 9, 10, 11)) + f" + barrier {t[:, 12].mean() / max(steps, 1):.1f}")
        print(f"   epilogue per tile: write halves + barriers {t[:, 13].mean() / tiles:.1f}  reads + stores {t[:, 14].mean() / tiles:.1f}  zeroing {t[:, 15].mean() / tiles:.1f}  rest {t[:, 3].mean() / tiles:.1f}")
        lib.pivlfn_conv_destroy(h)


if __name__ == "__main__":
    main()
