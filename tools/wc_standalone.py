#!/usr/bin/env python3
"""Standalone launches of one warp+correlation shape of the 1024x1024 PIV forward through the production library (for counter
passes: every warp_corr dispatch of the process is the shape asked for).  Smooth sub-pixel flow, as the network's own.
  python3 tools/wc_standalone.py --level 3 --batch 8 --launches 20"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "piv_liteflownet-pytorch_amd"))
from pivlfn import _lib  # noqa: E402

LEVELS = {1: (64, 1024, 2), 2: (64, 512, 2), 3: (64, 256, 2), 4: (96, 128, 1), 5: (128, 64, 1), 6: (192, 32, 1)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--level", type=int, default=3)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--size", type=int, default=1024)
    ap.add_argument("--launches", type=int, default=20)
    a = ap.parse_args()
    lib = _lib.load()
    dev = torch.device("cuda:0")
    C, n, s = LEVELS[a.level]
    n = n * a.size // 1024
    B = a.batch
    g = torch.Generator(device=dev).manual_seed(7)
    f1 = torch.randn(B, n, n, C, device=dev, generator=g)
    f2 = torch.randn(B, n, n, C, device=dev, generator=g)
    yy, xx = torch.meshgrid(torch.arange(n, device=dev, dtype=torch.float32), torch.arange(n, device=dev, dtype=torch.float32), indexing="ij")
    ph = torch.arange(B, device=dev, dtype=torch.float32).view(B, 1, 1)
    fl = torch.zeros(B, n, n, 4, device=dev)
    fl[..., 0] = 0.8 * torch.sin(yy * (6.2832 * 3 / n) + ph)
    fl[..., 1] = 0.8 * torch.cos(xx * (6.2832 * 2 / n) + 0.5 * ph)
    out = torch.empty(B, -(-n // s), -(-n // s), 56, device=dev)
    st = torch.cuda.current_stream(dev).cuda_stream
    torch.cuda.synchronize()
    for _ in range(a.launches):
        _lib.check(lib.pivlfn_warp_corr_nhwc(f1.data_ptr(), f2.data_ptr(), fl.data_ptr() if a.level < 6 else None, 1.25, out.data_ptr(),
                                             B, C, n, n, s, 1, st), "wc")
    torch.cuda.synchronize()
    print(f"level {a.level} batch {B}: {a.launches} launches, out checksum {out.double().abs().sum().item():.6e}")


if __name__ == "__main__":
    main()
