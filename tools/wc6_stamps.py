#!/usr/bin/env python3
"""Step timeline of the v6 warp+correlation kernel from in-kernel stamps (tools build): for one wave of every role, per step m:
  idx 3m-1 arrival at the barrier that opens the step, 3m leaving it, 3m+1 first wait served (consumers: dot products done).
Prints medians over workgroups in shader cycles: who arrives last, how long the barrier holds whom, how long the waits are."""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import _toolslib  # noqa: E402

LEVELS = {1: (64, 1024, 2), 2: (64, 512, 2), 3: (64, 256, 2), 4: (96, 128, 1), 5: (128, 64, 1)}
ROLES = ["consumer w0", "helper w7", "producer w8", "producer w15"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--level", type=int, default=3)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--mask", type=int, default=0)
    ap.add_argument("--steps", type=int, default=12)
    a = ap.parse_args()
    lib = _toolslib.load()
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream(dev).cuda_stream
    C, n, s = LEVELS[a.level]
    B = a.batch
    f1 = torch.randn(B, n, n, C, device=dev)
    f2 = torch.randn(B, n, n, C, device=dev)
    fl = torch.zeros(B, n, n, 4, device=dev)
    yy, xx = torch.meshgrid(torch.arange(n, device=dev, dtype=torch.float32), torch.arange(n, device=dev, dtype=torch.float32), indexing="ij")
    ph = torch.arange(B, device=dev, dtype=torch.float32).view(B, 1, 1)
    fl[..., 0] = 0.8 * torch.sin(yy * (6.2832 * 3 / n) + ph)
    fl[..., 1] = 0.8 * torch.cos(xx * (6.2832 * 2 / n) + 0.5 * ph)
    out = torch.empty(B, n // s, n // s, 56, device=dev)
    nwg = 256
    stamps = torch.zeros(nwg * 4 * 96, dtype=torch.int64, device=dev)

    def launch():
        _toolslib.check(lib, lib.pivlfn_warp_corr_nhwc(f1.data_ptr(), f2.data_ptr(), fl.data_ptr(), 1.25, out.data_ptr(), B, C, n, n, s, 1, st), "wc")
    lib.pivlfn_tune(2, a.mask | 32)             # 32: v6 stamps on
    for _ in range(5):
        launch()
    torch.cuda.synchronize()
    _toolslib.set_stamp_buffer(lib, stamps.data_ptr(), "wc")
    launch()
    torch.cuda.synchronize()
    _toolslib.set_stamp_buffer(lib, 0, "wc")
    lib.pivlfn_tune(2, 0)
    t = stamps.cpu().numpy().reshape(nwg, 4, 96).astype(np.int64)
    ok = t[:, 0, 0] > 0
    t = t[ok]
    print(f"level {a.level} batch {B} mask {a.mask}: {ok.sum()} workgroups stamped")
    base = t[:, :, 0].min(axis=1)            # first barrier leave of the workgroup
    for m in range(a.steps):
        line = [f"step {m:2d}"]
        for r, name in enumerate(ROLES):
            arr = t[:, r, 3 * m + 2] if 3 * m + 2 < 96 else None        # arrival at the barrier closing step m (= idx 3(m+1)-1)
            leave = t[:, r, 3 * m]
            w1 = t[:, r, 3 * m + 1]
            v = (leave > 0) & (w1 > 0) & (arr > 0)
            if not v.any():
                continue
            line.append(f"{name}: start {np.median((leave - base)[v]):7.0f} wait/dots {np.median((w1 - leave)[v]):6.0f} rest {np.median((arr - w1)[v]):6.0f} barrier {np.median((t[:, r, 3 * m + 3] - arr)[v]) if 3 * m + 3 < 96 else 0:6.0f}")
        print(" | ".join(line))


if __name__ == "__main__":
    main()
