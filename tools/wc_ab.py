#!/usr/bin/env python3
"""Interleaved same-box A/B of the channels-last warp+correlation launch between the current libpivlfn.so and another build (default
build/libpivlfn_prev.so: round 4's sources -- flow and output through one descriptor per launch, not per tile / per tap fetch):
level 1 and level 3 of a 1024 x 1024 pair, and the level-3 shape of 8 pairs.
  python tools/wc_ab.py [--other build/libpivlfn_prev.so] [--rounds 9] [--n 40]"""
import argparse
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "piv_liteflownet-pytorch_amd"))
from pivlfn import _lib  # noqa: E402


def load(path):
    lib = ctypes.CDLL(path)
    for name, (res, args) in _lib.SIGNATURES.items():
        if hasattr(lib, name):
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
    return lib


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--other", default=os.path.join(ROOT, "build", "libpivlfn_prev.so"))
    ap.add_argument("--rounds", type=int, default=9)
    ap.add_argument("--n", type=int, default=40)
    a = ap.parse_args()
    libs = {"current": load(_lib.LIB_PATH), "other": load(a.other)}
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream(dev).cuda_stream
    for name, B, n in (("level 1, one pair", 1, 1024), ("level 3, 8 pairs", 8, 256), ("level 3, one pair", 1, 256)):
        C, s = 64, 2
        g = torch.Generator(device=dev).manual_seed(n + B)
        f1 = torch.randn(B, n, n, C, device=dev, generator=g)
        f2 = torch.randn(B, n, n, C, device=dev, generator=g)
        fl = torch.zeros(B, n, n, 4, device=dev)
        yy, xx = torch.meshgrid(torch.arange(n, device=dev, dtype=torch.float32), torch.arange(n, device=dev, dtype=torch.float32), indexing="ij")
        fl[..., 0] = 0.8 * torch.sin(yy * (6.2832 * 3 / n))
        fl[..., 1] = 0.8 * torch.cos(xx * (6.2832 * 2 / n))
        outs = {k: torch.empty(B, n // s, n // s, 56, device=dev) for k in libs}
        alg = 4 * (C * (n // s) ** 2 + C * n * n + 2 * n * n + 49 * (n // s) ** 2) * B
        times = {k: [] for k in libs}

        def run(k):
            rc = libs[k].pivlfn_warp_corr_nhwc(f1.data_ptr(), f2.data_ptr(), fl.data_ptr(), 1.25, outs[k].data_ptr(), B, C, n, n, s, 1, st)
            assert rc == 0, libs[k].pivlfn_last_error()
        for k in libs:
            run(k)
        for rnd in range(a.rounds):
            for k in (list(libs) if rnd % 2 == 0 else list(libs)[::-1]):
                run(k)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(a.n):
                    run(k)
                e1.record()
                torch.cuda.synchronize()
                times[k].append(e0.elapsed_time(e1) / a.n * 1e3)
        med = {k: sorted(v)[len(v) // 2] for k, v in times.items()}
        print(f"{name} ({alg / 1e6:.1f} MB algorithmic): " + "   ".join(
            f"{k}: min {min(v):7.2f} med {med[k]:7.2f} us = {alg / med[k] / 8e6:.3f} of 8 TB/s" for k, v in times.items()) +
            f"   current / other {med['current'] / med['other']:.3f} (med)   bits equal: {bool(torch.equal(outs['current'], outs['other']))}", flush=True)


if __name__ == "__main__":
    main()
