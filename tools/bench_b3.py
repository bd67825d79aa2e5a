#!/usr/bin/env python3
"""Interleaved timing of the split-operand Winograd kernel (csrc/conv_wino_b3.hip) from several builds of libpivlfn.so, next to the
fp32-instruction Winograd kernel of the first one.
  python tools/bench_b3.py [--libs a.so,b.so] [--size 1024] [--levels 1,2] [--layers 128x128,...] [--terms 6]
The first library defaults to the production one; results of the others are compared with its bits."""
import argparse
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "piv_liteflownet-pytorch_amd"))
from pivlfn import _lib  # noqa: E402

LAYERS = [(49, 128), (128, 64), (130, 128), (131, 128), (128, 128), (64, 64)]


def load(path):
    lib = ctypes.CDLL(path)
    for name, (res, args) in _lib.SIGNATURES.items():
        if hasattr(lib, name):
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
    return lib


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--libs", default="")
    ap.add_argument("--size", type=int, default=1024)
    ap.add_argument("--levels", default="1")
    ap.add_argument("--layers", default="")
    ap.add_argument("--terms", type=int, default=6)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--n", type=int, default=10)
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--wino-only", action="store_true", help="time the fp32-instruction Winograd kernel of every library instead (any channel counts)")
    a = ap.parse_args()
    paths = [_lib.LIB_PATH] + [p for p in a.libs.split(",") if p]
    libs = [load(p) for p in paths]
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream(dev).cuda_stream
    layers = [tuple(int(v) for v in s.split("x")) for s in a.layers.split(",")] if a.layers else LAYERS
    for L in [int(x) for x in a.levels.split(",")]:
        n = a.size >> (L - 1)
        for ci, co in layers:
            g = torch.Generator().manual_seed(ci * 7 + co)
            w = (torch.randn(co, ci, 3, 3, generator=g) / (ci * 9) ** 0.5).contiguous()
            b = torch.randn(co, generator=g).contiguous()
            xs = -(-ci // 4) * 4
            x = torch.randn(a.batch, n, n, xs, device=dev)
            hs, ys, fns = [], {}, {}
            for i, lib in enumerate(libs):
                h = ctypes.c_void_p()
                assert lib.pivlfn_conv_create(w.data_ptr(), b.data_ptr(), co, ci, 3, 3, ctypes.byref(h)) == 0
                hs.append(h)
                if i == 0:
                    ys["wino"] = torch.empty(a.batch, n, n, co, device=dev)
                    fns["wino"] = lambda lib=lib, h=h: lib.pivlfn_conv2d_nhwc_wino(h, x.data_ptr(), xs, ys["wino"].data_ptr(), co, a.batch, n, n, 1, st)
                if a.wino_only:
                    if i:
                        k = f"wino[{i}]"
                        ys[k] = torch.empty(a.batch, n, n, co, device=dev)
                        fns[k] = lambda lib=lib, h=h, k=k: lib.pivlfn_conv2d_nhwc_wino(h, x.data_ptr(), xs, ys[k].data_ptr(), co, a.batch, n, n, 1, st)
                    continue
                k = f"b3[{i}]"
                ys[k] = torch.empty(a.batch, n, n, co, device=dev)
                fns[k] = lambda lib=lib, h=h, k=k: lib.pivlfn_conv2d_nhwc_wino_b3(h, x.data_ptr(), xs, ys[k].data_ptr(), co, a.batch, n, n, 1, a.terms, st)
            times = {k: [] for k in fns}
            for k in fns:
                assert fns[k]() == 0, (k, libs[0].pivlfn_last_error())
            for rnd in range(a.rounds):
                for k in (list(fns) if rnd % 2 == 0 else list(fns)[::-1]):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    fns[k]()
                    torch.cuda.synchronize()
                    e0.record()
                    for _ in range(a.n):
                        fns[k]()
                    e1.record()
                    torch.cuda.synchronize()
                    times[k].append(e0.elapsed_time(e1) / a.n * 1e3)
            ref = ys["wino"]
            out = f"L{L} {n}x{n} B={a.batch} {ci:3d}->{co:3d}:"
            for k, v in times.items():
                d = (ys[k] - ref).abs().max().item() / ref.abs().max().item()
                eq = "" if k in ("wino", "b3[0]") else (f" bits==wino: {bool(torch.equal(ys[k], ys['wino']))}" if a.wino_only else f" bits==b3[0]: {bool(torch.equal(ys[k], ys['b3[0]']))}")
                out += f"   {k} min {min(v):7.1f} med {sorted(v)[len(v) // 2]:7.1f} us (x{min(times['wino']) / min(v):4.2f}; diff {d:.1e}{eq})"
            print(out, flush=True)
            for lib, h in zip(libs, hs):
                lib.pivlfn_conv_destroy(h)


if __name__ == "__main__":
    main()
