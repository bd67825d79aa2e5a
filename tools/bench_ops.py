#!/usr/bin/env python3
"""A/B micro-benchmarks of single kernels at the shapes of the 1024x1024 PIV forward (one process, interleaved rounds).

  python tools/bench_ops.py warp_corr [--batch 1] [--variants 1,4,5,0]   (1 first generation, 4 v3 one pixel per lane, 5 v4, 0 shipped policy)
Times N back-to-back launches between two events on the current stream (so each figure includes one ~1.5 us
kernel boundary) and checks that all variants agree.
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "piv_liteflownet-pytorch_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import _toolslib  # noqa: E402


def _chk(rc, what=""):
    _toolslib.check(_toolslib.load(), rc, what)


LEVELS = {1: (64, 1024, 2), 2: (64, 512, 2), 3: (64, 256, 2), 4: (96, 128, 1), 5: (128, 64, 1), 6: (192, 32, 1)}


def time_it(fn, n=50, rounds=5):
    ts = []
    for _ in range(rounds):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        fn()
        torch.cuda.synchronize()
        a.record()
        for _ in range(n):
            fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) / n * 1e3)
    return min(ts), sorted(ts)[len(ts) // 2]


def bench_warp_corr(args):
    lib = _toolslib.load()
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream(dev).cuda_stream
    B = args.batch
    variants = [int(v) for v in args.variants.split(",")]
    for L in [int(x) for x in args.levels.split(",")]:
        C, n, s = LEVELS[L]
        n = n * args.size // 1024
        f1 = torch.randn(B, n, n, C, device=dev)
        f2 = torch.randn(B, n, n, C, device=dev)
        fl = torch.zeros(B, n, n, 4, device=dev)
        fl[..., :2] = torch.randn(B, n, n, 2, device=dev) * 0.8
        no = -(-n // s)
        flow_ptr = fl.data_ptr() if L < 6 else None
        outs = {}
        alg = 4 * B * (C * no * no + C * n * n + (2 * n * n if L < 6 else 0) + 49 * no * no)
        fns = {}
        for v in variants:
            out = torch.empty(B, no, no, 56, device=dev)
            outs[v] = out

            def fn(v=v, out=out):
                lib.pivlfn_tune(0, v)
                _chk(lib.pivlfn_warp_corr_nhwc(f1.data_ptr(), f2.data_ptr(), flow_ptr, 1.25, out.data_ptr(), B, C, n, n, s, 1, st), "wc")
            fns[v] = fn
        # interleaved rounds (clock state and neighbours are shared by all variants): min / median over the rounds
        times = {v: [] for v in variants}
        for _ in range(3):
            for v in variants:
                fns[v]()
        for rnd in range(args.rounds):
            for v in (variants if rnd % 2 == 0 else variants[::-1]):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                fns[v]()
                torch.cuda.synchronize()
                a.record()
                for _ in range(50):
                    fns[v]()
                b.record()
                torch.cuda.synchronize()
                times[v].append(a.elapsed_time(b) / 50 * 1e3)
        for v in variants:
            tmin, tmed = min(times[v]), sorted(times[v])[len(times[v]) // 2]
            print(f"L{L} B={B} C={C} {n}x{n} s={s} variant {v}: min {tmin:8.2f} us  med {tmed:8.2f} us   "
                  f"{alg / tmin / 1e3:8.1f} GB/s algorithmic ({alg / 1e6:.2f} MB)", flush=True)
        ref = outs[variants[0]]
        for v in variants[1:]:
            d = (outs[v] - ref).abs().max().item()
            print(f"    variant {v} vs {variants[0]}: max abs diff {d:.3e}")
    lib.pivlfn_tune(0, 0)


def bench_wc_ablate(args):
    """Ablation of the shipped warp+correlation kernel (mask bits: 1 no dot products, 2 no gathers, 4 no store, 8 empty)."""
    lib = _toolslib.load()
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream(dev).cuda_stream
    B = args.batch
    for L in [int(x) for x in args.levels.split(",")]:
        C, n, s = LEVELS[L]
        f1 = torch.randn(B, n, n, C, device=dev)
        f2 = torch.randn(B, n, n, C, device=dev)
        fl = torch.zeros(B, n, n, 4, device=dev)
        if args.smooth:      # a smooth sub-pixel flow, as the network's own are (bench.py's roofline_batch8 uses the same): every byte of f2 is touched
            yy, xx = torch.meshgrid(torch.arange(n, device=dev, dtype=torch.float32), torch.arange(n, device=dev, dtype=torch.float32), indexing="ij")
            ph = torch.arange(B, device=dev, dtype=torch.float32).view(B, 1, 1)
            fl[..., 0] = 0.8 * torch.sin(yy * (6.2832 * 3 / n) + ph)
            fl[..., 1] = 0.8 * torch.cos(xx * (6.2832 * 2 / n) + 0.5 * ph)
        else:
            fl[..., :2] = torch.randn(B, n, n, 2, device=dev) * 0.8
        no = -(-n // s)
        out = torch.empty(B, no, no, 56, device=dev)
        for mask in [int(m) for m in args.masks.split(",")]:
            def fn():
                _chk(lib.pivlfn_warp_corr_nhwc(f1.data_ptr(), f2.data_ptr(), fl.data_ptr() if L < 6 else None, 1.25, out.data_ptr(), B, C, n, n, s, 1, st), "wc")
            lib.pivlfn_tune(2, mask)
            tmin, tmed = time_it(fn)
            print(f"L{L} B={B} ablation mask {mask}: min {tmin:8.2f} us  med {tmed:8.2f} us", flush=True)
        lib.pivlfn_tune(2, 0)


def _level_layers(L, n):
    """Every dense conv of level L's Matching / Subpixel / Regularization stacks (src/models.py:154-163, 197-207, 236-267) at
    the level's resolution n x n; staged input channels (multiples of 4) as pivlfn_forward stages them."""
    return [(f"L{L} M.0 49->128", 128, 56, 3, 1, n, 1), (f"L{L} M.2 128->64", 64, 128, 3, 1, n, 1), (f"L{L} M.4 64->32", 32, 64, 3, 1, n, 1),
            (f"L{L} S.0 130->128", 128, 136, 3, 1, n, 1), (f"L{L} S.2 128->64", 64, 128, 3, 1, n, 1), (f"L{L} S.4 64->32", 32, 64, 3, 1, n, 1),
            (f"L{L} R.0 131->128", 128, 132, 3, 1, n, 1), (f"L{L} R.2 128->128", 128, 128, 3, 1, n, 1), (f"L{L} R.4 128->64", 64, 128, 3, 1, n, 1),
            (f"L{L} R.6 64->64", 64, 64, 3, 1, n, 1), (f"L{L} R.8 64->32", 32, 64, 3, 1, n, 1), (f"L{L} R.10 32->32", 32, 32, 3, 1, n, 1)]


CONV_SHAPES = _level_layers(1, 1024) + _level_layers(2, 512) + _level_layers(3, 256) + [
    # name, cout, cin, k, stride, H(=W) at the 1024x1024 PIV forward, batch multiplier
    ("L3 conv 128->128", 128, 128, 3, 1, 256, 1),
    ("L4 conv 128->128", 128, 128, 3, 1, 128, 1), ("L5 conv 128->128", 128, 128, 3, 1, 64, 1), ("L6 conv 128->128", 128, 128, 3, 1, 32, 1),
    ("L6 S.conv_S.0 386->128", 128, 392, 3, 1, 32, 1), ("L5 S.conv_S.0 258->128", 128, 264, 3, 1, 64, 1),
    ("NetC.conv1 7x7 3->32", 32, 4, 7, 1, 1024, 2), ("NetC.conv2.0 s2 32->32", 32, 32, 3, 2, 1024, 2),
    ("NetC.conv2.2 32->32 @512", 32, 32, 3, 1, 512, 2), ("NetC.conv3.0 s2 32->64", 64, 32, 3, 2, 512, 2),
    ("NetC.conv3.2 64->64 @256", 64, 64, 3, 1, 256, 2), ("NetC.conv5.0 s2 96->128", 128, 96, 3, 2, 128, 2),
    ("L1 moduleFeat 1x1 32->128", 128, 32, 1, 1, 1024, 1), ("L1 NetC_ext 1x1 32->64", 64, 32, 1, 1, 1024, 2),
    ("L1 dist 7x1 32->49", 49, 32, (7, 1), 1, 1024, 1),
    ("L1 dist 1x7 49->49", 49, 49, (1, 7), 1, 1024, 1),
    ("L2 dist 7x1 32->49", 49, 32, (7, 1), 1, 512, 1), ("L2 dist 1x7 49->49", 49, 49, (1, 7), 1, 512, 1),
]


def bench_conv(args):
    import ctypes
    lib = _toolslib.load()
    lib.pivlfn_tune(3, args.tune3)
    lib.pivlfn_tune(7, args.tune7)
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream(dev).cuda_stream
    variants = [int(v) for v in args.variants.split(",")]
    for name, co, ci, k, s, n, bm in CONV_SHAPES:
        if args.filter and args.filter not in name:
            continue
        kh, kw = (k, k) if isinstance(k, int) else k
        B = args.batch * bm
        w = (torch.randn(co, ci, kh, kw) / (ci * kh * kw) ** 0.5).contiguous()
        b = torch.randn(co).contiguous()
        h = ctypes.c_void_p()
        _chk(lib.pivlfn_conv_create(w.data_ptr(), b.data_ptr(), co, ci, kh, kw, ctypes.byref(h)), "create")
        xs = -(-ci // 4) * 4                      # stored lanes of the input (zeros behind the real channels)
        x = torch.randn(B, n, n, xs, device=dev)
        x[..., ci:] = 0
        no = (n + 2 * (kh // 2) - kh) // s + 1
        mo = (n + 2 * (kw // 2) - kw) // s + 1
        ys = -(-co // 4) * 4
        flop = 2.0 * B * no * mo * co * ci * kh * kw
        outs = {}
        xh = None
        for v in variants:
            # variants >= 100000 select the fp16-multiplicand kernel: even = fp32 in / fp32 out, odd = fp16 in / fp16 out;
            # (v - 100000) >> 1 goes to the tuning knob (pivlfn_tune(1, .)): 100000/100001 shipped policy, 100128/100129 = knob 64
            # variants >= 2000000: the split-operand fp32 kernel (conv_split.hip), stride-1 layers only; 2xxxxxx = six terms,
            # 3xxxxxx = three terms; v % 1000000 goes to the knob
            split = v >= 2000000
            if split and not (s == 1 or (s == 2 and v >= 3000000 and kh == 3 and kw == 3)):
                continue
            f16io = 100000 <= v < 2000000 and (v & 1) == 1
            y = torch.empty(B, no, mo, ys, device=dev, dtype=torch.float16 if f16io else torch.float32)
            outs[v] = y
            if f16io and xh is None and ci % 8 == 0:
                xh = x.half()

            def fn(v=v, y=y, f16io=f16io):
                lib.pivlfn_tune(1, (v % 1000000) if split else ((v - 100000) >> 1 if v >= 100000 else v))
                if split:
                    _chk(lib.pivlfn_conv2d_nhwc_split(h, x.data_ptr(), xs, y.data_ptr(), ys, B, n, n, s, kh // 2, kw // 2, 1, 3 if v >= 3000000 else 6, st), "conv")
                elif v >= 100000:
                    if f16io and ci % 8 == 0:
                        _chk(lib.pivlfn_conv2d_nhwc_f16(h, xh.data_ptr(), ci, 1, y.data_ptr(), ys, 1, B, n, n, s, kh // 2, kw // 2, 1, st), "conv")
                    else:
                        _chk(lib.pivlfn_conv2d_nhwc_f16(h, x.data_ptr(), xs, 0, y.data_ptr(), ys, 1 if f16io else 0, B, n, n, s, kh // 2, kw // 2, 1, st), "conv")
                else:
                    _chk(lib.pivlfn_conv2d_nhwc(h, x.data_ptr(), xs, y.data_ptr(), ys, None, 0, B, n, n, s, kh // 2, kw // 2, 1, st), "conv")
            tmin, tmed = time_it(fn, n=10 if flop > 2e10 else 30, rounds=4)
            print(f"{name:28s} B={B} variant {v}: min {tmin:9.1f} us  med {tmed:9.1f} us  {flop / tmin / 1e6:7.1f} TFLOP/s (staged K)", flush=True)
        for v in variants[1:]:
            if v not in outs or variants[0] not in outs:
                continue
            d = (outs[v].float() - outs[variants[0]].float()).abs().max().item()
            print(f"    variant {v} vs {variants[0]}: max abs diff {d:.3e}")
        lib.pivlfn_conv_destroy(h)
    lib.pivlfn_tune(1, 0)


def bench_conv_stamps(args):
    """Phase times of the fp16 conv kernel (wave 0 of every workgroup, s_memtime ticks): where a workgroup's time goes."""
    import ctypes
    lib = _toolslib.load()
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream(dev).cuda_stream
    for name, co, ci, k, s, n, bm in CONV_SHAPES:
        if args.filter and args.filter not in name:
            continue
        kh, kw = (k, k) if isinstance(k, int) else k
        B = args.batch * bm
        w = (torch.randn(co, ci, kh, kw) / (ci * kh * kw) ** 0.5).contiguous()
        b = torch.randn(co).contiguous()
        h = ctypes.c_void_p()
        _chk(lib.pivlfn_conv_create(w.data_ptr(), b.data_ptr(), co, ci, kh, kw, ctypes.byref(h)), "create")
        fp32 = args.tune3 == -1                     # --tune3 -1: stamp the fp32 kernel instead
        f16 = ci % 8 == 0 and not fp32
        x = torch.randn(B, n, n, ci, device=dev)
        xin = x.half() if f16 else x
        no = (n + 2 * (kh // 2) - kh) // s + 1
        mo = (n + 2 * (kw // 2) - kw) // s + 1
        ys = -(-co // 4) * 4
        y = torch.empty(B, no, mo, ys, device=dev, dtype=torch.float32 if fp32 else torch.float16)
        stamps = torch.zeros(65536 * 8, dtype=torch.int64, device=dev)
        ptr = stamps.data_ptr()

        def run():
            if fp32:
                _chk(lib.pivlfn_conv2d_nhwc(h, x.data_ptr(), ci, y.data_ptr(), ys, None, 0, B, n, n, s, kh // 2, kw // 2, 1, st), "conv")
            else:
                _chk(lib.pivlfn_conv2d_nhwc_f16(h, xin.data_ptr(), ci, 1 if f16 else 0, y.data_ptr(), ys, 1, B, n, n, s, kh // 2, kw // 2, 1, st), "conv")
        for _ in range(3):
            run()
        lib.pivlfn_tune(5, ctypes.c_int32(ptr & 0xFFFFFFFF).value)
        lib.pivlfn_tune(6, ctypes.c_int32((ptr >> 32) & 0xFFFFFFFF).value)
        run()
        torch.cuda.synchronize()
        lib.pivlfn_tune(5, 0)
        lib.pivlfn_tune(6, 0)
        t = stamps.view(-1, 8).cpu().numpy()
        t = t[t[:, 6] > 0]
        names = ["commit+wait", "barrier1", "load issue", "tap loop", "barrier2", "epilogue", "whole"]
        tot = t[:, 6].mean()
        print(f"{name}: {len(t)} workgroups, ticks per workgroup (mean): " +
              "  ".join(f"{nm} {t[:, i].mean():.0f} ({100 * t[:, i].mean() / tot:.0f}%)" for i, nm in enumerate(names)))
        span = (t[:, 7] + t[:, 6]).max() - t[:, 7].min()
        print(f"    kernel span {span} ticks; per-workgroup whole min {t[:, 6].min()} max {t[:, 6].max()}")
        lib.pivlfn_conv_destroy(h)


def bench_head(args):
    """The 32 -> 2 k x k flow heads (conv_M.6 / conv_S.6) at the fine levels: knob 524288 + 8192 = one pixel per lane (round 1),
    524288 = four rows per lane (round 2), 0 = shipped (7 x 7 on >= 256 x 256 images: the matrix-core formulation)."""
    import ctypes
    lib = _toolslib.load()
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream(dev).cuda_stream
    for name, n, k in (("L1 head 7x7", 1024, 7), ("L2 head 7x7", 512, 7), ("L3 head 5x5", 256, 5)):
        n = n * args.size // 1024
        B = args.batch
        w = (torch.randn(2, 32, k, k) / (32 * k * k) ** 0.5).contiguous()
        b = torch.randn(2).contiguous()
        h = ctypes.c_void_p()
        _chk(lib.pivlfn_conv_create(w.data_ptr(), b.data_ptr(), 2, 32, k, k, ctypes.byref(h)), "create")
        x = torch.randn(B, n, n, 32, device=dev)
        res = torch.randn(B, n, n, 4, device=dev)
        outs = {}
        for v in (524288 + 8192, 524288, 0):
            y = torch.empty(B, n, n, 4, device=dev)
            outs[v] = y

            def fn(v=v, y=y):
                lib.pivlfn_tune(1, v)
                _chk(lib.pivlfn_conv_head_nhwc(h, x.data_ptr(), res.data_ptr(), y.data_ptr(), B, n, n, st), "head")
            tmin, tmed = time_it(fn, n=20, rounds=4)
            flop = 2.0 * B * n * n * 2 * 32 * k * k
            print(f"{name} B={B} {n}x{n} knob {v}: min {tmin:8.1f} us  med {tmed:8.1f} us  {flop / tmin / 1e6:6.1f} TFLOP/s", flush=True)
        want = torch.nn.functional.conv2d(x[:1].permute(0, 3, 1, 2).double().cpu(), w.double(), b.double(), padding=k // 2).permute(0, 2, 3, 1)
        want = want + res[:1, ..., :2].double().cpu()
        for v in (524288 + 8192, 524288, 0):
            err = (outs[v][:1, ..., :2].double().cpu() - want).abs().max().item() / want.abs().max().item()
            pad = outs[v][..., 2:].abs().max().item()
            print(f"    knob {v}: max rel err vs float64 conv {err:.2e}, padding lanes max {pad:.1e}")
        lib.pivlfn_conv_destroy(h)
    lib.pivlfn_tune(1, 0)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("what", choices=["warp_corr", "wc_ablate", "conv", "conv_stamps", "head"])
    ap.add_argument("--filter", default="")
    ap.add_argument("--tune3", type=int, default=0, help="ablation mask of the fp16 conv kernel (pivlfn_tune(3, mask))")
    ap.add_argument("--tune7", type=int, default=0, help="ablation mask of the fp32 conv kernel in a -DPIVLFN_STAMPS build (pivlfn_tune(7, mask))")
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--size", type=int, default=1024, help="input image size the level shapes are derived from")
    ap.add_argument("--variants", default="6,8,5")
    ap.add_argument("--levels", default="3,1,2,4,5,6")
    ap.add_argument("--smooth", action="store_true", help="wc_ablate: smooth flow instead of per-pixel noise")
    ap.add_argument("--masks", default="0,8,1,2,4,3,7", help="wc_ablate: pivlfn_tune(2, .) masks")
    ap.add_argument("--rounds", type=int, default=8, help="interleaved timing rounds per variant (warp_corr)")
    a = ap.parse_args()
    {"conv_stamps": bench_conv_stamps, "warp_corr": bench_warp_corr, "wc_ablate": bench_wc_ablate, "conv": bench_conv, "head": bench_head}[a.what](a)
