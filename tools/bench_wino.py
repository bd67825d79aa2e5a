#!/usr/bin/env python3
"""Per-layer timing of the 3x3 stride-1 layers of a 1024x1024 PIV forward: Winograd F(2x2,3x3) kernel (conv_wino.hip) beside the
direct fp32 kernel (conv_mfma.hip), both on v_mfma_f32_32x32x2_f32, interleaved rounds in one process (production library).

  python tools/bench_wino.py [--size 1024] [--levels 1,2] [--rounds 5] [--layers 128x128,128x64,...]
TFLOP/s are those of the DIRECT algorithm's multiply count (2 * 9 * Cin * Cout per pixel) for both, so the ratio is the speed-up.
"""
import argparse
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "piv_liteflownet-pytorch_amd"))
from pivlfn import _lib  # noqa: E402

LAYERS = [(49, 128), (128, 64), (64, 32), (130, 128), (131, 128), (128, 128), (64, 64), (32, 32)]


def stamps(a):
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import _toolslib
    lib = _toolslib.load()
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream(dev).cuda_stream
    layers = [tuple(int(v) for v in s.split("x")) for s in a.layers.split(",")] if a.layers else [(128, 128)]
    names = ["prologue", "block 0 (+W0 issue)", "block 1 + reads + transform", "W1 issue + commit", "barrier", "epilogue", "whole"]
    for L in [int(x) for x in a.levels.split(",")]:
        n = a.size >> (L - 1)
        for ci, co in layers:
            g = torch.Generator().manual_seed(ci * 7 + co)
            w = (torch.randn(co, ci, 3, 3, generator=g) / (ci * 9) ** 0.5).contiguous()
            b = torch.randn(co, generator=g).contiguous()
            h = ctypes.c_void_p()
            _lib.check(lib.pivlfn_conv_create(w.data_ptr(), b.data_ptr(), co, ci, 3, 3, ctypes.byref(h)), "create")
            xs = -(-ci // 4) * 4
            x = torch.randn(a.batch, n, n, xs, device=dev)
            y = torch.empty(a.batch, n, n, co, device=dev)
            buf = torch.zeros(8192 * 8, dtype=torch.int64, device=dev)
            ptr = buf.data_ptr()
            for one in (0, 1):
                lib.pivlfn_tune(1, 1048576 if one else 0)
                for _ in range(3):
                    _lib.check(lib.pivlfn_conv2d_nhwc_wino(h, x.data_ptr(), xs, y.data_ptr(), co, a.batch, n, n, 1, st), "wino")
                buf.zero_()
                lib.pivlfn_tune(5, ctypes.c_int32(ptr & 0xFFFFFFFF).value)
                lib.pivlfn_tune(6, ctypes.c_int32((ptr >> 32) & 0xFFFFFFFF).value)
                _lib.check(lib.pivlfn_conv2d_nhwc_wino(h, x.data_ptr(), xs, y.data_ptr(), co, a.batch, n, n, 1, st), "wino")
                torch.cuda.synchronize()
                lib.pivlfn_tune(5, 0)
                lib.pivlfn_tune(6, 0)
                lib.pivlfn_tune(1, 0)
                t = buf.view(-1, 8).cpu().numpy()
                t = t[t[:, 6] > 0]
                nch = -(-ci // 8)
                tot = t[:, 6].mean()
                print(f"L{L} {ci}->{co} {'one workgroup' if one else 'two workgroups'} per CU: {len(t)} workgroups stamped, {nch} chunks; ticks per workgroup (mean): " +
                      "  ".join(f"{nm} {t[:, i].mean():.0f} ({100 * t[:, i].mean() / tot:.0f} %)" for i, nm in enumerate(names)), flush=True)
                print("    per chunk: " + "  ".join(f"{nm} {t[:, i].mean() / nch:.0f}" for i, nm in list(enumerate(names))[1:5]), flush=True)
            lib.pivlfn_conv_destroy(h)


def ws_stamps(a):
    """Wave-specialised kernel (conv_wino_ws.hip): ticks of consumer wave 0 and producer wave 4 of every workgroup."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import _toolslib
    lib = _toolslib.load()
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream(dev).cuda_stream
    layers = [tuple(int(v) for v in s.split("x")) for s in a.layers.split(",")] if a.layers else [(128, 128)]
    for L in [int(x) for x in a.levels.split(",")]:
        n = a.size >> (L - 1)
        for ci, co in layers:
            g = torch.Generator().manual_seed(ci * 7 + co)
            w = (torch.randn(co, ci, 3, 3, generator=g) / (ci * 9) ** 0.5).contiguous()
            b = torch.randn(co, generator=g).contiguous()
            h = ctypes.c_void_p()
            _lib.check(lib.pivlfn_conv_create(w.data_ptr(), b.data_ptr(), co, ci, 3, 3, ctypes.byref(h)), "create")
            xs = -(-ci // 4) * 4
            x = torch.randn(a.batch, n, n, xs, device=dev)
            y = torch.empty(a.batch, n, n, co, device=dev)
            buf = torch.zeros(4096 * 16, dtype=torch.int64, device=dev)
            lib.pivlfn_tune(14, 31)
            for dbg in [int(v) for v in (a.masks or "0").split(",")]:
                lib.pivlfn_tune(15, dbg)
                for _ in range(3):
                    _lib.check(lib.pivlfn_conv2d_nhwc_wino(h, x.data_ptr(), xs, y.data_ptr(), co, a.batch, n, n, 1, st), "wino")
                buf.zero_()
                _toolslib.set_stamp_buffer(lib, buf.data_ptr())
                _lib.check(lib.pivlfn_conv2d_nhwc_wino(h, x.data_ptr(), xs, y.data_ptr(), co, a.batch, n, n, 1, st), "wino")
                torch.cuda.synchronize()
                _toolslib.set_stamp_buffer(lib, 0)
                t = buf.view(-1, 16).cpu().numpy().astype(float)
                t = t[t[:, 3] > 0]
                nch = -(-ci // 8)
                steps = t[:, 5] * nch
                print(f"L{L} {ci}->{co} dbg {dbg}: {len(t)} workgroups, {t[:, 5].mean():.1f} items x {nch} chunks; per step (ticks, mean over workgroups): "
                      f"consumer: whole {(t[:, 3] / steps).mean():.0f}  MFMA+refills {(t[:, 0] / steps).mean():.0f}  column half {(t[:, 1] / steps).mean():.0f}  barrier {(t[:, 2] / steps).mean():.0f} | "
                      f"producer: whole {(t[:, 11] / steps).mean():.0f}  load wait+commit {(t[:, 9] / steps).mean():.0f}  rest of its work {(t[:, 8] / steps).mean():.0f}  barrier {(t[:, 10] / steps).mean():.0f}", flush=True)
            lib.pivlfn_tune(15, 0)
            lib.pivlfn_tune(14, 0)
            lib.pivlfn_conv_destroy(h)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=1024)
    ap.add_argument("--levels", default="1,2")
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--n", type=int, default=10)
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--layers", default="")
    ap.add_argument("--ab", default="", help="path of a second build of libpivlfn.so: its Winograd kernel is timed beside the current one (interleaved rounds)")
    ap.add_argument("--b3", default="", help="also time the split-operand Winograd kernel (conv_wino_b3.hip) with these piece-product counts, e.g. 6 or 6,8,9 (instead of F(4x4))")
    ap.add_argument("--w4", action="store_true", help="--masks are ablation masks of the F(4x4) kernel (1 no MFMAs, 2 no transform, 4 no weight loads, 8 no patch loads, 16 no epilogue)")
    ap.add_argument("--masks", default="", help="tools build only: per-variant masks of the Winograd kernel: (m >> 8) & 255 = forced tile shape (11, 21, 22, 14; 30 = never / 31 = always the wave-specialised kernel), 65536 = one workgroup per CU, m >> 24 = ablation mask of the wave-specialised kernel")
    ap.add_argument("--stamps", action="store_true", help="tools build only: phase times of wave 0 of every workgroup (s_memtime ticks) for the two-block shape, at two and at one workgroup per CU")
    ap.add_argument("--ws-stamps", action="store_true", help="tools build only: barrier / load-wait ticks of the wave-specialised kernel; --masks = its ablation masks")
    a = ap.parse_args()
    lib = _lib.load()
    if a.ws_stamps:
        return ws_stamps(a)
    if a.stamps:
        return stamps(a)
    if a.masks:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import _toolslib
        lib = _toolslib.load()
    lib_b = None
    if a.ab:
        lib_b = ctypes.CDLL(a.ab)
        for name, (res, args) in _lib.SIGNATURES.items():
            if hasattr(lib_b, name):
                fn = getattr(lib_b, name)
                fn.restype, fn.argtypes = res, args
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream(dev).cuda_stream
    layers = [tuple(int(v) for v in s.split("x")) for s in a.layers.split(",")] if a.layers else LAYERS
    for L in [int(x) for x in a.levels.split(",")]:
        n = a.size >> (L - 1)
        for ci, co in layers:
            g = torch.Generator().manual_seed(ci * 7 + co)
            w = (torch.randn(co, ci, 3, 3, generator=g) / (ci * 9) ** 0.5).contiguous()
            b = torch.randn(co, generator=g).contiguous()
            h = ctypes.c_void_p()
            _lib.check(lib.pivlfn_conv_create(w.data_ptr(), b.data_ptr(), co, ci, 3, 3, ctypes.byref(h)), "create")
            xs = -(-ci // 4) * 4
            x = torch.randn(a.batch, n, n, xs, device=dev)
            y = {k: torch.empty(a.batch, n, n, co, device=dev) for k in ("direct", "wino")}

            def direct():
                _lib.check(lib.pivlfn_conv2d_nhwc(h, x.data_ptr(), xs, y["direct"].data_ptr(), co, None, 0, a.batch, n, n, 1, 1, 1, 1, st), "direct")

            def wino():
                _lib.check(lib.pivlfn_conv2d_nhwc_wino(h, x.data_ptr(), xs, y["wino"].data_ptr(), co, a.batch, n, n, 1, st), "wino")
            y["wino4"] = torch.empty(a.batch, n, n, co, device=dev)

            def wino4():
                _lib.check(lib.pivlfn_conv2d_nhwc_wino4(h, x.data_ptr(), xs, y["wino4"].data_ptr(), co, a.batch, n, n, 1, st), "wino4")
            fns = {"direct": direct, "wino": wino, "wino4": wino4}
            if a.b3:
                fns = {"direct": direct, "wino": wino}
                if co % 64 == 0:
                    for t in [int(v) for v in a.b3.split(",")]:
                        y[f"b3_{t}"] = torch.empty(a.batch, n, n, co, device=dev)

                        def b3(t=t):
                            _lib.check(lib.pivlfn_conv2d_nhwc_wino_b3(h, x.data_ptr(), xs, y[f"b3_{t}"].data_ptr(), co, a.batch, n, n, 1, t, st), "wino_b3")
                        fns[f"b3_{t}"] = b3
            if lib_b is not None:
                hb = ctypes.c_void_p()
                _lib.check(lib_b.pivlfn_conv_create(w.data_ptr(), b.data_ptr(), co, ci, 3, 3, ctypes.byref(hb)), "create b")
                y["prev"] = torch.empty(a.batch, n, n, co, device=dev)

                def prev():
                    assert lib_b.pivlfn_conv2d_nhwc_wino(hb, x.data_ptr(), xs, y["prev"].data_ptr(), co, a.batch, n, n, 1, st) == 0
                fns = {"wino": wino, "prev": prev}
            if a.masks:
                fns = {}
                for m in [int(v) for v in a.masks.split(",")]:
                    def wm(m=m):
                        if a.w4:
                            lib.pivlfn_tune(7, m)
                            wino4()
                            lib.pivlfn_tune(7, 0)
                            return
                        lib.pivlfn_tune(13, m & 255)
                        lib.pivlfn_tune(14, (m >> 8) & 255)
                        lib.pivlfn_tune(1, ((m >> 16) & 255) << 20)
                        lib.pivlfn_tune(15, (m >> 24) & 255)
                        wino()
                        lib.pivlfn_tune(15, 0)
                        lib.pivlfn_tune(13, 0)
                        lib.pivlfn_tune(14, 0)
                        lib.pivlfn_tune(1, 0)
                    fns[f"mask{m}"] = wm
            times = {k: [] for k in fns}
            for k in fns:
                fns[k]()
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
            flop = 2.0 * 9 * ci * co * n * n * a.batch
            if lib_b is not None:
                tw, tp = min(times["wino"]), min(times["prev"])
                mw, mp = sorted(times["wino"])[len(times["wino"]) // 2], sorted(times["prev"])[len(times["prev"]) // 2]
                print(f"L{L} {n}x{n} B={a.batch} {ci:3d}->{co:3d}: current min {tw:8.1f} med {mw:8.1f} us   previous build min {tp:8.1f} med {mp:8.1f} us   "
                      f"current/previous {tw / tp:5.3f} (min) {mw / mp:5.3f} (med)   bits equal: {bool(torch.equal(y['wino'], y['prev']))}", flush=True)
                lib_b.pivlfn_conv_destroy(hb)
                lib.pivlfn_conv_destroy(h)
                continue
            if a.masks:
                direct()
                diffs = {}
                first = None
                for k in fns:
                    y["wino"].fill_(float("nan"))
                    fns[k]()
                    diffs[k] = (y["wino"] - y["direct"]).abs().max().item() / y["direct"].abs().max().item()
                    if first is None:
                        first = y["wino"].clone()
                    elif not torch.equal(first, y["wino"]):
                        bad = (first != y["wino"]).nonzero()
                        print(f"    {k}: {len(bad)} values differ from the first variant's; first at (b, y, x, c) = {bad[0].tolist()}, last {bad[-1].tolist()}, "
                              f"max |diff| {(first - y['wino']).abs().max().item():.3e}", flush=True)
                print(f"L{L} {n}x{n} B={a.batch} {ci:3d}->{co:3d}: " + "   ".join(f"{k} min {min(v):8.1f} med {sorted(v)[len(v) // 2]:8.1f} us (diff {diffs[k]:.1e})" for k, v in times.items()), flush=True)
                lib.pivlfn_conv_destroy(h)
                continue
            d = (y["wino"] - y["direct"]).abs().max().item() / y["direct"].abs().max().item()
            td, tw = min(times["direct"]), min(times["wino"])
            md, mw = sorted(times["direct"])[len(times["direct"]) // 2], sorted(times["wino"])[len(times["wino"]) // 2]
            print(f"L{L} {n}x{n} B={a.batch} {ci:3d}->{co:3d}: direct min {td:8.1f} med {md:8.1f} us ({flop / td / 1e6:6.1f} TF)   "
                  f"wino min {tw:8.1f} med {mw:8.1f} us ({flop / tw / 1e6:6.1f} TF-equiv, {flop / 2.25 / tw / 1e6:6.1f} TF executed)   "
                  f"x{td / tw:4.2f}   rel diff {d:.1e}", flush=True)
            for k in [k for k in times if k.startswith("b3_")]:
                db = (y[k] - y["direct"]).abs().max().item() / y["direct"].abs().max().item()
                tb, mb_ = min(times[k]), sorted(times[k])[len(times[k]) // 2]
                t_ = int(k[3:])
                print(f"        split bf16 x {t_}: min {tb:8.1f} med {mb_:8.1f} us ({flop / tb / 1e6:6.1f} TF-equiv, {flop / 2.25 * t_ / tb / 1e9:6.3f} PF bf16 executed)   "
                      f"x{td / tb:4.2f} vs direct, x{tw / tb:4.2f} vs fp32 Winograd   rel diff {db:.1e}", flush=True)
            if "wino4" in times:
                d4 = (y["wino4"] - y["direct"]).abs().max().item() / y["direct"].abs().max().item()
                t4, m4 = min(times["wino4"]), sorted(times["wino4"])[len(times["wino4"]) // 2]
                print(f"        F(4x4): min {t4:8.1f} med {m4:8.1f} us ({flop / t4 / 1e6:6.1f} TF-equiv, {flop / 4.0 / t4 / 1e6:6.1f} TF executed)   "
                      f"x{td / t4:4.2f} vs direct, x{tw / t4:4.2f} vs F(2x2)   rel diff {d4:.1e}", flush=True)
            lib.pivlfn_conv_destroy(h)


if __name__ == "__main__":
    main()
