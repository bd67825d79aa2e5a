#!/usr/bin/env python3
"""Accuracy and speed of the convolution arithmetics side by side (tools build):
  fp32   v_mfma_f32_32x32x2_f32 (conv_mfma.hip)
  x6     exact three-piece fp16 split, six partial products (conv_split.hip, precision 'fp32_split')
  x3     two pieces per operand, the three leading partial products (precision 'fp32_split3')
Per layer: max / mean error against a float64 convolution of the same fp32 data; end to end: error of a PIV forward against the
CPU oracle and ms per forward.   python tools/split_accuracy.py [--size 512]"""
import argparse
import ctypes
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import _toolslib  # noqa: E402
from pivlfn import _lib  # noqa: E402

LAYERS = [(128, 128, 3, 1, 256, 256), (64, 128, 3, 1, 256, 256), (64, 64, 3, 1, 256, 256), (32, 64, 3, 1, 256, 256), (128, 52, 3, 1, 256, 256)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--no-oracle", action="store_true")
    a = ap.parse_args()
    lib = _toolslib.load()
    _lib._lib = lib
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream(dev).cuda_stream
    for co, ci, k, B, H, W in LAYERS:
        g = torch.Generator().manual_seed(co + ci)
        w = (torch.randn(co, ci, k, k, generator=g) / (ci * k * k) ** 0.5).contiguous()
        b = torch.randn(co, generator=g).contiguous()
        x = torch.randn(B, H, W, ci, generator=g)
        want = F.leaky_relu(F.conv2d(x.permute(0, 3, 1, 2).double(), w.double(), b.double(), padding=k // 2), 0.1).permute(0, 2, 3, 1)
        scale = want.abs().max().item()
        h = ctypes.c_void_p()
        _toolslib.check(lib, lib.pivlfn_conv_create(w.data_ptr(), b.data_ptr(), co, ci, k, k, ctypes.byref(h)), "create")
        xd = x.to(dev)
        line = f"{ci:3d}->{co:3d} {k}x{k} {H}x{W}:"
        for tag, knob in (("fp32", None), ("x6", 6), ("x3", 3)):
            y = torch.empty(B, H, W, co, device=dev)
            if knob is None:
                lib.pivlfn_tune(1, 0)
                _toolslib.check(lib, lib.pivlfn_conv2d_nhwc(h, xd.data_ptr(), ci, y.data_ptr(), co, None, 0, B, H, W, 1, k // 2, k // 2, 1, st), tag)
            else:
                _toolslib.check(lib, lib.pivlfn_conv2d_nhwc_split(h, xd.data_ptr(), ci, y.data_ptr(), co, B, H, W, 1, k // 2, k // 2, 1, knob, st), tag)
            torch.cuda.synchronize()
            e = (y.cpu().double() - want).abs()
            line += f"  {tag} max {e.max().item() / scale:.2e} mean {e.mean().item() / scale:.2e}"
        lib.pivlfn_tune(1, 0)
        lib.pivlfn_conv_destroy(h)
        print(line, flush=True)

    import pivlfn
    from pivlfn import synth
    S = a.size
    p1, p2, _ = synth.particle_pair(S, S, 4321)
    x1, x2 = torch.from_numpy(synth.to_input(p1))[None], torch.from_numpy(synth.to_input(p2))[None]
    want = None
    if not a.no_oracle:
        import pivlfn_oracle as orc
        onet = orc.make_net("piv", synth.generate_weights("piv", 0), corr="c")
        with torch.no_grad():
            want = onet.forward(x1, x2).numpy().astype(np.float64)
    net = pivlfn.Network(model="piv", params=synth.generate_weights("piv", 0)).to(dev).eval()
    i1, i2 = x1.to(dev), x2.to(dev)
    base = None
    for tag, prec, knob in (("fp32", "fp32", 0), ("x6", "fp32_split", 0), ("x3", "fp32_split3", 0)):
        net.precision = prec
        lib.pivlfn_tune(1, knob)
        for _ in range(3):
            out = net(i1, i2)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            out = net(i1, i2)
        e1.record()
        torch.cuda.synchronize()
        got = out.cpu().numpy().astype(np.float64)
        if base is None:
            base = got
        msg = f"PIV {S}x{S} {tag:4s}: {e0.elapsed_time(e1) / 10:7.3f} ms / forward; vs fp32 instruction: max {np.abs(got - base).max():.2e}"
        if want is not None:
            err = np.abs(got - want)
            msg += f"; vs oracle: max {err.max():.2e} mean {err.mean():.2e} px (max|flow| {np.abs(want).max():.2f})"
        print(msg, flush=True)
    lib.pivlfn_tune(1, 0)


if __name__ == "__main__":
    main()
