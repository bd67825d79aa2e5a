#!/usr/bin/env python3
"""A/B of whole-forward variants in one process through the tools build: ms per 1024x1024 PIV forward for a list of
pivlfn_tune(1, .) masks (0 = shipped; 2048 = no side stream; 16 = no 16-row conv tiles; ...), interleaved rounds.
  python tools/net_ab.py --masks 0,2048 [--size 1024] [--batch 1]"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import _toolslib  # noqa: E402
from pivlfn import _lib  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--masks", default="0,2048")
    ap.add_argument("--size", type=int, default=1024)
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--knob", type=int, default=1, help="which pivlfn_tune knob the masks go to (1 = variant bits, 2 = warp+correlation debug mask)")
    ap.add_argument("--profile-level", type=int, default=0, help="also time this level's warp+correlation launch with dispatch events (pivlfn_profile_enable)")
    a = ap.parse_args()
    lib = _toolslib.load()
    _lib._lib = lib
    import pivlfn
    from pivlfn import synth
    dev = torch.device("cuda:0")
    x, y = synth.particle_batch(a.batch, a.size, a.size, seed=1234)
    i1, i2 = torch.from_numpy(x).to(dev), torch.from_numpy(y).to(dev)
    net = pivlfn.Network(model="piv", params=synth.generate_weights("piv", 0)).to(dev).eval()
    masks = [int(m) for m in a.masks.split(",")]
    ref = None
    for rnd in range(a.rounds):
        for m in masks:
            lib.pivlfn_tune(a.knob, m)
            for _ in range(3):
                out = net(i1, i2)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            if a.profile_level:
                net.profile_enable(a.profile_level)
            e0.record()
            for _ in range(a.steps):
                out = net(i1, i2)
            e1.record()
            torch.cuda.synchronize()
            extra = ""
            if a.profile_level:
                k_ms, _, k_n = net.profile_read()
                net.profile_enable(0)
                extra = f"   level-{a.profile_level} warp+correlation {1e3 * k_ms / max(1, k_n):6.2f} us (n={k_n})"
            if ref is None:
                ref = out.clone()
            print(f"round {rnd} mask {m:5d}: {e0.elapsed_time(e1) / a.steps:8.3f} ms / forward   max|diff vs first| {(out - ref).abs().max().item():.2e}" + extra, flush=True)
    lib.pivlfn_tune(a.knob, 0)


if __name__ == "__main__":
    main()
