#!/usr/bin/env python3
"""ms per PIV forward for each precision mode and each minimum-output-size threshold of the split kernels (tools build,
pivlfn_tune(11, pixels)), interleaved rounds.   python tools/split_threshold.py --sizes 1024,512,256"""
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
    ap.add_argument("--sizes", default="1024,512,256")
    ap.add_argument("--thresholds", default="4096,16384,65536,262144")
    ap.add_argument("--modes", default="fp32,fp32_split,fp32_split3")
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    a = ap.parse_args()
    lib = _toolslib.load()
    _lib._lib = lib
    import pivlfn
    from pivlfn import synth
    dev = torch.device("cuda:0")
    for S in [int(x) for x in a.sizes.split(",")]:
        x, y = synth.particle_batch(a.batch, S, S, seed=1234)
        i1, i2 = torch.from_numpy(x).to(dev), torch.from_numpy(y).to(dev)
        net = pivlfn.Network(model="piv", params=synth.generate_weights("piv", 0)).to(dev).eval()
        for rnd in range(2):
            for mode in a.modes.split(","):
                net.precision = mode
                for th in ([0] if mode == "fp32" else [int(t) for t in a.thresholds.split(",")]):
                    lib.pivlfn_tune(11, th)
                    for _ in range(3):
                        net(i1, i2)
                    torch.cuda.synchronize()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(a.steps):
                        net(i1, i2)
                    e1.record()
                    torch.cuda.synchronize()
                    print(f"{S}x{S} B={a.batch} round {rnd} {mode:12s} min output px {th:7d}: {e0.elapsed_time(e1) / a.steps:8.3f} ms / forward", flush=True)
        lib.pivlfn_tune(11, 0)


if __name__ == "__main__":
    main()
