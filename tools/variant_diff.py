#!/usr/bin/env python3
"""Which level first differs between two warp+correlation kernel variants (tools build) on a real pair?
  python tools/variant_diff.py --pair 6 --variants 5,6"""
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
    ap.add_argument("--pair", type=int, default=6)
    ap.add_argument("--size", type=int, default=1024)
    ap.add_argument("--seed", type=int, default=99)
    ap.add_argument("--variants", default="5,6")
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--chunk", type=int, default=0, help="compare a batch of CHUNK consecutive pairs starting at --pair with the pairs alone")
    a = ap.parse_args()
    lib = _toolslib.load()
    _lib._lib = lib
    import pivlfn
    from pivlfn import synth
    dev = torch.device("cuda:0")
    fr = synth.ParticleSequence(a.size, a.size, seed=a.seed, device=dev).frames(a.pair, a.pair + 2)
    x = fr.to(torch.float32).div_(255.0)[:, None].expand(-1, 3, -1, -1).contiguous()
    i1, i2 = x[0:1].repeat(a.batch, 1, 1, 1), x[1:2].repeat(a.batch, 1, 1, 1)
    net = pivlfn.piv_liteflownet(synth.generate_weights("piv", 0)).to(dev).eval()
    if a.chunk:
        fr = synth.ParticleSequence(a.size, a.size, seed=a.seed, device=dev).frames(a.pair, a.pair + a.chunk + 1)
        x = fr.to(torch.float32).div_(255.0)[:, None].expand(-1, 3, -1, -1).contiguous()
        fb, lb = net.forward_levels(x[:-1], x[1:])
        for k in range(a.chunk):
            f1, l1 = net.forward_levels(x[k:k + 1], x[k + 1:k + 2])
            print(f"pair {a.pair + k} in the batch vs alone: final flow max abs diff {(fb[k:k + 1] - f1).abs().max().item():.3e}")
            for j, (ta, tb) in enumerate(zip(lb, l1)):
                for name, p, q in zip("MSR", ta, tb):
                    d = (p[k:k + 1] - q).abs()
                    n = int((d > 0).sum().item())
                    if n:
                        idx = torch.nonzero(d[0].amax(0) > 0)[:5].tolist()
                        print(f"   level {6 - j} {name}: {n} differing values, max {d.max().item():.3e}, first at (y,x) {idx}")
        return
    res = {}
    for v in [int(t) for t in a.variants.split(",")]:
        lib.pivlfn_tune(0, v)
        flow, levels = net.forward_levels(i1, i2)
        res[v] = (flow.clone(), [[t.clone() for t in trio] for trio in levels])
    lib.pivlfn_tune(0, 0)
    if a.batch > 1:        # the same pair alone
        flow, levels = net.forward_levels(i1[:1], i2[:1])
        fb, lb = res[list(res)[0]]
        print(f"batch {a.batch} vs batch 1 (shipped policy for batch 1, variant {list(res)[0]} for the batch): final flow max abs diff {(fb[:1] - flow).abs().max().item():.3e}")
        for j, (ta, tb) in enumerate(zip(lb, levels)):
            for name, p, q in zip("MSR", ta, tb):
                d = (p[:1] - q).abs()
                n = int((d > 0).sum().item())
                if n:
                    idx = torch.nonzero(d[0].amax(0) > 0)[:5].tolist()
                    print(f"   level {6 - j} {name}: {n} differing values, max {d.max().item():.3e}, first at (y,x) {idx}")
    vs = list(res)
    f0, l0 = res[vs[0]]
    for v in vs[1:]:
        f1, l1 = res[v]
        print(f"variant {v} vs {vs[0]}: final flow max abs diff {(f1 - f0).abs().max().item():.3e}")
        for j, (ta, tb) in enumerate(zip(l0, l1)):
            for name, p, q in zip("MSR", ta, tb):
                d = (p - q).abs()
                n = int((d > 0).sum().item())
                if n:
                    idx = torch.nonzero(d[0].amax(0) > 0)[:5].tolist()
                    print(f"   level {6 - j} {name}: {n} differing values, max {d.max().item():.3e}, first at (y,x) {idx}")


if __name__ == "__main__":
    main()
