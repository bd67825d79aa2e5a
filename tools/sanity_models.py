#!/usr/bin/env python3
"""GPU sanity sweep: every model family at a mid-size image against the CPU oracle (fp32 mode) and fp16-mode EPE.
  python tools/sanity_models.py [size]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "piv_liteflownet-pytorch_amd"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import torch

import pivlfn
import pivlfn_oracle as orc
from pivlfn import synth

S = int(sys.argv[1]) if len(sys.argv) > 1 else 384
dev = torch.device("cuda:0")
torch.set_num_threads(16)
for model, version in [("piv", 1), ("hui", 1), ("piv", 2), ("hui", 2)]:
    tag = model + ("2" if version == 2 else "")
    wts = synth.generate_weights(tag, 0)
    a, b, _ = synth.particle_pair(S, S + 64, 11)
    i1 = torch.from_numpy(synth.to_input(a))[None]
    i2 = torch.from_numpy(synth.to_input(b))[None]
    net = pivlfn.Network(model=model, params=wts, version=version).to(dev).eval()
    got = net(i1.to(dev), i2.to(dev)).cpu()
    t0 = time.time()
    with torch.no_grad():
        want = orc.make_net(tag, wts, corr="c").forward(i1.clone(), i2.clone())
    t_cpu = time.time() - t0
    err = float((got - want).abs().max())
    scale = max(1.0, float(want.abs().max()))
    net.precision = "fp16"
    g16 = net(i1.to(dev), i2.to(dev)).cpu()
    epe = (g16 - got).pow(2).sum(1).sqrt()
    ok = err <= 1e-4 * scale and float(epe.mean()) <= 0.05
    print(f"{tag:5s} {S}x{S + 64}: out {tuple(got.shape)} max|flow| {float(want.abs().max()):6.2f}  fp32 vs oracle max-abs {err:.2e} "
          f"(tol {1e-4 * scale:.1e})  fp16-mode EPE mean {float(epe.mean()):.2e} max {float(epe.max()):.2e}  oracle {t_cpu:.1f} s  {'OK' if ok else 'FAIL'}",
          flush=True)
    assert ok
print("all model families OK")
