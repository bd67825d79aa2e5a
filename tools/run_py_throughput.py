#!/usr/bin/env python3
"""End-to-end rate of run.py on a folder of PNG frames (decode -> H2D -> estimate -> D2H -> .flo): N2/N1 of SURVEY section 8(f).
  python tools/run_py_throughput.py [frames] [size]"""
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "piv_liteflownet-pytorch_amd"))
import PIL.Image
import torch

import run as runpy
from pivlfn import synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 33
S = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
d = tempfile.mkdtemp(prefix="seq_")
fr = synth.ParticleSequence(S, S, seed=3, device="cuda:0").frames(0, n).cpu().numpy()
for k in range(n):
    PIL.Image.fromarray(fr[k]).save(os.path.join(d, f"frame_{k:05d}.png"))
out = tempfile.mkdtemp(prefix="flo_")
dev = torch.device("cuda:0")
net = runpy.Network(model="piv", params=synth.generate_weights("piv", 0)).to(dev).eval()
for precision in ("fp32", "fp16"):
    net.precision = precision
    for batch in (1, 4):
        runpy.main_dl(net, d, out, False, 0, 5, dev, batch)                     # warm-up (workspace, caches)
        t0 = time.perf_counter()
        pairs = runpy.main_dl(net, d, out, False, 0, -1, dev, batch)            # run.py's per-directory loop (run.py:137-168)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"run.py main_dl {S}x{S} PNG sequence, {pairs} pairs, --batch {batch}, {precision}: {dt:.2f} s = {pairs / dt:.1f} pairs/s "
              f"end to end (PNG decode -> H2D -> estimate -> D2H -> .flo files closed)", flush=True)
