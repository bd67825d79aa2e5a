#!/usr/bin/env python3
"""Where a frame of pivlfn.synth.ParticleSequence goes (host vs device): python tools/render_time.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "piv_liteflownet-pytorch_amd"))
from pivlfn import synth  # noqa: E402

dev = torch.device("cuda:0")
seq = synth.ParticleSequence(1024, 1024, seed=7, device=dev)
seq.frames(0, 2)
torch.cuda.synchronize()
t0 = time.perf_counter()
seq.frames(2, 34)
t_issue = time.perf_counter() - t0
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"renderer alone: {dt / 32 * 1e3:.2f} ms per frame wall, {t_issue / 32 * 1e3:.2f} ms per frame until the host has issued everything")
t0 = time.perf_counter()
for k in range(34, 66):
    seq._positions(k)
print(f"host advection (numpy float64): {(time.perf_counter() - t0) / 32 * 1e3:.2f} ms per frame")
import cProfile
import pstats
pr = cProfile.Profile()
pr.enable()
seq.frames(66, 82)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
