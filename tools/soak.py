#!/usr/bin/env python3
"""Soak: many forwards over changing shapes, batches, models and precisions; every configuration must reproduce its first
result bit for bit (determinism, no state leaking between calls / workspaces / the side stream), and device memory must not grow.
  python tools/soak.py [rounds]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "piv_liteflownet-pytorch_amd"))
import torch

import pivlfn
from pivlfn import synth

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 12
dev = torch.device("cuda:0")
nets = {(m, v): pivlfn.Network(model=m, params=synth.generate_weights(m + ("2" if v == 2 else ""), 0), version=v).to(dev).eval()
        for m, v in [("piv", 1), ("hui", 1), ("piv", 2)]}
cfgs = [(("piv", 1), 1, 256, 256, "fp32_split3"), (("piv", 1), 3, 96, 160, "fp32"), (("hui", 1), 2, 128, 192, "fp32_split3"),
        (("piv", 2), 1, 192, 128, "fp32_split"), (("piv", 1), 1, 512, 384, "fp16"), (("piv", 1), 2, 64, 64, "fp32_split3"),
        (("hui", 1), 1, 320, 256, "fp16"), (("piv", 1), 5, 128, 128, "fp32_split3"), (("piv", 1), 1, 1024, 1024, "fp32_split3"),
        (("piv", 1), 2, 1024, 512, "fp32_split"), (("hui", 1), 1, 544, 800, "fp32_split3"), (("piv", 2), 2, 512, 512, "fp32_split3"),
        # the default arithmetic at sizes that take round 4's kernels (conv1 with the level-1 1 x 1 layers inside, the whole-line
        # stride-2 kernel, the streaming distance convolutions, warp+correlation v6 / v7) and at sizes just below their bounds
        (("piv", 1), 1, 1024, 1024, "fp32"), (("piv", 1), 2, 512, 768, "fp32"), (("hui", 1), 1, 1024, 512, "fp32"),
        (("piv", 1), 1, 480, 544, "fp32"), (("piv", 1), 3, 256, 256, "fp32"), (("piv", 1), 1, 1024, 1024, "fp32_direct"),
        # round 6: the default now runs the split-operand Winograd kernel on the large layers; the fp32-instruction mode beside it
        (("piv", 1), 1, 1024, 1024, "fp32_wino_mfma32"), (("piv", 1), 2, 544, 800, "fp32"), (("hui", 1), 2, 256, 288, "fp32_wino_mfma32")]
inputs, first = {}, {}
for i, (key, B, H, W, prec) in enumerate(cfgs):
    a, b = synth.particle_batch(B, H, W, seed=70 + i)
    inputs[i] = (torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev))
torch.cuda.synchronize()
t0 = time.time()
n = 0
mem0 = None
for r in range(rounds):
    order = list(range(len(cfgs)))
    if r % 2:
        order.reverse()
    for i in order:
        key, B, H, W, prec = cfgs[i]
        net = nets[key]
        net.precision = prec
        out = net(*inputs[i])
        n += 1
        assert torch.isfinite(out).all(), (r, cfgs[i])
        if i not in first:
            first[i] = out.clone()
        else:
            assert torch.equal(out, first[i]), f"round {r}: configuration {cfgs[i]} changed its result"
    del out
    torch.cuda.synchronize()
    mem = torch.cuda.memory_allocated(dev)
    if r == 1:
        mem0 = mem
    if mem0 is not None:
        assert mem <= mem0, f"device memory grew: {mem0} -> {mem}"
    print(f"round {r}: {n} forwards, {mem / 2**20:.0f} MiB allocated, {time.time() - t0:.1f} s", flush=True)
print("soak OK")
