import sys, time
sys.path.insert(0, "piv_liteflownet-pytorch_amd")
import numpy as np, torch
import pivlfn
from pivlfn import synth
dev = torch.device("cuda:0")
net = pivlfn.Network(model="piv", params=synth.generate_weights("piv", 0)).to(dev).eval()
for (B, S) in [(32, 512), (8, 1024)]:
    a, b = synth.particle_batch(2, S, S, seed=5)
    i1 = torch.from_numpy(a).to(dev); i2 = torch.from_numpy(b).to(dev)
    ref = net(i1, i2)
    big1 = torch.empty(B, 3, S, S, device=dev); big2 = torch.empty_like(big1)
    for k in range(B):
        big1[k] = i1[k % 2]; big2[k] = i2[k % 2]
    torch.cuda.synchronize(); t0 = time.time()
    out = net(big1, big2)
    torch.cuda.synchronize(); dt = time.time() - t0
    ok = all(torch.equal(out[k], ref[k % 2]) for k in range(B))
    print(f"B={B} {S}x{S}: {dt*1e3:.1f} ms first call ({B/dt:.1f} pairs/s incl. workspace alloc), batch-consistent bit-for-bit: {ok}, max|flow| {float(out.abs().max()):.2f}", flush=True)
    t0 = time.time()
    for _ in range(3): out = net(big1, big2)
    torch.cuda.synchronize(); dt = (time.time() - t0) / 3
    print(f"   steady (default arithmetic, {net.precision}): {dt*1e3:.1f} ms/batch = {B/dt:.1f} pairs/s", flush=True)
    net.precision = "fp32_split3"
    o32 = net(big1, big2); torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(3): o32 = net(big1, big2)
    torch.cuda.synchronize(); dt = (time.time() - t0) / 3
    print(f"   fp32_split3 (opt-in, 22-23 operand bits): {dt*1e3:.1f} ms/batch = {B/dt:.1f} pairs/s, max |diff| vs default {float((o32 - out).abs().max()):.2e}", flush=True)
    del o32
    net.precision = "fp16"
    o16 = net(big1, big2); torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(3): o16 = net(big1, big2)
    torch.cuda.synchronize(); dt = (time.time() - t0) / 3
    e = (o16 - out).pow(2).sum(1).sqrt()
    print(f"   fp16 mode: {dt*1e3:.1f} ms/batch = {B/dt:.1f} pairs/s, EPE vs fp32 mean {float(e.mean()):.2e} max {float(e.max()):.2e}", flush=True)
    net.precision = "fp32"
    del big1, big2, out, o16
    torch.cuda.empty_cache()
