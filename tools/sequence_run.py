#!/usr/bin/env python3
"""BASELINE config #4 in miniature: a synthetic PIV frame SEQUENCE generated on the device, pairs (k, k+1) sharded over the
ranks (contiguous shards, one halo frame each), flows estimated chunk by chunk, reassembled with an asynchronous all-gather
(RCCL under torch.distributed.run; a no-op on one rank), and optionally written as .flo files by rank 0's background writer.

  python tools/sequence_run.py --frames 65 --size 1024 [--chunk 8] [--write DIR] [--precision fp16]
  python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 tools/sequence_run.py --frames 10000 ...

Prints one JSON line on rank 0: pairs/s of the estimation alone and of the whole loop (generation + estimation + gather +
device-to-host + .flo writing).  10 000 frames x 1024^2 are 84 GB of .flo: writing is what N1's asynchronous writer is for.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "piv_liteflownet-pytorch_amd"))
import torch

import pivlfn
from pivlfn import synth
from pivlfn.dist import gather_flows, shard_bounds
from pivlfn.flo import FloWriter


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=65)
    ap.add_argument("--size", type=int, default=1024)
    ap.add_argument("--chunk", type=int, default=8, help="pairs per forward / per gather on every rank")
    ap.add_argument("--write", default=None, help="directory for rank 0's .flo files")
    ap.add_argument("--precision", default="fp32", choices=["fp32", "fp16"])
    ap.add_argument("--model", default="piv")
    a = ap.parse_args()
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    backend = os.environ.get("PIVLFN_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local %= torch.cuda.device_count()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group(backend, **({"device_id": dev} if backend == "nccl" else {}))
    net = pivlfn.Network(model=a.model, params=synth.generate_weights(a.model, 0)).to(dev).eval()
    net.precision = a.precision
    n_pairs = a.frames - 1
    lo, hi = shard_bounds(n_pairs, rank, world)
    per = -(-n_pairs // world)
    seq = synth.ParticleSequence(a.size, a.size, seed=99, device=dev)
    writer = FloWriter() if (a.write and rank == 0) else None
    if writer:
        os.makedirs(a.write, exist_ok=True)
    S = a.size
    t_est = 0.0
    pending = None          # (first global pair of the chunk position, finish(), valid rows per rank)
    written = 0

    def drain(item):
        nonlocal written
        c0, finish, rows = item
        full = finish() if finish is not None else None
        if writer is None or full is None:
            return
        host = full.permute(0, 2, 3, 1).contiguous().cpu().numpy()          # [world*chunk, H, W, 2], rank-major
        for r in range(world):
            for j in range(rows[r]):
                gi = r * per + c0 + j
                writer.submit(host[r * a.chunk + j], os.path.join(a.write, f"frame_{gi:06d}_out.flo"))
                written += 1

    torch.cuda.synchronize(dev)
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    nchunks = -(-per // a.chunk)
    prev = None             # the halo: last frame of the previous chunk
    for c in range(nchunks):
        i0 = lo + c * a.chunk
        i1 = min(hi, i0 + a.chunk)
        n = max(0, i1 - i0)
        flows = torch.zeros(a.chunk, 2, S, S, device=dev)
        if n > 0:
            fr = seq.frames(i0 if prev is None else i0 + 1, i1 + 1)          # every frame is rendered once
            if prev is not None:
                fr = torch.cat([prev, fr])
            prev = fr[-1:].clone()
            x = fr.to(torch.float32).div_(255.0)[:, None].expand(-1, 3, -1, -1).contiguous()
            torch.cuda.synchronize(dev)
            t1 = time.perf_counter()
            flows[:n] = pivlfn.estimate(net, x[:-1], x[1:], tensor=True)
            torch.cuda.synchronize(dev)
            t_est += time.perf_counter() - t1
        rows = [max(0, min(min(n_pairs, (r + 1) * per), r * per + c * a.chunk + a.chunk) - (r * per + c * a.chunk)) for r in range(world)]
        if world > 1:
            _, finish = gather_flows(flows, world * a.chunk, async_op=True)   # overlaps the next chunk's rendering + estimation
        else:
            finish = (lambda f=flows: f)
        if pending is not None:
            drain(pending)
        pending = (c * a.chunk, finish, rows)
    if pending is not None:
        drain(pending)
    if writer:
        writer.close()
    torch.cuda.synchronize(dev)
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    if rank == 0:
        print(json.dumps({"workload": f"{a.frames} frames {S}x{S} ({n_pairs} pairs), {world} rank(s), chunk {a.chunk}, {a.precision}",
                          "pairs_per_s_estimation_only_rank0": round((hi - lo) / t_est, 2) if t_est else None,
                          "pairs_per_s_whole_loop": round(n_pairs / dt, 2), "seconds": round(dt, 2),
                          "flo_files_written": written,
                          "flo_gb_written": round(written * (12 + S * S * 8) / 1e9, 2)}), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
