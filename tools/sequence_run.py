#!/usr/bin/env python3
"""BASELINE config #4 in miniature, from the command line: a synthetic PIV frame sequence rendered on the device, sharded over
the ranks and estimated by pivlfn.sequence.run_sequence (contiguous shards + one halo frame, asynchronous all-gather of the
flows, .flo files from rank 0's background writer).

  python tools/sequence_run.py --frames 65 --size 1024 [--chunk 8] [--write DIR] [--precision fp16]
  python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 tools/sequence_run.py --frames 10000 ...

Prints one JSON line on rank 0: pairs/s of the estimation alone and of the whole loop (rendering + estimation + gather +
device-to-host + .flo writing).  10 000 frames x 1024^2 are 84 GB of .flo.
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "piv_liteflownet-pytorch_amd"))
import torch

import pivlfn
from pivlfn import synth
from pivlfn.sequence import run_sequence


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=65)
    ap.add_argument("--size", type=int, default=1024)
    ap.add_argument("--chunk", type=int, default=8, help="pairs per forward / per gather on every rank")
    ap.add_argument("--write", default=None, help="directory for rank 0's .flo files")
    ap.add_argument("--null-sink", action="store_true", help="hand every flow to a sink that drops it (the loop without the disk)")
    ap.add_argument("--no-gather", action="store_true", help="no all-gather: every rank hands its own shard to its own sink / writer")
    ap.add_argument("--precision", default="fp32", choices=["fp32", "fp32_wino_mfma32", "fp32_direct", "fp32_split", "fp32_split3", "fp16"])
    ap.add_argument("--model", default="piv")
    ap.add_argument("--seed", type=int, default=99)
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
    seq = synth.ParticleSequence(a.size, a.size, seed=a.seed, device=dev)
    if dist is not None:
        dist.barrier()
    import resource
    rss0 = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
    seen = [0]

    def null_sink(gi, flow):
        seen[0] += 1
    st = run_sequence(net, seq.frames, a.frames, a.chunk, dev, write_dir=a.write, sink=null_sink if a.null_sink else None, rank=rank, world=world,
                      gather=not a.no_gather)
    rss1 = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
    if dist is not None:
        dist.barrier()
    if rank == 0:
        S = a.size
        print(json.dumps({"workload": f"{a.frames} frames {S}x{S} ({st['pairs_total']} pairs), {world} rank(s), chunk {a.chunk}, {a.precision}",
                          "pairs_per_s_estimation_only_rank0": round(st["pairs_this_rank"] / st["seconds_estimation"], 2) if st["seconds_estimation"] else None,
                          "pairs_per_s_whole_loop": round(st["pairs_total"] / st["seconds"], 2), "seconds": round(st["seconds"], 2),
                          "flows_handed_to_the_sink": st["flows_emitted"], "max_rss_mb_before_after": [round(rss0 / 1024), round(rss1 / 1024)],
                          "flo_files_written": st["flows_emitted"] if a.write else 0,
                          "flo_gb_written": round(st["flows_emitted"] * (12 + S * S * 8) / 1e9, 2) if a.write else 0.0}), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
