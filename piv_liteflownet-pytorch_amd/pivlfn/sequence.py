"""A frame SEQUENCE through the path, sharded over ranks: BASELINE config #4 (10 000 frames of 1024x1024 over 8 GPUs).

Pair i = (frame i, frame i+1) (the reference's sequence mode, src/datasets.py:456-463).  Rank r of R owns the contiguous
pairs `shard_bounds(n_pairs, r, R)`, renders / reads only the frames of its shard plus one halo frame, estimates them chunk
by chunk, and the flows of every chunk position are reassembled with one all-gather (RCCL on GPUs, issued asynchronously so
it overlaps the next chunk; no other collective touches the data path).  Rank 0 hands the gathered flows to a sink -- by
default a background `.flo` writer naming the files `frame_<pair index, 6 digits>_out.flo`.

    stats = run_sequence(net, frames_fn, n_frames, chunk=8, device=dev, write_dir="out/flow")

`frames_fn(f0, f1)` returns frames f0..f1-1 as a uint8 [n,H,W] (grey) or [n,H,W,3] (RGB) tensor on `device`; every frame
is asked for exactly once per rank, in increasing order (`pivlfn.synth.ParticleSequence.frames` is such a function).
"""
from __future__ import annotations

import os
import time
from typing import Callable, Dict, List, Optional

import torch

from .dist import gather_flows, shard_bounds
from .flo import FloWriter
from .inference import estimate
from .pipeline import u8_to_input


def frames_to_input(frames: torch.Tensor) -> torch.Tensor:
    """uint8 [n,H,W] or [n,H,W,3] -> float32 [n,3,H,W] in [0,1] (grey frames replicated to three channels), the same bits as
    ToTensor / run.py's input path: the table of correctly divided k/255 of `pipeline.u8_to_input`, not the GPU's `t * (1/255)`."""
    if frames.dim() == 3:
        frames = frames[..., None].expand(-1, -1, -1, 3)
    return u8_to_input(frames.contiguous())


def flow_file_name(pair_index: int) -> str:
    return f"frame_{pair_index:06d}_out.flo"


def run_sequence(net, frames_fn: Callable[[int, int], torch.Tensor], n_frames: int, chunk: int, device: torch.device,
                 write_dir: Optional[str] = None, sink: Optional[Callable[[int, "object"], None]] = None,
                 rank: int = 0, world: int = 1, estimate_fn: Callable = estimate) -> Dict[str, float]:
    """Estimate all n_frames-1 pairs.  With world > 1 a process group must be initialised (`nccl` on GPUs, `gloo` for
    rehearsals).  On rank 0, `sink(pair_index, flow_hw2_numpy)` -- or the `.flo` writer when `write_dir` is given -- sees
    every pair exactly once.  Returns timing / count statistics of this rank."""
    if n_frames < 2 or chunk < 1:
        raise ValueError("run_sequence: need at least two frames and chunk >= 1")
    n_pairs = n_frames - 1
    lo, hi = shard_bounds(n_pairs, rank, world)
    per = -(-n_pairs // world)
    nchunks = -(-per // chunk)                     # identical on every rank: the gather is collective
    on_gpu = device.type == "cuda"
    writer = None
    if rank == 0 and write_dir is not None:
        os.makedirs(write_dir, exist_ok=True)
        writer = FloWriter()
        sink = lambda gi, flow: writer.submit(flow, os.path.join(write_dir, flow_file_name(gi)))      # noqa: E731
    emitted = 0

    def drain(item) -> None:
        nonlocal emitted
        c0, finish, rows = item
        full = finish()
        if rank != 0 or sink is None:
            return
        host = full.permute(0, 2, 3, 1).contiguous().cpu().numpy()          # [world*chunk, H, W, 2], rank-major
        for r in range(world):
            for j in range(rows[r]):
                sink(r * per + c0 + j, host[r * chunk + j])
                emitted += 1

    def sync() -> None:
        # the compute stream only: a device-wide synchronize would also wait for the pending asynchronous all-gather (its own
        # stream) and serialise it with the next chunk, which is exactly what the gather is asynchronous to avoid
        if on_gpu:
            torch.cuda.current_stream(device).synchronize()

    t_est = 0.0
    sync()
    t0 = time.perf_counter()
    pending = None
    halo = None                                    # last frame of the previous chunk: shared by two consecutive pairs
    shape = None
    for c in range(nchunks):
        i0 = lo + c * chunk
        i1 = min(hi, i0 + chunk)
        n = max(0, i1 - i0)
        flows = None
        if n > 0:
            fr = frames_fn(i0 if halo is None else i0 + 1, i1 + 1)
            if halo is not None:
                fr = torch.cat([halo, fr])
            halo = fr[-1:].clone()
            x = frames_to_input(fr)
            sync()
            t1 = time.perf_counter()
            out = estimate_fn(net, x[:-1], x[1:], tensor=True)
            sync()
            t_est += time.perf_counter() - t1
            shape = tuple(out.shape[1:])
            flows = out if n == chunk else torch.cat([out, out.new_zeros((chunk - n,) + shape)])
        if world > 1:
            if flows is None:                      # a rank past the end of its shard still joins the collective
                if shape is None:
                    probe = frames_fn(0, 1)
                    shape = (2,) + tuple(probe.shape[1:3])
                flows = torch.zeros((chunk,) + shape, device=device)
            _, finish = gather_flows(flows, world * chunk, async_op=True)
        else:
            finish = (lambda f=flows: f)
        rows: List[int] = [max(0, min(min(n_pairs, (r + 1) * per), r * per + (c + 1) * chunk) - (r * per + c * chunk))
                           for r in range(world)]
        if pending is not None:
            drain(pending)
        pending = (c * chunk, finish, rows)
    if pending is not None:
        drain(pending)
    if writer is not None:
        writer.close()
    if on_gpu:
        torch.cuda.synchronize(device)
    dt = time.perf_counter() - t0
    return {"pairs_total": n_pairs, "pairs_this_rank": hi - lo, "seconds": dt, "seconds_estimation": t_est,
            "flows_emitted": emitted}
