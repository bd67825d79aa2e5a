"""Multi-GPU sharding of an image-pair sequence: one process per GPU, RCCL only to reassemble the flows.

Every pair is independent (the reference processes them one by one, run.py:159-166), so the path shards with no
data-path collective: rank r of R owns the contiguous pairs [r*ceil(N/R), min(N, (r+1)*ceil(N/R))).  In sequence mode
pair i = (frame i, frame i+1) (src/datasets.py:456-463), so a rank needs one halo frame past its last pair.  The only
exchange is the all-gather that rebuilds the [N,2,H,W] flow sequence (8.4 MB per 1024x1024 pair: far below one xGMI
link), issued per chunk and asynchronously so it overlaps the next chunk's compute.

Backend-agnostic on purpose: `nccl` (= RCCL on ROCm) on GPUs, `gloo` on CPU for the tests (with a stub network).
"""
from __future__ import annotations

from typing import Callable, Iterable, List, Optional, Tuple

import torch
import torch.distributed as dist


def shard_bounds(n_pairs: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous shard [lo, hi) of rank `rank`; later ranks may be short or empty."""
    if n_pairs < 0 or world <= 0 or not (0 <= rank < world):
        raise ValueError(f"shard_bounds({n_pairs}, {rank}, {world})")
    per = -(-n_pairs // world) if n_pairs else 0
    lo = min(n_pairs, rank * per)
    return lo, min(n_pairs, lo + per)


def frames_needed(lo: int, hi: int, is_pair: bool) -> Tuple[int, int]:
    """Frame index range [f0, f1) a rank must read for pairs [lo, hi).

    is_pair=True : files come as (img1, img2) couples, pair i = frames (2i, 2i+1)   (src/datasets.py:447-455)
    is_pair=False: a sequence, pair i = frames (i, i+1): one halo frame            (src/datasets.py:456-463)
    """
    if hi <= lo:
        return 0, 0
    return (2 * lo, 2 * hi) if is_pair else (lo, hi + 1)


def gather_flows(local: torch.Tensor, n_total: int, group=None, async_op: bool = False):
    """All-gather per-rank flow shards [n_r,2,H,W] (contiguous shards as given by shard_bounds) into [n_total,2,H,W].

    Returns the full tensor (async_op=False) or (work_handle, finish) where finish() returns it after work.wait().
    Short / empty shards are padded to ceil(n_total/world) rows for the collective and trimmed afterwards.
    """
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    per = -(-n_total // world) if n_total else 0
    lo, hi = shard_bounds(n_total, rank, world)
    if local.shape[0] != hi - lo:
        raise ValueError(f"rank {rank} holds {local.shape[0]} flows, its shard is [{lo},{hi})")
    if dist.get_backend(group) != "nccl" and local.is_cuda:
        local = local.cpu()                    # gloo rehearsals on a GPU box: the collective runs on host copies
    tail = tuple(local.shape[1:])
    send = local.contiguous()
    if send.shape[0] != per:
        pad = torch.zeros((per,) + tail, dtype=local.dtype, device=local.device)
        pad[: send.shape[0]] = send
        send = pad
    out = torch.empty((world * per,) + tail, dtype=local.dtype, device=local.device)
    if per == 0:
        return out[:0] if not async_op else (None, lambda: out[:0])
    if dist.get_backend(group) == "nccl":
        work = dist.all_gather_into_tensor(out, send, group=group, async_op=async_op)
    else:
        work = dist.all_gather(list(out.view((world, per) + tail).unbind(0)), send, group=group, async_op=async_op)
    if not async_op:
        return out[:n_total]

    def finish():
        work.wait()
        return out[:n_total]
    return work, finish


def run_sharded(flow_fn: Callable[[torch.Tensor, torch.Tensor], torch.Tensor], load_pair_batch: Callable[[int, int], Tuple[torch.Tensor, torch.Tensor]],
                n_pairs: int, batch: int = 1, group=None) -> torch.Tensor:
    """Process `n_pairs` pairs over all ranks and return the reassembled [n_pairs,2,H,W] flow sequence on every rank.

    flow_fn(img1, img2) -> [b,2,H,W]  (e.g. lambda a, b: estimate(net, a, b, tensor=True))
    load_pair_batch(i0, i1) -> (img1, img2) for global pair indices [i0, i1) on this rank's device.
    """
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    lo, hi = shard_bounds(n_pairs, rank, world)
    outs: List[torch.Tensor] = []
    for i0 in range(lo, hi, batch):
        a, b = load_pair_batch(i0, min(hi, i0 + batch))
        outs.append(flow_fn(a, b))
    if outs:
        local = torch.cat(outs, 0)
        shape = torch.tensor(list(local.shape[1:]), dtype=torch.int64, device=local.device)
    else:
        local, shape = None, None
    # ranks with an empty shard learn the flow shape from rank 0 (which is never empty when n_pairs > 0)
    if n_pairs == 0:
        return torch.empty(0)
    dev = local.device if local is not None else _default_device(group)
    if shape is None:
        shape = torch.zeros(3, dtype=torch.int64, device=dev)
    dist.broadcast(shape, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
    if local is None:
        local = torch.empty((0,) + tuple(int(v) for v in shape.tolist()), dtype=torch.float32, device=dev)
    return gather_flows(local, n_pairs, group=group)


def _default_device(group):
    if dist.get_backend(group) == "nccl":
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")
