"""A frame SEQUENCE through the path, sharded over ranks: BASELINE config #4 (10 000 frames of 1024x1024 over 8 GPUs).

Pair i = (frame i, frame i+1) (the reference's sequence mode, src/datasets.py:456-463).  Rank r of R owns the contiguous
pairs `shard_bounds(n_pairs, r, R)`, renders / reads only the frames of its shard plus one halo frame, estimates them chunk
by chunk, and the flows of every chunk position are reassembled with one all-gather (RCCL on GPUs, issued asynchronously so
it overlaps the next chunk; no other collective touches the data path).  Rank 0 hands the gathered flows to a sink -- by
default a background `.flo` writer naming the files `frame_<pair index, 6 digits>_out.flo`.  The loop is a pipeline: frames of the
next chunk are produced on a side stream while the current one is estimated, flows leave through a pinned ring on a copy stream.

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


class _FramePrefetcher:
    """Asks `frames_fn` for the chunks' frame ranges one chunk ahead of the estimation, in a thread of its own and (on a GPU) on a side
    stream: the renderer / decoder of chunk c + 1 runs while chunk c is estimated.  Every frame is still asked for exactly once, in
    increasing order (one producer).  get() returns (frames, event) -- the consumer's stream waits for the event."""

    def __init__(self, frames_fn, ranges, device, depth: int = 2):
        import queue
        import threading
        self._fn, self._ranges, self._dev = frames_fn, list(ranges), device
        self._q: "queue.Queue" = queue.Queue(maxsize=depth)
        self._stop = threading.Event()
        self._err = None
        self._t = threading.Thread(target=self._run, daemon=True)
        self._t.start()

    def _run(self):
        try:
            on_gpu = self._dev.type == "cuda"
            if on_gpu:
                torch.cuda.set_device(self._dev)
                side = torch.cuda.Stream(self._dev)
            for rng in self._ranges:
                if self._stop.is_set():
                    return
                if rng is None:
                    self._q.put((None, None))
                    continue
                if on_gpu:
                    with torch.cuda.stream(side):
                        fr = self._fn(*rng)
                        ev = torch.cuda.Event()
                        ev.record(side)
                else:
                    fr, ev = self._fn(*rng), None
                self._q.put((fr, ev))
        except BaseException as e:      # noqa: BLE001 -- handed to the consumer
            self._err = e
            self._q.put((None, None))

    def get(self):
        fr, ev = self._q.get()
        if self._err is not None:
            raise self._err
        return fr, ev

    def close(self):
        self._stop.set()
        while self._t.is_alive():
            try:
                self._q.get_nowait()
            except Exception:       # noqa: BLE001
                pass
            self._t.join(timeout=0.05)


class _FlowDrain:
    """Rank 0's output side: the (gathered) flows of a chunk go device -> pinned host buffer on a copy stream (a ring of `depth` buffers
    allocated once), and a thread of its own waits for the copy's event, turns the rows into [H,W,2] arrays and hands them to the sink.
    The main thread only enqueues."""

    def __init__(self, sink, device, depth: int = 3):
        import queue
        import threading
        self._sink, self._dev = sink, device
        self._on_gpu = device.type == "cuda"
        self._copy = torch.cuda.Stream(device) if self._on_gpu else None
        self._free: "queue.Queue" = queue.Queue()
        self._work: "queue.Queue" = queue.Queue()
        self._ring = [None] * depth
        for i in range(depth):
            self._free.put(i)
        self._err = None
        self.emitted = 0
        self._t = threading.Thread(target=self._run, daemon=True)
        self._t.start()

    def _run(self):
        import numpy as np
        while True:
            item = self._work.get()
            if item is None:
                return
            slot, ev, n_rows, targets = item
            try:
                if ev is not None:
                    ev.synchronize()
                host = self._ring[slot][:n_rows].numpy()                 # [rows, 2, H, W]
                for row, gi in targets:
                    self._sink(gi, np.ascontiguousarray(host[row].transpose(1, 2, 0)))
                    self.emitted += 1
            except BaseException as e:      # noqa: BLE001
                self._err = e
            finally:
                self._free.put(slot)

    def submit(self, finish, targets, ready_event=None):
        """finish() -> the chunk's flows [rows,2,H,W] (it may wait for an asynchronous all-gather); targets = [(row, pair index)]."""
        slot = self._free.get()              # back-pressure: at most `depth` chunks between the GPU and the sink
        if self._on_gpu:
            with torch.cuda.stream(self._copy):
                if ready_event is not None:
                    self._copy.wait_event(ready_event)
                full = finish()
                if full.is_cuda:
                    if self._ring[slot] is None or self._ring[slot].shape[0] < full.shape[0] or self._ring[slot].shape[1:] != full.shape[1:]:
                        self._ring[slot] = torch.empty(full.shape, dtype=full.dtype).pin_memory()
                    self._ring[slot][:full.shape[0]].copy_(full, non_blocking=True)
                    full.record_stream(self._copy)
                    ev = torch.cuda.Event()
                    ev.record(self._copy)
                else:
                    self._ring[slot], ev = full, None
        else:
            self._ring[slot], ev = finish(), None
        self._work.put((slot, ev, self._ring[slot].shape[0], targets))

    def close(self):
        self._work.put(None)
        self._t.join()
        if self._err is not None:
            raise self._err


def run_sequence(net, frames_fn: Callable[[int, int], torch.Tensor], n_frames: int, chunk: int, device: torch.device,
                 write_dir: Optional[str] = None, sink: Optional[Callable[[int, "object"], None]] = None,
                 rank: int = 0, world: int = 1, estimate_fn: Callable = estimate, gather: bool = True) -> Dict[str, float]:
    """Estimate all n_frames-1 pairs.  With world > 1 a process group must be initialised (`nccl` on GPUs, `gloo` for
    rehearsals).  `gather=True` (BASELINE config #4): the flows of every chunk position are reassembled with one asynchronous
    all-gather and rank 0's `sink(pair_index, flow_hw2_numpy)` -- or its `.flo` writer when `write_dir` is given -- sees every pair
    exactly once.  `gather=False`: no collective at all; every rank hands ITS OWN shard to its own sink / writer (same file names, so
    a shared directory ends up with the same set) -- rank 0 then does not receive world x the device-to-host traffic.
    A pipeline (round 6): the frames of chunk c + 1 are rendered / decoded on a side stream by a producer thread while chunk c is
    estimated; nothing synchronises the host with the compute stream inside the loop (the estimation time is taken with events);
    finished chunks leave through a pinned ring on a copy stream and a sink thread.  Returns timing / count statistics of this rank."""
    if n_frames < 2 or chunk < 1:
        raise ValueError("run_sequence: need at least two frames and chunk >= 1")
    n_pairs = n_frames - 1
    lo, hi = shard_bounds(n_pairs, rank, world)
    per = -(-n_pairs // world)
    nchunks = -(-per // chunk)                     # identical on every rank: the gather is collective
    on_gpu = device.type == "cuda"
    use_gather = gather and world > 1
    writer = None
    if write_dir is not None and (rank == 0 or not gather):
        os.makedirs(write_dir, exist_ok=True)
        writer = FloWriter()
        sink = lambda gi, flow: writer.submit(flow, os.path.join(write_dir, flow_file_name(gi)))      # noqa: E731
    i_sink = sink if (rank == 0 or not gather) else None
    drain = _FlowDrain(i_sink, device) if i_sink is not None else None

    # frame ranges the producer asks for, chunk by chunk (the halo frame of a chunk is the last frame of the previous one)
    ranges, first = [], True
    for c in range(nchunks):
        i0 = lo + c * chunk
        i1 = min(hi, i0 + chunk)
        if i1 - i0 <= 0:
            ranges.append(None)
            continue
        ranges.append((i0 if first else i0 + 1, i1 + 1))
        first = False
    frames = _FramePrefetcher(frames_fn, ranges, device)

    timing: List = []          # (start event, stop event) per estimate call on a GPU; seconds on the CPU
    t_est_cpu = 0.0
    if on_gpu:
        torch.cuda.current_stream(device).synchronize()
    t0 = time.perf_counter()
    halo = None                                    # last frame of the previous chunk: shared by two consecutive pairs
    shape = None
    try:
        for c in range(nchunks):
            i0 = lo + c * chunk
            i1 = min(hi, i0 + chunk)
            n = max(0, i1 - i0)
            fr, ev = frames.get()
            flows = None
            if n > 0:
                if ev is not None:
                    torch.cuda.current_stream(device).wait_event(ev)
                    fr.record_stream(torch.cuda.current_stream(device))
                if halo is not None:
                    fr = torch.cat([halo, fr])
                halo = fr[-1:].clone()
                x = frames_to_input(fr)
                if on_gpu:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    out = estimate_fn(net, x[:-1], x[1:], tensor=True)
                    e1.record()
                    timing.append((e0, e1))
                else:
                    t1 = time.perf_counter()
                    out = estimate_fn(net, x[:-1], x[1:], tensor=True)
                    t_est_cpu += time.perf_counter() - t1
                shape = tuple(out.shape[1:])
                flows = out if (n == chunk or not use_gather) else torch.cat([out, out.new_zeros((chunk - n,) + shape)])
            if use_gather:
                if flows is None:                      # a rank past the end of its shard still joins the collective
                    if shape is None:
                        probe = frames_fn(0, 1)
                        shape = (2,) + tuple(probe.shape[1:3])
                    flows = torch.zeros((chunk,) + shape, device=device)
                _, finish = gather_flows(flows, world * chunk, async_op=True)
                rows = [max(0, min(min(n_pairs, (r + 1) * per), r * per + (c + 1) * chunk) - (r * per + c * chunk)) for r in range(world)]
                targets = [(r * chunk + j, r * per + c * chunk + j) for r in range(world) for j in range(rows[r])]
                ready = None
            else:
                finish = (lambda f=flows: f)
                targets = [(j, i0 + j) for j in range(n)]
                ready = None
                if on_gpu and flows is not None:
                    ready = torch.cuda.Event()
                    ready.record()
            if drain is not None and (use_gather or flows is not None):
                drain.submit(finish, targets, ready)
            elif use_gather:
                finish()                               # ranks without a sink still complete the collective
    finally:
        frames.close()
    if drain is not None:
        drain.close()
    if writer is not None:
        writer.close()
    if on_gpu:
        torch.cuda.synchronize(device)
    dt = time.perf_counter() - t0
    t_est = t_est_cpu + sum(a.elapsed_time(b) for a, b in timing) * 1e-3
    return {"pairs_total": n_pairs, "pairs_this_rank": hi - lo, "seconds": dt, "seconds_estimation": t_est,
            "flows_emitted": drain.emitted if drain is not None else 0}
