"""Streaming input/output pipeline around `estimate` for run.py (reference loop: run.py:137-168).

The reference reads, converts, uploads, infers, downloads and writes one pair at a time on one thread
(`for images, fname in tqdm(dataset)`: run.py:155-166).  Here the three stages overlap:

  loader thread : decode each frame ONCE (sequence mode reuses frame i+1 of pair i as frame i of pair i+1), as uint8 HWC,
                  on a small pool of decode threads, one batch ahead; group equal-sized pairs into batches, stage them
                  in pinned host memory                                                              (PairLoader)
  main thread   : H2D of the uint8 batch on a copy stream, uint8 -> fp32 NCHW / 255 on the device (the arithmetic of
                  torchvision's ToTensor that the reference uses, src/datasets.py:452-453), `estimate`
  D2H + writer  : flow -> pinned buffer on the copy stream, an event per batch; the writer threads wait on the event and
                  write the .flo files while the next batch is being computed                      (drain / FloWriter)

Nothing here touches the numerical path: `estimate(net, a, b)` receives exactly the tensors the reference would build.
"""
from __future__ import annotations

import queue
import threading
from typing import Callable, Iterator, List, Optional, Sequence, Tuple

import numpy as np
import torch


def read_image_u8(path: str) -> np.ndarray:
    """PIL -> RGB -> uint8 [H,W,3] (read_gen of the reference, src/utils_data.py:46-56, before ToTensor)."""
    import PIL.Image
    return np.array(PIL.Image.open(path).convert("RGB"), dtype=np.uint8)


_LUT = {}


def u8_to_input(x: torch.Tensor) -> torch.Tensor:
    """uint8 [n,H,W,3] -> float32 [n,3,H,W] in [0,1], bit-identical to ToTensor's `byte -> float -> div(255)` on the CPU.
    On the GPU torch evaluates `t / 255` as `t * (1/255)`, which differs from the division in the last bit for some of the
    256 values, so the 256 correctly-divided results are tabulated once on the host and gathered on the device."""
    key = str(x.device)
    if key not in _LUT:
        _LUT[key] = (torch.arange(256, dtype=torch.float32) / 255.0).to(x.device)
    n, H, W, C = x.shape
    flat = _LUT[key].index_select(0, x.reshape(-1).to(torch.int32))
    return flat.view(n, H, W, C).permute(0, 3, 1, 2).contiguous()


class PairLoader:
    """Iterates `(names, img1_u8, img2_u8)` batches ([n,H,W,3] uint8 torch tensors, pinned when `pin`), in dataset order,
    over pairs [lo, hi) of a `Run`-like dataset (anything with `.image_list[i] = [path1, path2]` and `.name_list[i]`).
    A batch never mixes image sizes.  Decoding runs on a background thread, `depth` batches ahead."""

    def __init__(self, dataset, lo: int, hi: int, batch: int, depth: int = 2, pin: bool = False,
                 reader: Callable[[str], np.ndarray] = read_image_u8, workers: int = 4):
        if batch < 1 or depth < 1 or workers < 1:
            raise ValueError("PairLoader: batch, depth and workers must be >= 1")
        self.ds, self.lo, self.hi, self.batch, self.pin, self.reader = dataset, lo, hi, batch, pin, reader
        self.workers = workers                             # decode threads (PIL releases the GIL while it decodes)
        self.last_slot = None
        self.decoded = 0                                   # frames actually decoded (tests: sequence mode decodes n+1, not 2n)
        self._rings: dict = {}
        self._ring_size = depth + 4                        # > batches that can be between production and the end of their H2D
        self._q: "queue.Queue" = queue.Queue(maxsize=depth)
        self._stop = threading.Event()
        self._thread = threading.Thread(target=self._produce, daemon=True)
        self._thread.start()

    # ---- producer ------------------------------------------------------------------------------------------------
    def _staging(self, n: int, shape) -> Tuple[torch.Tensor, torch.Tensor, Optional[list]]:
        """Two uint8 [n,H,W,3] staging tensors.  Pinned staging comes from a small ring allocated once per shape
        (`Tensor.pin_memory()` per batch page-locks fresh memory every time: measured 27.7 vs 40 pairs/s at batch 1); a slot is
        reused only after the consumer's H2D copy out of it has completed (the event the consumer stores in slot[2])."""
        if not self.pin:
            return torch.empty((n,) + shape, dtype=torch.uint8), torch.empty((n,) + shape, dtype=torch.uint8), None
        key = (self.batch,) + tuple(shape)
        ring = self._rings.setdefault(key, {"slots": [], "next": 0})
        if len(ring["slots"]) < self._ring_size:
            slot = [torch.empty((self.batch,) + shape, dtype=torch.uint8).pin_memory(),
                    torch.empty((self.batch,) + shape, dtype=torch.uint8).pin_memory(), None]
            ring["slots"].append(slot)
        else:
            slot = ring["slots"][ring["next"] % self._ring_size]
            ring["next"] += 1
            if slot[2] is not None:
                slot[2].synchronize()
                slot[2] = None
        return slot[0][:n], slot[1][:n], slot

    def _emit(self, names: List[str], f1: List[np.ndarray], f2: List[np.ndarray]) -> bool:
        a, b, slot = self._staging(len(names), f1[0].shape)
        for k in range(len(names)):
            a[k].copy_(torch.from_numpy(f1[k]))
            b[k].copy_(torch.from_numpy(f2[k]))
        item = (names, a, b, slot)
        while not self._stop.is_set():
            try:
                self._q.put(item, timeout=0.1)
                return True
            except queue.Full:
                continue
        return False

    def _produce(self) -> None:
        from concurrent.futures import ThreadPoolExecutor
        try:
            with ThreadPoolExecutor(max_workers=self.workers) as pool:
                # decode ahead: the distinct frames of the next max(batch, 2 x workers) pairs are in flight, each submitted once
                pending: dict = {}

                def want(path):
                    if path not in pending:
                        pending[path] = pool.submit(self.reader, path)
                        self.decoded += 1

                names: List[str] = []
                f1: List[np.ndarray] = []
                f2: List[np.ndarray] = []
                ahead = self.lo
                for i in range(self.lo, self.hi):
                    while ahead < min(self.hi, i + max(self.batch, 2 * self.workers)):      # keep every decode thread busy
                        for path in self.ds.image_list[ahead]:
                            want(path)
                        ahead += 1
                    p1, p2 = self.ds.image_list[i]
                    a, b = pending[p1].result(), pending[p2].result()
                    # a frame can only be shared with the following pair (src/datasets.py:456-463): drop everything older
                    nxt = set(self.ds.image_list[i + 1]) if i + 1 < self.hi else set()
                    for path in (p1, p2):
                        if path not in nxt:
                            pending.pop(path, None)
                    if a.shape != b.shape:
                        raise ValueError(f"pair '{self.ds.name_list[i]}': image sizes differ {a.shape} vs {b.shape}")
                    if names and (len(names) == self.batch or a.shape != f1[0].shape):
                        if not self._emit(names, f1, f2):
                            return
                        names, f1, f2 = [], [], []
                    names.append(self.ds.name_list[i])
                    f1.append(a)
                    f2.append(b)
                if names and not self._emit(names, f1, f2):
                    return
            self._q.put(None)
        except BaseException as e:                         # noqa: BLE001  -- re-raised on the consumer side
            self._q.put(e)

    # ---- consumer ------------------------------------------------------------------------------------------------
    def __iter__(self) -> Iterator[Tuple[List[str], torch.Tensor, torch.Tensor]]:
        while True:
            item = self._q.get()
            if item is None:
                return
            if isinstance(item, BaseException):
                raise item
            names, a, b, slot = item
            self.last_slot = slot                          # stream_pairs stores the H2D-complete event here
            yield names, a, b

    def close(self) -> None:
        self._stop.set()
        try:
            while True:
                self._q.get_nowait()
        except queue.Empty:
            pass
        self._thread.join(timeout=5.0)


def stream_pairs(net, loader: PairLoader, device: torch.device, sink: Callable[..., None],
                 estimate_fn: Optional[Callable] = None, in_flight: int = 3,
                 mods: Optional[Sequence[Tuple[float, float]]] = None) -> int:
    """Drive `estimate` over a PairLoader.  `sink(flow_hw2, name)` is called once per pair, in order; the numpy view it gets
    owns a reference to its (pinned) batch buffer, so an asynchronous writer may keep it.  On a GPU the uploads and
    downloads run on a copy stream and overlap with compute; on the CPU (tests of the host logic, with a stand-in
    `estimate_fn`) the same code runs synchronously.
    `mods`: (brightness, contrast) factors of run.py -b/-c; every uploaded batch is then estimated once per entry, with both
    frames modified on the device (pivlfn.imagemod), and the sink is called as `sink(flow_hw2, name, (brightness, contrast))`."""
    if estimate_fn is None:
        from .inference import estimate as estimate_fn      # noqa: N813
    from .imagemod import image_mod
    on_gpu = device.type == "cuda"
    copy = torch.cuda.Stream(device) if on_gpu else None
    pending: List[Tuple[Optional[torch.cuda.Event], torch.Tensor, Sequence[str], Optional[Tuple[float, float]]]] = []
    done = 0

    def drain(keep: int) -> None:
        nonlocal done
        while len(pending) > keep:
            ev, host, names, mod = pending.pop(0)
            if ev is not None:
                ev.synchronize()
            arr = host.numpy()
            for k, name in enumerate(names):
                if mod is None:
                    sink(arr[k], name)
                else:
                    sink(arr[k], name, mod)
            done += len(names)

    for names, a8, b8 in loader:
        if on_gpu:
            main = torch.cuda.current_stream(device)
            with torch.cuda.stream(copy):
                a_dev = a8.to(device, non_blocking=True)
                b_dev = b8.to(device, non_blocking=True)
                slot = getattr(loader, "last_slot", None)
                if slot is not None:                       # the loader may refill this pinned staging slot once the copies are done
                    ev_h2d = torch.cuda.Event()
                    ev_h2d.record(copy)
                    slot[2] = ev_h2d
            main.wait_stream(copy)
            a_dev.record_stream(main)
            b_dev.record_stream(main)
        else:
            a_dev, b_dev = a8, b8
        for mod in (mods if mods is not None else (None,)):
            am, bm = (a_dev, b_dev) if mod is None else (image_mod(a_dev, *mod), image_mod(b_dev, *mod))
            flow = estimate_fn(net, u8_to_input(am), u8_to_input(bm), tensor=True)           # [n,2,H,W]
            out = flow.permute(0, 2, 3, 1).contiguous()                                      # [n,H,W,2], the .flo layout
            if on_gpu:
                host = torch.empty(out.shape, dtype=out.dtype, pin_memory=True)
                copy.wait_stream(main)
                with torch.cuda.stream(copy):
                    host.copy_(out, non_blocking=True)
                    ev = torch.cuda.Event()
                    ev.record(copy)
                out.record_stream(copy)
                pending.append((ev, host, names, mod))
            else:
                pending.append((None, out, names, mod))
            drain(in_flight - 1)
    drain(0)
    return done
