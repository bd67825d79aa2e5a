"""Middlebury `.flo` files: the wire format of the flow output (reference: src/utils_plot.py:26-73, 120-158, 310-318).

Layout: float32 tag 202021.25 ('PIEH'), int32 width, int32 height, then height x width x {2,3} float32, interleaved
(u, v[, w]) in row-major order, little endian.  `FloWriter` is the batched asynchronous writer the multi-GPU sequence
runs use (84 GB of .flo for 10 000 frames at 1024^2 is I/O-bound once the GPU path is fast).
"""
from __future__ import annotations

import io
import os
import queue
import threading

import numpy as np

TAG_FLOAT = 202021.25                  # src/utils_plot.py:15
_HEADER = np.dtype([("tag", "<f4"), ("width", "<i4"), ("height", "<i4")])      # 12 bytes, little endian
_MAX_SIDE = 100000


def _fail(msg: str):
    # the reference signals every malformed-file condition with AssertionError; callers that catch it keep working
    raise AssertionError(msg)


def read_flow(source, use_stereo: bool = False) -> np.ndarray:
    """Middlebury .flo -> float32 [H, W, 2] (3 bands with use_stereo).  `source` is a path ending in .flo or an open binary
    stream (closed on return, as the reference does).  Same contract as src/utils_plot.py:26-73 for well-formed files; a
    payload shorter than the header promises is an error here (the reference silently tiles what it got)."""
    if isinstance(source, io.IOBase):
        stream = source
    else:
        path = os.fspath(source) if isinstance(source, (str, os.PathLike)) else _fail(f"read_flow: {source!r} is neither a path nor a binary stream")
        if not os.path.isfile(path):
            _fail(f"read_flow: no such file: {path}")
        if os.path.splitext(path)[1] != ".flo":
            _fail(f"read_flow: expected a .flo file, got {path}")
        stream = open(path, "rb")
    with stream:
        raw = stream.read(_HEADER.itemsize)
        if len(raw) != _HEADER.itemsize:
            _fail("read_flow: file too short for a .flo header")
        head = np.frombuffer(raw, dtype=_HEADER)[0]
        if float(head["tag"]) != TAG_FLOAT:
            _fail(f"read_flow: bad magic number {float(head['tag'])!r} (want {TAG_FLOAT})")
        width, height = int(head["width"]), int(head["height"])
        if not (0 < width < _MAX_SIDE and 0 < height < _MAX_SIDE):
            _fail(f"read_flow: implausible size {width} x {height}")
        bands = 3 if use_stereo else 2
        want = bands * width * height
        payload = np.frombuffer(stream.read(4 * want), dtype="<f4")
    if payload.size != want:
        _fail(f"read_flow: payload has {payload.size} of {want} values")
    return payload.reshape(height, width, bands).astype(np.float32, copy=True)


def write_flow(flow: np.ndarray, filename: str) -> None:
    """[H, W, 2|3] array -> Middlebury .flo (src/utils_plot.py:120-158 without its optional normalisation).  The payload is
    always written as little-endian float32 (the reference writes the array's own dtype, which only round-trips for float32)."""
    name = os.fspath(filename)
    if not name.endswith(".flo"):
        raise AssertionError(f"write_flow: output name must end in .flo, got {name!r}")
    arr = np.asarray(flow)
    if arr.ndim != 3 or arr.shape[2] not in (2, 3):
        raise AssertionError(f"write_flow: expected [H, W, 2 or 3], got shape {arr.shape}")
    head = np.zeros(1, dtype=_HEADER)
    head["tag"], head["width"], head["height"] = TAG_FLOAT, arr.shape[1], arr.shape[0]
    with open(name, "wb") as f:
        f.write(head.tobytes())
        f.write(np.ascontiguousarray(arr, dtype="<f4").tobytes())


def flowname_modifier(indir: str, outdir: str, ext: str = "_out.flo", pair: bool = True) -> str:
    """Where the flow of input image `indir` goes (src/utils_plot.py:310-318): <outdir>/<stem><ext>, the stem losing its last
    `_suffix` (`_img1`) for paired inputs."""
    stem = os.path.splitext(os.path.basename(indir))[0]
    if pair:
        head, sep, _ = stem.rpartition("_")
        if sep:
            stem = head
    return os.path.join(outdir, stem + ext)


class FloWriter:
    """Background writer: submit(flow_hw2, path) returns immediately; close() drains.  Errors surface on close()."""

    def __init__(self, workers: int = 4, depth: int = 64):
        self._q: "queue.Queue" = queue.Queue(maxsize=depth)
        self._err = []
        self._threads = [threading.Thread(target=self._run, daemon=True) for _ in range(max(1, workers))]
        for t in self._threads:
            t.start()

    def _run(self):
        while True:
            item = self._q.get()
            if item is None:
                return
            try:
                write_flow(item[0], item[1])
            except Exception as e:          # noqa: BLE001
                self._err.append(e)

    def submit(self, flow: np.ndarray, filename: str) -> None:
        self._q.put((flow, filename))

    def close(self) -> None:
        for _ in self._threads:
            self._q.put(None)
        for t in self._threads:
            t.join()
        if self._err:
            raise self._err[0]

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
