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
from typing import Union

import numpy as np

TAG_FLOAT = 202021.25                  # src/utils_plot.py:15


def read_flow(filename: Union[str, io.BufferedReader], use_stereo: bool = False) -> np.ndarray:
    """Returns [H, W, 2] (or 3 with use_stereo) float32.  Error behaviour of src/utils_plot.py:26-73 (AssertionError)."""
    if not isinstance(filename, io.BufferedReader):
        if not isinstance(filename, str):
            raise AssertionError(f"Input [{filename}] is not a string")
        if not os.path.isfile(filename):
            raise AssertionError(f"Path [{filename}] does not exist")
        if filename.split(".")[-1] != "flo":
            raise AssertionError(f"File extension [flo] required, [{filename.split('.')[-1]}] given")
        f = open(filename, "rb")
    else:
        f = filename
    try:
        tag = np.frombuffer(f.read(4), np.float32, count=1)[0]
        if not TAG_FLOAT == tag:
            raise AssertionError(f"Wrong Tag [{tag}]")
        width = int(np.frombuffer(f.read(4), np.int32, count=1)[0])
        if not (0 < width < 100000):
            raise AssertionError(f"Illegal width [{width}]")
        height = int(np.frombuffer(f.read(4), np.int32, count=1)[0])
        if not (0 < height < 100000):
            raise AssertionError(f"Illegal height [{height}]")
        bands = 3 if use_stereo else 2
        data = np.frombuffer(f.read(bands * width * height * 4), np.float32, count=bands * width * height)
    finally:
        f.close()
    return np.array(data, dtype=np.float32).reshape(height, width, bands)


def write_flow(flow: np.ndarray, filename: str) -> None:
    """flow: [H, W, 2|3] float32 (src/utils_plot.py:120-158, without the optional normalisation)."""
    assert type(filename) is str, "file is not str (%r)" % str(filename)
    assert filename[-4:] == ".flo", "file ending is not .flo (%r)" % filename[-4:]
    height, width, bands = flow.shape
    assert bands == 2 or bands == 3, "Number of bands = %r != (2 or 3)" % bands
    with open(filename, "wb") as f:
        np.array([TAG_FLOAT], dtype=np.float32).tofile(f)
        np.array([width], dtype=np.int32).tofile(f)
        np.array([height], dtype=np.int32).tofile(f)
        np.ascontiguousarray(flow, dtype=np.float32).tofile(f)


def flowname_modifier(indir: str, outdir: str, ext: str = "_out.flo", pair: bool = True) -> str:
    """Output file name for an input image name (src/utils_plot.py:310-318)."""
    out_name = os.path.splitext(os.path.basename(indir))[0]
    if pair:
        out_name = str(out_name.rsplit("_", 1)[0]) + ext
    else:
        out_name += ext
    return os.path.join(outdir, out_name)


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
