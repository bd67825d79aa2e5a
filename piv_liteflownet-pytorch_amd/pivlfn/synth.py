"""Synthetic inputs for tests and benchmarks: network weights and PIV particle-image pairs.

No pretrained weights ship with the reference (`.MISSING_LARGE_BLOBS:1-5`), so parity is
established on *generated* weights loaded through the same state-dict layout the reference's
factories build (`src/models.py:719-766`).  Everything here is numpy-Philox seeded and keyed
by the state-dict name, so the same tensors come out on every machine and torch version.

* `state_dict_spec(model)`   -- ordered {name: shape}, same keys/shapes as the reference nets.
* `generate_weights(model)`  -- flow-calibrated random weights (see below).
* `particle_pair(H, W, seed)`-- Gaussian-blob particle images following the image model of
  `src/particle_image_generator.py:30-58` (density 0.05 ppp, d = 1.5 + U[0,1) px,
  I = 240 exp(-z^2), uint8), second frame displaced by a Lamb-Oseen vortex + uniform shift.

Calibration: with torch's default init the flow stays ~0.005 px and every warp is an identity,
which tests nothing.  The generator therefore uses variance-preserving conv weights, near-bilinear
(unit-gain) depthwise deconvolutions, near-one regulariser scale weights and O(0.3) flow heads, so
that every pyramid level produces a flow update of a visible fraction of a pixel and the final
field is a few pixels.
"""
from __future__ import annotations

import zlib
from collections import OrderedDict
from typing import Dict, Tuple

import numpy as np

K_LEVEL = [0, 7, 7, 5, 5, 3, 3]          # src/models.py:161,205,225
C_FEAT = [0, 32, 32, 64, 96, 128, 192]   # src/models.py:70-106
C_MATCH = [0, 64, 64, 64, 96, 128, 192]  # NetC_ext lifts L1/L2 to 64 (src/models.py:124,353-357)

MODEL_CFG = {
    # src/models.py:729-730 (hui) and :754-755 (piv)
    "hui": dict(starting_scale=40.0, lowest_level=2,
                rgb_mean=(0.411618, 0.434631, 0.454253, 0.410782, 0.433645, 0.452793)),
    "piv": dict(starting_scale=10.0, lowest_level=1,
                rgb_mean=(0.173935, 0.180594, 0.192608, 0.172978, 0.179518, 0.191300)),
    # LiteFlowNet2 backbones: src/models.py:731-732 (defaults :374-375) and :756-758
    "hui2": dict(starting_scale=40.0, lowest_level=3,
                 rgb_mean=(0.411618, 0.434631, 0.454253, 0.410782, 0.433645, 0.452793)),
    "piv2": dict(starting_scale=10.0, lowest_level=2,
                 rgb_mean=(0.194286, 0.190633, 0.191766, 0.194220, 0.190595, 0.191701)),
}
STACK = {1: [128, 64, 32], 2: [128, 128, 96, 64, 32]}      # hidden widths of conv_M / conv_S (src/models.py:154-163 vs :487-500)


def model_version(model: str) -> int:
    return 2 if model.endswith("2") else 1


def state_dict_spec(model: str = "piv", lowest_level: int | None = None, version: int | None = None
                    ) -> "OrderedDict[str, Tuple[int, ...]]":
    """Ordered name -> shape map of `LiteFlowNet.state_dict()` / `LiteFlowNet2.state_dict()` (src/models.py:305-317, 651-663)."""
    if lowest_level is None:
        lowest_level = MODEL_CFG[model]["lowest_level"]
    if version is None:
        version = model_version(model)
    widths = STACK[version]
    spec: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()

    def conv(name, cout, cin, kh, kw, bias=True):
        spec[name + ".weight"] = (cout, cin, kh, kw)
        if bias:
            spec[name + ".bias"] = (cout,)

    # NetC (src/models.py:70-106)
    conv("NetC.conv1.0", 32, 3, 7, 7)
    conv("NetC.conv2.0", 32, 32, 3, 3); conv("NetC.conv2.2", 32, 32, 3, 3); conv("NetC.conv2.4", 32, 32, 3, 3)
    conv("NetC.conv3.0", 64, 32, 3, 3); conv("NetC.conv3.2", 64, 64, 3, 3)
    conv("NetC.conv4.0", 96, 64, 3, 3); conv("NetC.conv4.2", 96, 96, 3, 3)
    conv("NetC.conv5.0", 128, 96, 3, 3)
    conv("NetC.conv6.0", 192, 128, 3, 3)
    # NetC_ext (src/models.py:309-311): one per level in [lowest_level, 2]
    for i in range(len(range(lowest_level - 1, 2))):
        conv(f"NetC_ext.{i}.conv_ext.0", 64, 32, 1, 1)
    levels = list(range(lowest_level, 7))
    for i, L in enumerate(levels):            # NetE_M (src/models.py:134-163)
        k = K_LEVEL[L]
        if L != 6:
            spec[f"NetE_M.{i}.upConv_M.weight"] = (2, 1, 4, 4)
        if L < 4:
            spec[f"NetE_M.{i}.upCorr_M.weight"] = (49, 1, 4, 4)
        cin = 49
        for j, wd in enumerate(widths):
            conv(f"NetE_M.{i}.conv_M.{2 * j}", wd, cin, 3, 3)
            cin = wd
        conv(f"NetE_M.{i}.conv_M.{2 * len(widths)}", 2, 32, k, k)
    for i, L in enumerate(levels):            # NetE_S (src/models.py:190-207)
        k = K_LEVEL[L]
        cin = 2 * C_MATCH[L] + 2
        for j, wd in enumerate(widths):
            conv(f"NetE_S.{i}.conv_S.{2 * j}", wd, cin, 3, 3)
            cin = wd
        conv(f"NetE_S.{i}.conv_S.{2 * len(widths)}", 2, 32, k, k)
    for i, L in enumerate(levels):            # NetE_R (src/models.py:220-272)
        k = K_LEVEL[L]
        if L < 5:
            conv(f"NetE_R.{i}.moduleFeat.0", 128, C_FEAT[L], 1, 1)
        cin = 195 if L == 6 else 131
        conv(f"NetE_R.{i}.conv_R.0", 128, cin, 3, 3); conv(f"NetE_R.{i}.conv_R.2", 128, 128, 3, 3)
        conv(f"NetE_R.{i}.conv_R.4", 64, 128, 3, 3); conv(f"NetE_R.{i}.conv_R.6", 64, 64, 3, 3)
        conv(f"NetE_R.{i}.conv_R.8", 32, 64, 3, 3); conv(f"NetE_R.{i}.conv_R.10", 32, 32, 3, 3)
        if L < 5:
            conv(f"NetE_R.{i}.conv_dist_R.0", k * k, 32, k, 1); conv(f"NetE_R.{i}.conv_dist_R.1", k * k, k * k, 1, k)
        else:
            conv(f"NetE_R.{i}.conv_dist_R.0", k * k, 32, k, k)
        conv(f"NetE_R.{i}.moduleScaleX", 1, k * k, 1, 1); conv(f"NetE_R.{i}.moduleScaleY", 1, k * k, 1, 1)
    return spec


def _rng(seed: int, name: str) -> np.random.Generator:
    return np.random.Generator(np.random.Philox(key=[seed & 0xFFFFFFFF, zlib.crc32(name.encode())]))


_BILIN = np.outer([0.25, 0.75, 0.75, 0.25], [0.25, 0.75, 0.75, 0.25]).astype(np.float64)


def generate_weights_np(model: str = "piv", seed: int = 0, lowest_level: int | None = None
                        ) -> "OrderedDict[str, np.ndarray]":
    """Flow-calibrated random weights as float32 numpy arrays, in state-dict order."""
    out: "OrderedDict[str, np.ndarray]" = OrderedDict()
    for name, shape in state_dict_spec(model, lowest_level).items():
        g = _rng(seed, name)
        if name.endswith("upConv_M.weight") or name.endswith("upCorr_M.weight"):
            w = _BILIN[None, None] * (1.0 + 0.05 * g.standard_normal(shape))
        elif ".moduleScale" in name:
            w = (1.0 + 0.1 * g.standard_normal(shape)) if name.endswith("weight") else 0.02 * g.standard_normal(shape)
        elif name.endswith(".bias") and shape == (2,):     # flow-head biases
            w = 0.12 * g.standard_normal(shape)
        elif name.endswith(".bias"):
            w = 0.05 * g.standard_normal(shape)
        else:
            cout, cin, kh, kw = shape
            gain = 1.4
            if cout == 2 and cin == 32:      # flow heads (conv_M.6 / conv_S.6, or .10 in LiteFlowNet2)
                gain = 1.0           # flow heads: O(0.3-1) updates in normalised flow units
            if ".conv_dist_R." in name:
                gain = 0.9           # distances of O(1): exp(-d^2) spreads over the patch
            w = g.standard_normal(shape) * (gain / np.sqrt(cin * kh * kw))
        out[name] = np.ascontiguousarray(w, dtype=np.float32)
    return out


def generate_weights(model: str = "piv", seed: int = 0, lowest_level: int | None = None):
    """Same as `generate_weights_np` but as an OrderedDict of torch tensors (a loadable state dict)."""
    import torch
    return OrderedDict((k, torch.from_numpy(v.copy())) for k, v in generate_weights_np(model, seed, lowest_level).items())


def displacement_field(x: np.ndarray, y: np.ndarray, H: int, W: int, peak: float = 4.0,
                       shift: Tuple[float, float] = (1.5, -0.75)) -> Tuple[np.ndarray, np.ndarray]:
    """Lamb-Oseen vortex (peak tangential displacement `peak` px) plus a uniform shift, at points (x, y)."""
    cx, cy = 0.5 * (W - 1), 0.5 * (H - 1)
    rc = 0.18 * min(H, W)
    dx, dy = x - cx, y - cy
    r2 = dx * dx + dy * dy
    r = np.sqrt(r2) + 1e-12
    # v_theta(r) = G (1 - exp(-r^2/rc^2)) / r, max at r = 1.1209 rc with value 0.6382 G / rc
    G = peak * rc / 0.6382
    vt = G * (1.0 - np.exp(-r2 / (rc * rc))) / r
    return shift[0] - vt * dy / r, shift[1] + vt * dx / r


def _render(xp, yp, zp, dp, H, W) -> np.ndarray:
    img = np.zeros((H, W), dtype=np.float64)
    inten = 240.0 * np.exp(-(zp ** 2))
    rad = 4
    for x0, y0, a, d in zip(xp, yp, inten, dp):
        ix, iy = int(np.floor(x0)), int(np.floor(y0))
        xa, xb = max(ix - rad, 0), min(ix + rad + 2, W)
        ya, yb = max(iy - rad, 0), min(iy + rad + 2, H)
        if xa >= xb or ya >= yb:
            continue
        xs = np.arange(xa, xb)[None, :] - x0
        ys = np.arange(ya, yb)[:, None] - y0
        img[ya:yb, xa:xb] += a * np.exp(-(xs * xs + ys * ys) / ((0.5 * d) ** 2))
    return np.clip(img, 0, 255).astype(np.uint8)


def particle_pair(H: int, W: int, seed: int = 1234, density: float = 0.05, peak: float = 4.0,
                  shift: Tuple[float, float] = (1.5, -0.75)):
    """Returns (img1, img2, flow): uint8 [H,W] frames and the true displacement [2,H,W] (u, v) in px."""
    g = np.random.Generator(np.random.Philox(key=[seed & 0xFFFFFFFF, 0x9E3779B9]))
    m = 16
    n = int(np.floor(density * (H + 2 * m) * (W + 2 * m)))
    xp = g.random(n) * (W + 2 * m) - m
    yp = g.random(n) * (H + 2 * m) - m
    zp = g.random(n) - 0.5
    dp = 1.5 + g.random(n)
    u, v = displacement_field(xp, yp, H, W, peak, shift)
    img1 = _render(xp, yp, zp, dp, H, W)
    img2 = _render(xp + u, yp + v, zp, dp, H, W)
    yy, xx = np.meshgrid(np.arange(H, dtype=np.float64), np.arange(W, dtype=np.float64), indexing="ij")
    fu, fv = displacement_field(xx, yy, H, W, peak, shift)
    return img1, img2, np.stack([fu, fv]).astype(np.float32)


def to_input(img_u8: np.ndarray) -> np.ndarray:
    """uint8 [H,W] -> float32 [3,H,W] in [0,1] (what torchvision's ToTensor gives for a gray image -> RGB)."""
    f = img_u8.astype(np.float32) / np.float32(255.0)
    return np.ascontiguousarray(np.broadcast_to(f[None], (3,) + f.shape))


def particle_batch(B: int, H: int, W: int, seed: int = 1234):
    """float32 arrays img1, img2 of shape [B,3,H,W] in [0,1]; frame b uses seed + b."""
    a, c = [], []
    for b in range(B):
        i1, i2, _ = particle_pair(H, W, seed + b)
        a.append(to_input(i1)); c.append(to_input(i2))
    return np.stack(a), np.stack(c)


# ---- frame sequences generated on the device (BASELINE config #4: a long synthetic PIV sequence, sharded over ranks) ----------
class ParticleSequence:
    """A deterministic particle-image SEQUENCE rendered with torch ops on any device: the same particle model as
    `particle_pair` (density, diameter, intensity), particles advected frame to frame by the Lamb-Oseen + uniform-shift field,
    wrapped around a margin so the seeding density stays constant.  Frame k is a pure function of (seed, k): every rank can
    render exactly the frames of its shard (plus the halo frame) and all ranks agree on them.

    frames(k0, k1) -> uint8 [k1-k0, H, W].  Rendering is a scatter-add of each particle's 9x9 Gaussian footprint, in fixed
    point so that it is bit-reproducible on any device."""

    def __init__(self, H: int, W: int, seed: int = 1234, density: float = 0.05, peak: float = 4.0,
                 shift: Tuple[float, float] = (1.5, -0.75), device="cpu"):
        import torch
        self.H, self.W, self.peak, self.shift, self.m = H, W, peak, shift, 16
        self.device = torch.device(device)
        g = np.random.Generator(np.random.Philox(key=[seed & 0xFFFFFFFF, 0x51ED270B]))
        n = int(np.floor(density * (H + 2 * self.m) * (W + 2 * self.m)))
        self._x0 = g.random(n) * (W + 2 * self.m) - self.m
        self._y0 = g.random(n) * (H + 2 * self.m) - self.m
        self.z = torch.from_numpy(g.random(n) - 0.5).to(self.device, torch.float32)
        self.d = torch.from_numpy(1.5 + g.random(n)).to(self.device, torch.float32)
        self._k = 0
        self._x, self._y = self._x0.copy(), self._y0.copy()

    def _positions(self, k: int):
        if k < self._k:
            self._k, self._x, self._y = 0, self._x0.copy(), self._y0.copy()
        W2, H2, m = self.W + 2 * self.m, self.H + 2 * self.m, self.m
        while self._k < k:                                 # float64 on the host: 52 k particles per step, negligible
            u, v = displacement_field(self._x, self._y, self.H, self.W, self.peak, self.shift)
            self._x = (self._x + u + m) % W2 - m
            self._y = (self._y + v + m) % H2 - m
            self._k += 1
        return self._x, self._y

    def frames(self, k0: int, k1: int):
        import torch
        out = torch.empty(k1 - k0, self.H, self.W, dtype=torch.uint8, device=self.device)
        rad = 4
        oy, ox = torch.meshgrid(torch.arange(-rad, rad + 2, device=self.device), torch.arange(-rad, rad + 2, device=self.device), indexing="ij")
        oy, ox = oy.reshape(1, -1), ox.reshape(1, -1)
        inten = 240.0 * torch.exp(-(self.z ** 2))
        for k in range(k0, k1):
            xs, ys = self._positions(k)
            # (float32 by numpy: a torch CPU conversion wakes torch's thread pool, whose spinning workers then slow the next frame's
            #  numpy advection down from 0.6 to 10 ms -- measured with tools/render_time.py)
            x = torch.from_numpy(xs.astype(np.float32)).to(self.device)[:, None]
            y = torch.from_numpy(ys.astype(np.float32)).to(self.device)[:, None]
            ix, iy = torch.floor(x).long() + ox, torch.floor(y).long() + oy
            val = inten[:, None] * torch.exp(-((ix - x) ** 2 + (iy - y) ** 2) / ((0.5 * self.d[:, None]) ** 2))
            ok = (ix >= 0) & (ix < self.W) & (iy >= 0) & (iy < self.H)
            # accumulate in 2^-16 fixed point: integer adds commute, so the frame does not depend on the order in which the
            # device's atomic adds land (float index_add_ on a GPU does, and a frame could differ by one grey level between
            # two renderings -- ranks must agree on the halo frame they both render).  Footprint pixels outside the image add
            # an exact 0 to a clamped index instead of being filtered out: boolean indexing is a device-to-host round trip per
            # frame (15 ms per 1024 x 1024 frame in round 1 -- longer than the network's forward), and the sums are the same integers.
            add = torch.round(val * 65536.0).to(torch.int64) * ok
            idx = iy.clamp(0, self.H - 1) * self.W + ix.clamp(0, self.W - 1)
            img = torch.zeros(self.H * self.W, dtype=torch.int64, device=self.device)
            img.index_add_(0, idx.reshape(-1), add.reshape(-1))
            out[k - k0] = (img.view(self.H, self.W).to(torch.float64) / 65536.0).clamp_(0, 255).to(torch.uint8)
        return out
