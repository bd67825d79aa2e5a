"""Drop-in for the reference's `src/models.py` inference surface.

Same names, argument meaning and state-dict layout as the reference:
  backwarp(tensorInput, tensorFlow)                       src/models.py:20-35
  LiteFlowNet(starting_scale, lowest_level, rgb_mean)     src/models.py:39-317   (.forward :319-370)
  hui_liteflownet(params, version) / piv_liteflownet(...) src/models.py:719-766
  Network(model='piv'|'hui')                              alias named by the project's north star
`LiteFlowNet` holds ordinary torch Parameters under the reference's key names, so real `.paramOnly`
state dicts load with `load_state_dict(strict=True)`; its forward is ONE call into libpivlfn.so
(`pivlfn_forward`), which runs the whole coarse-to-fine pipeline as hand-written gfx950 kernels.

Differences, all deliberate and documented in DESIGN.md:
  * forward does not mutate its inputs (the reference subtracts the mean in place, :321-323; every
    caller passes temporaries, inference.py:46-54);
  * training mode raises NotImplementedError (the reference returns per-level flows, :365-367; they are
    available for tests through `forward_levels`);
LiteFlowNet2 (src/models.py:373-716, `version=2` in the factories) is the same pipeline with five hidden layers in
conv_M / conv_S; it shares every kernel and differs only in the state-dict layout and the default levels.
"""
from __future__ import annotations

import ctypes
import math
from collections import OrderedDict
from typing import List, Optional, Tuple, Union

import torch

from . import _lib
from .synth import MODEL_CFG, state_dict_spec

__all__ = ["hui_liteflownet", "piv_liteflownet", "LiteFlowNet", "LiteFlowNet2", "Network", "backwarp"]


def backwarp(tensorInput: torch.Tensor, tensorFlow: torch.Tensor) -> torch.Tensor:
    """out[b,c,y,x] = bilinear(tensorInput[b,c], x + flow[b,0,y,x], y + flow[b,1,y,x]); zeros outside."""
    if not tensorInput.is_cuda or not tensorFlow.is_cuda:
        raise NotImplementedError("backwarp: GPU tensors only (the reference calls .cuda() unconditionally, src/models.py:27)")
    x = tensorInput.contiguous().float()
    f = tensorFlow.contiguous().float()
    B, C, H, W = x.shape
    if f.shape != (B, 2, H, W):
        raise ValueError(f"backwarp: flow shape {tuple(f.shape)} does not match input {tuple(x.shape)}")
    out = torch.empty_like(x)
    if out.numel() == 0:
        return out
    with torch.cuda.device(x.device):
        _lib.check(_lib.load().pivlfn_backwarp(x.data_ptr(), f.data_ptr(), out.data_ptr(), B, C, H, W,
                                               _lib.stream_ptr(x.device)), "backwarp")
    return out


_PARAM_GEN = [0]      # bumped whenever any parameter object is (re)registered on any holder: invalidates cached parameter lists


class _Holder(torch.nn.Module):
    """Parameter container; exists only so that state_dict() has the reference's key names."""

    def register_parameter(self, name, param):
        _PARAM_GEN[0] += 1
        return super().register_parameter(name, param)


class LiteFlowNet(torch.nn.Module):
    VERSION = 1

    def __init__(self, starting_scale: int = 40, lowest_level: int = 2,
                 rgb_mean: Union[Tuple[float, ...], List[float]] = (0.411618, 0.434631, 0.454253, 0.410782, 0.433645, 0.452793)
                 ) -> None:
        super(LiteFlowNet, self).__init__()
        rgb_mean = list(rgb_mean)
        self.MEAN = [rgb_mean[:3], rgb_mean[3:]]
        self.lowest_level = int(lowest_level)
        self.PLEVELS = 6
        self.starting_scale = float(starting_scale)
        self.SCALEFACTOR = [float(starting_scale) / (2.0 ** level) for level in range(self.PLEVELS + 1)]
        self.level2use = list(range(self.lowest_level, self.PLEVELS + 1))
        gen = torch.Generator().manual_seed(0)
        for name, shape in state_dict_spec(lowest_level=self.lowest_level, version=self.VERSION).items():
            fan_in = (shape[1] * shape[2] * shape[3]) if len(shape) == 4 else self._fan_in_of_bias(name)
            bound = 1.0 / math.sqrt(max(1, fan_in))
            value = (torch.rand(shape, generator=gen) * 2.0 - 1.0) * bound
            self._register(name, torch.nn.Parameter(value, requires_grad=False))
        self._handle = None
        self._handle_key = None
        self._ws = None
        self._precision = "fp32"

    # -- precision of the conv stacks ---------------------------------------------------------------------
    _PRECISIONS = {"fp32": 0, "fp16": 1, "fp32_split": 2, "fp32_split3": 3, "fp32_direct": 4, "fp32_wino_mfma32": 5}

    @property
    def precision(self) -> str:
        """How the large convolutions multiply (everything else -- correlation, warps, heads, small levels -- is fp32 throughout):
        'fp32'        (default) fp32 operands (all 24 bits), fp32 products and accumulation.  The 3x3 stride-1 layers by Winograd F(2x2, 3x3):
                      those with whole 64-channel output groups and >= 48 input channels with every operand split exactly into three
                      bf16 pieces on v_mfma_f32_32x32x16_bf16 (csrc/conv_wino_b3.hip: six exact piece products per product, what is
                      dropped is <= 2^-23 of it; fp32's input domain), the others on the fp32 matrix-core instruction
                      v_mfma_f32_32x32x2_f32 (csrc/conv_wino.hip); every other layer by direct convolution on that instruction;
        'fp32_wino_mfma32' the default of rounds 3-5: as 'fp32' with every Winograd layer on the fp32 instruction;
        'fp32_direct' the same instruction, direct convolution for every layer (2.25 x the multiplies of 'fp32' in the 3x3 layers);
        'fp32_split'  fp32 operands as three fp16 pieces each (all 24 bits), six partial products on the fp16 matrix cores, fp32
                      accumulation: products exact to 2^-32 (csrc/conv_split.hip); inputs of those layers must stay below 65504;
        'fp32_split3' two pieces, three partial products: 22-23 operand bits, products good to 2^-21 -- narrower than fp32, opt-in;
        'fp16'        operands rounded to fp16 (BASELINE config #5): reduced precision, its own tolerance."""
        return self._precision

    @precision.setter
    def precision(self, value: str) -> None:
        if value not in self._PRECISIONS:
            raise ValueError("precision must be 'fp32', 'fp32_wino_mfma32', 'fp32_direct', 'fp32_split', 'fp32_split3' or 'fp16'")
        self._precision = value
        if self.__dict__.get("_handle") is not None:
            _lib.check(_lib.load().pivlfn_set_precision(self._handle, self._PRECISIONS[value]), "set_precision")

    # -- parameter tree --------------------------------------------------------------------------------
    def _fan_in_of_bias(self, name: str) -> int:
        w = dict(state_dict_spec(lowest_level=self.lowest_level, version=self.VERSION))[name[:-len("bias")] + "weight"]
        return w[1] * w[2] * w[3]

    def _register(self, dotted: str, param: torch.nn.Parameter) -> None:
        mod = self
        parts = dotted.split(".")
        for p in parts[:-1]:
            if p not in mod._modules:
                mod.add_module(p, _Holder())
            mod = mod._modules[p]
        mod.register_parameter(parts[-1], param)

    # -- native handle ---------------------------------------------------------------------------------
    def _key(self):
        # the parameter list is cached: walking the 250-module tree on every forward costs 2 ms of host time, which is what
        # a small or fp16-mode forward takes on the GPU (load_state_dict / .to() change data in place or swap .data: both
        # show up in _version / data_ptr, and _apply() drops the cache); the key itself costs ~50 us for 252 parameters
        ps = self.__dict__.get("_plist")
        if ps is None or self.__dict__.get("_plist_gen") != _PARAM_GEN[0]:
            ps = list(self.parameters())
            self.__dict__["_plist"] = ps
            self.__dict__["_plist_gen"] = _PARAM_GEN[0]
        # every parameter's storage address is part of the key: `p.data = new_tensor` (weight surgery, EMA swaps) bumps no version
        return (ps[0].device, tuple(p._version for p in ps), tuple(p.data_ptr() for p in ps))

    def _apply(self, fn, *args, **kwargs):
        self.__dict__["_plist"] = None
        return super()._apply(fn, *args, **kwargs)

    def _native(self):
        key = self._key()
        if self._handle is not None and key == self._handle_key:
            return self._handle
        self._release()
        dev = key[0]
        if dev.type != "cuda":
            raise NotImplementedError("LiteFlowNet.forward: the network must live on a GPU (net.to('cuda')); "
                                      "there is no CPU path")
        lib = _lib.load()
        sd = self.state_dict()
        host = [(k, v.detach().to("cpu", torch.float32).contiguous()) for k, v in sd.items()]
        arr = (_lib.Tensor * len(host))()
        for i, (k, v) in enumerate(host):
            arr[i].name = k.encode()
            arr[i].data = v.data_ptr()
            arr[i].ndim = v.dim()
            for d in range(v.dim()):
                arr[i].shape[d] = v.shape[d]
        mean = (ctypes.c_float * 6)(*(self.MEAN[0] + self.MEAN[1]))
        h = ctypes.c_void_p()
        with torch.cuda.device(dev):
            _lib.check(lib.pivlfn_create(arr, len(host), self.starting_scale, self.lowest_level, mean, ctypes.byref(h)),
                       "LiteFlowNet: weight upload")
        self._handle, self._handle_key = h, key
        _lib.check(lib.pivlfn_set_precision(h, self._PRECISIONS[self._precision]), "set_precision")
        return h

    def _release(self):
        h = self.__dict__.get("_handle")
        if h is not None:
            try:
                _lib.load().pivlfn_destroy(h)
            except Exception:
                pass
        self.__dict__["_handle"] = None
        self.__dict__["_handle_key"] = None

    def __del__(self):
        try:
            self._release()
        except Exception:
            pass

    def _workspace(self, nbytes: int, device) -> torch.Tensor:
        if self._ws is None or self._ws.numel() < nbytes or self._ws.device != device:
            self._ws = None
            self._ws = torch.empty(nbytes, dtype=torch.uint8, device=device)
        return self._ws

    # -- forward ---------------------------------------------------------------------------------------
    def _run(self, img1: torch.Tensor, img2: torch.Tensor, want_levels: bool):
        if img1.shape != img2.shape or img1.dim() != 4 or img1.shape[1] != 3:
            raise ValueError(f"LiteFlowNet.forward: expected two [B,3,H,W] tensors, got {tuple(img1.shape)} and {tuple(img2.shape)}")
        h = self._native()
        dev = self._handle_key[0]
        if img1.device != dev or img2.device != dev:
            raise ValueError(f"LiteFlowNet.forward: inputs on {img1.device}, network on {dev}")
        a = img1.detach().contiguous().float()
        b = img2.detach().contiguous().float()
        B, _, H, W = a.shape
        if H % 32 or W % 32 or H < 32 or W < 32:
            raise ValueError(f"LiteFlowNet.forward: H={H}, W={W} must be multiples of 32 (estimate() adapts other sizes, "
                             "as inference.py:39-49 does)")
        lib = _lib.load()
        div = 2 ** (self.lowest_level - 1)
        out = torch.empty([B, 2, H // div, W // div], dtype=torch.float32, device=dev)
        lv = None
        with torch.cuda.device(dev):
            ws = self._workspace(lib.pivlfn_workspace_bytes(h, B, H, W), dev)
            if want_levels:
                lv = torch.empty(lib.pivlfn_levels_floats(h, B, H, W), dtype=torch.float32, device=dev)
            _lib.check(lib.pivlfn_forward(h, a.data_ptr(), b.data_ptr(), out.data_ptr(), lv.data_ptr() if want_levels else None,
                                          B, H, W, ws.data_ptr(), ws.numel(), _lib.stream_ptr(dev)), "LiteFlowNet.forward")
        return out, lv

    def forward(self, img1: torch.Tensor, img2: torch.Tensor) -> torch.Tensor:
        if self.training:
            raise NotImplementedError("LiteFlowNet: inference only -- call net.eval() first (training-mode per-level "
                                      "flows are exposed by forward_levels for tests)")
        return self._run(img1, img2, False)[0]

    def forward_levels(self, img1: torch.Tensor, img2: torch.Tensor):
        """(flow, [[M,S,R] per level, coarsest first]) -- the training-mode return of src/models.py:363-367."""
        out, lv = self._run(img1, img2, True)
        B, _, H, W = img1.shape
        res, off = [], 0
        for L in range(6, self.lowest_level - 1, -1):
            h, w = H >> (L - 1), W >> (L - 1)
            trio = []
            for _ in range(3):
                n = B * 2 * h * w
                trio.append(lv[off:off + n].view(B, 2, h, w))
                off += n
            res.append(trio)
        return out, res

    # -- measurement hooks (bench.py) -------------------------------------------------------------------
    def profile_enable(self, level: int) -> None:
        _lib.check(_lib.load().pivlfn_profile_enable(self._native(), int(level)), "profile_enable")

    def profile_read(self, reset: bool = True):
        ms, empty, n = ctypes.c_double(), ctypes.c_double(), ctypes.c_long()
        _lib.check(_lib.load().pivlfn_profile_read(self._native(), ctypes.byref(ms), ctypes.byref(empty), ctypes.byref(n), int(reset)),
                   "profile_read")
        return ms.value, empty.value, n.value


class LiteFlowNet2(LiteFlowNet):
    """LiteFlowNet2 (Hui 2020), src/models.py:373-716: defaults starting_scale 40, lowest level 3 (quarter-resolution flow)."""
    VERSION = 2

    def __init__(self, starting_scale: int = 40, lowest_level: int = 3,
                 rgb_mean: Union[Tuple[float, ...], List[float]] = (0.411618, 0.434631, 0.454253, 0.410782, 0.433645, 0.452793)
                 ) -> None:
        super(LiteFlowNet2, self).__init__(starting_scale, lowest_level, rgb_mean)


def _build(model: str, params: Optional[OrderedDict], version: int) -> LiteFlowNet:
    if version not in (1, 2):
        raise ValueError(f'Wrong input of model version (input = {version})! Choose between version 1 or 2 only!')
    cfg = MODEL_CFG[model + ("2" if version == 2 else "")]
    cls = LiteFlowNet2 if version == 2 else LiteFlowNet
    net = cls(starting_scale=cfg["starting_scale"], lowest_level=cfg["lowest_level"], rgb_mean=cfg["rgb_mean"])
    if params is not None:
        net.load_state_dict(params)
    return net


def hui_liteflownet(params: Optional[OrderedDict] = None, version: int = 1) -> LiteFlowNet:
    """LiteFlowNet (Hui 2018): starting_scale 40, lowest level 2, half-resolution flow (src/models.py:719-740)."""
    return _build("hui", params, version)


def piv_liteflownet(params: Optional[OrderedDict] = None, version: int = 1) -> LiteFlowNet:
    """PIV-LiteFlowNet-en (Cai 2019): starting_scale 10, lowest level 1, full-resolution flow (src/models.py:743-766)."""
    return _build("piv", params, version)


def Network(model: str = "piv", params: Optional[OrderedDict] = None, version: int = 1) -> LiteFlowNet:
    """`Network(model=...)` spelling used by the project's north star; same objects as the two factories."""
    if model not in ("piv", "hui"):
        raise ValueError(f"model must be 'piv' or 'hui', got {model!r}")
    return _build(model, params, version)
