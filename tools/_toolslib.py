"""Loader of tools/libpivlfn_tools.so: the same sources as libpivlfn.so compiled with -DPIVLFN_TOOLS -DPIVLFN_STAMPS
(`bash piv_liteflownet-pytorch_amd/csrc/build.sh tools`).  It adds pivlfn_tune (kernel-variant knobs, ablation masks, the
address of an in-kernel stamp buffer) for A/B timing inside one process; the product never loads it."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "piv_liteflownet-pytorch_amd"))
from pivlfn import _lib  # noqa: E402

PATH = os.environ.get("PIVLFN_TOOLS_LIB") or os.path.join(ROOT, "tools", "libpivlfn_tools.so")       # the override: variant builds of tools/ab_variant.sh
_tools = None


def load() -> ctypes.CDLL:
    global _tools
    if _tools is None:
        if not os.path.exists(PATH):
            raise ImportError(f"{PATH} is missing: bash piv_liteflownet-pytorch_amd/csrc/build.sh tools")
        lib = ctypes.CDLL(PATH)
        sigs = dict(_lib.SIGNATURES)
        sigs["pivlfn_tune"] = (ctypes.c_int, [ctypes.c_int, ctypes.c_int])
        sigs["pivlfn_conv2d_nhwc_wino4"] = (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int] + [ctypes.c_int] * 4 + [ctypes.c_void_p])
        for name, (res, args) in sigs.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _tools = lib
    return _tools


def check(lib, rc: int, what: str = "") -> None:
    if rc:
        raise RuntimeError(f"{what}: {lib.pivlfn_last_error().decode(errors='replace')} (code {rc})")


def set_stamp_buffer(lib, ptr: int, which: str = "conv") -> None:
    """64-bit device address of a stamp buffer (0 = off): knobs 5/6 for the conv kernels, 9/10 for warp+correlation."""
    lo, hi = (5, 6) if which == "conv" else (9, 10)
    lib.pivlfn_tune(lo, ctypes.c_int32(ptr & 0xFFFFFFFF).value)
    lib.pivlfn_tune(hi, ctypes.c_int32((ptr >> 32) & 0xFFFFFFFF).value)
