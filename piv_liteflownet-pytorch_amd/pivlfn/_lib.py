"""ctypes binding of libpivlfn.so (C ABI declared in include/pivlfn.h).  Fails loudly when the library is absent."""
from __future__ import annotations

import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libpivlfn.so")

c_float_p = ctypes.POINTER(ctypes.c_float)


class Tensor(ctypes.Structure):
    """struct pivlfn_tensor"""
    _fields_ = [("name", ctypes.c_char_p), ("data", ctypes.c_void_p), ("ndim", ctypes.c_int), ("shape", ctypes.c_int * 4)]


# name -> (restype, argtypes); one entry per symbol declared in include/pivlfn.h
SIGNATURES = {
    "pivlfn_last_error": (ctypes.c_char_p, []),
    "pivlfn_abi_version": (ctypes.c_int, []),
    "pivlfn_corr_fwd": (ctypes.c_int, [ctypes.c_void_p] * 3 + [ctypes.c_int] * 5 + [ctypes.c_void_p]),
    "pivlfn_corr_bwd": (ctypes.c_int, [ctypes.c_void_p] * 5 + [ctypes.c_int] * 5 + [ctypes.c_void_p]),
    "pivlfn_conv2d_nhwc_f16": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int] + [ctypes.c_int] * 7 + [ctypes.c_void_p]),
    "pivlfn_conv2d_nhwc_split": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int] + [ctypes.c_int] * 8 + [ctypes.c_void_p]),
    "pivlfn_conv2d_nhwc_wino": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int] + [ctypes.c_int] * 4 + [ctypes.c_void_p]),
    "pivlfn_conv2d_nhwc_wino_b3": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int] + [ctypes.c_int] * 5 + [ctypes.c_void_p]),
    "pivlfn_conv_create_cat": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    "pivlfn_conv2d_nhwc_cat": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int] + [ctypes.c_int] * 4 + [ctypes.c_void_p]),
    "pivlfn_set_precision": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int]),
    "pivlfn_backwarp": (ctypes.c_int, [ctypes.c_void_p] * 3 + [ctypes.c_int] * 4 + [ctypes.c_void_p]),
    "pivlfn_warp_corr_fwd": (ctypes.c_int, [ctypes.c_void_p] * 3 + [ctypes.c_float, ctypes.c_void_p] + [ctypes.c_int] * 6 + [ctypes.c_void_p]),
    "pivlfn_warp_corr_nhwc": (ctypes.c_int, [ctypes.c_void_p] * 3 + [ctypes.c_float, ctypes.c_void_p] + [ctypes.c_int] * 6 + [ctypes.c_void_p]),
    "pivlfn_warp_corr_nhwc_timed": (ctypes.c_int, [ctypes.c_void_p] * 3 + [ctypes.c_float, ctypes.c_void_p] + [ctypes.c_int] * 7
                                    + [ctypes.POINTER(ctypes.c_double), ctypes.c_void_p]),
    "pivlfn_resize_bilinear": (ctypes.c_int, [ctypes.c_void_p] * 2 + [ctypes.c_int] * 6 + [c_float_p, ctypes.c_void_p]),
    "pivlfn_create": (ctypes.c_int, [ctypes.POINTER(Tensor), ctypes.c_int, ctypes.c_float, ctypes.c_int, c_float_p, ctypes.POINTER(ctypes.c_void_p)]),
    "pivlfn_destroy": (ctypes.c_int, [ctypes.c_void_p]),
    "pivlfn_workspace_bytes": (ctypes.c_size_t, [ctypes.c_void_p] + [ctypes.c_int] * 3),
    "pivlfn_levels_floats": (ctypes.c_size_t, [ctypes.c_void_p] + [ctypes.c_int] * 3),
    "pivlfn_forward": (ctypes.c_int, [ctypes.c_void_p] * 5 + [ctypes.c_int] * 3 + [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    "pivlfn_conv_create": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p] + [ctypes.c_int] * 4 + [ctypes.POINTER(ctypes.c_void_p)]),
    "pivlfn_conv_destroy": (ctypes.c_int, [ctypes.c_void_p]),
    "pivlfn_conv2d_nhwc": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int] + [ctypes.c_int] * 7 + [ctypes.c_void_p]),
    "pivlfn_conv_head_nhwc": (ctypes.c_int, [ctypes.c_void_p] * 4 + [ctypes.c_int] * 3 + [ctypes.c_void_p]),
    "pivlfn_profile_enable": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int]),
    "pivlfn_profile_read": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_long), ctypes.c_int]),
}

_lib = None


def load() -> ctypes.CDLL:
    """Load libpivlfn.so and attach prototypes.  Raises ImportError (never falls back) if it is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} is missing: build it with piv_liteflownet-pytorch_amd/csrc/build.sh "
                              "(or python -c 'import __graft_entry__ as g; g.build()'). There is no CPU fallback.")
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def check(rc: int, what: str = "") -> None:
    if rc == 0:
        return
    msg = load().pivlfn_last_error().decode(errors="replace")
    if rc in (1, 4):
        raise ValueError(f"{what}: {msg}")
    raise RuntimeError(f"{what}: {msg} (code {rc})")


def stream_ptr(device) -> int:
    import torch
    return torch.cuda.current_stream(device).cuda_stream
