"""CPU: the C-ABI library loads and exports every symbol include/pivlfn.h declares (no compute calls)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "pivlfn.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"#ifdef PIVLFN_TOOLS.*?#endif", "", text, flags=re.S)      # tools-build-only declarations are not the boundary
    return sorted(set(re.findall(r"\b(pivlfn_[a-z_0-9]+)\s*\(", text)))


def test_header_declares_the_boundary():
    names = _declared()
    for must in ("pivlfn_corr_fwd", "pivlfn_backwarp", "pivlfn_warp_corr_fwd", "pivlfn_create", "pivlfn_destroy",
                 "pivlfn_workspace_bytes", "pivlfn_forward", "pivlfn_last_error", "pivlfn_resize_bilinear"):
        assert must in names


def test_library_exports_every_declared_symbol():
    from pivlfn import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as ge
        ge.build()
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in _declared():
        assert hasattr(lib, name), f"{name} declared in include/pivlfn.h but not exported"
    assert set(_declared()) == set(_lib.SIGNATURES), "ctypes prototypes out of sync with the header"
    assert not hasattr(lib, "pivlfn_tune"), "the production library must not export the tools-only knob setter"
    loaded = _lib.load()
    assert loaded.pivlfn_abi_version() == 3
    assert loaded.pivlfn_last_error() is not None


def test_argument_errors_are_reported_without_a_gpu():
    from pivlfn import _lib
    lib = _lib.load()
    # null pointers / bad shapes are rejected on the host before any launch
    assert lib.pivlfn_corr_fwd(None, None, None, 1, 8, 4, 4, 1, None) != 0
    assert b"null" in lib.pivlfn_last_error()
    assert lib.pivlfn_workspace_bytes(None, 1, 64, 64) == 0
    with pytest.raises(ValueError):
        _lib.check(lib.pivlfn_backwarp(None, None, None, 1, 1, 1, 1, None), "backwarp")
