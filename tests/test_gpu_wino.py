"""GPU: the Winograd F(2x2, 3x3) kernel of the default fp32 mode (csrc/conv_wino.hip) against a float64 convolution of the same
fp32 data, next to the direct kernel on the same layer.  Replaces the 3x3 / stride 1 torch.nn.Conv2d call sites of
/root/reference/src/models.py:77-101, 154-160, 197-204, 236-250.  Tolerance (fp32): max-abs <= 1e-5 * max|out| per layer, the
bar the direct kernel is held to in test_gpu_conv.py (2e-5); the measured errors are printed."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from pivlfn import _lib
from test_gpu_conv import Conv, run as run_direct

pytestmark = pytest.mark.gpu


def _tools():
    """tools/libpivlfn_tools.so: the research kernels of tools/kernels/ (F(4x4), the wave-specialised kernel) are compiled into it only."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import _toolslib
    return _toolslib.load()


class ToolsConv:
    """A layer object of the tools library (its handles are not the production library's: other fields, F(4x4) weights packed)."""
    def __init__(self, w, b):
        import ctypes
        self.w, self.b = w.contiguous(), b.contiguous()
        self.h = ctypes.c_void_p()
        co, ci, kh, kw = w.shape
        assert _tools().pivlfn_conv_create(self.w.data_ptr(), self.b.data_ptr(), co, ci, kh, kw, ctypes.byref(self.h)) == 0

    def __del__(self):
        _tools().pivlfn_conv_destroy(self.h)


def run_wino(conv, x_nchw, leaky, dev, x_lanes=None, y_lanes=None, tile=2):
    B, C, H, W = x_nchw.shape
    co = conv.w.shape[0]
    xs = x_lanes or -(-C // 4) * 4
    x = torch.zeros(B, H, W, xs)
    x[..., :C] = x_nchw.permute(0, 2, 3, 1)
    x = x.to(dev)
    ys = y_lanes or -(-co // 4) * 4
    y = torch.full((B, H, W, ys), float("nan"), device=dev)
    if tile == 2:
        _lib.check(_lib.load().pivlfn_conv2d_nhwc_wino(conv.h, x.data_ptr(), xs, y.data_ptr(), ys, B, H, W, int(leaky), torch.cuda.current_stream(dev).cuda_stream), "conv2d_wino")
    else:
        assert isinstance(conv, ToolsConv)
        rc = _tools().pivlfn_conv2d_nhwc_wino4(conv.h, x.data_ptr(), xs, y.data_ptr(), ys, B, H, W, int(leaky), torch.cuda.current_stream(dev).cuda_stream)
        assert rc == 0, _tools().pivlfn_last_error()
    y = y.cpu()
    cs = min(-(-co // 4) * 4, ys)
    assert torch.all(y[..., co:cs] == 0)            # padding lanes are exact zeros
    if ys > cs:
        assert torch.isnan(y[..., cs:]).all()       # lanes beyond the stored ones are never touched
    return y[..., :co].permute(0, 3, 1, 2).contiguous()


CASES = [
    # cout, cin, H, W, B
    (128, 128, 32, 48, 1),       # conv_R.2
    (128, 49, 32, 64, 1),        # conv_M.0: 6 chunks + a 4-lane tail (one real channel in it)
    (64, 128, 19, 35, 2),        # odd sizes: half tiles at the right and bottom edges
    (32, 64, 16, 16, 3),
    (32, 32, 64, 64, 1),         # NetC.conv2.x
    (96, 96, 24, 40, 1),         # NetC.conv4.2: three N blocks
    (64, 64, 1, 1, 1),           # a single pixel
    (32, 36, 5, 3, 1),           # 4-lane tail, image smaller than a tile
    (128, 386, 8, 8, 1),         # conv_S.0 level-6 width
    (9, 32, 8, 24, 2),           # cout not a multiple of 4: lanes 9..11 zero
    (128, 128, 130, 70, 1),      # several workgroups in both directions, ragged
    # full-size launches (more workgroups than the chip holds at once, two co-resident per CU), one per tile shape
    (64, 64, 512, 250, 1),       # two channel blocks per wave (1, 2): 64 x 16 = 1024 workgroups
    (32, 32, 380, 384, 4),       # 16-row tiles (2, 1): 4 x 24 x 24 = 2304 workgroups
    (32, 16, 320, 320, 1),       # 8-row tiles, one block (1, 1): 800 workgroups
    (96, 40, 300, 310, 1),       # odd number of channel blocks, 4-lane tail: (2, 1) x 3 groups
]


FULL_OCCUPANCY = [c for c in CASES if c[2] * c[3] * c[4] >= 100000]


@pytest.mark.parametrize("case", CASES + [c + (1,) for c in FULL_OCCUPANCY])      # the full-occupancy shapes under a second seed
def test_wino_matches_float64_conv(case, dev):
    co, ci, H, W, B = case[:5]
    g = torch.Generator().manual_seed(co * 1000 + ci + H + (7919 * case[5] if len(case) > 5 else 0))
    w = torch.randn(co, ci, 3, 3, generator=g) / (ci * 9) ** 0.5
    b = torch.randn(co, generator=g) * 0.1
    x = torch.randn(B, ci, H, W, generator=g)
    conv = Conv(w, b)
    for leaky in (False, True):
        want = F.conv2d(x.double(), w.double(), b.double(), padding=1)
        if leaky:
            want = F.leaky_relu(want, 0.1)
        got = run_wino(conv, x, leaky, dev)
        err = (got.double() - want).abs().max().item()
        assert err < 1e-5 * max(1.0, want.abs().max().item()), (case, err)


@pytest.mark.parametrize("case", CASES)
def test_wino4_matches_float64_conv(case, dev):
    """The F(4x4, 3x3) kernel (tools/kernels/conv_wino4.hip, compiled into the tools library only since round 6: no user path launches
    it) on the same shapes:
    its transforms multiply by 2, 4, 5 and 8, so a layer's error against float64 is ~8e-6 of max |out| where F(2x2) and the direct
    kernel stay below 5e-7 (measured on the CPU restatement before the kernel existed: DESIGN.md 4.2c); the per-layer bound here is
    3e-5, and what decides is the end-to-end bound every oracle test holds the network to (1e-4 of the flow scale)."""
    co, ci, H, W, B = case[:5]
    g = torch.Generator().manual_seed(co * 1000 + ci + H + 17)
    w = torch.randn(co, ci, 3, 3, generator=g) / (ci * 9) ** 0.5
    b = torch.randn(co, generator=g) * 0.1
    x = torch.randn(B, ci, H, W, generator=g)
    conv = ToolsConv(w, b)
    for leaky in (False, True):
        want = F.conv2d(x.double(), w.double(), b.double(), padding=1)
        if leaky:
            want = F.leaky_relu(want, 0.1)
        got = run_wino(conv, x, leaky, dev, tile=4)
        err = (got.double() - want).abs().max().item()
        assert err < 3e-5 * max(1.0, want.abs().max().item()), (case, err)


def test_wino_error_beside_the_direct_kernel(dev):
    """A cancellation-heavy layer -- post-LeakyReLU-like activations with a positive mean (|N(0,1)| + 10) against zero-mean
    weights, error normalised by the RMS of the output (not its maximum): Winograd in fp32 stays within 2x the direct fp32
    kernel's error against float64."""
    g = torch.Generator().manual_seed(11)
    w = torch.randn(128, 128, 3, 3, generator=g) / (128 * 9) ** 0.5
    b = torch.zeros(128)
    x = torch.randn(1, 128, 64, 96, generator=g).abs() + 10.0
    conv = Conv(w, b)
    want = F.conv2d(x.double(), w.double(), padding=1)
    rms = want.pow(2).mean().sqrt().item()
    ew = (run_wino(conv, x, False, dev).double() - want).abs()
    ed = (run_direct(conv, x, 1, (1, 1), False, dev).double() - want).abs()
    print(f"128->128 3x3, x = |N|+10: rms(out) {rms:.3f}; Winograd max {ew.max().item() / rms:.2e} mean {ew.mean().item() / rms:.2e}; "
          f"direct max {ed.max().item() / rms:.2e} mean {ed.mean().item() / rms:.2e} (of rms)")
    assert ew.mean().item() <= 2.0 * ed.mean().item() + 1e-9
    assert ew.max().item() <= 1e-5 * rms * 10        # absolute bar: 1e-4 of the output's rms on a layer whose inputs are 10x its outputs


def test_wino_wide_lanes_and_batch_invariance(dev):
    """Input living in a wider tensor, output into a wider tensor; a sample's bits do not depend on its batch mates or on the tile
    shape the launch picks (8-row blocks for small launches, 16-row blocks for large ones)."""
    g = torch.Generator().manual_seed(5)
    w = torch.randn(64, 32, 3, 3, generator=g) / 17
    b = torch.randn(64, generator=g)
    conv = Conv(w, b)
    x = torch.randn(40, 32, 64, 64, generator=g)          # 40 x 4 x 4 x 2 = 1280 16-row blocks -> MB = 2
    full = run_wino(conv, x, True, dev, x_lanes=40, y_lanes=72)
    one = run_wino(conv, x[7:8], True, dev, x_lanes=40, y_lanes=72)      # 32 blocks -> MB = 1
    assert torch.equal(one[0], full[7])
    want = F.leaky_relu(F.conv2d(x[7:8].double(), w.double(), b.double(), padding=1), 0.1)
    assert (one.double() - want).abs().max().item() < 1e-5 * want.abs().max().item()


def test_wino_refuses_what_it_does_not_cover(dev):
    g = torch.Generator().manual_seed(1)
    conv = Conv(torch.randn(32, 32, 1, 1, generator=g), torch.zeros(32))
    x = torch.zeros(1, 8, 8, 32, device=dev)
    y = torch.zeros(1, 8, 8, 32, device=dev)
    rc = _lib.load().pivlfn_conv2d_nhwc_wino(conv.h, x.data_ptr(), 32, y.data_ptr(), 32, 1, 8, 8, 0, torch.cuda.current_stream(dev).cuda_stream)
    assert rc != 0 and b"3 x 3" in _lib.load().pivlfn_last_error()


def test_default_mode_is_fp32_with_winograd(dev):
    """The network's default precision is 'fp32'; it differs from 'fp32_direct' only in summation order (both meet the oracle
    tolerance in test_gpu_net.py), and both reproduce themselves bit for bit."""
    import pivlfn
    from pivlfn import synth
    net = pivlfn.piv_liteflownet(synth.generate_weights("piv", 0)).to(dev).eval()
    assert net.precision == "fp32"
    a, b, _ = synth.particle_pair(256, 256, 77)
    x1, x2 = torch.from_numpy(synth.to_input(a))[None].to(dev), torch.from_numpy(synth.to_input(b))[None].to(dev)
    fw = net(x1, x2)
    net.precision = "fp32_direct"
    fd = net(x1, x2)
    net.precision = "fp32"
    assert torch.equal(net(x1, x2), fw)
    assert not torch.equal(fw, fd)
    d = (fw - fd).abs().max().item()
    print(f"piv 256x256: max |flow(fp32, Winograd) - flow(fp32_direct)| = {d:.2e} px at max |flow| {fd.abs().max().item():.2f}")
    assert d < 1e-4 * max(1.0, fd.abs().max().item())


# ---- several sources: torch.cat + Conv2d through the fp32 dispatch (pivlfn_conv2d_nhwc_cat) -------------------------------------------
# The front layers of Matching / Subpixel / Regularization read a channel concatenation (src/models.py:171-187, 209-217, 280).  The
# kernels stage it source by source: per-source descriptors, a source switch inside the staging loop, a 4-lane tail that only the
# last source may have.  Sizes: below 64 x 64 the direct kernel runs, from there up the Winograd kernel, at 256 x 272 with every CU
# holding two workgroups.
CAT_CASES = [
    # cout, channels per source, lanes per source (None = channels rounded up to 4), H, W, B
    (128, (64, 64, 2), (64, 72, 4), 32, 40, 1),            # conv_S.0 of levels 1-3 below the Winograd bound: direct kernel, a wider second source
    (128, (64, 64, 2), None, 80, 96, 1),                   # conv_S.0 on the Winograd kernel: three sources, 4-lane tail in the last
    (128, (128, 3), (128, 4), 72, 64, 2),                  # conv_R.0 (moduleFeat + 3 flow channels), batch 2
    (128, (96, 96, 2), None, 67, 83, 1),                   # conv_S.0 of level 4 (C = 96): sources that are not multiples of 64, ragged size
    (128, (64, 64, 2), None, 256, 272, 1),                 # full occupancy
    (128, (128, 3), (136, 4), 261, 280, 1),                # full occupancy, ragged, the first source inside a wider tensor
]


@pytest.mark.parametrize("case", CAT_CASES)
def test_concatenated_sources_match_float64_conv(case, dev):
    import ctypes
    co, chans, lanes, H, W, B = case
    lib = _lib.load()
    g = torch.Generator().manual_seed(co + sum(chans) + H)
    cin = sum(chans)
    w = (torch.randn(co, cin, 3, 3, generator=g) / (cin * 9) ** 0.5).contiguous()
    b = (torch.randn(co, generator=g) * 0.1).contiguous()
    srcs = [torch.randn(B, c, H, W, generator=g) for c in chans]
    want = F.leaky_relu(F.conv2d(torch.cat(srcs, 1).double(), w.double(), b.double(), padding=1), 0.1)
    h = ctypes.c_void_p()
    ch = (ctypes.c_int * len(chans))(*chans)
    _lib.check(lib.pivlfn_conv_create_cat(w.data_ptr(), b.data_ptr(), co, len(chans), ch, 3, 3, ctypes.byref(h)), "conv_create_cat")
    try:
        lanes = lanes or tuple(-(-c // 4) * 4 for c in chans)
        dev_srcs = []
        for s_, c, l in zip(srcs, chans, lanes):
            t = torch.zeros(B, H, W, l)
            t[..., :c] = s_.permute(0, 2, 3, 1)
            if l > -(-c // 4) * 4:
                t[..., -(-c // 4) * 4:] = float("nan")          # lanes beyond the source's own are never read
            dev_srcs.append(t.to(dev))
        ptrs = (ctypes.c_void_p * len(chans))(*[t.data_ptr() for t in dev_srcs])
        strides = (ctypes.c_int * len(chans))(*lanes)
        y = torch.full((B, H, W, co), float("nan"), device=dev)
        _lib.check(lib.pivlfn_conv2d_nhwc_cat(h, len(chans), ptrs, strides, y.data_ptr(), co, B, H, W, 1,
                                              torch.cuda.current_stream(dev).cuda_stream), "conv2d_cat")
        got = y.cpu().permute(0, 3, 1, 2).double()
        err = (got - want).abs().max().item()
        print(f"cat {chans} -> {co} at {H}x{W} B={B}: max abs err {err:.2e} at max |out| {want.abs().max().item():.2f}")
        assert err < 1e-5 * max(1.0, want.abs().max().item()), (case, err)
    finally:
        lib.pivlfn_conv_destroy(h)


WS_CASES = [
    # cout, cin (per source), H, W, B -- the shapes the bench never covered: ragged tiles, several images, 4-channel tail chunks, an odd
    # number of channel blocks, two and three sources
    (64, (64,), 37, 45, 2),
    (128, (36,), 70, 130, 1),          # cin % 8 == 4
    (96, (40,), 33, 47, 3),            # three channel blocks: one block per wave
    (128, (64, 64, 4), 50, 66, 1),     # conv_S.0's three sources
    (128, (128, 4), 41, 39, 2),        # conv_R.0's two sources
    (32, (32,), 128, 160, 1),
]


@pytest.mark.parametrize("case", WS_CASES)
def test_wave_specialised_kernel_is_bit_identical(case, dev):
    """tools/kernels/conv_wino_ws.hip (round 5: persistent workgroups, producer / consumer waves; slower than conv_wino.hip, kept in the
    tools library for A/B runs) returns the shipped kernel's bits -- same packed weights, same summation order -- on ragged images,
    several images, 4-channel tail chunks, odd channel-block counts and multi-source layers."""
    import ctypes
    lib = _tools()
    co, cins, H, W, B = case
    g = torch.Generator().manual_seed(co + sum(cins) + H)
    ci = sum(cins)
    w = (torch.randn(co, ci, 3, 3, generator=g) / (ci * 9) ** 0.5).contiguous()
    b = torch.randn(co, generator=g).contiguous()
    h = ctypes.c_void_p()
    ch = (ctypes.c_int * len(cins))(*cins)
    assert lib.pivlfn_conv_create_cat(w.data_ptr(), b.data_ptr(), co, len(cins), ch, 3, 3, ctypes.byref(h)) == 0
    xs = [torch.randn(B, H, W, -(-c // 4) * 4, generator=g) for c in cins]
    for x, c in zip(xs, cins):
        x[..., c:] = 0
    xd = [x.to(dev) for x in xs]
    ptrs = (ctypes.c_void_p * len(cins))(*[x.data_ptr() for x in xd])
    strides = (ctypes.c_int * len(cins))(*[x.shape[-1] for x in xd])
    outs = []
    for knob in (0, 31):
        lib.pivlfn_tune(12, 1)          # Winograd from one output pixel up (the network's dispatch starts it at 64 x 64)
        lib.pivlfn_tune(14, knob)
        y = torch.full((B, H, W, co), float("nan"), device=dev)
        rc = lib.pivlfn_conv2d_nhwc_cat(h, len(cins), ptrs, strides, y.data_ptr(), co, B, H, W, 1, torch.cuda.current_stream(dev).cuda_stream)
        lib.pivlfn_tune(14, 0)
        lib.pivlfn_tune(12, 0)
        assert rc == 0, lib.pivlfn_last_error()
        outs.append(y.cpu())
    lib.pivlfn_conv_destroy(h)
    assert torch.isfinite(outs[0]).all()
    assert torch.equal(outs[0], outs[1])
