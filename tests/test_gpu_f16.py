"""GPU: the optional fp16-multiplicand conv mode (BASELINE config #5; csrc/conv_f16.hip, pivlfn_set_precision).

Two kinds of statement:
  * the kernel is EXACT up to fp32 summation order once the operand rounding is taken out: against a float64 convolution of the
    fp16-rounded inputs and weights, max-abs <= 2e-5 * max|out| (fp32 output) -- the same bar as the fp32 kernel;
  * the mode's end-to-end effect on the flow is bounded: end-point error against the fp32 mode (mean <= 0.05 px, the bound
    SURVEY section 8c states for this variant) and the fp32 oracle.
"""
import ctypes

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import pivlfn
import pivlfn_oracle as orc
from pivlfn import _lib, synth

pytestmark = pytest.mark.gpu

# cout, cin, kh, kw, stride, B, H, W
LAYERS = [(128, 128, 3, 3, 1, 2, 256, 256),    # <4,4> 16-row tiles, one workgroup per CU
          (128, 128, 3, 3, 1, 1, 64, 96),      # <2,4>
          (64, 128, 3, 3, 1, 2, 256, 256),     # <4,2>
          (64, 64, 3, 3, 1, 1, 72, 100),       # <2,2>, ragged edges
          (32, 64, 3, 3, 1, 4, 256, 256),      # <4,1>
          (32, 32, 3, 3, 1, 1, 37, 45),        # <2,1>, ragged
          (128, 130, 3, 3, 1, 1, 64, 64),      # K not a multiple of 16 (the 130-channel Subpixel input)
          (32, 3, 7, 7, 1, 1, 96, 128),        # NetC.conv1
          (64, 32, 3, 3, 2, 1, 128, 160),      # stride 2
          (192, 128, 3, 3, 2, 1, 64, 64),      # stride 2, cout 192 -> two 64-wide... (cout_pad 192: nt=2)
          (64, 32, 1, 1, 1, 1, 128, 128),      # 1x1 (NetC_ext)
          (49, 32, 7, 1, 1, 1, 96, 96),        # separable k x 1, cout 49 -> stored as 52 lanes
          (49, 49, 1, 7, 1, 1, 96, 96),        # 1 x k with a 52-lane source
          (25, 32, 5, 1, 1, 1, 64, 64)]


def _ref(x16, w16, b, stride, kh, kw, leaky):
    y = F.conv2d(x16.double(), w16.double(), b.double(), stride=stride, padding=(kh // 2, kw // 2))
    return F.leaky_relu(y, 0.1) if leaky else y


@pytest.mark.parametrize("layer", LAYERS, ids=[f"{c[0]}x{c[1]}k{c[2]}x{c[3]}s{c[4]}_{c[5]}x{c[6]}x{c[7]}" for c in LAYERS])
@pytest.mark.parametrize("x_f16,y_f16", [(0, 0), (1, 1)])
def test_conv_f16_kernel_is_exact_up_to_summation_order(layer, x_f16, y_f16, dev):
    co, ci, kh, kw, s, B, H, W = layer
    g = torch.Generator().manual_seed(co * 1000 + ci * 10 + kh + s + H)
    w = (torch.randn(co, ci, kh, kw, generator=g) / (ci * kh * kw) ** 0.5).contiguous()
    b = torch.randn(co, generator=g).contiguous()
    gran = 8 if x_f16 else 4
    xs = -(-ci // gran) * gran
    x = torch.randn(B, H, W, xs, generator=g)
    x[..., ci:] = 0.0
    lib = _lib.load()
    h = ctypes.c_void_p()
    _lib.check(lib.pivlfn_conv_create(w.data_ptr(), b.data_ptr(), co, ci, kh, kw, ctypes.byref(h)), "create")
    ys = -(-co // 4) * 4
    xd = x.to(dev).half() if x_f16 else x.to(dev)
    y = torch.full((B, (H + 2 * (kh // 2) - kh) // s + 1, (W + 2 * (kw // 2) - kw) // s + 1, ys), float("nan"),
                   device=dev, dtype=torch.float16 if y_f16 else torch.float32)
    st = torch.cuda.current_stream(dev).cuda_stream
    _lib.check(lib.pivlfn_conv2d_nhwc_f16(h, xd.data_ptr(), xs, x_f16, y.data_ptr(), ys, y_f16, B, H, W, s, kh // 2, kw // 2, 1, st), "conv")
    torch.cuda.synchronize()
    lib.pivlfn_conv_destroy(h)
    x16 = x[..., :ci].half().float().permute(0, 3, 1, 2)
    want = _ref(x16, w.half().float(), b, s, kh, kw, True).permute(0, 2, 3, 1)
    got = y.float().cpu()
    assert torch.isfinite(got).all()
    if ys > co:
        assert (got[..., co:] == 0).all()                      # padding lanes are exact zeros
    err = (got[..., :co].double() - want).abs().max().item() / want.abs().max().item()
    assert err < (1e-3 if y_f16 else 2e-5), err                 # fp16 output adds its own 2^-11 rounding


def _epe(a, b):
    return np.sqrt(((np.asarray(a, np.float64) - np.asarray(b, np.float64)) ** 2).sum(1))


@pytest.mark.parametrize("model,size", [("piv", (256, 256)), ("hui", (256, 320)), ("piv", (512, 512))])
def test_fp16_mode_end_point_error(model, size, dev):
    H, W = size
    a, b, _ = synth.particle_pair(H, W, 77)
    i1 = torch.from_numpy(synth.to_input(a))[None].to(dev)
    i2 = torch.from_numpy(synth.to_input(b))[None].to(dev)
    net = pivlfn.Network(model=model, params=synth.generate_weights(model, 0)).to(dev).eval()
    net.precision = "fp32"
    f32 = net(i1, i2).cpu().numpy()
    net.precision = "fp16"
    assert net.precision == "fp16"
    f16 = net(i1, i2).cpu().numpy()
    net.precision = "fp32"
    again = net(i1, i2).cpu().numpy()
    assert np.array_equal(again, f32)                           # switching back restores the fp32 path bit for bit
    assert not np.array_equal(f16, f32)                         # the mode really ran
    e = _epe(f16, f32)
    print(f"{model} {H}x{W}: fp16-mode EPE vs fp32 mode: mean {e.mean():.2e} px, max {e.max():.2e} px, max|flow| {np.abs(f32).max():.2f}")
    assert e.mean() <= 0.05 and e.max() <= 0.5


def test_fp16_mode_against_the_oracle(gold, dev):
    """96x160 golden case (reference flows): level 1 is large enough to run on the fp16 kernel."""
    g = gold["e2e_cases"]
    tag = "piv_2x96x160"
    i1 = torch.from_numpy(np.stack([synth.to_input(a) for a in g[f"{tag}_img1"]])).to(dev)
    i2 = torch.from_numpy(np.stack([synth.to_input(a) for a in g[f"{tag}_img2"]])).to(dev)
    want = g[f"{tag}_flow"]
    net = pivlfn.piv_liteflownet(synth.generate_weights("piv", 0)).to(dev).eval()
    net.precision = "fp16"
    got = net(i1, i2).cpu().numpy()
    e = _epe(got, want)
    print(f"fp16 mode vs reference flows: EPE mean {e.mean():.2e} max {e.max():.2e} at max|flow| {np.abs(want).max():.2f}")
    assert e.mean() <= 0.05


def test_precision_argument_is_validated(dev):
    net = pivlfn.piv_liteflownet(synth.generate_weights("piv", 0)).to(dev).eval()
    with pytest.raises(ValueError):
        net.precision = "bf16"
