"""GPU: Winograd F(2x2, 3x3) with exactly split operands on the bf16 matrix cores (csrc/conv_wino_b3.hip) against a float64
convolution of the same fp32 data, next to the fp32-instruction Winograd kernel and the direct fp32 kernel on the same layer.
Replaces the 3x3 / stride 1 torch.nn.Conv2d call sites of /root/reference/src/models.py:77-101, 154-160, 197-204, 236-250 whose
output channels come in whole groups of 64.

What makes the kernel creditable as fp32 arithmetic (and what this file asserts):
  * every operand's 24 significand bits enter the products (three bf16 pieces, x = h + m + l exactly);
  * per-layer error against float64 <= the direct fp32-MFMA kernel's, mean and max, on every shape, the cancellation-heavy layer included;
  * no input domain narrower than fp32's: activations of 1e30 and of 1e-30, and 40 binades mixed inside one layer;
  * a sample's bits do not depend on its batch mates."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from pivlfn import _lib
from test_gpu_conv import Conv, run as run_direct
from test_gpu_wino import run_wino

pytestmark = pytest.mark.gpu


def run_b3(conv, x_nchw, leaky, dev, terms=6, x_lanes=None, y_lanes=None):
    B, C, H, W = x_nchw.shape
    co = conv.w.shape[0]
    xs = x_lanes or -(-C // 4) * 4
    x = torch.zeros(B, H, W, xs)
    x[..., :C] = x_nchw.permute(0, 2, 3, 1)
    x = x.to(dev)
    ys = y_lanes or -(-co // 4) * 4
    y = torch.full((B, H, W, ys), float("nan"), device=dev)
    _lib.check(_lib.load().pivlfn_conv2d_nhwc_wino_b3(conv.h, x.data_ptr(), xs, y.data_ptr(), ys, B, H, W, int(leaky), terms,
                                                     torch.cuda.current_stream(dev).cuda_stream), "conv2d_wino_b3")
    y = y.cpu()
    cs = min(-(-co // 4) * 4, ys)
    assert torch.all(y[..., co:cs] == 0)            # padding lanes are exact zeros
    if ys > cs:
        assert torch.isnan(y[..., cs:]).all()       # lanes beyond the stored ones are never touched
    return y[..., :co].permute(0, 3, 1, 2).contiguous()


CASES = [
    # cout, cin, H, W, B
    (128, 128, 32, 48, 1),       # conv_R.2
    (128, 49, 32, 64, 1),        # conv_M.0: 3 steps + a step with one real channel
    (64, 128, 19, 35, 2),        # odd sizes: half tiles at the right and bottom edges
    (64, 64, 1, 1, 1),           # a single pixel
    (64, 36, 5, 3, 1),           # 4-lane tail, image smaller than a tile
    (128, 386, 8, 8, 1),         # conv_S.0 level-6 width: 25 steps
    (57, 32, 8, 24, 2),          # cout not a multiple of 4: lanes 57..59 zero, channels 60..63 never stored
    (128, 128, 130, 70, 1),      # several workgroups in both directions, ragged
    (64, 64, 512, 250, 1),       # full-size launches: more workgroups than the chip holds at once
    (128, 16, 300, 310, 2),      # one K step
    (192, 40, 100, 90, 1),       # three channel groups
]


@pytest.mark.parametrize("terms", [6, 8, 9])
@pytest.mark.parametrize("case", CASES)
def test_b3_matches_float64_conv(case, terms, dev):
    """max-abs error <= 1e-5 of max |out| (the bar of test_gpu_wino.py), and the error against float64 is not above that of the kernel it
    replaces (Winograd on the fp32 matrix instruction), mean and max; on shapes of the network's regime (>= 64 x 64 outputs per image, where
    neither kernel is a split-K launch whose shorter chains round less) it is not above the direct fp32 kernel's either."""
    co, ci, H, W, B = case
    g = torch.Generator().manual_seed(co * 1000 + ci + H)
    w = torch.randn(co, ci, 3, 3, generator=g) / (ci * 9) ** 0.5
    b = torch.randn(co, generator=g) * 0.1
    x = torch.randn(B, ci, H, W, generator=g)
    conv = Conv(w, b)
    for leaky in (False, True):
        want = F.conv2d(x.double(), w.double(), b.double(), padding=1)
        if leaky:
            want = F.leaky_relu(want, 0.1)
        got = run_b3(conv, x, leaky, dev, terms)
        e = (got.double() - want).abs()
        scale = max(1.0, want.abs().max().item())
        assert e.max().item() < 1e-5 * scale, (case, terms, e.max().item())
        if not leaky:
            ew = (run_wino(conv, x, False, dev).double() - want).abs()
            assert e.mean().item() <= ew.mean().item() * 1.02 + 1e-12, (case, terms, e.mean().item(), ew.mean().item())
            assert e.max().item() <= ew.max().item() * 1.25 + 1e-12, (case, terms, e.max().item(), ew.max().item())
            if H * W >= 64 * 64:
                ed = (run_direct(conv, x, 1, (1, 1), False, dev).double() - want).abs()
                assert e.mean().item() <= ed.mean().item() * 1.02 + 1e-12, (case, terms, e.mean().item(), ed.mean().item())
                assert e.max().item() <= ed.max().item() * 1.25 + 1e-12, (case, terms, e.max().item(), ed.max().item())


def _errors(conv, x, want, dev):
    out = {"direct": run_direct(conv, x, 1, (1, 1), False, dev), "wino_fp32": run_wino(conv, x, False, dev)}
    for t in (6, 8, 9):
        out[f"b3_{t}"] = run_b3(conv, x, False, dev, t)
    return {k: (v.double() - want).abs() for k, v in out.items()}


def test_b3_error_beside_the_fp32_kernels(dev, capsys):
    """Random layer and the cancellation-heavy layer of test_gpu_wino.py (activations |N(0,1)| + 10 against zero-mean weights): the
    split kernel's error against float64, mean and max, is at or below the direct fp32 kernel's for 6, 8 and 9 piece products."""
    g = torch.Generator().manual_seed(11)
    w = torch.randn(128, 128, 3, 3, generator=g) / (128 * 9) ** 0.5
    b = torch.zeros(128)
    conv = Conv(w, b)
    for name, x in (("random", torch.randn(1, 128, 64, 96, generator=g)), ("cancellation", torch.randn(1, 128, 64, 96, generator=g).abs() + 10.0)):
        want = F.conv2d(x.double(), w.double(), padding=1)
        rms = want.pow(2).mean().sqrt().item()
        e = _errors(conv, x, want, dev)
        with capsys.disabled():
            print(f"\n128->128 3x3, {name}: rms(out) {rms:.3f}; error / rms  " +
                  "  ".join(f"{k}: max {v.max().item() / rms:.2e} mean {v.mean().item() / rms:.2e}" for k, v in e.items()))
        for t in (6, 8, 9):
            assert e[f"b3_{t}"].mean().item() <= e["direct"].mean().item(), (name, t)
            assert e[f"b3_{t}"].max().item() <= e["direct"].max().item() * 1.1, (name, t)


@pytest.mark.parametrize("scale", [1e30, 1e-30, 3e4, 1e5])
def test_b3_has_fp32s_input_domain(scale, dev):
    """Activations far outside fp16's range (the domain of round 2's fp16 splitting ended at 65504): same relative error."""
    g = torch.Generator().manual_seed(3)
    w = torch.randn(64, 64, 3, 3, generator=g) / (64 * 9) ** 0.5
    b = torch.zeros(64)
    x = torch.randn(1, 64, 40, 40, generator=g) * scale
    conv = Conv(w, b)
    want = F.conv2d(x.double(), w.double(), padding=1)
    got = run_b3(conv, x, False, dev, 6)
    assert torch.isfinite(got).all()
    assert (got.double() - want).abs().max().item() < 1e-5 * want.abs().max().item()


def test_b3_mixed_magnitudes_inside_one_layer(dev):
    """Channels whose magnitudes span 40 binades (2^-20 ... 2^20) with weights scaled the other way, so that every channel contributes
    equally to the output: the small channels' bits must survive (a piece scheme with a shared scale would lose them)."""
    g = torch.Generator().manual_seed(4)
    ci = 64
    s = torch.tensor([2.0 ** (k % 41 - 20) for k in range(ci)])
    w = torch.randn(64, ci, 3, 3, generator=g) / (ci * 9) ** 0.5 / s.view(1, ci, 1, 1)
    x = torch.randn(1, ci, 33, 47, generator=g) * s.view(1, ci, 1, 1)
    b = torch.zeros(64)
    conv = Conv(w, b)
    want = F.conv2d(x.double(), w.double(), padding=1)
    e6 = (run_b3(conv, x, False, dev, 6).double() - want).abs()
    ed = (run_direct(conv, x, 1, (1, 1), False, dev).double() - want).abs()
    assert e6.mean().item() <= ed.mean().item() * 1.02
    assert e6.max().item() < 1e-5 * want.abs().max().item()


def test_b3_wide_lanes_and_batch_invariance(dev):
    """Input living in a wider tensor, output into a wider tensor; a sample's bits do not depend on its batch mates."""
    g = torch.Generator().manual_seed(5)
    w = torch.randn(64, 20, 3, 3, generator=g) / (20 * 9) ** 0.5
    b = torch.randn(64, generator=g)
    x = torch.randn(3, 20, 37, 41, generator=g)
    conv = Conv(w, b)
    full = run_b3(conv, x, True, dev, 6, x_lanes=32, y_lanes=72)
    for i in range(3):
        one = run_b3(conv, x[i:i + 1], True, dev, 6)
        assert torch.equal(one[0], full[i])


def test_b3_refuses_layers_without_whole_channel_groups(dev):
    g = torch.Generator().manual_seed(6)
    conv = Conv(torch.randn(32, 32, 3, 3, generator=g), torch.zeros(32))
    with pytest.raises(ValueError, match="64-channel"):
        run_b3(conv, torch.randn(1, 32, 8, 8, generator=g), False, dev)
