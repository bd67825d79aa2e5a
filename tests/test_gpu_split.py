"""GPU: fp32 convolution on the fp16 matrix cores by operand splitting (csrc/conv_split.hip): precision 'fp32_split3' (the
default: two fp16 pieces per operand, three partial products) and 'fp32_split' (three pieces, six partial products).

The statements, for both:
  * per layer, against a float64 convolution of the SAME fp32 inputs and weights, the split kernel's error is of the size of the
    fp32-instruction kernel's (fp32 summation-order noise; the split adds < 2^-32 / < 2^-21 per product) -- the bar is the fp32
    kernel's own bar (2e-5 of max|out|), the mean error may not exceed twice the fp32 instruction's (measured: 0.6 x), and the
    two are printed next to each other;
  * the accuracy does not depend on the magnitudes: weights of 1e-6 and 1e+3, activations of 1e-4 and 1e+3, mixed magnitudes
    inside one layer (fp16 pieces would underflow / overflow if the scales were not handled);
  * end to end the mode meets the fp32 tolerance against the oracle and the reference's golden flows (the same 1e-4 / 1e-5 bars
    as the fp32 instruction path).
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

# cout, cin, kh, kw, B, H, W
LAYERS = [(128, 128, 3, 3, 1, 128, 128),   # <2,2> twice over the channels
          (128, 128, 3, 3, 2, 272, 1024),  # enough 16-row tiles for the one-workgroup-per-CU kernel (three terms), ragged in y
          (64, 132, 3, 3, 1, 544, 512),    # the same with 64 channels and a 4-channel K tail
          (256, 96, 3, 3, 1, 256, 544),    # two 128-channel blocks
          (128, 100, 3, 3, 1, 530, 500),   # the same kernel with ragged tiles in x and y and a 4-channel K tail
          (64, 128, 3, 3, 2, 64, 96),
          (64, 64, 3, 3, 1, 72, 100),      # ragged edges
          (32, 64, 3, 3, 1, 64, 64),       # <2,1>
          (32, 32, 3, 3, 1, 37, 45),       # ragged
          (128, 49, 3, 3, 1, 64, 64),      # conv_M.0: 49 channels in 52 lanes, K padded to 64
          (128, 130, 3, 3, 1, 64, 64),     # 130 channels: 8 full chunks + a chunk of 2 (+2 zero lanes)
          (64, 32, 1, 1, 1, 128, 128),     # 1x1 (NetC_ext)
          (128, 32, 1, 1, 1, 64, 64),      # 1x1 (moduleFeat)
          (49, 32, 7, 1, 1, 96, 96),       # separable k x 1, cout 49 -> 52 lanes
          (49, 49, 1, 7, 1, 96, 96),       # 1 x k with a 52-lane source
          (25, 32, 5, 5, 1, 64, 64),       # 5 x 5 (conv_dist_R.0 of the Hui layout)
          (9, 32, 3, 3, 1, 64, 64)]


def _run(lib, h, fn, x, xs, co, B, H, W, kh, kw, dev, terms=6):
    ys = -(-co // 4) * 4
    y = torch.full((B, H, W, ys), float("nan"), device=dev)
    st = torch.cuda.current_stream(dev).cuda_stream
    if fn == "split":
        _lib.check(lib.pivlfn_conv2d_nhwc_split(h, x.data_ptr(), xs, y.data_ptr(), ys, B, H, W, 1, kh // 2, kw // 2, 1, terms, st), "split")
    else:
        _lib.check(lib.pivlfn_conv2d_nhwc(h, x.data_ptr(), xs, y.data_ptr(), ys, None, 0, B, H, W, 1, kh // 2, kw // 2, 1, st), "fp32")
    torch.cuda.synchronize()
    return y.cpu()


def _layer_errors(layer, dev, wscale=1.0, xscale=1.0, mixed=False, terms=6):
    co, ci, kh, kw, B, H, W = layer
    g = torch.Generator().manual_seed(co * 1000 + ci * 10 + kh + H)
    w = (torch.randn(co, ci, kh, kw, generator=g) / (ci * kh * kw) ** 0.5 * wscale).contiguous()
    b = (torch.randn(co, generator=g) * wscale * xscale).contiguous()
    xs = -(-ci // 4) * 4
    x = torch.randn(B, H, W, xs, generator=g) * xscale
    if mixed:                                   # magnitudes spread over 12 binades inside one layer
        w = (w * torch.exp2(torch.randint(-12, 1, w.shape, generator=g).float())).contiguous()
        x = x * torch.exp2(torch.randint(-12, 1, x.shape, generator=g).float())
    x[..., ci:] = 0.0
    lib = _lib.load()
    h = ctypes.c_void_p()
    _lib.check(lib.pivlfn_conv_create(w.data_ptr(), b.data_ptr(), co, ci, kh, kw, ctypes.byref(h)), "create")
    xd = x.to(dev)
    got_s = _run(lib, h, "split", xd, xs, co, B, H, W, kh, kw, dev, terms)
    got_f = _run(lib, h, "fp32", xd, xs, co, B, H, W, kh, kw, dev)
    lib.pivlfn_conv_destroy(h)
    want = F.leaky_relu(F.conv2d(x[..., :ci].permute(0, 3, 1, 2).double(), w.double(), b.double(), padding=(kh // 2, kw // 2)), 0.1).permute(0, 2, 3, 1)
    assert torch.isfinite(got_s).all()
    ys = got_s.shape[-1]
    if ys > co:
        assert (got_s[..., co:] == 0).all()                     # padding lanes are exact zeros
    scale = want.abs().max().item()
    es = (got_s[..., :co].double() - want).abs()
    ef = (got_f[..., :co].double() - want).abs()
    return es.max().item() / scale, ef.max().item() / scale, es.mean().item() / scale, ef.mean().item() / scale


@pytest.mark.parametrize("terms", [6, 3])
@pytest.mark.parametrize("layer", LAYERS, ids=[f"{c[0]}x{c[1]}k{c[2]}x{c[3]}_{c[4]}x{c[5]}x{c[6]}" for c in LAYERS])
def test_split_kernel_has_the_fp32_kernels_error(layer, terms, dev):
    es, ef, ms, mf = _layer_errors(layer, dev, terms=terms)
    print(f"{terms} terms, max-abs error / max|out| against float64: split {es:.2e}, fp32 instruction {ef:.2e}; mean {ms:.2e} vs {mf:.2e}")
    assert es < 2e-5, es                                        # the fp32 kernel's own bar (tests/test_gpu_conv.py)
    assert ms < 2.0 * mf + 1e-9, (ms, mf)                       # and on average no worse than twice the fp32 instruction's


@pytest.mark.parametrize("terms", [6, 3])
@pytest.mark.parametrize("wscale,xscale", [(1e-6, 1.0), (1e3, 1.0), (1.0, 1e-4), (1.0, 1e3), (1e-5, 1e3), (3e2, 3e-3)])
def test_split_kernel_accuracy_is_scale_free(wscale, xscale, terms, dev):
    es, ef, ms, mf = _layer_errors((64, 128, 3, 3, 1, 64, 96), dev, wscale, xscale, terms=terms)
    print(f"{terms} terms, w x {wscale:g}, x x {xscale:g}: split {es:.2e} (mean {ms:.2e}), fp32 instruction {ef:.2e} (mean {mf:.2e})")
    assert es < 2e-5 and ms < 2.0 * mf + 1e-9


@pytest.mark.parametrize("terms", [6, 3])
def test_split_kernel_mixed_magnitudes(terms, dev):
    es, ef, ms, mf = _layer_errors((64, 128, 3, 3, 1, 64, 96), dev, mixed=True, terms=terms)
    print(f"{terms} terms, 12 binades of magnitudes in one layer: split {es:.2e} (mean {ms:.2e}), fp32 instruction {ef:.2e} (mean {mf:.2e})")
    assert es < 2e-5 and ms < 2.0 * mf + 1e-9


@pytest.mark.parametrize("layer", [(32, 32, 1, 128, 160), (64, 32, 2, 96, 128), (64, 64, 1, 70, 90)], ids=["32x32", "64x32", "64x64ragged"])
def test_three_term_kernel_stride_2(layer, dev):
    """3 x 3 stride-2 layers (NetC.conv2.0 / conv3.0) on the three-term kernel's 4-row tiles, against float64."""
    co, ci, B, H, W = layer
    g = torch.Generator().manual_seed(co + ci + H)
    w = (torch.randn(co, ci, 3, 3, generator=g) / (ci * 9) ** 0.5).contiguous()
    b = torch.randn(co, generator=g).contiguous()
    x = torch.randn(B, H, W, ci, generator=g)
    lib = _lib.load()
    h = ctypes.c_void_p()
    _lib.check(lib.pivlfn_conv_create(w.data_ptr(), b.data_ptr(), co, ci, 3, 3, ctypes.byref(h)), "create")
    Ho, Wo = (H + 2 - 3) // 2 + 1, (W + 2 - 3) // 2 + 1
    xd = x.to(dev)
    st = torch.cuda.current_stream(dev).cuda_stream
    ys, yf = torch.full((B, Ho, Wo, co), float("nan"), device=dev), torch.full((B, Ho, Wo, co), float("nan"), device=dev)
    _lib.check(lib.pivlfn_conv2d_nhwc_split(h, xd.data_ptr(), ci, ys.data_ptr(), co, B, H, W, 2, 1, 1, 1, 3, st), "split s2")
    _lib.check(lib.pivlfn_conv2d_nhwc(h, xd.data_ptr(), ci, yf.data_ptr(), co, None, 0, B, H, W, 2, 1, 1, 1, st), "fp32 s2")
    torch.cuda.synchronize()
    lib.pivlfn_conv_destroy(h)
    want = F.leaky_relu(F.conv2d(x.permute(0, 3, 1, 2).double(), w.double(), b.double(), stride=2, padding=1), 0.1).permute(0, 2, 3, 1)
    scale = want.abs().max().item()
    es, ef = (ys.cpu().double() - want).abs(), (yf.cpu().double() - want).abs()
    print(f"stride 2: split max {es.max().item() / scale:.2e} mean {es.mean().item() / scale:.2e}; fp32 instruction max {ef.max().item() / scale:.2e} mean {ef.mean().item() / scale:.2e}")
    assert es.max().item() / scale < 2e-5 and es.mean().item() < 2.0 * ef.mean().item() + 1e-9 * scale


def test_split_kernel_rejects_strided_layers(dev):
    lib = _lib.load()
    w, b = torch.randn(32, 32, 3, 3).contiguous(), torch.zeros(32)
    h = ctypes.c_void_p()
    _lib.check(lib.pivlfn_conv_create(w.data_ptr(), b.data_ptr(), 32, 32, 3, 3, ctypes.byref(h)), "create")
    x = torch.zeros(1, 64, 64, 32, device=dev)
    y = torch.zeros(1, 32, 32, 32, device=dev)
    rc = lib.pivlfn_conv2d_nhwc_split(h, x.data_ptr(), 32, y.data_ptr(), 32, 1, 64, 64, 2, 1, 1, 1, 6, torch.cuda.current_stream(dev).cuda_stream)
    lib.pivlfn_conv_destroy(h)
    assert rc != 0 and b"stride" in lib.pivlfn_last_error()


E2E_MAX, E2E_MEAN = 1e-4, 1e-5          # the fp32 end-to-end tolerance, relative to max(1, max|flow|)


def _check(got, want, what):
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    scale = max(1.0, np.abs(want).max())
    err = np.abs(got - want)
    print(f"{what}: max-abs {err.max():.2e}, mean-abs {err.mean():.2e} px at flow scale {scale:.2f}")
    assert err.max() <= E2E_MAX * scale, f"{what}: max-abs {err.max():.3e} at flow scale {scale:.2f}"
    assert err.mean() <= E2E_MEAN * scale, f"{what}: mean-abs {err.mean():.3e}"


def test_split_modes_are_opt_in(dev):
    """The default is the fp32 instruction ('fp32'); the split modes have to be asked for."""
    net = pivlfn.piv_liteflownet(synth.generate_weights("piv", 0)).to(dev).eval()
    assert net.precision == "fp32"


@pytest.mark.parametrize("mode", ["fp32_split3", "fp32_split"])
@pytest.mark.parametrize("model,size", [("piv", (256, 256)), ("hui", (256, 320)), ("piv", (512, 384)), ("piv", (512, 512))])
def test_split_modes_meet_the_fp32_tolerance_against_the_oracle(model, size, mode, dev):
    """Level 1 (and level 2 of the 512 x 512 pair) run on the split kernel; the bars are those of the fp32 instruction path."""
    H, W = size
    a, b, _ = synth.particle_pair(H, W, 1234 + H)
    x1, x2 = torch.from_numpy(synth.to_input(a))[None], torch.from_numpy(synth.to_input(b))[None]
    net = pivlfn.Network(model=model, params=synth.generate_weights(model, 0)).to(dev).eval()
    net.precision = "fp32"
    f32 = net(x1.to(dev), x2.to(dev)).cpu().numpy()
    net.precision = mode
    assert net.precision == mode
    got = net(x1.to(dev), x2.to(dev)).cpu().numpy()
    assert not np.array_equal(got, f32)                         # the mode really ran (summation orders differ)
    onet = orc.make_net(model, synth.generate_weights(model, 0), corr="c")
    with torch.no_grad():
        want = onet.forward(x1, x2).numpy()
    _check(f32, want, f"{model} {H}x{W} fp32 instruction vs oracle")
    _check(got, want, f"{model} {H}x{W} {mode} vs oracle")
    net.precision = "fp32"
    assert np.array_equal(net(x1.to(dev), x2.to(dev)).cpu().numpy(), f32)     # switching back restores the bits


def test_golden_case_small_grids(gold, dev):
    """Golden 96 x 160 case (flows produced by the reference itself).  No level has 256 x 256 outputs, so the six-term mode is the
    fp32 instruction path bit for bit; the three-term mode runs level 1 (96 x 160 >= 64 x 64) on its 4-row tiles and meets the
    same bars against the reference's flows."""
    g = gold["e2e_cases"]
    tag = "piv_2x96x160"
    i1 = torch.from_numpy(np.stack([synth.to_input(a) for a in g[f"{tag}_img1"]])).to(dev)
    i2 = torch.from_numpy(np.stack([synth.to_input(a) for a in g[f"{tag}_img2"]])).to(dev)
    net = pivlfn.Network(model="piv", params=synth.generate_weights("piv", 0)).to(dev).eval()
    outs = {}
    for mode in ("fp32_direct", "fp32_split", "fp32_split3"):
        net.precision = mode
        outs[mode] = net(i1, i2).cpu().numpy()
    assert np.array_equal(outs["fp32_direct"], outs["fp32_split"]) and not np.array_equal(outs["fp32_direct"], outs["fp32_split3"])
    _check(outs["fp32_direct"], g[f"{tag}_flow"], f"{tag} fp32 instruction (direct) vs reference flows")
    _check(outs["fp32_split3"], g[f"{tag}_flow"], f"{tag} three-term split vs reference flows")


@pytest.mark.parametrize("size", [(64, 64), (128, 160), (192, 256)])
def test_three_term_mode_small_levels_split_k(size, dev):
    """Coarse levels on the three-term kernel's split-K path (64 x 64 ... 128 x 128 grids): against the oracle, and the same bits
    alone and in a batch of three."""
    H, W = size
    a, b = synth.particle_batch(3, H * 2, W * 2, seed=31 + H)
    i1, i2 = torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev)
    net = pivlfn.piv_liteflownet(synth.generate_weights("piv", 0)).to(dev).eval()
    net.precision = "fp32_split3"
    full = net(i1, i2)
    for k in range(3):
        assert torch.equal(net(i1[k:k + 1], i2[k:k + 1])[0], full[k])
    onet = orc.make_net("piv", synth.generate_weights("piv", 0), corr="c")
    with torch.no_grad():
        want = onet.forward(torch.from_numpy(a[:1]), torch.from_numpy(b[:1])).numpy()
    _check(full[:1].cpu().numpy(), want, f"piv {2 * H}x{2 * W} three-term split (levels down to {H // 8}x{W // 8} on split-K) vs oracle")


@pytest.mark.parametrize("mode", ["fp32_split3", "fp32_split"])
def test_split_mode_batch_consistency(mode, dev):
    """A pair's flow is the same bits alone and inside a batch (the kernel choice is made per image, never from the batch)."""
    a, b = synth.particle_batch(3, 256, 256, seed=5)
    i1, i2 = torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev)
    net = pivlfn.piv_liteflownet(synth.generate_weights("piv", 0)).to(dev).eval()
    net.precision = mode
    full = net(i1, i2)
    for k in range(3):
        assert torch.equal(net(i1[k:k + 1], i2[k:k + 1])[0], full[k])


# ---- exactness and size-independent properties ---------------------------------------------------------------------------
def _conv_split(w, b, x, terms, dev, leaky=0):
    co, ci, kh, kw = w.shape
    B, H, W, xs = x.shape
    lib = _lib.load()
    h = ctypes.c_void_p()
    _lib.check(lib.pivlfn_conv_create(w.contiguous().data_ptr(), b.contiguous().data_ptr(), co, ci, kh, kw, ctypes.byref(h)), "create")
    ys = -(-co // 4) * 4
    y = torch.full((B, H, W, ys), float("nan"), device=dev)
    xd = x.to(dev)
    _lib.check(lib.pivlfn_conv2d_nhwc_split(h, xd.data_ptr(), xs, y.data_ptr(), ys, B, H, W, 1, kh // 2, kw // 2, leaky, terms,
                                            torch.cuda.current_stream(dev).cuda_stream), "split")
    torch.cuda.synchronize()
    lib.pivlfn_conv_destroy(h)
    return y.cpu()


def test_six_term_identity_layer_returns_its_fp32_input_bit_for_bit(dev):
    """x = h + m 2^-11 + l 2^-22 exactly for 2^-14 <= |x| < 65504: a 1 x 1 layer with the identity matrix hands back every such fp32
    input bit -- 24 binades of magnitudes, both signs, zeros.  The three-term form returns h + m 2^-11: within 2^-21 of x.  Below
    2^-14 (fp16's subnormal range, which the conversion flushes) the leading piece is zero and the remaining ones carry 22 (11) bits:
    the absolute deviation stays below 2^-37 (2^-26)."""
    C = 64
    g = torch.Generator().manual_seed(11)
    x = torch.randn(1, 64, 96, C, generator=g).clamp(-3.9, 3.9)
    x = torch.where(x.abs() < 0.25, torch.full_like(x, 0.5), x) * torch.exp2(torch.randint(-12, 13, (1, 64, 96, C), generator=g).float())
    x[0, :4] = 0.0
    assert float(x.abs().max()) < 65504 and float(x[0, 4:].abs().min()) >= 2.0 ** -14
    w = torch.eye(C).reshape(C, C, 1, 1)
    b = torch.zeros(C)
    y6 = _conv_split(w, b, x, 6, dev)
    assert torch.equal(y6, x)
    y3 = _conv_split(w, b, x, 3, dev)
    rel = ((y3 - x).abs() / x.abs().clamp_min(1e-30)).max().item()
    print(f"identity layer, three terms: max relative deviation {rel:.2e} (bound 2^-21 = 4.8e-7)")
    assert rel <= 2.0 ** -21
    tiny = torch.randn(1, 64, 96, C, generator=g) * torch.exp2(torch.randint(-30, -14, (1, 64, 96, C), generator=g).float())
    tiny = tiny.clamp(-2.0 ** -14 * 0.999, 2.0 ** -14 * 0.999)
    d6 = (_conv_split(w, b, tiny, 6, dev) - tiny).abs().max().item()
    d3 = (_conv_split(w, b, tiny, 3, dev) - tiny).abs().max().item()
    print(f"inputs below 2^-14: max absolute deviation six terms {d6:.2e} (2^-37 = 7.3e-12), three terms {d3:.2e} (2^-26 = 1.5e-8)")
    assert d6 <= 2.0 ** -37 and d3 <= 2.0 ** -26


@pytest.mark.parametrize("terms", [6, 3])
def test_split_kernel_properties_at_full_size(terms, dev):
    """1024 x 1024, 128 -> 128, 3 x 3 (the forward's dominant layer; the 16-row kernel in the three-term form) through properties that
    need no reference of that size: zero weights give the bias; an input shifted by a tile-unaligned offset gives the shifted output
    in the interior, bit for bit (every tile computes every pixel the same way); scaling the input by 2 scales the pre-bias output by 2
    exactly."""
    C, S = 128, 1024
    g = torch.Generator().manual_seed(21)
    w = (torch.randn(C, C, 3, 3, generator=g) / (C * 9) ** 0.5)
    b = torch.randn(C, generator=g)
    x = torch.randn(1, S, S, C, generator=g)
    x = torch.where(x.abs() < 1e-3, torch.full_like(x, 1e-3), x)       # away from the fp16 flush threshold (2^-14): see the identity test
    y0 = _conv_split(torch.zeros_like(w), b, x, terms, dev)
    assert torch.equal(y0, b.expand_as(y0))
    zero_b = torch.zeros(C)
    y = _conv_split(w, zero_b, x, terms, dev)
    assert torch.isfinite(y).all()
    dy, dx = 5, 37
    xs = torch.zeros_like(x)
    xs[:, dy:, dx:] = x[:, :S - dy, :S - dx]
    ysft = _conv_split(w, zero_b, xs, terms, dev)
    assert torch.equal(ysft[:, dy + 1:S - 1, dx + 1:S - 1], y[:, 1:S - dy - 1, 1:S - dx - 1])     # (the last row / column see the zero padding)
    y2 = _conv_split(w, zero_b, x * 2.0, terms, dev)
    assert torch.equal(y2, y * 2.0)


def test_split_modes_outside_the_fp16_range_give_nan_not_a_wrong_number(dev):
    """The split kernels' leading piece is an fp16: an input of magnitude >= 65504 overflows it and the affected outputs are NaN /
    non-finite (include/pivlfn.h) -- never a silently saturated, plausible-looking value; the fp32 kernels (direct and Winograd)
    take the same layer in their stride.  The split modes are opt-in for this reason (the library default is 'fp32')."""
    g = torch.Generator().manual_seed(9)
    w = torch.randn(64, 32, 3, 3, generator=g) / 17
    b = torch.zeros(64)
    x = torch.randn(1, 64, 64, 32, generator=g)
    x[0, 20, 30, 5] = 1.0e5                                   # one out-of-range activation
    for terms in (3, 6):
        y = _conv_split(w, b, x, terms, dev)
        hit = y[0, 19:22, 29:32, :]                           # its 3 x 3 receptive neighbourhood
        assert not torch.isfinite(hit).all(), f"{terms}-term split returned finite values for an input outside the fp16 range"
        far = y[0, :10, :10, :]
        assert torch.isfinite(far).all()
    from test_gpu_wino import run_wino
    from test_gpu_conv import Conv
    yw = run_wino(Conv(w, b), x.permute(0, 3, 1, 2).contiguous(), False, dev)
    want = torch.nn.functional.conv2d(x.permute(0, 3, 1, 2).double(), w.double(), padding=1)
    assert torch.isfinite(yw).all() and (yw.double() - want).abs().max().item() < 1e-5 * want.abs().max().item()
