"""CPU: the oracle against the committed golden vectors (which were produced by the reference itself)."""
import json
import os

import numpy as np
import torch

import pivlfn_oracle as orc
from pivlfn import synth

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def rel(a, b):
    return float(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).max() / max(1e-30, np.abs(b).max()))


def test_pin_report_says_bit_identical():
    rep = json.load(open(os.path.join(GOLD, "pin_report.json")))
    for k, v in rep.items():
        if k.startswith("e2e_"):
            assert v["bit_identical"] and v["oracle_vs_reference_rel"] == 0.0
            assert v["max_abs_flow_px"] > 1.0          # calibrated weights: the warps are not identities


def test_correlation_c_and_numpy_match_golden(gold):
    g = gold["corr_cases"]
    n = 0
    while f"f1_{n}" in g:
        f1, f2, s, want = g[f"f1_{n}"], g[f"f2_{n}"], int(g[f"stride_{n}"]), g[f"out_{n}"]
        assert np.array_equal(orc.correlation_c(f1, f2, s), want)          # same C code, same bits
        assert rel(orc.correlation_np(f1.astype(np.float64), f2.astype(np.float64), s), want) < 2e-6
        assert want.shape == (f1.shape[0], 49, -(-f1.shape[2] // s), -(-f1.shape[3] // s))
        n += 1
    assert n >= 6


def test_correlation_known_answers():
    # identical one-hot features: the centre displacement (channel 24) is 1/C at every pixel, borders are zero-padded
    f = np.zeros((1, 4, 5, 6), np.float32)
    f[0, 0] = 1.0
    out = orc.correlation_c(f, f, 1)
    assert np.allclose(out[0, 24], 0.25)
    assert out[0, 0, 0, 0] == 0.0 and np.isclose(out[0, 0, 3, 3], 0.25)     # (dy,dx)=(-3,-3)
    # a shifted copy peaks at the matching displacement: f2[y, x] = f1[y, x-2]  ->  dx = +2
    g = np.random.default_rng(0).standard_normal((1, 8, 9, 9)).astype(np.float32)
    g2 = np.roll(g, 2, axis=3)
    o = orc.correlation_np(g.astype(np.float64), g2.astype(np.float64), 1)
    assert int(o[0, :, 4, 4].argmax()) == 7 * 3 + (2 + 3)


def test_backwarp_matches_golden(gold):
    g = gold["backwarp_cases"]
    n = 0
    while f"x_{n}" in g:
        x, fl, want = g[f"x_{n}"], g[f"flow_{n}"], g[f"out_{n}"]
        got = orc.backwarp(torch.from_numpy(x), torch.from_numpy(fl)).numpy()
        assert np.array_equal(got, want)
        assert rel(orc.backwarp_np(x.astype(np.float64), fl.astype(np.float64)), want) < 2e-5
        assert rel(orc.backwarp_c(x, fl), want) < 2e-5
        n += 1
    assert n >= 3


def test_backwarp_zero_flow_is_identity_and_far_flow_is_zero():
    x = np.random.default_rng(1).standard_normal((1, 3, 7, 9)).astype(np.float32)
    assert np.allclose(orc.backwarp_c(x, np.zeros((1, 2, 7, 9), np.float32)), x)
    assert np.all(orc.backwarp_c(x, np.full((1, 2, 7, 9), 100.0, np.float32)) == 0)


def test_state_dict_spec_matches_reference_layout():
    for model in ("piv", "hui"):
        spec = json.load(open(os.path.join(GOLD, f"state_dict_spec_{model}.json")))
        mine = [[k, list(v)] for k, v in synth.state_dict_spec(model).items()]
        assert mine == spec
    assert sum(int(np.prod(s)) for _, s in json.load(open(os.path.join(GOLD, "state_dict_spec_piv.json")))) == 6249298


def _inputs(g, tag):
    i1 = torch.from_numpy(np.stack([synth.to_input(a) for a in g[f"{tag}_img1"]]))
    i2 = torch.from_numpy(np.stack([synth.to_input(a) for a in g[f"{tag}_img2"]]))
    return i1, i2


def test_oracle_end_to_end_matches_reference_flows(gold):
    g = gold["e2e_cases"]
    torch.set_num_threads(8)
    for model, tag in (("piv", "piv_1x64x64"), ("hui", "hui_1x64x64")):
        net = orc.make_net(model, synth.generate_weights(model, 0), corr="c")
        i1, i2 = _inputs(g, tag)
        with torch.no_grad():
            out, lv = net.forward(i1, i2, return_levels=True)
        want = g[f"{tag}_flow"]
        assert rel(out.numpy(), want) < 1e-5
        for j, trio in enumerate(lv):
            for name, t in zip("MSR", trio):
                assert rel(t.numpy(), g[f"{tag}_lv{j}_{name}"]) < 1e-4
        assert i1.min() >= 0.0                      # oracle does not mutate its inputs


def test_oracle_liteflownet2_matches_reference_flows():
    g = np.load(os.path.join(GOLD, "e2e_v2_cases.npz"))
    rep = json.load(open(os.path.join(GOLD, "pin_report_v2.json")))
    assert all(v["bit_identical"] for k, v in rep.items() if k.startswith("e2e_"))
    for model, tag in (("piv2", "piv2_1x64x64"), ("hui2", "hui2_1x64x96")):
        net = orc.make_net(model, synth.generate_weights(model, 0), corr="c")
        i1, i2 = _inputs(g, tag)
        with torch.no_grad():
            out = net.forward(i1, i2).numpy()
        assert out.shape == g[f"{tag}_flow"].shape
        assert rel(out, g[f"{tag}_flow"]) < 1e-5
    for model in ("piv2", "hui2"):
        spec = json.load(open(os.path.join(GOLD, f"state_dict_spec_{model}.json")))
        assert [[k, list(v)] for k, v in synth.state_dict_spec(model).items()] == spec


def test_oracle_estimate_non_multiple_of_32(gold):
    g = gold["e2e_cases"]
    net = orc.make_net("piv", synth.generate_weights("piv", 0), corr="c")
    i1, i2 = _inputs(g, "est_piv_100x76")
    out = orc.estimate(net, i1, i2, tensor=True).numpy()
    assert out.shape == (1, 2, 100, 76)
    assert rel(out, g["est_piv_100x76_flow"]) < 1e-5


def test_correlation_backward_restatement_matches_golden_and_autograd(gold):
    """Backward (src/correlation.py:106-234, 348-405): the loop-for-loop C restatement reproduces its fixture bit for bit and
    agrees with float64 autograd through the pinned forward restatement; off-grid gradients are exact zeros."""
    g = gold["corr_bwd_cases"]
    n = 0
    while f"f1_{n}" in g:
        f1, f2, go, s = g[f"f1_{n}"], g[f"f2_{n}"], g[f"go_{n}"], int(g[f"stride_{n}"])
        g1, g2 = orc.correlation_backward_c(f1, f2, go, s)
        assert np.array_equal(g1, g[f"g1_{n}"]) and np.array_equal(g2, g[f"g2_{n}"])
        a1, a2 = orc.correlation_backward_autograd(torch.from_numpy(f1).double(), torch.from_numpy(f2).double(),
                                                   torch.from_numpy(go).double(), s)
        assert np.abs(g1 - a1.numpy()).max() <= 1e-6 * np.abs(a1.numpy()).max()
        assert np.abs(g2 - a2.numpy()).max() <= 1e-6 * np.abs(a2.numpy()).max()
        if s > 1:
            off = np.ones(f1.shape[2:], bool)
            off[::s, ::s] = False
            assert not g1[:, :, off].any() and not g2[:, :, off].any()
        n += 1
    assert n >= 8


def test_correlation_backward_known_answer():
    """C=1, stride 1, a single unit gradient at displacement (dy,dx)=(1,-2), pixel (3,4): gradFirst[3,4] = second[4,2],
    gradSecond[4,2] = first[3,4], everything else zero."""
    f1 = np.arange(48, dtype=np.float32).reshape(1, 1, 6, 8) + 1
    f2 = -np.arange(48, dtype=np.float32).reshape(1, 1, 6, 8) - 100
    go = np.zeros((1, 49, 6, 8), np.float32)
    go[0, 7 * (1 + 3) + (-2 + 3), 3, 4] = 1.0
    g1, g2 = orc.correlation_backward_c(f1, f2, go, 1)
    w1, w2 = np.zeros_like(f1), np.zeros_like(f2)
    w1[0, 0, 3, 4] = f2[0, 0, 4, 2]
    w2[0, 0, 4, 2] = f1[0, 0, 3, 4]
    assert np.array_equal(g1, w1) and np.array_equal(g2, w2)


def test_reference_arithmetic_is_nan_when_a_level_collapses_to_one_pixel():
    """Documented difference: for inputs whose padded size is 32 in a dimension the level-6 maps are one
    pixel wide and the reference's backwarp divides the flow by (W - 1) / 2 = 0 (src/models.py:28-30): its output is NaN.  The
    oracle restates that arithmetic, so it is NaN too; the HIP path works in pixel units and returns finite flow
    (tests/test_gpu_net.py::test_degenerate_sizes_stay_finite)."""
    wts = synth.generate_weights("piv", 0)
    net = orc.make_net("piv", wts, corr="c")
    g = torch.Generator().manual_seed(1)
    i1, i2 = torch.rand(1, 3, 32, 64, generator=g), torch.rand(1, 3, 32, 64, generator=g)
    out = orc.estimate(net, i1, i2, tensor=True)
    assert out.shape == (1, 2, 32, 64) and not torch.isfinite(out).all()
