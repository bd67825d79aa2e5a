"""GPU: pivlfn_forward (the whole level pipeline as HIP kernels) against golden flows of the reference and the oracle."""
import os

import numpy as np
import pytest
import torch

import pivlfn
import pivlfn_oracle as orc
from pivlfn import synth

pytestmark = pytest.mark.gpu

# End-to-end fp32 tolerance: max-abs <= 1e-4 * max(1, max|flow|) px, and the mean
# absolute error an order of magnitude below that.  The fp32-vs-fp64 noise floor of the reference itself on
# these cases is 2e-6 .. 1.3e-5 px max-abs (tests/golden/pin_report.json).
E2E_MAX = 1e-4
E2E_MEAN = 1e-5


def _inputs(g, tag, dev):
    i1 = torch.from_numpy(np.stack([synth.to_input(a) for a in g[f"{tag}_img1"]])).to(dev)
    i2 = torch.from_numpy(np.stack([synth.to_input(a) for a in g[f"{tag}_img2"]])).to(dev)
    return i1, i2


def _check(got, want, what):
    got = np.asarray(got, np.float64)
    want = np.asarray(want, np.float64)
    scale = max(1.0, np.abs(want).max())
    err = np.abs(got - want)
    assert err.max() <= E2E_MAX * scale, f"{what}: max-abs {err.max():.3e} vs flow scale {scale:.2f}"
    assert err.mean() <= E2E_MEAN * scale, f"{what}: mean-abs {err.mean():.3e}"


@pytest.fixture(scope="module")
def nets(dev):
    out = {}
    for model in ("piv", "hui"):
        n = pivlfn.Network(model=model, params=synth.generate_weights(model, 0)).to(dev)
        n.eval()
        out[model] = n
    return out


@pytest.mark.parametrize("tag", ["piv_1x64x64", "piv_2x96x160", "hui_1x64x64", "hui_2x96x160"])
def test_forward_matches_reference_golden(tag, gold, nets, dev):
    g = gold["e2e_cases"]
    model = tag[:3]
    i1, i2 = _inputs(g, tag, dev)
    keep = i1.clone()
    flow, levels = nets[model].forward_levels(i1, i2)
    assert torch.equal(i1, keep)                              # inputs are not mutated
    want = g[f"{tag}_flow"]
    assert tuple(flow.shape) == want.shape
    # localise first: per-level M, S, R flows (training-mode return of the reference)
    for j, trio in enumerate(levels):
        for name, t in zip("MSR", trio):
            _check(t.cpu().numpy(), g[f"{tag}_lv{j}_{name}"], f"{tag} level#{j} {name}")
    _check(flow.cpu().numpy(), want, tag)
    again = nets[model](i1, i2)
    assert torch.equal(again, flow)                           # run-to-run bitwise determinism


def test_estimate_non_multiple_of_32(gold, nets, dev):
    g = gold["e2e_cases"]
    i1, i2 = _inputs(g, "est_piv_100x76", dev)
    out = pivlfn.estimate(nets["piv"], i1, i2, tensor=True)
    assert tuple(out.shape) == (1, 2, 100, 76)
    _check(out.cpu().numpy(), g["est_piv_100x76_flow"], "estimate 100x76")
    arr = pivlfn.estimate(nets["piv"], i1, i2, tensor=False)
    assert arr.shape == (100, 76, 2) and arr.dtype == np.float32
    assert np.array_equal(arr, out[0].permute(1, 2, 0).cpu().numpy())


def test_estimate_hui_upsamples_half_resolution_flow(gold, nets, dev):
    g = gold["e2e_cases"]
    i1, i2 = _inputs(g, "hui_1x64x64", dev)
    out = pivlfn.estimate(nets["hui"], i1, i2, tensor=True).cpu()
    raw = torch.from_numpy(g["hui_1x64x64_flow"])
    want = torch.nn.functional.interpolate(raw, size=(64, 64), mode="bilinear", align_corners=False)
    _check(out.numpy(), want.numpy(), "estimate hui 64x64")


def test_forward_vs_oracle_fresh_seed(nets, dev):
    """A case that is NOT in the fixtures: 128x96, batch 2, new frames, oracle run here on the CPU."""
    a, b = synth.particle_batch(2, 128, 96, seed=777)
    i1, i2 = torch.from_numpy(a), torch.from_numpy(b)
    for model in ("piv", "hui"):
        onet = orc.make_net(model, synth.generate_weights(model, 0), corr="c")
        with torch.no_grad():
            want = onet.forward(i1, i2).numpy()
        got = nets[model](i1.to(dev), i2.to(dev)).cpu().numpy()
        _check(got, want, f"{model} 2x128x96")


def test_forward_vs_oracle_mid_size(nets, dev):
    """512 x 448, one pair, oracle run here on the CPU (~2 s): the size from which levels 1 and 2 launch every tile shape of the
    Winograd kernel at full occupancy (two workgroups per CU, several generations) and the 7 x 1 / 7 x 7 layers run on their
    streaming matrix-core kernels -- the small fixtures never get there.  Same tolerance as every end-to-end case."""
    a, b = synth.particle_batch(1, 512, 448, seed=4242)
    i1, i2 = torch.from_numpy(a), torch.from_numpy(b)
    onet = orc.make_net("piv", synth.generate_weights("piv", 0), corr="c")
    with torch.no_grad():
        want = onet.forward(i1, i2).numpy()
    got = nets["piv"](i1.to(dev), i2.to(dev)).cpu().numpy()
    _check(got, want, "piv 1x512x448")


def test_batch_consistency_and_argument_errors(nets, dev):
    a, b = synth.particle_batch(3, 64, 96, seed=31)
    i1, i2 = torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev)
    full = nets["piv"](i1, i2)
    for k in range(3):
        one = nets["piv"](i1[k:k + 1], i2[k:k + 1])
        assert torch.equal(one[0], full[k])                   # every pair is independent of its batch mates
    # (bit-for-bit as long as the batch does not move a level's warp+correlation launch between the latency and the throughput
    #  variant of the kernel -- the conv split-K factor never depends on the batch; the next test bounds the other case)
    with pytest.raises(ValueError):
        nets["piv"](i1[:, :, :50], i2[:, :, :50])            # not a multiple of 32 -> estimate() territory
    with pytest.raises(ValueError):
        nets["piv"](i1, i2[:2])
    nets["piv"].train()
    with pytest.raises(NotImplementedError):
        nets["piv"](i1, i2)
    nets["piv"].eval()


def test_batch_consistency_across_kernel_variants(nets, dev):
    """A batch large enough to move the warp+correlation launches of several levels from the latency kernel (a whole CU per
    tile) to the throughput kernel: both use one summation order (two fma chains per displacement over the interleaved
    16-channel halves), so every pair still equals its single-pair result bit for bit."""
    a, b = synth.particle_batch(2, 256, 256, seed=32)
    i1, i2 = torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev)
    single = torch.cat([nets["piv"](i1[k:k + 1], i2[k:k + 1]) for k in range(2)])
    B = 24
    big = nets["piv"](torch.stack([i1[k % 2] for k in range(B)]), torch.stack([i2[k % 2] for k in range(B)]))
    for k in range(B):
        assert torch.equal(big[k], single[k % 2]), f"pair {k} of the batch differs from its single-pair flow"


def test_forward_is_graph_capturable(nets, dev):
    """No allocation and no host synchronisation inside pivlfn_forward, and the side stream joins through events: the whole
    forward can be captured into a HIP graph and replayed (measured on MI355X: same speed as eager launches -- the path is bound by
    the kernels, not by launch overhead -- so bench.py does not use a graph)."""
    a, b = synth.particle_batch(1, 128, 160, seed=41)
    i1, i2 = torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev)
    net = nets["piv"]
    ref = net(i1, i2)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        net(i1, i2)                                   # workspace allocated outside the capture
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = net(i1, i2)
    i1.copy_(torch.from_numpy(b).to(dev))             # new inputs in the captured buffers
    i2.copy_(torch.from_numpy(a).to(dev))
    g.replay()
    torch.cuda.synchronize()
    swapped = out.clone()
    i1.copy_(torch.from_numpy(a).to(dev))
    i2.copy_(torch.from_numpy(b).to(dev))
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, ref)
    assert torch.equal(swapped, net(torch.from_numpy(b).to(dev), torch.from_numpy(a).to(dev)))


def test_degenerate_sizes_stay_finite(nets, dev):
    """Sizes for which the reference itself returns NaN (a level-6 map one pixel wide: tests/test_oracle.py): finite here, and
    the right shape."""
    for H, W in [(32, 32), (20, 28), (64, 32), (1, 1)]:
        g = torch.Generator().manual_seed(H * 100 + W)
        i1, i2 = torch.rand(1, 3, H, W, generator=g).to(dev), torch.rand(1, 3, H, W, generator=g).to(dev)
        out = pivlfn.estimate(nets["piv"], i1, i2, tensor=True)
        assert out.shape == (1, 2, H, W) and torch.isfinite(out).all()


def test_reloading_weights_takes_effect(dev):
    net = pivlfn.piv_liteflownet(synth.generate_weights("piv", 0)).to(dev).eval()
    a, b = synth.particle_batch(1, 64, 64, seed=5)
    i1, i2 = torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev)
    f0 = net(i1, i2)
    net.load_state_dict({k: v.to(dev) for k, v in synth.generate_weights("piv", 1).items()})
    f1 = net(i1, i2)
    assert not torch.equal(f0, f1)
    net.load_state_dict(synth.generate_weights("piv", 0))
    assert torch.equal(net(i1, i2), f0)


def test_full_size_1024_properties(nets, dev):
    """BASELINE config #2 size: no oracle run (12 s on CPU); size-independent properties instead."""
    a, b = synth.particle_batch(1, 1024, 1024, seed=1234)
    i1, i2 = torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev)
    f = nets["piv"](i1, i2)
    assert tuple(f.shape) == (1, 2, 1024, 1024) and torch.isfinite(f).all()
    assert torch.equal(nets["piv"](i1, i2), f)                # deterministic
    # translation covariance away from the borders: the top-left 512x512 crop sees the same data inside its
    # receptive field only near the crop centre; check the crop's interior agrees to fp32 noise.
    fc = nets["piv"](i1[:, :, :512, :512].contiguous(), i2[:, :, :512, :512].contiguous())
    d = (fc[:, :, 192:320, 192:320] - f[:, :, 192:320, 192:320]).abs().max().item()
    assert d < 0.5, d                                         # far inside: border effects decay, fields stay close
    assert f.abs().max().item() > 1.0


def test_full_size_1024_vs_oracle(nets, dev):
    """BASELINE configs[1] itself against the CPU oracle (one pair, ~10 s of CPU): the headline size takes the level-1 / level-2
    sliding-window warp+correlation launches, the one-tile-per-CU launch of level 3 and every full-occupancy convolution shape
    inside one forward; same tolerance as the golden cases."""
    a, b = synth.particle_batch(1, 1024, 1024, seed=1234)
    i1, i2 = torch.from_numpy(a), torch.from_numpy(b)
    onet = orc.make_net("piv", synth.generate_weights("piv", 0), corr="c")
    with torch.no_grad():
        want = onet.forward(i1, i2).numpy()
    got = nets["piv"](i1.to(dev), i2.to(dev)).cpu().numpy()
    assert np.abs(want).max() > 1.0
    _check(got, want, "piv 1x1024x1024 vs oracle")


def test_run_py_end_to_end(tmp_path, dev):
    """run.py counterpart: a 6-frame sequence (5 pairs, 3 batches through the copy-stream pipeline) -> .flo files equal to
    estimate() on the same pairs."""
    import PIL.Image
    import run as runpy
    from pivlfn.flo import read_flow
    seq = tmp_path / "seq"
    seq.mkdir()
    frames = []
    for k in range(6):
        a, _, _ = synth.particle_pair(64, 96, 400 + k)
        frames.append(a)
        PIL.Image.fromarray(a).save(str(seq / f"frame_{k:04d}.png"))
    out = tmp_path / "out"
    n = runpy.main(["--model", "piv", "-i", str(seq), "-o", str(out), "--batch", "2"])
    assert n == 5
    flodir = out / "piv-synthetic" / "seq" / "flow"
    assert (out / "piv-synthetic" / "seq" / "args.txt").exists()
    net = pivlfn.piv_liteflownet(synth.generate_weights("piv", 0)).to(dev).eval()
    for k in range(5):
        got = read_flow(str(flodir / f"frame_{k:04d}_out.flo"))
        i1 = torch.from_numpy(synth.to_input(frames[k]))[None].to(dev)
        i2 = torch.from_numpy(synth.to_input(frames[k + 1]))[None].to(dev)
        want = pivlfn.estimate(net, i1, i2, tensor=False)
        assert got.shape == (64, 96, 2) and np.array_equal(got, want)


def test_run_py_version2_pair_folder(tmp_path, dev):
    """run.py -p (pair folder, `*_img1/_img2` naming, src/datasets.py:447-455) with --version 2 --model hui: the v2 Hui
    network returns quarter-resolution flow, which estimate() resizes to the input size; .flo equals estimate() bit for bit."""
    import PIL.Image
    import run as runpy
    from pivlfn.flo import read_flow
    d = tmp_path / "left"
    d.mkdir()
    pairs = []
    for k in range(2):
        a, b, _ = synth.particle_pair(96, 64, 500 + k)
        PIL.Image.fromarray(a).save(str(d / f"shot{k}_img1.png"))
        PIL.Image.fromarray(b).save(str(d / f"shot{k}_img2.png"))
        pairs.append((a, b))
    out = tmp_path / "out"
    n = runpy.main(["--model", "hui", "-v", "2", "-p", "-i", str(d), "-o", str(out), "--batch", "2"])
    assert n == 2
    # a folder called left/right goes under flow/left of its parent's name (run.py:241-249)
    flodir = out / "hui2-synthetic" / os.path.basename(str(tmp_path)) / "flow" / "left"
    assert (out / "hui2-synthetic" / os.path.basename(str(tmp_path)) / "args_left.txt").exists()
    net = pivlfn.Network(model="hui", params=synth.generate_weights("hui2", 0), version=2).to(dev).eval()
    for k, (a, b) in enumerate(pairs):
        got = read_flow(str(flodir / f"shot{k}_out.flo"))
        want = pivlfn.estimate(net, torch.from_numpy(synth.to_input(a))[None].to(dev), torch.from_numpy(synth.to_input(b))[None].to(dev), tensor=False)
        assert got.shape == (96, 64, 2) and np.array_equal(got, want)


@pytest.mark.parametrize("tag", ["piv2_1x64x64", "piv2_2x96x160", "hui2_1x64x96"])
def test_liteflownet2_matches_reference_golden(tag, dev):
    """`--version 2` (LiteFlowNet2 backbones, src/models.py:373-716) against flows of the reference itself."""
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "e2e_v2_cases.npz"))
    model = tag[:4]
    net = pivlfn.Network(model=model[:3], params=synth.generate_weights(model, 0), version=2).to(dev).eval()
    i1, i2 = _inputs(g, tag, dev)
    flow, levels = net.forward_levels(i1, i2)
    want = g[f"{tag}_flow"]
    assert tuple(flow.shape) == want.shape                      # piv2: half resolution, hui2: quarter resolution
    for j, trio in enumerate(levels):
        for name, t in zip("MSR", trio):
            _check(t.cpu().numpy(), g[f"{tag}_lv{j}_{name}"], f"{tag} level#{j} {name}")
    _check(flow.cpu().numpy(), want, tag)
    full = pivlfn.estimate(net, i1, i2, tensor=True)
    assert tuple(full.shape) == (i1.shape[0], 2, i1.shape[2], i1.shape[3])


def test_replacing_a_parameter_object_takes_effect(dev):
    """The native handle is keyed on a cached parameter list; assigning a NEW Parameter object must invalidate it."""
    net = pivlfn.piv_liteflownet(synth.generate_weights("piv", 0)).to(dev).eval()
    a, b = synth.particle_batch(1, 64, 64, seed=8)
    i1, i2 = torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev)
    before = net(i1, i2).clone()
    mod = net.NetC.conv1._modules["0"]
    mod.weight = torch.nn.Parameter(mod.weight.detach() * 0.5, requires_grad=False)
    after = net(i1, i2)
    assert not torch.equal(before, after)


def test_swapping_parameter_data_takes_effect(dev):
    """`p.data = new_tensor` on a LATE parameter (no _version bump, a new storage address): the native weights must be rebuilt."""
    net = pivlfn.piv_liteflownet(synth.generate_weights("piv", 0)).to(dev).eval()
    a, b = synth.particle_batch(1, 64, 64, seed=9)
    i1, i2 = torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev)
    before = net(i1, i2).clone()
    late = list(net.parameters())[-8]
    late.data = (late.data * 0.25).clone()
    after = net(i1, i2)
    assert not torch.equal(before, after)
