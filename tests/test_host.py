"""CPU: host-side mirror of the reference interface (names, state dict, error behaviour), synthetic generators."""
import json
import os

import numpy as np
import pytest
import torch

import pivlfn
from pivlfn import synth

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_public_names_match_reference_surface():
    import src.models as sm
    import src.correlation as sc
    import inference
    assert sm.__all__[:2] == ["hui_liteflownet", "piv_liteflownet"]
    for n in ("LiteFlowNet", "backwarp", "piv_liteflownet", "hui_liteflownet", "Network"):
        assert hasattr(sm, n)
    assert hasattr(sc, "FunctionCorrelation") and hasattr(sc, "ModuleCorrelation")
    assert inference.estimate is pivlfn.estimate


@pytest.mark.parametrize("model", ["piv", "hui", "piv2", "hui2"])
def test_state_dict_roundtrip_strict(model):
    spec = json.load(open(os.path.join(GOLD, f"state_dict_spec_{model}.json")))
    wts = synth.generate_weights(model, seed=3)
    net = pivlfn.Network(model=model[:3], params=wts, version=2 if model.endswith("2") else 1)
    sd = net.state_dict()
    assert [[k, list(v.shape)] for k, v in sd.items()] == spec
    for k in wts:
        assert torch.equal(sd[k], wts[k])
    bad = dict(wts)
    bad.pop(next(iter(bad)))
    with pytest.raises(RuntimeError):
        net.load_state_dict(bad)                       # strict, like the reference's load_state_dict


def test_factories_configure_like_reference():
    p, h = pivlfn.piv_liteflownet(), pivlfn.hui_liteflownet()
    assert p.lowest_level == 1 and h.lowest_level == 2
    assert p.SCALEFACTOR[1] == 5.0 and h.SCALEFACTOR[1] == 20.0 and p.SCALEFACTOR[6] == 10.0 / 64
    assert p.MEAN[0] == [0.173935, 0.180594, 0.192608] and h.MEAN[1] == [0.410782, 0.433645, 0.452793]
    with pytest.raises(ValueError):
        pivlfn.piv_liteflownet(version=3)
    p2, h2 = pivlfn.piv_liteflownet(version=2), pivlfn.hui_liteflownet(version=2)      # src/models.py:731-732, 756-758
    assert isinstance(p2, pivlfn.LiteFlowNet2) and p2.lowest_level == 2 and h2.lowest_level == 3
    assert p2.SCALEFACTOR[1] == 5.0 and p2.MEAN[0] == [0.194286, 0.190633, 0.191766]
    with pytest.raises(ValueError):
        pivlfn.Network(model="foo")


def test_no_cpu_fallback_anywhere():
    a = torch.zeros(1, 8, 4, 4)
    with pytest.raises(NotImplementedError):
        pivlfn.FunctionCorrelation(a, a, 1)            # src/correlation.py:339-340
    with pytest.raises(AssertionError):
        pivlfn.FunctionCorrelation(a.permute(0, 1, 3, 2), a, 1)    # contiguity assert :297-298
    with pytest.raises(NotImplementedError):
        pivlfn.backwarp(a, torch.zeros(1, 2, 4, 4))
    net = pivlfn.piv_liteflownet()
    with pytest.raises(NotImplementedError):
        net(torch.zeros(1, 3, 64, 64), torch.zeros(1, 3, 64, 64))          # training mode
    net.eval()
    with pytest.raises(NotImplementedError):
        net(torch.zeros(1, 3, 64, 64), torch.zeros(1, 3, 64, 64))          # CPU module
    with pytest.raises(NotImplementedError):
        pivlfn.estimate(net, torch.zeros(1, 3, 64, 64), torch.zeros(1, 3, 64, 64))


def test_generators_are_deterministic_and_calibrated():
    w1, w2 = synth.generate_weights_np("piv", 0), synth.generate_weights_np("piv", 0)
    assert all(np.array_equal(w1[k], w2[k]) for k in w1)
    assert not np.array_equal(w1["NetC.conv1.0.weight"], synth.generate_weights_np("piv", 1)["NetC.conv1.0.weight"])
    a1, b1, f1 = synth.particle_pair(64, 96, 5)
    a2, b2, f2 = synth.particle_pair(64, 96, 5)
    assert np.array_equal(a1, a2) and np.array_equal(b1, b2) and a1.dtype == np.uint8 and a1.shape == (64, 96)
    assert 2.0 < np.abs(f1).max() < 8.0 and a1.max() > 100
    x = synth.to_input(a1)
    assert x.shape == (3, 64, 96) and x.dtype == np.float32 and 0.0 <= x.min() and x.max() <= 1.0


def test_golden_inputs_are_reproducible_from_seeds(gold):
    rep = json.load(open(os.path.join(GOLD, "pin_report.json")))
    seed = rep["e2e_piv_1x64x64"]["seed"]
    a, b, _ = synth.particle_pair(64, 64, seed)
    assert np.array_equal(a, gold["e2e_cases"]["piv_1x64x64_img1"][0])
    assert np.array_equal(b, gold["e2e_cases"]["piv_1x64x64_img2"][0])


def test_particle_sequence_is_a_pure_function_of_seed_and_frame_index():
    """BASELINE config #4's synthetic sequence: every rank renders only its shard's frames, so frame k must not depend on
    which frames were rendered before it; consecutive frames must differ by the advection (a pair has a flow to find)."""
    import torch
    from pivlfn import synth
    a = synth.ParticleSequence(64, 96, seed=5).frames(0, 4)
    b = synth.ParticleSequence(64, 96, seed=5).frames(2, 4)
    assert a.dtype == torch.uint8 and a.shape == (4, 64, 96)
    assert torch.equal(a[2:], b)
    s = synth.ParticleSequence(64, 96, seed=5)
    later = s.frames(3, 4)
    earlier = s.frames(1, 2)                                   # going backwards restarts the advection
    assert torch.equal(later[0], a[3]) and torch.equal(earlier[0], a[1])
    assert not torch.equal(a[0], a[1])
    assert 10.0 < a.float().mean().item() < 80.0               # same seeding density / intensity model as particle_pair
    assert not torch.equal(synth.ParticleSequence(64, 96, seed=6).frames(0, 1), a[:1])
