"""GPU: every model family at a mid-size image against the CPU oracle (fp32 mode, the tolerance of the end-to-end tests) and the
fp16 mode's end-point error -- PIV / Hui, LiteFlowNet and LiteFlowNet2 layouts (src/models.py:39-370, 373-716, 719-766)."""
import pytest
import torch

import pivlfn
import pivlfn_oracle as orc
from pivlfn import synth

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("model,version", [("piv", 1), ("hui", 1), ("piv", 2), ("hui", 2)])
def test_model_family_mid_size(model, version, dev):
    tag = model + ("2" if version == 2 else "")
    wts = synth.generate_weights(tag, 0)
    a, b, _ = synth.particle_pair(384, 448, 11)
    i1 = torch.from_numpy(synth.to_input(a))[None]
    i2 = torch.from_numpy(synth.to_input(b))[None]
    net = pivlfn.Network(model=model, params=wts, version=version).to(dev).eval()
    got = net(i1.to(dev), i2.to(dev)).cpu()
    with torch.no_grad():
        want = orc.make_net(tag, wts, corr="c").forward(i1.clone(), i2.clone())
    assert got.shape == want.shape
    scale = max(1.0, float(want.abs().max()))
    assert float((got - want).abs().max()) <= 1e-4 * scale
    net.precision = "fp16"
    g16 = net(i1.to(dev), i2.to(dev)).cpu()
    epe = (g16 - got).pow(2).sum(1).sqrt()
    assert float(epe.mean()) <= 0.05
