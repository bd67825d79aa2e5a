"""ORACLE -- TEST INFRASTRUCTURE ONLY.  Generates tests/golden/*.npz|json and pins the oracle.

Runs ONLY in the build container, where /root/reference exists.  It imports the reference's own
`src/models.py` with in-memory shims:
  1. a stub module named `cupy` whose `util.memoize(**kw)` is an identity decorator
     (only src/correlation.py:278 runs at import time);
  2. `src.models.FunctionCorrelation` is pointed at the oracle's C correlation (the CuPy CUDA
     kernels cannot run without CUDA; the reference raises NotImplementedError on CPU,
     src/correlation.py:339-340);
  3. `torch.Tensor.cuda` becomes a no-op so the reference's own `backwarp` body
     (src/models.py:20-35) runs unmodified on CPU.
Nothing from the reference is written to disk: fixtures hold inputs (seeded uint8 particle frames,
seeded random tensors) and the reference's numeric outputs only.  Weights are NOT stored: both sides
regenerate them from `pivlfn.synth.generate_weights(model, seed)`.

Usage:  python oracle/gen_golden.py [--only-v2]   (writes tests/golden/, prints the pin report)
"""
from __future__ import annotations

import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(ROOT, "piv_liteflownet-pytorch_amd"))
sys.path.insert(0, HERE)
GOLD = os.path.join(ROOT, "tests", "golden")
REF = "/root/reference"

import pivlfn_oracle as orc                      # noqa: E402
from pivlfn import synth                         # noqa: E402


def import_reference():
    cupy = types.ModuleType("cupy")
    cupy.util = types.SimpleNamespace(memoize=lambda **kw: (lambda f: f))
    cupy.cuda = types.SimpleNamespace(compile_with_cache=None)
    sys.modules["cupy"] = cupy
    sys.path.insert(0, REF)
    import src.models as ref_models              # the reference's file, imported where it lies
    torch.Tensor.cuda = lambda self, *a, **k: self

    def corr_cpu(tensorFirst, tensorSecond, intStride):
        return torch.from_numpy(orc.correlation_c(tensorFirst.numpy(), tensorSecond.numpy(), intStride))

    ref_models.FunctionCorrelation = corr_cpu
    return ref_models


def relerr(a, b):
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / max(1e-30, np.abs(b).max()))


def main_v2(ref):
    """LiteFlowNet2 backbones (`--version 2`, src/models.py:373-716): layout pin + end-to-end fixtures."""
    report = {}
    e2e = {}
    for model, factory in (("piv2", ref.piv_liteflownet), ("hui2", ref.hui_liteflownet)):
        sd = factory(version=2).state_dict()
        spec = [[k, list(v.shape)] for k, v in sd.items()]
        mine = [[k, list(v)] for k, v in synth.state_dict_spec(model).items()]
        assert spec == mine, f"state dict layout mismatch for {model}"
        with open(os.path.join(GOLD, f"state_dict_spec_{model}.json"), "w") as f:
            json.dump(spec, f)
        report[f"spec_{model}"] = {"tensors": len(spec), "params": int(sum(int(np.prod(s)) for _, s in spec))}
    for model, B, H, W, seed in [("piv2", 1, 64, 64, 61), ("piv2", 2, 96, 160, 71), ("hui2", 1, 64, 96, 81)]:
        wts = synth.generate_weights(model, seed=0)
        factory = ref.piv_liteflownet if model == "piv2" else ref.hui_liteflownet
        net = factory(wts, version=2)
        fr1, fr2 = [], []
        for b in range(B):
            a, c, _ = synth.particle_pair(H, W, seed + b)
            fr1.append(a); fr2.append(c)
        fr1 = np.stack(fr1); fr2 = np.stack(fr2)
        i1 = torch.from_numpy(np.stack([synth.to_input(a) for a in fr1]))
        i2 = torch.from_numpy(np.stack([synth.to_input(a) for a in fr2]))
        with torch.no_grad():
            ref.backwarp_tensorGrid.clear()
            net.eval()
            out_ref = net(i1.clone(), i2.clone()).numpy()
            ref.backwarp_tensorGrid.clear()
            net.train()
            lv_ref = net(i1.clone(), i2.clone())[:-1]          # LiteFlowNet2 appends the upsampled flow in training mode (:709-712)
            net.eval()
            out_orc, lv_orc = orc.make_net(model, wts, torch.float32, corr="c").forward(i1, i2, return_levels=True)
        tag = f"{model}_{B}x{H}x{W}"
        bit = bool(np.array_equal(out_orc.numpy(), out_ref))
        assert relerr(out_orc.numpy(), out_ref) < 1e-5, tag
        e2e[f"{tag}_img1"] = fr1; e2e[f"{tag}_img2"] = fr2; e2e[f"{tag}_flow"] = out_ref
        for j, (m, s_, r) in enumerate(lv_ref):
            e2e[f"{tag}_lv{j}_M"] = m.numpy(); e2e[f"{tag}_lv{j}_S"] = s_.numpy(); e2e[f"{tag}_lv{j}_R"] = r.numpy()
            for a, bb in zip((m, s_, r), lv_orc[j]):
                assert relerr(bb.numpy(), a.numpy()) < 1e-4
        report[f"e2e_{tag}"] = {"bit_identical": bit, "max_abs_flow_px": float(np.abs(out_ref).max()), "seed": seed,
                                "out_shape": list(out_ref.shape)}
    np.savez_compressed(os.path.join(GOLD, "e2e_v2_cases.npz"), **e2e)
    with open(os.path.join(GOLD, "pin_report_v2.json"), "w") as f:
        json.dump(report, f, indent=1)
    print(json.dumps(report, indent=1))


def main():
    os.makedirs(GOLD, exist_ok=True)
    if "--only-v2" in sys.argv:
        torch.set_num_threads(8)
        main_v2(import_reference())
        return
    torch.manual_seed(0)
    torch.set_num_threads(8)
    ref = import_reference()
    report = {}

    # ---- state-dict layout pinned from the reference's constructors --------------------------------
    for model, factory in (("piv", ref.piv_liteflownet), ("hui", ref.hui_liteflownet)):
        sd = factory().state_dict()
        spec = [[k, list(v.shape)] for k, v in sd.items()]
        mine = [[k, list(v)] for k, v in synth.state_dict_spec(model).items()]
        assert spec == mine, f"state dict layout mismatch for {model}"
        with open(os.path.join(GOLD, f"state_dict_spec_{model}.json"), "w") as f:
            json.dump(spec, f)
        report[f"spec_{model}"] = {"tensors": len(spec), "params": int(sum(int(np.prod(s)) for _, s in spec))}

    # ---- (i) correlation: two independent restatements must agree ------------------------------------
    g = np.random.Generator(np.random.Philox(key=[7, 7]))
    corr_cases = [(1, 64, 12, 20, 2), (2, 96, 9, 7, 1), (1, 192, 8, 8, 1), (1, 64, 13, 11, 2), (1, 128, 6, 10, 1),
                  (1, 32, 5, 5, 2)]
    cc = {}
    for n, (B, C, H, W, s) in enumerate(corr_cases):
        f1 = g.standard_normal((B, C, H, W)).astype(np.float32)
        f2 = g.standard_normal((B, C, H, W)).astype(np.float32)
        o_c = orc.correlation_c(f1, f2, s)
        o_np = orc.correlation_np(f1.astype(np.float64), f2.astype(np.float64), s)
        e = relerr(o_c, o_np)
        assert e < 2e-6, (n, e)
        cc[f"f1_{n}"] = f1; cc[f"f2_{n}"] = f2; cc[f"out_{n}"] = o_c; cc[f"stride_{n}"] = np.int32(s)
        report[f"corr_case_{n}"] = {"shape": [B, C, H, W, s], "c_vs_np64_rel": e}
    np.savez_compressed(os.path.join(GOLD, "corr_cases.npz"), **cc)

    # ---- (ii) backwarp: the reference's own function body on CPU --------------------------------------
    bw = {}
    for n, (B, C, H, W, amp) in enumerate([(1, 8, 9, 13, 2.5), (2, 3, 16, 12, 6.0), (1, 64, 8, 8, 0.7)]):
        x = g.standard_normal((B, C, H, W)).astype(np.float32)
        fl = (amp * g.standard_normal((B, 2, H, W))).astype(np.float32)
        ref.backwarp_tensorGrid.clear()
        o_ref = ref.backwarp(tensorInput=torch.from_numpy(x), tensorFlow=torch.from_numpy(fl)).numpy()
        o_orc = orc.backwarp(torch.from_numpy(x), torch.from_numpy(fl)).numpy()
        o_np = orc.backwarp_np(x.astype(np.float64), fl.astype(np.float64))
        o_c = orc.backwarp_c(x, fl)
        assert np.array_equal(o_ref, o_orc), "oracle backwarp must be bit-identical to the reference's body"
        e1, e2 = relerr(o_ref, o_np), relerr(o_c, o_np)
        assert e1 < 2e-5 and e2 < 2e-6, (e1, e2)
        bw[f"x_{n}"] = x; bw[f"flow_{n}"] = fl; bw[f"out_{n}"] = o_ref
        report[f"backwarp_case_{n}"] = {"shape": [B, C, H, W], "ref_vs_pixel_np64_rel": e1, "c_vs_np64_rel": e2}
    np.savez_compressed(os.path.join(GOLD, "backwarp_cases.npz"), **bw)

    # ---- (iv,v) end-to-end: reference net (shimmed) vs OracleNet, same generated weights ---------------
    e2e = {}
    cases = [("piv", 1, 64, 64, 11), ("piv", 2, 96, 160, 21), ("hui", 1, 64, 64, 31), ("hui", 2, 96, 160, 41)]
    for model, B, H, W, seed in cases:
        wts = synth.generate_weights(model, seed=0)
        factory = ref.piv_liteflownet if model == "piv" else ref.hui_liteflownet
        net = factory(wts)
        fr1, fr2 = [], []
        for b in range(B):
            a, c, _ = synth.particle_pair(H, W, seed + b)
            fr1.append(a); fr2.append(c)
        fr1 = np.stack(fr1); fr2 = np.stack(fr2)
        i1 = torch.from_numpy(np.stack([synth.to_input(a) for a in fr1]))
        i2 = torch.from_numpy(np.stack([synth.to_input(a) for a in fr2]))
        with torch.no_grad():
            ref.backwarp_tensorGrid.clear()
            net.eval()
            out_ref = net(i1.clone(), i2.clone()).numpy()
            ref.backwarp_tensorGrid.clear()
            net.train()
            lv_ref = net(i1.clone(), i2.clone())
            net.eval()
            onet = orc.make_net(model, wts, torch.float32, corr="c")
            out_orc, lv_orc = onet.forward(i1, i2, return_levels=True)
            o64 = orc.make_net(model, wts, torch.float64, corr="torch").forward(i1, i2).numpy()
        tag = f"{model}_{B}x{H}x{W}"
        e = relerr(out_orc.numpy(), out_ref)
        bit = bool(np.array_equal(out_orc.numpy(), out_ref))
        e2e[f"{tag}_img1"] = fr1; e2e[f"{tag}_img2"] = fr2; e2e[f"{tag}_flow"] = out_ref
        for j, (m, s, r) in enumerate(lv_ref):
            e2e[f"{tag}_lv{j}_M"] = m.numpy(); e2e[f"{tag}_lv{j}_S"] = s.numpy(); e2e[f"{tag}_lv{j}_R"] = r.numpy()
            for a, bb in zip((m, s, r), lv_orc[j]):
                assert relerr(bb.numpy(), a.numpy()) < 1e-4
        lvmax = [float(np.abs(r.numpy()).max()) for (_, _, r) in lv_ref]
        report[f"e2e_{tag}"] = {"oracle_vs_reference_rel": e, "bit_identical": bit,
                                "fp32_vs_fp64_maxabs": float(np.abs(out_ref - o64).max()),
                                "fp32_vs_fp64_meanabs": float(np.abs(out_ref - o64).mean()),
                                "max_abs_flow_px": float(np.abs(out_ref).max()),
                                "per_level_max_norm_flow(L6..)": lvmax, "seed": seed}
        assert e < 1e-5, (tag, e)

    # ---- estimate() at a non-/32 size: restated estimate around the reference net -------------------------
    model, H, W, seed = "piv", 100, 76, 51
    wts = synth.generate_weights(model, seed=0)
    net = ref.piv_liteflownet(wts)
    a, c, _ = synth.particle_pair(H, W, seed)
    i1 = torch.from_numpy(synth.to_input(a))[None]; i2 = torch.from_numpy(synth.to_input(c))[None]
    ref.backwarp_tensorGrid.clear()
    est_ref = orc.estimate(net, i1.clone(), i2.clone(), tensor=True).numpy()
    est_orc = orc.estimate(orc.make_net(model, wts, corr="c"), i1.clone(), i2.clone(), tensor=True).numpy()
    e = relerr(est_orc, est_ref)
    assert e < 1e-5, e
    e2e["est_piv_100x76_img1"] = a[None]; e2e["est_piv_100x76_img2"] = c[None]; e2e["est_piv_100x76_flow"] = est_ref
    report["estimate_piv_100x76"] = {"oracle_vs_reference_rel": e, "max_abs_flow_px": float(np.abs(est_ref).max())}
    np.savez_compressed(os.path.join(GOLD, "e2e_cases.npz"), **e2e)

    with open(os.path.join(GOLD, "pin_report.json"), "w") as f:
        json.dump(report, f, indent=1)
    print(json.dumps(report, indent=1))
    main_v2(ref)


if __name__ == "__main__":
    main()
