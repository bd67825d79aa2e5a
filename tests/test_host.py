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


def test_inference_alias_exports_the_reference_names():
    """run.py:16 of the reference does `from inference import Inference, estimate`."""
    import inference
    assert inference.Inference is pivlfn.Inference and callable(inference.Inference.parser)
    inf = inference.Inference(net=None, netname="models/pretrain_torch/PIV-LiteFlowNet-en.paramOnly", output_dir="/o")
    assert inf.netname == "PIV-LiteFlowNet-en" and inf.default == os.path.join("/o", "PIV-LiteFlowNet-en")
    with pytest.raises(NotImplementedError):
        inf.video_parsing(0)
    with pytest.raises(AssertionError):                         # size mismatch is asserted before anything runs (inference.py:204)
        import PIL.Image
        inference.Inference.parser(None, PIL.Image.new("RGB", (8, 8)), PIL.Image.new("RGB", (8, 9)))


def test_image_mod_is_bit_identical_to_pil_enhance():
    """run.py -b/-c: the reference modifies the PIL image with torchvision's adjust_brightness / adjust_contrast, which for PIL
    inputs are ImageEnhance.Brightness / .Contrast; pivlfn.imagemod restates them on uint8 tensors."""
    import PIL.Image
    import PIL.ImageEnhance
    from pivlfn.imagemod import image_mod, mod_name
    rng = np.random.default_rng(0)
    for trial in range(3):
        a = rng.integers(0, 256, (23, 31, 3), dtype=np.uint8)
        if trial == 1:
            a[...] = a[..., :1]                                  # grey frames, as PIV images are
        for b in (0.0, 0.3, 1.0, 1.5, 2.7):
            for c in (0.0, 0.25, 1.0, 1.3, 3.0):
                want = np.asarray(PIL.ImageEnhance.Contrast(PIL.ImageEnhance.Brightness(PIL.Image.fromarray(a)).enhance(b)).enhance(c))
                got = image_mod(torch.from_numpy(a), b, c).numpy()
                assert np.array_equal(want, got), (trial, b, c)
    batch = torch.from_numpy(rng.integers(0, 256, (2, 9, 7, 3), dtype=np.uint8))
    out = image_mod(batch, 1.2, 0.6)                             # the contrast mean is per frame
    assert torch.equal(out[0], image_mod(batch[0], 1.2, 0.6)) and torch.equal(out[1], image_mod(batch[1], 1.2, 0.6))
    assert mod_name(1.0, 1.0) == "100_100" and mod_name(0.5, 1.25) == "050_125"      # run.py:125
    with pytest.raises(ValueError):
        image_mod(torch.zeros(4, 4, 3), 1.0, 1.0)


def test_run_py_output_layout_and_mod_names():
    """Output tree of run.py:232-266 and the -b/-c naming of run.py:125-131."""
    import run
    lay = run.OutputLayout.of("/out", "PIV-LiteFlowNet-en", "/data/exp7", 0, -1)
    assert (lay.save, lay.flow, lay.args_file) == ("/out/PIV-LiteFlowNet-en/exp7", "/out/PIV-LiteFlowNet-en/exp7/flow",
                                                   "/out/PIV-LiteFlowNet-en/exp7/args.txt")
    lay = run.OutputLayout.of("/out", "net", "/data/exp7/", 5, -1)
    assert lay.save == "/out/net/exp7-5_end"
    lay = run.OutputLayout.of("/out", "net", "/data/exp7", 0, 12)
    assert lay.save == "/out/net/exp7-0_12"
    lay = run.OutputLayout.of("/out", "net", "/data/stereo3/Left", 0, -1)
    assert (lay.save, lay.flow, lay.args_file) == ("/out/net/stereo3", "/out/net/stereo3/flow/left", "/out/net/stereo3/args_left.txt")
    assert run.mod_flow_name("/d/with_under/img_0007.png", "/o", (1.0, 0.5)) == "/o/img_100_050_0007_out.flo"
    assert run.mod_flow_name("/d/frame7.png", "/o", (1.5, 1.0)) == "/o/frame7_150_100_out.flo"
    args = run.parser.parse_args(["-i", "a", "b", "-b", "0.5", "1.5", "-c", "2"])
    assert args.brightness == [0.5, 1.5] and args.contrast == [2.0] and args.model == "piv" and args.input == ["a", "b"]


def test_precision_property_without_a_gpu():
    """`Network.precision` is host state until the weights are uploaded: default, accepted names, refusal of anything else."""
    import pivlfn
    net = pivlfn.Network(model="piv")
    assert net.precision == "fp32"
    for mode in ("fp32_direct", "fp32_split", "fp16", "fp32_split3", "fp32"):
        net.precision = mode
        assert net.precision == mode
    with pytest.raises(ValueError):
        net.precision = "bf16"
    assert net.precision == "fp32"


def test_no_kernel_spills_to_scratch():
    """The gfx950 code objects inside libpivlfn.so: no kernel keeps live values in scratch memory, bar one listed exception.  (A
    Winograd build that needed 24 bytes of scratch per lane returned wrong tiles at full occupancy in round 3; spills are also a
    serialising round trip to memory in the middle of a software pipeline.)  Read from the AMDGPU metadata note of every embedded
    code object with llvm-readelf."""
    import re
    import shutil
    import struct
    import subprocess
    import tempfile
    from pivlfn import _lib
    readelf = shutil.which("llvm-readelf") or "/opt/rocm/lib/llvm/bin/llvm-readelf"
    if not os.path.exists(readelf):
        pytest.skip("llvm-readelf not found")
    data = open(_lib.LIB_PATH, "rb").read()
    allowed = {}              # kernel-name fragment -> spilled registers tolerated (none at present)
    seen = 0
    with tempfile.TemporaryDirectory() as tmp:
        for m in re.finditer(b"__CLANG_OFFLOAD_BUNDLE__", data):
            i = m.start()
            nb = struct.unpack_from("<Q", data, i + 24)[0]
            off = i + 32
            for _ in range(nb):
                o, sz, tl = struct.unpack_from("<QQQ", data, off)
                off += 24
                triple = data[off:off + tl].decode()
                off += tl
                if "gfx950" not in triple or sz == 0:
                    continue
                path = os.path.join(tmp, "co.elf")
                with open(path, "wb") as f:
                    f.write(data[i + o:i + o + sz])
                out = subprocess.run([readelf, "--notes", path], capture_output=True, text=True, check=True).stdout
                name = None
                for line in out.splitlines():
                    t = line.strip()
                    if t.startswith(".name:"):
                        name = t.split(":", 1)[1].strip()
                    elif t.startswith(".vgpr_spill_count:"):
                        seen += 1
                        n = int(t.split(":")[1])
                        limit = max([v for k, v in allowed.items() if k in name] + [0])
                        assert n <= limit, f"{name} spills {n} registers to scratch"
    assert seen > 50          # every kernel of the library was looked at


def test_bench_py_refuses_more_ranks_than_devices():
    """`python bench.py --gpus 8` with no outer launcher starts its own ranks; with fewer devices visible than ranks asked for (none
    in this container) the parent refuses before it starts anything and before it touches a GPU itself."""
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "PIVLFN_BENCH_BACKEND")}
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    import torch
    if torch.cuda.device_count() >= 8:
        pytest.skip("eight devices are visible here")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--steps", "1", "--warmup", "0"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 2 and "device(s) visible" in r.stderr and not r.stdout.strip()


def test_bench_py_parent_reports_a_dead_rank_quickly():
    """No GPU needed: with the gloo backend the parent starts two ranks; rank 1 exits 3 on request (rank 0 stops at its own "needs a GPU"
    assertion here).  The parent must return non-zero within seconds, name a rank with its exit code and show that rank's stderr --
    not sit in a wait for the other one."""
    import subprocess
    import sys
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    t0 = time.monotonic()
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=dict(env, PIVLFN_BENCH_BACKEND="gloo", PIVLFN_BENCH_FAIL_RANK="1"), capture_output=True, text=True, timeout=120)
    assert r.returncode != 0
    assert time.monotonic() - t0 < 60
    assert "exited with code" in r.stderr and "--- stderr tail of rank" in r.stderr, r.stderr[-1500:]
