"""Host logic of the streaming pipeline (pivlfn.pipeline) on the CPU: batching, ordering, single decode per
frame in sequence mode, error propagation, and agreement of the uint8 device conversion with the reference's ToTensor."""
import os

import numpy as np
import pytest
import torch

from pivlfn.datasets import Run, read_image
from pivlfn.pipeline import PairLoader, read_image_u8, stream_pairs, u8_to_input


def _write_frames(folder, n, size=(24, 32), pair=False, seed=0):
    import PIL.Image
    rng = np.random.default_rng(seed)
    os.makedirs(folder, exist_ok=True)
    for i in range(n):
        a = rng.integers(0, 256, size=(size[0], size[1]), dtype=np.uint8)
        if pair:
            PIL.Image.fromarray(a).save(os.path.join(folder, f"p{i:03d}_img1.png"))
            PIL.Image.fromarray(255 - a).save(os.path.join(folder, f"p{i:03d}_img2.png"))
        else:
            PIL.Image.fromarray(a).save(os.path.join(folder, f"f{i:03d}.png"))


def _fake_estimate(net, a, b, tensor=True):
    # a stand-in with the boundary's shapes: [n,3,H,W] x2 -> [n,2,H,W]
    return torch.stack([(b - a).mean(1), (b + a).mean(1)], dim=1)


def test_u8_conversion_matches_to_tensor(tmp_path):
    _write_frames(str(tmp_path), 1)
    p = os.path.join(str(tmp_path), "f000.png")
    ref = read_image(p)                                            # float path: uint8 -> float32 / 255 (ToTensor)
    got = u8_to_input(torch.from_numpy(read_image_u8(p))[None])[0]
    assert got.dtype == torch.float32 and got.shape == ref.shape
    assert torch.equal(got, ref)


@pytest.mark.parametrize("batch", [1, 3, 4])
def test_sequence_mode_decodes_each_frame_once(tmp_path, batch):
    _write_frames(str(tmp_path), 8)
    ds = Run(str(tmp_path), is_pair=False)
    assert len(ds) == 7
    loader = PairLoader(ds, 0, len(ds), batch)
    names, n1, n2 = [], [], []
    for nm, a, b in loader:
        assert a.dtype == torch.uint8 and a.shape == b.shape and a.shape[1:] == (24, 32, 3) and len(nm) <= batch
        names += nm
        n1.append(a)
        n2.append(b)
    loader.close()
    assert names == ds.name_list
    assert loader.decoded == 8                                     # 7 pairs, 8 frames: the halo frame is reused
    a_all, b_all = torch.cat(n1), torch.cat(n2)
    assert torch.equal(a_all[1:], b_all[:-1])                      # pair i = (frame i, frame i+1)
    for i in range(7):
        assert torch.equal(u8_to_input(a_all[i:i + 1])[0], ds[i][0][0])
        assert torch.equal(u8_to_input(b_all[i:i + 1])[0], ds[i][0][1])


def test_pair_mode_and_shard_range(tmp_path):
    _write_frames(str(tmp_path), 5, pair=True)
    ds = Run(str(tmp_path), is_pair=True)
    loader = PairLoader(ds, 1, 4, 2)
    got = [(nm, a.clone(), b.clone()) for nm, a, b in loader]
    loader.close()
    assert [n for g in got for n in g[0]] == ds.name_list[1:4]
    assert [len(g[0]) for g in got] == [2, 1]
    assert loader.decoded == 6
    assert torch.equal(got[0][1][0], 255 - got[0][2][0])


def test_batches_never_mix_sizes(tmp_path):
    _write_frames(os.path.join(str(tmp_path)), 2, size=(16, 16), pair=True, seed=1)
    import PIL.Image
    big = np.zeros((32, 16), np.uint8)
    for tag in ("img1", "img2"):
        PIL.Image.fromarray(big).save(os.path.join(str(tmp_path), f"p001b_{tag}.png"))      # sorts between p001 and p002..
    ds = Run(str(tmp_path), is_pair=True)
    loader = PairLoader(ds, 0, len(ds), 8)
    shapes = [tuple(a.shape) for _, a, _ in loader]
    loader.close()
    assert sum(s[0] for s in shapes) == len(ds) == 3
    assert all(len({s[1:]}) == 1 for s in shapes) and len(shapes) >= 2


def test_stream_pairs_order_and_values(tmp_path):
    _write_frames(str(tmp_path), 6)
    ds = Run(str(tmp_path), is_pair=False)
    loader = PairLoader(ds, 0, len(ds), 2)
    seen = []
    n = stream_pairs(None, loader, torch.device("cpu"), lambda f, name: seen.append((name, f.copy())), estimate_fn=_fake_estimate)
    loader.close()
    assert n == 5 and [s[0] for s in seen] == ds.name_list
    for i, (_, f) in enumerate(seen):
        (a, b), _ = ds[i]
        want = _fake_estimate(None, a[None], b[None])[0].permute(1, 2, 0).numpy()
        assert f.shape == (24, 32, 2)
        np.testing.assert_array_equal(f, want)


def test_reader_errors_surface_in_the_consumer(tmp_path):
    _write_frames(str(tmp_path), 4)
    ds = Run(str(tmp_path), is_pair=False)

    def bad_reader(path):
        if path.endswith("f002.png"):
            raise OSError("boom")
        return read_image_u8(path)

    loader = PairLoader(ds, 0, len(ds), 1, reader=bad_reader)
    with pytest.raises(OSError, match="boom"):
        for _ in loader:
            pass
    loader.close()


def test_close_unblocks_a_full_queue(tmp_path):
    _write_frames(str(tmp_path), 10)
    ds = Run(str(tmp_path), is_pair=False)
    loader = PairLoader(ds, 0, len(ds), 1, depth=1)
    next(iter(loader))
    loader.close()                                                 # producer is blocked on put(); must exit
    assert not loader._thread.is_alive()


def test_stream_pairs_with_brightness_contrast_mods(tmp_path):
    """run.py -b/-c: every pair is estimated once per (brightness, contrast) combination on frames modified on the device
    (here: the CPU stand-in), in pair-major order, and the sink learns which combination a flow belongs to."""
    from pivlfn.imagemod import image_mod
    _write_frames(str(tmp_path), 4)
    ds = Run(str(tmp_path), is_pair=False)
    mods = [(1.0, 1.0), (0.5, 1.0), (1.5, 2.0)]
    got = []
    loader = PairLoader(ds, 0, len(ds), 2)
    n = stream_pairs(None, loader, torch.device("cpu"), lambda flow, name, mod: got.append((name, mod, flow.copy())),
                     estimate_fn=_fake_estimate, mods=mods)
    loader.close()
    assert n == 3 * len(mods) and len(got) == 9
    for name, mod, flow in got:
        i = ds.name_list.index(name)
        a = torch.from_numpy(read_image_u8(ds.image_list[i][0]))[None]
        b = torch.from_numpy(read_image_u8(ds.image_list[i][1]))[None]
        want = _fake_estimate(None, u8_to_input(image_mod(a, *mod)), u8_to_input(image_mod(b, *mod)))[0].permute(1, 2, 0).numpy()
        assert np.array_equal(flow, want), (name, mod)
    assert sorted({m for _, m, _ in got}) == sorted(mods)
