"""CPU: .flo wire format, output naming, Run dataset semantics."""
import os
import struct

import numpy as np
import pytest
import torch

from pivlfn.datasets import Run, image_files_from_folder
from pivlfn.flo import FloWriter, flowname_modifier, read_flow, write_flow


def test_flo_roundtrip_and_header(tmp_path):
    flow = np.random.default_rng(0).standard_normal((7, 11, 2)).astype(np.float32)
    p = str(tmp_path / "a_out.flo")
    write_flow(flow, p)
    raw = open(p, "rb").read()
    assert len(raw) == 12 + 7 * 11 * 2 * 4
    tag, w, h = struct.unpack("<fii", raw[:12])
    assert tag == 202021.25 and raw[:4] == b"PIEH" and (w, h) == (11, 7)          # src/utils_plot.py:14-15
    assert np.array_equal(np.frombuffer(raw[12:], "<f4").reshape(7, 11, 2), flow)   # interleaved u,v row-major
    assert np.array_equal(read_flow(p), flow)
    f3 = np.random.default_rng(1).standard_normal((4, 5, 3)).astype(np.float32)
    write_flow(f3, str(tmp_path / "b.flo"))
    assert np.array_equal(read_flow(str(tmp_path / "b.flo"), use_stereo=True), f3)


def test_flo_errors(tmp_path):
    with pytest.raises(AssertionError):
        read_flow(str(tmp_path / "missing.flo"))
    bad = tmp_path / "bad.flo"
    bad.write_bytes(struct.pack("<fii", 1.0, 2, 2) + b"\0" * 32)
    with pytest.raises(AssertionError):
        read_flow(str(bad))
    with pytest.raises(AssertionError):
        write_flow(np.zeros((2, 2, 2), np.float32), str(tmp_path / "x.txt"))
    with pytest.raises(AssertionError):
        write_flow(np.zeros((2, 2, 4), np.float32), str(tmp_path / "x.flo"))


def test_flowname_modifier():
    assert flowname_modifier("/d/DNS_turbulence_img1.tif", "/o") == os.path.join("/o", "DNS_turbulence_out.flo")
    assert flowname_modifier("/d/frame_0001.png", "/o", pair=False) == os.path.join("/o", "frame_0001_out.flo")
    assert flowname_modifier("frame_0001", "/o", pair=False) == os.path.join("/o", "frame_0001_out.flo")


def test_async_writer(tmp_path):
    flows = [np.full((3, 4, 2), i, np.float32) for i in range(20)]
    with FloWriter(workers=3, depth=4) as w:
        for i, f in enumerate(flows):
            w.submit(f, str(tmp_path / f"f{i:03d}_out.flo"))
    for i, f in enumerate(flows):
        assert np.array_equal(read_flow(str(tmp_path / f"f{i:03d}_out.flo")), f)


def _png(path, value, size=(6, 5)):
    import PIL.Image
    PIL.Image.fromarray(np.full(size, value, np.uint8)).save(path)


def test_run_dataset_pairs_and_sequence(tmp_path):
    d = tmp_path / "pairs"
    d.mkdir()
    for name in ("a_img1", "a_img2", "b_img1", "b_img2", "c_img1"):       # c has no partner -> skipped
        _png(str(d / f"{name}.png"), 10)
    ds = Run(str(d), is_pair=True)
    assert len(ds) == 2 and ds.name_list == ["a", "b"]
    (i1, i2), name = ds[0]
    assert name == "a" and i1.shape == (3, 6, 5) and i1.dtype == torch.float32
    assert torch.allclose(i1, torch.full((3, 6, 5), 10 / 255.0))
    s = tmp_path / "seq"
    s.mkdir()
    for k in range(5):
        _png(str(s / f"frame_{k:04d}.png"), k)
    ds = Run(str(s), is_pair=False)
    assert len(ds) == 4 and ds.name_list[0] == "frame_0000"              # pair i = (frame i, frame i+1)
    (i1, i2), _ = ds[3]
    assert torch.allclose(i1, torch.full((3, 6, 5), 3 / 255.0)) and torch.allclose(i2, torch.full((3, 6, 5), 4 / 255.0))
    assert len(Run(str(s), is_pair=False, n_images=3, start_at=1)) == 2
    assert image_files_from_folder(str(s), pair=False, n_images=2, start_at=1) == [str(s / "frame_0001.png"), str(s / "frame_0002.png")]
    with pytest.raises(ValueError):
        Run(str(tmp_path / "nope"))


GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_reference_demo_flo_fixture():
    """tests/golden/DNS_turbulence_out.flo is the reference's own network output for its demo pair
    (/root/reference/images/demo/, a data file, committed as a fixture): the reader must take a real .flo as it is."""
    p = os.path.join(GOLDEN, "DNS_turbulence_out.flo")
    f = read_flow(p)
    assert f.shape == (256, 256, 2) and f.dtype == np.float32
    assert abs(float(f[..., 0].min()) - (-2.5557063)) < 1e-6 and abs(float(f[..., 0].max()) - 2.216687) < 1e-6
    assert abs(float(f[..., 1].min()) - (-2.0943394)) < 1e-6 and abs(float(f[..., 1].max()) - 2.4587212) < 1e-6
    with open(p, "rb") as stream:                       # an open binary stream is accepted too (and closed)
        g = read_flow(stream)
    assert stream.closed and np.array_equal(f, g)
    ref = "/root/reference/images/demo/DNS_turbulence_out.flo"
    if os.path.exists(ref):
        assert open(ref, "rb").read() == open(p, "rb").read()


def test_flo_truncated_payload_is_an_error(tmp_path):
    p = tmp_path / "short.flo"
    p.write_bytes(struct.pack("<fii", 202021.25, 4, 4) + b"\0" * 40)       # 16 of the 128 payload bytes missing
    with pytest.raises(AssertionError):
        read_flow(str(p))
    q = tmp_path / "dims.flo"
    q.write_bytes(struct.pack("<fii", 202021.25, 0, 4))
    with pytest.raises(AssertionError):
        read_flow(str(q))


def test_write_flow_casts_to_float32(tmp_path):
    flow64 = np.random.default_rng(3).standard_normal((3, 5, 2))
    write_flow(flow64, str(tmp_path / "d.flo"))
    assert os.path.getsize(tmp_path / "d.flo") == 12 + 3 * 5 * 2 * 4
    assert np.array_equal(read_flow(str(tmp_path / "d.flo")), flow64.astype(np.float32))


def test_file_listing_order_and_case(tmp_path):
    """Extension by extension, lower-case spelling before the upper-case one, sorted inside a group; hidden files skipped
    (glob semantics of src/utils_data.py:13-33)."""
    for n in ("b.png", "a.png", "c.PNG", "z.jpg", ".hidden.png", "note.txt", "a_img1.tif", "a_img2.tif", "B_img1.TIF"):
        (tmp_path / n).write_bytes(b"")
    d = str(tmp_path)
    got = [os.path.basename(x) for x in image_files_from_folder(d, pair=False)]
    assert got == ["z.jpg", "a.png", "b.png", "c.PNG", "a_img1.tif", "a_img2.tif", "B_img1.TIF"]
    assert [os.path.basename(x) for x in image_files_from_folder(d, pair=False, upper=False)] == ["z.jpg", "a.png", "b.png", "a_img1.tif", "a_img2.tif"]
    assert [os.path.basename(x) for x in image_files_from_folder(d, pair=True)] == ["a_img1.tif", "B_img1.TIF"]
    assert image_files_from_folder(d, pair=False, n_images=0) == []


def test_flowname_modifier_without_underscore():
    assert flowname_modifier("/d/frame.png", "/o") == os.path.join("/o", "frame_out.flo")
    assert flowname_modifier("/d/a_b_img1.png", "/o") == os.path.join("/o", "a_b_out.flo")
