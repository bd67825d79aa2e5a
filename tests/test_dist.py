"""CPU, gloo, world_size 2: the multi-GPU sharding and the flow all-gather (with a stub network -- the real one needs a GPU)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from pivlfn.dist import frames_needed, shard_bounds


def test_shard_bounds_cover_everything_once():
    for n in (0, 1, 2, 7, 8, 9, 9999):
        for world in (1, 2, 3, 8):
            seen = []
            for r in range(world):
                lo, hi = shard_bounds(n, r, world)
                assert 0 <= lo <= hi <= n
                seen += list(range(lo, hi))
            assert seen == list(range(n))
    assert shard_bounds(9999, 7, 8) == (8750, 9999)          # BASELINE config #4: 10 000 frames -> 9 999 pairs on 8 ranks
    with pytest.raises(ValueError):
        shard_bounds(4, 2, 2)


def test_frames_needed_halo():
    assert frames_needed(0, 5, is_pair=False) == (0, 6)       # sequence: one halo frame (src/datasets.py:456-463)
    assert frames_needed(5, 9, is_pair=False) == (5, 10)
    assert frames_needed(2, 4, is_pair=True) == (4, 8)        # couples: frames 2i, 2i+1
    assert frames_needed(3, 3, is_pair=False) == (0, 0)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_pairs, batch, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from pivlfn.dist import gather_flows, run_sharded, shard_bounds
    H, W = 6, 10

    def load(i0, i1):          # deterministic "frames": pair i is filled with i and i + 0.5
        idx = torch.arange(i0, i1, dtype=torch.float32).view(-1, 1, 1, 1)
        return idx.expand(-1, 3, H, W).clone(), (idx + 0.5).expand(-1, 3, H, W).clone()

    def stub_flow(a, b):       # stands in for estimate(net, a, b, tensor=True)
        return torch.stack([a[:, 0] * 2.0, b[:, 0] - a[:, 0]], 1)

    full = run_sharded(stub_flow, load, n_pairs, batch=batch)
    want_u = torch.arange(n_pairs, dtype=torch.float32) * 2.0
    ok = full.shape == (n_pairs, 2, H, W) and torch.equal(full[:, 0, 0, 0], want_u) and bool((full[:, 1] == 0.5).all())
    # async form, uneven shards
    lo, hi = shard_bounds(n_pairs, rank, world)
    local = torch.full((hi - lo, 2, 2, 2), float(rank))
    work, finish = gather_flows(local, n_pairs, async_op=True)
    g = finish()
    owners = torch.cat([torch.full((shard_bounds(n_pairs, r, world)[1] - shard_bounds(n_pairs, r, world)[0],), float(r)) for r in range(world)])
    ok = ok and torch.equal(g[:, 0, 0, 0], owners)
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_pairs,batch", [(7, 2), (8, 3), (1, 1)])
def test_two_rank_gloo_run_sharded(n_pairs, batch):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_pairs, batch, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res == [(0, True), (1, True)]


def _seq_worker(rank, world, port, n_frames, chunk, outdir, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from pivlfn import synth
    from pivlfn.sequence import run_sequence
    seq = synth.ParticleSequence(32, 48, seed=3, device="cpu")
    asked = []

    def frames(f0, f1):
        asked.extend(range(f0, f1))
        return seq.frames(f0, f1)

    def stub_estimate(net, a, b, tensor=True):          # stands in for pivlfn.estimate: [n,3,H,W] x2 -> [n,2,H,W]
        return torch.stack([b[:, 0] - a[:, 0], a[:, 0] + 2.0 * b[:, 0]], 1)

    st = run_sequence(None, frames, n_frames, chunk, torch.device("cpu"), write_dir=outdir, rank=rank, world=world,
                      estimate_fn=stub_estimate)
    q.put((rank, st["pairs_this_rank"], st["flows_emitted"], asked))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_frames,chunk", [(8, 3), (6, 4), (3, 2)])
def test_two_rank_gloo_sequence(tmp_path, n_frames, chunk):
    """pivlfn.sequence.run_sequence (BASELINE config #4's loop) on two gloo ranks with a stub estimate: contiguous shards, one
    halo frame per rank, every frame rendered once per rank, rank 0 writes every pair's .flo exactly once."""
    import numpy as np
    from pivlfn import synth
    from pivlfn.flo import read_flow
    from pivlfn.sequence import flow_file_name
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    out = str(tmp_path / "flow")
    procs = [ctx.Process(target=_seq_worker, args=(r, 2, port, n_frames, chunk, out, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    n_pairs = n_frames - 1
    assert res[0][1] + res[1][1] == n_pairs and res[0][2] == n_pairs and res[1][2] == 0
    for r in range(2):
        lo, hi = shard_bounds(n_pairs, r, 2)
        assert res[r][3] == (list(range(lo, hi + 1)) if hi > lo else [])       # its shard's frames + the halo, each once, in order
    fr = synth.ParticleSequence(32, 48, seed=3, device="cpu").frames(0, n_frames).to(torch.float32) / 255.0
    assert sorted(os.listdir(out)) == [flow_file_name(k) for k in range(n_pairs)]
    for k in range(n_pairs):
        want = torch.stack([fr[k + 1] - fr[k], fr[k] + 2.0 * fr[k + 1]], -1).numpy()
        assert np.array_equal(read_flow(os.path.join(out, flow_file_name(k))), want)
