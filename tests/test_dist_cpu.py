"""CPU: the N>1 host path (contiguous sharding + all-gather of SMPL records) with 2 gloo processes."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from poco_amd import dist as pdist


def test_shard_ranges_cover_exactly():
    for n in (0, 1, 7, 64, 65, 511):
        for w in (1, 2, 3, 8):
            spans = [pdist.shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def _fake_outputs(lo, hi):
    idx = torch.arange(lo, hi, dtype=torch.float32)
    B = hi - lo
    return {"pred_pose": (idx.view(B, 1, 1, 1) + torch.arange(216.).view(1, 24, 3, 3) / 1000).contiguous(),
            "pred_shape": idx.view(B, 1) * 2 + torch.arange(10.).view(1, 10),
            "pred_cam": idx.view(B, 1) * 3 + torch.arange(3.).view(1, 3),
            "var_pose": idx.view(B, 1) / 100 + torch.arange(24.).view(1, 24) / 1e4}


def _worker(rank, world, port, n_total, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = pdist.shard_range(n_total, rank, world)
    rec = pdist.pack_records(_fake_outputs(lo, hi))
    full = pdist.all_gather_records(rec, n_total)
    expect = pdist.pack_records(_fake_outputs(0, n_total))
    q.put((rank, bool(torch.equal(full, expect)), tuple(full.shape)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_total", [8, 7])
def test_two_rank_gather_gloo(n_total):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_total, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res), res
    assert all(shape == (n_total, pdist.REC) for _, _, shape in res)


def test_unpack_roundtrip():
    o = _fake_outputs(0, 5)
    d = pdist.unpack_records(pdist.pack_records(o))
    for k in ("pred_pose", "pred_shape", "pred_cam", "var_pose"):
        assert torch.equal(d[k], o[k])


def test_shard_tracks_partition_and_balance():
    tracks = {f"p{i}": {"frames": list(range(n)), "bbox": [[0, 0, 1, 1]] * n} for i, n in enumerate([50, 3, 17, 17, 40, 1, 9, 28])}
    for world in (1, 2, 3, 8):
        parts = [pdist.shard_tracks(tracks, r, world) for r in range(world)]
        keys = [k for p in parts for k in p]
        assert sorted(keys) == sorted(tracks)                       # every track on exactly one rank
        loads = [sum(len(v["frames"]) for v in p.values()) for p in parts]
        assert max(loads) - min(loads) <= 50                        # balanced to within the longest track
        assert parts == [pdist.shard_tracks(tracks, r, world) for r in range(world)]   # deterministic


def _track_worker(rank, world, port, q, lengths=(5, 2, 7, 1), width=None):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    tracks = {f"p{i}": {"frames": list(range(n))} for i, n in enumerate(lengths)}
    mine = pdist.shard_tracks(tracks, rank, world)
    width = width or pdist.REC
    rec = lambda k, t: torch.full((t, width), float(int(k[1:]) + 1)) + torch.arange(t).view(t, 1) + 0.001 * torch.arange(width)   # noqa: E731
    local = {k: rec(k, len(v["frames"])) for k, v in mine.items()}
    full = pdist.gather_track_records(local, width=width)
    ok = set(full) == set(tracks) and all(torch.equal(full[k], rec(k, len(v["frames"]))) for k, v in tracks.items())
    q.put((rank, ok))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,lengths,width", [(2, (5, 2, 7, 1), None), (3, (4,), None), (2, (), None), (3, (3, 6, 2, 2, 1), pdist.VIDEO_REC)])
def test_track_gather_gloo(world, lengths, width):
    """Incl. more ranks than tracks (ranks 1, 2 own nothing: ADVICE r1), no track at all, and the 352-float video record
    (final pose | betas | cam | var | confidence | raw 2-D joints) of the round-3 rank-0 merge."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_track_worker, args=(r, world, port, q, lengths, width)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok in res), res
