"""The N > 1 path on CPU: two processes over gloo partition the streams and gather the tracks."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from lbm_amd.sharding import gather_tracks, max_over_ranks, split_streams


def test_split_streams_covers_everything_once():
    for n in (0, 1, 7, 8, 512, 513):
        for world in (1, 2, 3, 8):
            blocks = [split_streams(n, world, r) for r in range(world)]
            assert blocks[0][0] == 0 and blocks[-1][1] == n
            assert all(blocks[i][1] == blocks[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in blocks]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        split_streams(4, 2, 2)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_streams, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lo, hi = split_streams(n_streams, world, rank)
        T, N = 3, 5
        # what a rank would produce: (streams, T, N, 4) = x, y, vis, rho; value encodes the global stream id
        sid = torch.arange(lo, hi, dtype=torch.float32).view(-1, 1, 1, 1)
        local = sid * 1000 + torch.arange(T * N * 4, dtype=torch.float32).view(1, T, N, 4)
        got = gather_tracks(local, n_streams, dst=0)
        slow = max_over_ranks(1.0 + rank, torch.device("cpu"))
        if rank == 0:
            q.put((got.clone(), slow))
        else:
            assert got is None
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_streams", [5, 8])
def test_two_ranks_gather_tracks_in_stream_order(n_streams):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_streams, q)) for r in range(2)]
    for p in procs:
        p.start()
    got, slow = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    want = torch.arange(n_streams, dtype=torch.float32).view(-1, 1, 1, 1) * 1000 + torch.arange(60, dtype=torch.float32).view(1, 3, 5, 4)
    assert torch.equal(got, want)
    assert slow == 2.0


def _record_worker(rank, world, port, path, q):
    from lbm_amd.sharding import TrackWriter, pack_tracks
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        S, T, N = 3, 4, 6
        lo, hi = split_streams(S, world, rank)
        g = torch.Generator().manual_seed(5)
        coords = torch.rand(S, T, N, 2, generator=g) * 100
        vis = torch.rand(S, T, N, generator=g) > 0.4
        rho = torch.rand(S, T, N, generator=g)
        writer = TrackWriter(path, S, N, (480, 854)) if rank == 0 else None
        for t in range(T):                       # one gather per frame: what a live consumer on rank 0 would see
            rec = gather_tracks(pack_tracks(coords[lo:hi, t], vis[lo:hi, t], rho[lo:hi, t]), S, dst=0)
            if rank == 0:
                writer.append(rec)
        if rank == 0:
            writer.close()
            q.put((coords, vis, rho))
    finally:
        dist.destroy_process_group()


def test_track_records_through_the_gather_into_the_track_file(tmp_path):
    """(x, y, visible, rho) records: packed per rank, gathered per frame, appended to the track file, read back in the
    layout of LBM.inference's result; a file cut inside a frame yields the complete frames."""
    from lbm_amd.sharding import read_tracks
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    path = str(tmp_path / "tracks.lbmt")
    procs = [ctx.Process(target=_record_worker, args=(r, 2, port, path, q)) for r in range(2)]
    for p in procs:
        p.start()
    coords, vis, rho = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    got = read_tracks(path)
    assert got["points"].shape == (3, 4, 6, 2) and got["frame_hw"] == (480.0, 854.0)
    assert torch.equal(got["points"], coords) and torch.equal(got["visibility"], vis) and torch.equal(got["rho"], rho)
    raw = open(path, "rb").read()
    open(path, "wb").write(raw[:-20])
    cut = read_tracks(path)
    assert cut["points"].shape == (3, 3, 6, 2) and torch.equal(cut["points"], coords[:, :3])
    open(path, "wb").write(b"garbage")
    with pytest.raises(ValueError):
        read_tracks(path)
