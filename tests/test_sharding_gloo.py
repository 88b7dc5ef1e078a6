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
