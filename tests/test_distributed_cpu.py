"""world_size-2 gloo test of the window sharding / anchor exchange (CPU, runs everywhere)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from seva.distributed import exchange_anchor_latents, gather_window_outputs, shard_windows
    mine = shard_windows(10)
    local = torch.full((2, 4, 3, 3), float(rank + 1))
    allv = exchange_anchor_latents(local)
    outs = gather_window_outputs(torch.full((3,), float(rank)), dst=0)
    q.put((rank, mine, allv[:, 0, 0, 0].tolist(), None if outs is None else [o[0].item() for o in outs]))
    dist.destroy_process_group()


def test_two_rank_sharding_and_exchange():
    import sys
    from conftest import PKG
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    os.environ["PYTHONPATH"] = PKG + os.pathsep + os.environ.get("PYTHONPATH", "")
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert res[0][1] == [0, 2, 4, 6, 8] and res[1][1] == [1, 3, 5, 7, 9]
    assert res[0][2] == res[1][2] == [1.0, 1.0, 2.0, 2.0]
    assert res[0][3] == [0.0, 1.0] and res[1][3] is None


def test_single_process_identity():
    from seva.distributed import exchange_anchor_latents, shard_windows
    t = torch.arange(8.0).view(2, 4, 1, 1)
    assert exchange_anchor_latents(t) is t
    assert shard_windows(10, 3, 8) == [3] and shard_windows(11, 0, 1) == list(range(11))
