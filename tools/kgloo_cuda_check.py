#!/usr/bin/env python3
"""Do gloo's collectives carry CUDA tensors correctly when two ranks share one card (the rehearsal set-up of the CFG-split tests)?
all_gather_into_tensor on the world and on a 2-rank subgroup, gather to rank 0, with data produced by a HIP kernel of this
library (raw stream pointer) right before the collective."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "stable-virtual-camera_amd"))
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def worker(rank, world, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    from seva import ops
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    grp = dist.new_group([0, 1])
    bad = 0
    for it in range(20):
        x = torch.full((21, 4, 72, 72), float(rank * 100 + it), device=dev)
        s = torch.full((21,), 2.0, device=dev)
        y = torch.empty_like(x)
        ops.scale_rows(x, s, y)  # HIP kernel on the current stream: y = 2 x
        both = torch.empty((42, 4, 72, 72), device=dev)
        dist.all_gather_into_tensor(both, y, group=grp)
        z = torch.empty_like(both)
        ops.scale_rows(both, torch.full((42,), 0.5, device=dev), z)  # consumer kernel right after
        want = torch.cat([torch.full_like(x, float(0 * 100 + it)), torch.full_like(x, float(1 * 100 + it))])
        if not torch.equal(z, want):
            bad += 1
        bufs = [torch.empty_like(y) for _ in range(world)] if rank == 0 else None
        dist.gather(y.contiguous(), bufs, dst=0)
        if rank == 0 and not (torch.equal(bufs[0], 2 * torch.full_like(x, float(it))) and torch.equal(bufs[1], 2 * torch.full_like(x, float(100 + it)))):
            bad += 100
    print(f"rank {rank}: mismatches {bad}", flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(worker, args=(2, port), nprocs=2, join=True)
