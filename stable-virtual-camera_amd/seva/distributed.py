"""Multi-GPU sharding of the denoising hot path (one process per GPU, RCCL over xGMI).

The reference is single-process (no distributed code).  The path shards at window
granularity: second-pass windows of a long trajectory read only {input views, first-pass anchors}
(reference seva/eval.py:1890-1906), so they are independent work units.  Weights (2.5 GB fp16) are
replicated; the only exchange is one all-gather of the anchor latents that adjacent windows share
(<= 20 x 4 x 72 x 72 fp32 = 1.66 MB) -- latency-, not bandwidth-bound on xGMI, so a single
direct all-gather is used and nothing is bucketed or overlapped.
"""

from __future__ import annotations

import torch
import torch.distributed as dist


def shard_windows(num_windows: int, rank: int | None = None, world: int | None = None) -> list[int]:
    """Window indices owned by `rank`: round-robin (rank, rank+world, ...), matching SURVEY §8e."""
    if rank is None:
        rank = dist.get_rank() if dist.is_initialized() else 0
    if world is None:
        world = dist.get_world_size() if dist.is_initialized() else 1
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world of {world}")
    return list(range(rank, num_windows, world))


def exchange_anchor_latents(local: torch.Tensor) -> torch.Tensor:
    """All-gather per-rank anchor latents [a, c, h, w] -> [world * a, c, h, w] (rank-major).

    Every rank must pass the same shape.  With a single process it is the identity."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return local
    local = local.contiguous()
    out = torch.empty((dist.get_world_size() * local.shape[0],) + tuple(local.shape[1:]),
                      dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, local)
    return out


def gather_window_outputs(local: torch.Tensor, dst: int = 0) -> list[torch.Tensor] | None:
    """Collect each rank's finished window latents on `dst` (for decoding/saving there)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return [local]
    bufs = [torch.empty_like(local) for _ in range(dist.get_world_size())] if dist.get_rank() == dst else None
    dist.gather(local.contiguous(), bufs, dst=dst)
    return bufs
