"""Multi-GPU plumbing: streams are independent (per-stream state only, lib/libbackscrub.cc:46-48),
so the job shards contiguous blocks of streams across ranks with NO data-path collective.  The one
collective is the all-reduce of the throughput counters {frames (sum), elapsed (max), checksum (sum)}
— RCCL over xGMI on GPUs (backend "nccl"), gloo in the CPU tests.  SURVEY.md §8(e)."""
from __future__ import annotations

import os


def env_rank_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


def shard_streams(total_streams: int, world: int, rank: int):
    """Contiguous block of global stream ids owned by `rank` (first `total % world` ranks get one extra)."""
    base, extra = divmod(total_streams, world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def reduce_counters(frames: float, elapsed: float, checksum: int, device=None):
    """→ (total frames, max elapsed over ranks, checksum sum mod 2^40).  Works with any initialised backend;
    with no process group it returns the local values."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(frames), float(elapsed), int(checksum) % (1 << 40)
    s = torch.tensor([float(frames), float(int(checksum) % (1 << 40))], dtype=torch.float64, device=device)
    m = torch.tensor([float(elapsed)], dtype=torch.float64, device=device)
    dist.all_reduce(s, op=dist.ReduceOp.SUM)
    dist.all_reduce(m, op=dist.ReduceOp.MAX)
    return s[0].item(), m[0].item(), int(s[1].item()) % (1 << 40)
