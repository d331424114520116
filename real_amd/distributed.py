"""Read sharding across the GPUs of one node and the one collective of the path.

The path shards by reads: rank g of N takes the contiguous range
[g*R/N, (g+1)*R/N) of the batch, matches it against its own replica of the index, and the
per-read records (UniqueMatchInfo word + float score = 12 B/read) are gathered to the root with
one collective.  No reduction: no read is seen by two ranks.  One process per GPU;
torch.distributed backend "nccl" is RCCL on ROCm (xGMI point-to-point links), "gloo" on CPU.
"""
from __future__ import annotations

from typing import List, Optional, Tuple


def shard_range(n_total: int, rank: int, world: int) -> Tuple[int, int]:
    """contiguous read range of `rank` (sizes differ by at most one)."""
    base, rem = divmod(n_total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def gather_records(info, score, dst: int = 0):
    """Gather the shards' records to `dst` in rank order.  info: int64/uint64 tensor, score: float32
    tensor or None (device tensors with nccl, CPU tensors with gloo).  Returns (info_all, score_all)
    on dst, (None, None) elsewhere.  Shards may differ in length by one read: sizes are exchanged
    first and the payload is padded to the longest shard."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size()
    rank = dist.get_rank()
    n = torch.tensor([info.shape[0]], dtype=torch.int64, device=info.device)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n)
    sizes = [int(s.item()) for s in sizes]
    m = max(sizes)

    def pad(t):
        if t.shape[0] == m:
            return t.contiguous()
        out = torch.zeros(m, dtype=t.dtype, device=t.device)
        out[:t.shape[0]] = t
        return out

    def gather(t):
        out: Optional[List] = [torch.empty(m, dtype=t.dtype, device=t.device) for _ in range(world)] if rank == dst else None
        dist.gather(pad(t), out, dst=dst)
        return out

    gi = gather(info)
    gs = gather(score) if score is not None else None
    if rank != dst:
        return None, None
    info_all = torch.cat([g[:s] for g, s in zip(gi, sizes)])
    score_all = torch.cat([g[:s] for g, s in zip(gs, sizes)]) if gs is not None else None
    return info_all, score_all
