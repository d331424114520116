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


class RecordGatherer:
    """The same gather, set up once and pipelined: shard sizes are exchanged at construction, the root's
    receive buffers are the final arrays themselves (no concatenation), and `start` returns at once so that
    the collective of step k runs while step k+1 is matched.  Two (or more) record buffers alternate; a
    buffer is reused only after `wait` on its gather.

        g = RecordGatherer(n_local, device, scores=True)
        for k in range(steps):
            info, score = bufs[k % 2]
            g.wait(k % 2)                   # the gather that last read this buffer
            ...match into info, score...
            g.start(k % 2, info, score)
        g.wait_all()                        # records of the last step are on the root: g.info_all, g.score_all
    """

    def __init__(self, n_local: int, device, scores: bool = True, dst: int = 0, slots: int = 2):
        import torch
        import torch.distributed as dist
        self.dist, self.torch = dist, torch
        self.world, self.rank, self.dst = dist.get_world_size(), dist.get_rank(), dst
        n = torch.tensor([n_local], dtype=torch.int64, device=device)
        sizes = [torch.zeros_like(n) for _ in range(self.world)]
        dist.all_gather(sizes, n)
        self.sizes = [int(x.item()) for x in sizes]
        self.m = max(self.sizes)
        self.equal = min(self.sizes) == self.m
        self.scores = scores
        self.pending = [[] for _ in range(slots)]
        self.info_all = self.score_all = None
        self._pad = {}
        if self.rank == dst:
            # one receive area per slot: with equal shards the gather writes the final arrays directly
            self._recv_i = [torch.empty(self.world * self.m, dtype=torch.int64, device=device) for _ in range(slots)]
            self._recv_s = [torch.empty(self.world * self.m, dtype=torch.float32, device=device) for _ in range(slots)] if scores else None
        self._last = None

    def _padded(self, t, key):
        if t.shape[0] == self.m:
            return t
        buf = self._pad.get(key)
        if buf is None or buf.dtype != t.dtype:
            buf = self._pad[key] = self.torch.zeros(self.m, dtype=t.dtype, device=t.device)
        buf[:t.shape[0]] = t
        return buf

    def start(self, slot: int, info, score=None):
        outs_i = list(self._recv_i[slot].view(self.world, self.m).unbind(0)) if self.rank == self.dst else None
        self.pending[slot].append(self.dist.gather(self._padded(info, ("i", slot)), outs_i, dst=self.dst, async_op=True))
        if self.scores:
            outs_s = list(self._recv_s[slot].view(self.world, self.m).unbind(0)) if self.rank == self.dst else None
            self.pending[slot].append(self.dist.gather(self._padded(score, ("s", slot)), outs_s, dst=self.dst, async_op=True))
        self._last = slot

    def wait(self, slot: int):
        for w in self.pending[slot]:
            w.wait()
        self.pending[slot] = []

    def wait_all(self):
        for s in range(len(self.pending)):
            self.wait(s)
        if self.rank == self.dst and self._last is not None:
            ri = self._recv_i[self._last].view(self.world, self.m)
            rs = self._recv_s[self._last].view(self.world, self.m) if self.scores else None
            if self.equal:
                self.info_all = ri.reshape(-1)
                self.score_all = rs.reshape(-1) if self.scores else None
            else:
                self.info_all = self.torch.cat([ri[g, :n] for g, n in enumerate(self.sizes)])
                self.score_all = self.torch.cat([rs[g, :n] for g, n in enumerate(self.sizes)]) if self.scores else None
