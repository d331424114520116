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


def gather_hits(hits, hit_offsets, dst: int = 0):
    """matchAll across ranks: gather the shards' variable-length hit lists to `dst` -- counts first, then payload
    (SURVEY 8e).  Every rank holds the hits of its own contiguous shard of reads as the library returned them
    (matchAllImplementation.cpp:451-535 emits the unified hit list of every read of a block; here the block's reads
    are spread over the ranks):

      hits         int32 tensor [n_hits, 4]: the 16-byte real_hip_hit records viewed as four dwords; dword 0 is the
                   read index INSIDE the shard
      hit_offsets  int64 tensor [n_local + 1]: hits[hit_offsets[i] : hit_offsets[i+1]] belong to read i of the shard

    Returns on dst (hits_all [H, 4], offsets_all [R + 1]) with read indices rebased to the whole batch (rank order =
    read order, as shard_range cuts it) and offsets rebased by the hits of the ranks in front; (None, None) elsewhere.
    Two collectives carry data: one all_gather of {n_local, n_hits} per rank, then one gather of the payload padded to
    the longest shard (device tensors with nccl = RCCL over xGMI, CPU tensors with gloo)."""
    import torch
    import torch.distributed as dist
    world, rank = dist.get_world_size(), dist.get_rank()
    dev = hits.device
    n_local, n_hits = int(hit_offsets.shape[0]) - 1, int(hits.shape[0])
    mine = torch.tensor([n_local, n_hits], dtype=torch.int64, device=dev)
    sizes = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(sizes, mine)                                  # counts first
    sizes = [(int(s[0].item()), int(s[1].item())) for s in sizes]
    max_r, max_h = max(s[0] for s in sizes), max(s[1] for s in sizes)

    # payload: hit records, and the per-read hit counts (offsets are rebuilt from them on the root)
    pay_h = torch.zeros((max(max_h, 1), 4), dtype=torch.int32, device=dev)
    pay_h[:n_hits] = hits.view(-1, 4)[:n_hits]
    pay_c = torch.zeros(max(max_r, 1), dtype=torch.int64, device=dev)
    pay_c[:n_local] = hit_offsets[1:] - hit_offsets[:-1]
    out_h = [torch.empty_like(pay_h) for _ in range(world)] if rank == dst else None
    out_c = [torch.empty_like(pay_c) for _ in range(world)] if rank == dst else None
    dist.gather(pay_h, out_h, dst=dst)
    dist.gather(pay_c, out_c, dst=dst)
    if rank != dst:
        return None, None
    parts, counts, first = [], [], 0
    for g, (nr, nh) in enumerate(sizes):
        h = out_h[g][:nh].clone()
        h[:, 0] += first                                          # read index inside the shard -> inside the batch
        parts.append(h)
        counts.append(out_c[g][:nr])
        first += nr
    hits_all = torch.cat(parts) if parts else torch.zeros((0, 4), dtype=torch.int32, device=dev)
    offsets_all = torch.zeros(first + 1, dtype=torch.int64, device=dev)
    torch.cumsum(torch.cat(counts), 0, out=offsets_all[1:])
    return hits_all, offsets_all
