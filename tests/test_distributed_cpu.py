"""The N>1 path on CPU: world_size-2 gloo.  Each rank takes its contiguous shard of the
reads, matches it (here with the oracle standing in for the device matcher -- the oracle is the
checker, and on a CPU-only box the only matcher there is), the records are gathered to rank 0
with real_amd.distributed.gather_records, and rank 0 compares with a one-process run."""
import os
import socket
import sys

import numpy as np
import pytest

from real_amd.distributed import shard_range


def test_shard_range_partitions():
    for n in (0, 1, 7, 50, 1001):
        for w in (1, 2, 3, 8):
            r = [shard_range(n, g, w) for g in range(w)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(r[g][1] == r[g + 1][0] for g in range(w - 1))
            assert max(b - a for a, b in r) - min(b - a for a, b in r) <= 1


def _worker(rank, world, port, out_path):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import oracle_lib as ora
    from real_amd import synth
    from real_amd.distributed import RecordGatherer, gather_records, shard_range
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    g = synth.random_genome(60_000, seed=21, n_frag=2, repeats=10)
    b = synth.sample_reads(g, 1001, 100, 0.02, seed=22)            # odd count: unequal shards
    og = ora.Genome(g.sym, g.frag_start)
    ix = ora.Index(og, 32)
    p = ora.make_params(seedl=32, seedkmax=2, totalkmax=3, scores=1, threads=1)
    lo, hi = shard_range(b.n_reads, rank, world)
    off = b.offsets[lo:hi + 1] - b.offsets[lo]
    info, score, _ = ora.match_unique(og, ix, p, b.bases[int(b.offsets[lo]):int(b.offsets[hi])],
                                      b.qual[int(b.offsets[lo]):int(b.offsets[hi])], off)
    ti = torch.from_numpy(info.view(np.int64).copy())
    ts = torch.from_numpy(score.copy())
    gi, gs = gather_records(ti, ts, dst=0)
    # the pipelined form: two steps through alternating buffers, the second one carries the records
    rg = RecordGatherer(ti.shape[0], "cpu", scores=True)
    bufs = [(torch.zeros_like(ti), torch.zeros_like(ts)), (ti.clone(), ts.clone())]
    for k in range(2):
        rg.wait(k % 2)
        rg.start(k % 2, *bufs[k])
    rg.wait_all()
    if rank == 0:
        assert torch.equal(rg.info_all, gi) and torch.equal(rg.score_all.view(torch.int32), gs.view(torch.int32))
    if rank == 0:
        full_i, full_s, _ = ora.match_unique(og, ix, p, b.bases, b.qual, b.offsets)
        ok = np.array_equal(gi.numpy().view(np.uint64), full_i) and np.array_equal(gs.numpy().view(np.uint32), full_s.view(np.uint32))
        with open(out_path, "w") as f:
            f.write("ok" if ok else "mismatch")
    else:
        assert gi is None and gs is None
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_shard_and_gather(tmp_path):
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    out = str(tmp_path / "result.txt")
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    assert open(out).read() == "ok"
