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
    # matchAll: variable-length hit lists, counts first, then payload
    from real_amd.distributed import gather_hits
    from real_amd.lib import HIT_DTYPE

    def as_abi(oh, ooff):      # oracle hit records -> the ABI's 16-byte real_hip_hit, read index inside the given batch
        h = np.zeros(oh.shape[0], dtype=HIT_DTYPE)
        for f in ("pos", "score", "frag", "k", "inverted"):
            h[f] = oh[f]
        h["read"] = np.repeat(np.arange(ooff.shape[0] - 1, dtype=np.uint32), np.diff(ooff.astype(np.int64)))
        return h

    p2 = ora.make_params(seedl=32, seedkmax=2, totalkmax=2, scores=1, threads=1)
    oh, ooff, _ = ora.match_all(og, ix, p2, b.bases[int(b.offsets[lo]):int(b.offsets[hi])],
                                b.qual[int(b.offsets[lo]):int(b.offsets[hi])], off)
    th = torch.from_numpy(as_abi(oh, ooff).view(np.int32).reshape(-1, 4).copy())
    to = torch.from_numpy(ooff.astype(np.int64))
    gh, go = gather_hits(th, to, dst=0)
    if rank == 0:
        full_i, full_s, _ = ora.match_unique(og, ix, p, b.bases, b.qual, b.offsets)
        ok = np.array_equal(gi.numpy().view(np.uint64), full_i) and np.array_equal(gs.numpy().view(np.uint32), full_s.view(np.uint32))
        fh, foff, _ = ora.match_all(og, ix, p2, b.bases, b.qual, b.offsets)
        want = as_abi(fh, foff)
        got = gh.numpy().reshape(-1).view(HIT_DTYPE)
        ok = ok and np.array_equal(go.numpy(), foff.astype(np.int64)) and got.shape == want.shape and got.tobytes() == want.tobytes()
        ok = ok and want.shape[0] > b.n_reads // 2 and int(np.diff(foff.astype(np.int64)).max()) > 1      # (the case is not trivial)
        with open(out_path, "w") as f:
            f.write("ok" if ok else "mismatch")
    else:
        assert gi is None and gs is None and gh is None and go is None
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_shard_and_gather(tmp_path):
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    out = str(tmp_path / "result.txt")
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    assert open(out).read() == "ok"


def _bench_json(cmd, timeout=600):
    import json
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + cmd, capture_output=True, text=True, timeout=timeout, env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout                  # ONE JSON line, from rank 0
    return json.loads(lines[0])


def test_bench_gpus_n_starts_n_ranks():
    """`python bench.py --gpus 2` (no torch.distributed environment) must start two ranks itself: the ranks meet
    over gloo and count each other (--launch-check makes no GPU call, so this runs on the CPU box)."""
    j = _bench_json(["--gpus", "2", "--launch-check"])
    assert j["launch_check"] and j["n_gpus"] == 2 and j["world_size_env"] == 2 and j["requested"] == 2
    j = _bench_json(["--gpus", "1", "--launch-check"])
    assert j["n_gpus"] == 1


def test_bench_refuses_a_world_size_that_contradicts_gpus():
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--launch-check"], capture_output=True, text=True, env=env, cwd=root)
    assert r.returncode != 0 and "WORLD_SIZE" in r.stderr


@pytest.mark.gpu
def test_bench_two_rank_rehearsal_through_the_launcher():
    """the N > 1 bench path end to end through the exact launcher, on a 1-GPU box: both ranks on cuda:0, gather over
    gloo; matchUnique records and (second run) matchAll hit lists reach the root"""
    small = ["--gpus", "2", "--rehearse-on-one-gpu", "--genome-mbp", "20", "--reads", "200000", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"]
    j = _bench_json(small)
    assert j["n_gpus"] == 2 and j["config"]["distributed"]["world_size"] == 2 and j["config"]["distributed"]["backend"] == "gloo"
    assert j["value"] > 0 and 0.5 < j["config"]["uniquely_aligned_frac_rank0"] < 1
    # the N > 1 line diagnoses itself: what the gather cost outside the shadow of the next step, and what the step rate asks of a link
    d = j["config"]["distributed"]
    for key in ("gather_exposed_ms_per_step", "gather_wait_between_steps_ms_per_step", "gather_final_drain_ms", "gather_bytes_per_rank_per_step",
                "gather_GBps_per_link_needed", "gather_GBps_into_root_needed"):
        assert key in d and d[key] >= 0, key
    assert d["gather_bytes_per_rank_per_step"] == 12 * 200000 and abs(j["value_per_gpu"] * 2 - j["value"]) < 1e-6 * j["value"]
    assert d["gather_exposed_ms_per_step"] < j["ms_per_step"]
    j = _bench_json(small + ["--mode", "all", "--totalk", "2"])
    assert j["n_gpus"] == 2 and j["config"]["hits_gathered_on_root_per_step"] > j["config"]["hits_per_step_rank0"] > 0
