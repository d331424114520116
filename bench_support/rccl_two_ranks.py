#!/usr/bin/env python3
"""The N = 2 gathers of the path -- torch.distributed's (real_amd/distributed.py) and the C ABI's own
(real_hip_comm_*, real_hip_gather_records / _hits) -- over real RCCL, two ranks started by the script itself.

    python bench_support/rccl_two_ranks.py

On a box with two or more GPUs rank r uses GPU r.  On the one-GPU development box both ranks land on cuda:0 and RCCL
refuses the communicator ("Duplicate GPU detected : rank 0 and rank 1 both on CUDA device ...", tried in round 2): there
RCCL can only be run with one rank (tests/test_gpu_pipeline.py::test_c_abi_rccl_gather_single_rank), the two-rank logic
over gloo (tests/test_distributed_cpu.py), and N > 1 over RCCL is left to the driver's multi-GPU run of bench.py.
"""
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    if "RANK" not in os.environ:
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", NCCL_DEBUG="WARN")
        sys.exit(subprocess.call([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                                  "--master-port", str(port), os.path.abspath(__file__)], env=env))
    import numpy as np
    import torch
    import torch.distributed as dist
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    local = int(os.environ.get("LOCAL_RANK", "0")) % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist.init_process_group("nccl")
    t = torch.ones(1, device=dev) * (rank + 1)
    dist.all_reduce(t)
    assert int(t.item()) == 3
    print("rank %d: all_reduce over RCCL with two ranks on one GPU works" % rank, flush=True)
    from real_amd.distributed import RecordGatherer, gather_hits, gather_records
    n = 1000 + rank
    info = torch.arange(n, dtype=torch.int64, device=dev) + 10_000 * rank
    score = torch.arange(n, dtype=torch.float32, device=dev) + 0.5 * rank
    gi, gs = gather_records(info, score, dst=0)
    if rank == 0:
        assert gi.shape[0] == 2001 and int(gi[1000].item()) == 10_000 and float(gs[1000].item()) == 0.5
    hits = torch.zeros((n * 2, 4), dtype=torch.int32, device=dev)
    hits[:, 0] = torch.arange(n * 2, device=dev) // 2
    hits[:, 1] = 7 + rank
    off = torch.arange(n + 1, dtype=torch.int64, device=dev) * 2
    gh, go = gather_hits(hits, off, dst=0)
    if rank == 0:
        assert gh.shape[0] == 2 * 2001 and int(gh[2000, 0].item()) == 1000 and int(gh[2000, 1].item()) == 8 and int(go[-1].item()) == 4002
    print("rank %d: torch.distributed gathers ok" % rank, flush=True)
    # the C ABI's own communicator: the id travels through torch's store
    from real_amd.matcher import HipMatcher, RealOptions
    m = HipMatcher(RealOptions(seedl=32, seedkmax=2, totalkmax=3, scores=True).normalise(), device=local)
    ids = [HipMatcher.comm_id() if rank == 0 else None]
    dist.broadcast_object_list(ids, src=0)
    m.comm_init(ids[0], rank, world)
    ai = torch.zeros(2001, dtype=torch.int64, device=dev)
    asc = torch.zeros(2001, dtype=torch.float32, device=dev)
    got = m.gather_records(0, info, score, ai, asc)
    assert got == 2001
    if rank == 0:
        assert torch.equal(ai, gi) and torch.equal(asc, gs)
    ah = torch.zeros((4002, 4), dtype=torch.int32, device=dev)
    ao = torch.zeros(2002, dtype=torch.int64, device=dev)
    assert m.gather_hits(0, hits, off, 2 * n, ah, ao) == (2001, 4002)
    if rank == 0:
        assert torch.equal(ah, gh) and torch.equal(ao, go)
    print("rank %d: C ABI gathers (real_hip_gather_records / real_hip_gather_hits) ok" % rank, flush=True)
    dist.barrier()
    m.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
