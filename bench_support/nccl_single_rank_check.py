import os, sys
sys.path.insert(0, os.getcwd())
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
import torch, torch.distributed as dist
dist.init_process_group("nccl")
torch.cuda.set_device(0)
from real_amd.distributed import RecordGatherer, gather_records
n = 1000
info = torch.arange(n, dtype=torch.int64, device="cuda")
score = torch.arange(n, dtype=torch.float32, device="cuda")
g = RecordGatherer(n, torch.device("cuda", 0), scores=True)
for k in range(3):
    g.wait(k % 2)
    g.start(k % 2, info + k, score + k)
g.wait_all()
torch.cuda.synchronize()
assert torch.equal(g.info_all, info + 2) and torch.equal(g.score_all, score + 2)
gi, gs = gather_records(info, score)
assert torch.equal(gi, info)
dist.barrier(); dist.destroy_process_group()
print("nccl single-rank gather ok")
