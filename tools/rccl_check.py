"""One-rank RCCL rehearsal of bench.py's exchange (GPU box has one device): process group over nccl (= RCCL), the all-gather of
sharding.all_gather_times with device tensors, barrier.  python3 tools/rccl_check.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29517")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
import torch, torch.distributed as dist
from dsurftomo_amd import sharding
torch.cuda.set_device(0)
t0 = time.time()
dist.init_process_group(backend="nccl", device_id=torch.device("cuda", 0))
mine = torch.arange(512000, dtype=torch.float32, device="cuda")
full = sharding.all_gather_times(dist, mine, [mine.numel()])
dist.barrier(); torch.cuda.synchronize()
print("RCCL one-rank all-gather ok: %d values, equal %s, %.2f s incl. init" % (full.numel(), bool(torch.equal(full, mine)), time.time() - t0))
dist.destroy_process_group()
