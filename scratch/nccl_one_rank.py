"""Validates the torch.distributed calls bench.py makes for N > 1 with a one-rank RCCL group (the N > 1 path itself can only be
rehearsed with gloo on a one-GPU box)."""
import os, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
dist.barrier()
t = torch.tensor([1.25], dtype=torch.float64, device="cuda:0")
dist.all_reduce(t, op=dist.ReduceOp.MAX)
print("rccl one-rank ok", float(t.item()), dist.get_backend())
# the records gather used by the eval loop
import sys; sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from ovmono3d_amd.evaluation.distributed import gather_records
rec = torch.arange(96, dtype=torch.float32, device="cuda:0").view(2, 48)
out = gather_records(rec)
print("gather ok", type(out).__name__)
dist.destroy_process_group()
