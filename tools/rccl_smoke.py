"""One-rank RCCL check of the statistics reduction bench.py uses between ranks (float64 SUM and MAX vectors on the device)."""
import os, sys
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
sys.path.insert(0, ".")
import torch, torch.distributed as dist
import bench
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
from mpc_amd import sharding
sv = torch.tensor([1.0, 2.0], dtype=torch.float64, device="cuda"); mv = torch.tensor([3.0, -1.0], dtype=torch.float64, device="cuda")
dist.all_reduce(sv, op=dist.ReduceOp.SUM); dist.all_reduce(mv, op=dist.ReduceOp.MAX)      # what allreduce_stats issues with more than one rank
dist.barrier()
torch.cuda.synchronize()
print("rccl ok", sv.tolist(), mv.tolist())
dist.destroy_process_group()
