"""One rank, RCCL backend: exercises the torch.distributed calls of the slab layer that a world of one normally skips
(all_gather_into_tensor on fp64 / fp32 device tensors, barrier, all_reduce MAX) -- an API-usage check for the N > 1 path."""
import os, sys
import torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
t = torch.tensor([1.5, 2.5], dtype=torch.float64, device="cuda")
out = torch.empty((1, 2), dtype=torch.float64, device="cuda")
dist.all_gather_into_tensor(out, t)
m = torch.tensor([3.0], dtype=torch.float64, device="cuda"); dist.all_reduce(m, op=dist.ReduceOp.MAX)
lst = [torch.empty_like(t)]; dist.all_gather(lst, t)
dist.barrier()
torch.cuda.synchronize()
print("rccl world-1 collectives ok:", out.tolist(), m.item(), lst[0].tolist())
dist.destroy_process_group()
