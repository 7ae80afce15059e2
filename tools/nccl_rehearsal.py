"""The collective calls bench.py makes at N > 1 (RCCL init with device_id, barrier, MAX all-reduce of the elapsed
time), run with however many ranks torchrun starts — one rank on a one-GPU box checks that they work at all."""
import os, torch, torch.distributed as dist
local = int(os.environ.get("LOCAL_RANK", "0"))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
torch.cuda.set_device(local)
dist.init_process_group("nccl", device_id=torch.device("cuda", local))
dist.barrier()
torch.cuda.synchronize()
t = torch.tensor([1.5 + dist.get_rank()], dtype=torch.float64, device="cuda")
dist.all_reduce(t, op=dist.ReduceOp.MAX)
print("rank", dist.get_rank(), "of", dist.get_world_size(), "max", float(t))
dist.destroy_process_group()
