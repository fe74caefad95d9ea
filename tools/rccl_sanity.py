import os, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR","127.0.0.1"); os.environ.setdefault("MASTER_PORT","29533")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda",0))
t = torch.arange(8, dtype=torch.float64, device="cuda")
dist.all_reduce(t); dist.barrier(); torch.cuda.synchronize()
print("rccl ok", t.sum().item())
dist.destroy_process_group()
