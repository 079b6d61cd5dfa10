import os, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
t = torch.ones(1024, device="cuda") * 3
dist.all_reduce(t, op=dist.ReduceOp.MAX); dist.barrier()
ops = []  # batched P2P with nobody to talk to: just make sure the API objects build
print("nccl world 1 ok", float(t[0]), torch.cuda.nccl.version() if hasattr(torch.cuda, "nccl") else "")
dist.destroy_process_group()
