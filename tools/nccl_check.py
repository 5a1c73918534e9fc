import os, torch, torch.distributed as dist
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", device_id=dev)
t = torch.ones(4, device=dev, dtype=torch.float64)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
dist.barrier()
v = torch.arange(4, device=dev, dtype=torch.float32)
out = [torch.zeros_like(v)]
dist.all_gather(out, v)
torch.cuda.synchronize()
print("nccl world=1 ok", t.tolist(), out[0].tolist())
dist.destroy_process_group()
