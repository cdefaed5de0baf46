"""dev tool (GPU box): the torch.distributed calls bench.py makes, on a 1-rank RCCL group."""
import os, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
x = torch.arange(4096, dtype=torch.uint8, device=dev)
out = [torch.zeros_like(x)]
dist.gather(x, out, dst=0)
tt = torch.tensor([1.5], device=dev, dtype=torch.float64); dist.all_reduce(tt, op=dist.ReduceOp.MAX)
big = torch.zeros(1 * 4096, dtype=torch.uint8, device=dev); dist.all_gather_into_tensor(big, x)
dist.barrier(); torch.cuda.synchronize()
print("rccl ok:", bool(torch.equal(out[0], x)), float(tt), bool(torch.equal(big, x)), dist.get_backend())
dist.destroy_process_group()
