"""dev tool (GPU box): can two RCCL ranks share one device here?  (If so the real N > 1 path can be rehearsed.)"""
import os, sys, torch, torch.distributed as dist
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
try:
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
    x = torch.full((1024,), rank, dtype=torch.uint8, device="cuda")
    outs = [torch.zeros_like(x) for _ in range(world)] if rank == 0 else None
    dist.gather(x, outs, dst=0)
    torch.cuda.synchronize()
    if rank == 0:
        print("same-device RCCL gather ok:", [int(o[0]) for o in outs], flush=True)
    dist.destroy_process_group()
except Exception as e:
    print(f"rank {rank}: {type(e).__name__}: {str(e)[:300]}", flush=True)
    sys.exit(3)
