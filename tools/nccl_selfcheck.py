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

# the pipelined FrameSharder step (async gather under the next render) through the same group
import sys; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import relativisticraytracer_amd as rrt
from relativisticraytracer_amd import sharding
from relativisticraytracer_amd.sky import synthetic_sky
w, h, R = 640, 360, 16
tex = rrt.SkyTexture(synthetic_sky()); cam = rrt.CameraState.default(); fx = rrt.CameraEffects()
ws = rrt.Workspace(1 << 30); prm = rrt.RenderParams(spin=0.9, workspace=ws.id)
times = [1.0, 3.0, 5.0, 7.0]; n = {"i": 0}
pools = [ws, rrt.Workspace(1 << 30)]; prms = [rrt.RenderParams(spin=0.9, workspace=p.id) for p in pools]
def render(buf, slot):
    rrt.launch_raymarch_tiles(buf, w, h, R, 0, 1, times[n["i"]], cam, tex, fx, prms[slot]); n["i"] += 1
fs = sharding.FrameSharder(w, h, R, 0, 1, dev, render, None, pipeline=True, collective_at_world1=True,
                           assemble_all=lambda f, b, st: rrt.assemble_all_tiles(f, b, st, w, h, R, 1))
got = []
for k in range(len(times)):
    f = fs.step()
    if f is not None: got.append(f.clone())
got.append(fs.flush().clone())
ok = len(got) == len(times)
ref = torch.zeros(h * w * 4, dtype=torch.uint8, device=dev)
for k, t in enumerate(times):
    rrt.launch_raymarch(ref, w, h, t, cam, tex, fx, rrt.RenderParams(spin=0.9)); torch.cuda.synchronize()
    ok = ok and bool(torch.equal(ref, got[k]))
print("pipelined sharder over rccl (1 rank):", ok)

# step rate of one rank's share (shard 0 of 8 of the 4K bench frame) through the same machinery, pipelined or not
import time
W, H = 3840, 2160
big = [rrt.Workspace(3 << 30), rrt.Workspace(3 << 30)]
bprm = [rrt.RenderParams(spin=0.9, workspace=p.id) for p in big]
def render8(buf, slot):
    rrt.launch_raymarch_tiles(buf, W, H, 16, 0, 8, 1.0, cam, tex, fx, bprm[slot])
for pipe in (False, True):
    fs8 = sharding.FrameSharder(W, H // 8, 16, 0, 1, dev, render8, None, pipeline=pipe, collective_at_world1=True,
                                assemble_all=lambda f, b, st: rrt.assemble_all_tiles(f, b, st, W, H // 8, 16, 1))
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for k in range(24):
            fs8.step()
        fs8.flush(); torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 24 * 1e3
    print(f"1/8 of the 4K frame through FrameSharder over rccl, pipeline={pipe}: {dt:.3f} ms/step")
dist.destroy_process_group()
