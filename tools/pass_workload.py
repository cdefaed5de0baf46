"""dev tool (GPU, under rocprofv3): ONE view at 4K through one path, a few times -- the workload of tools/pass_counters.sh.

    pass_workload.py <view> <path> [reps]
    path: single | single_ordered | three_pass (full frame, one chain, pool large enough for one round) |
          shard0of8 (rank 0's share of the frame the way a rank of an 8-GPU run renders it: 16-row tiles t mod 8, 2 GiB pool, two chains)
"""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import relativisticraytracer_amd as rrt
from relativisticraytracer_amd.sky import synthetic_sky
VIEWS = {"default": ((0.0, 10.0, -60.0), 0.0, -10.0, 1.0), "skimmer": ((4.2, 0.6, 4.2), -90.0, -5.7, 14.0),
         "key1": ((15.0, 3.0, -30.0), -26.6, -5.1, 6.0), "grazing": ((35.0, 0.8, 10.0), -106.0, -1.2, 12.0)}
view, path = sys.argv[1], sys.argv[2]
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
W, H = 3840, 2160
pos, yaw, pitch, t = VIEWS[view]
cam = rrt.CameraState.from_angles(pos, yaw, pitch)
tex = rrt.SkyTexture(synthetic_sky()); fx = rrt.CameraEffects(); nt = rrt.NoiseTable(32.0)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ws = order = None
if path in ("shard0of8", "shard0of8_ordered", "shard0of8_one_chain"):
    ws = rrt.Workspace(2 << 30)
    rows = rrt.tile_shard_rows(H, 16, 0, 8)
    buf = torch.zeros(rows * W * 4, dtype=torch.uint8, device="cuda")
    order = rrt.TileOrder() if path == "shard0of8_ordered" else None
    prm = rrt.RenderParams(spin=0.9, noise_table=nt.id, workspace=ws.id, tile_order=order.id if order else 0,
                           pass_chains=1 if path == "shard0of8_one_chain" else 0)
    go = lambda: rrt.launch_raymarch_tiles(buf, W, H, 16, 0, 8, t, cam, tex, fx, prm)
else:
    buf = torch.zeros(H * W * 4, dtype=torch.uint8, device="cuda")
    if path == "three_pass":
        ws = rrt.Workspace(int(os.environ.get("RRT_POOL_GIB", "48")) << 30)
        prm = rrt.RenderParams(spin=0.9, noise_table=nt.id, workspace=ws.id, path_policy=2, pass_chains=int(os.environ.get("RRT_CHAINS", "1")),
                               pool_rounds=int(os.environ.get("RRT_ROUNDS", "1")))
    elif path == "single_ordered":
        order = rrt.TileOrder()
        prm = rrt.RenderParams(spin=0.9, noise_table=nt.id, tile_order=order.id)
    else:
        prm = rrt.RenderParams(spin=0.9, noise_table=nt.id)
    go = lambda: rrt.launch_raymarch(buf, W, H, t, cam, tex, fx, prm)
for r in range(reps):
    e0.record(); go(); e1.record(); torch.cuda.synchronize()
    print(f"{view} {path}: {e0.elapsed_time(e1):.3f} ms", flush=True)
if ws:
    print(ws.stats())
