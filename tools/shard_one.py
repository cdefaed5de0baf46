"""dev tool (GPU): render shard `s` of N of the 4K bench frame K times through the three-pass path (for rocprofv3 --kernel-trace --stats)."""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import relativisticraytracer_amd as rrt
from relativisticraytracer_amd.sky import synthetic_sky
s = int(sys.argv[1]) if len(sys.argv) > 1 else 0
N = int(sys.argv[2]) if len(sys.argv) > 2 else 8
K = int(sys.argv[3]) if len(sys.argv) > 3 else 10
view = sys.argv[4] if len(sys.argv) > 4 else "default"
order_on = len(sys.argv) > 5 and sys.argv[5] == "order"
chains = int(os.environ.get("RRT_CHAINS", "0"))
V = {"default": ((0.0, 10.0, -60.0), 0.0, -10.0, 1.0), "skimmer": ((4.2, 0.6, 4.2), -90.0, -5.7, 14.0),
     "key1": ((15.0, 3.0, -30.0), -26.6, -5.1, 6.0), "grazing": ((35.0, 0.8, 10.0), -106.0, -1.2, 12.0)}[view]
W, H, R = 3840, 2160, 16
cam = rrt.CameraState.from_angles(*V[:3]); t = V[3]
tex = rrt.SkyTexture(synthetic_sky()); fx = rrt.CameraEffects(); nt = rrt.NoiseTable(32.0)
ws = rrt.Workspace(2048 << 20)
order = rrt.TileOrder() if order_on else None
prm = rrt.RenderParams(spin=0.9, noise_table=nt.id, workspace=ws.id, path_policy=2, tile_order=order.id if order else 0, pass_chains=chains)
buf = torch.zeros(H * W * 4 // N + W * 64 * 4, dtype=torch.uint8, device="cuda")
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for k in range(K):
    e0.record(); rrt.launch_raymarch_tiles(buf, W, H, R, s, N, t, cam, tex, fx, prm); e1.record(); torch.cuda.synchronize()
    print(f"frame {k}: {e0.elapsed_time(e1):.3f} ms  {ws.stats()}")
