"""dev tool (GPU, under rocprofv3 --kernel-trace --stats): does pass 2's time per row depend on how spatially compact the launch is?
An eighth of a 4K view through the three-pass path (one chain, ample pool) as (band) one contiguous band of rows around the middle
or (tiles) the interleaved 16-row tiles t mod 8 == 4.  usage: eval_locality.py <view> band|tiles"""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import relativisticraytracer_amd as rrt
from relativisticraytracer_amd.sky import synthetic_sky
VIEWS = {"default": ((0.0, 10.0, -60.0), 0.0, -10.0, 1.0), "skimmer": ((4.2, 0.6, 4.2), -90.0, -5.7, 14.0),
         "key1": ((15.0, 3.0, -30.0), -26.6, -5.1, 6.0), "grazing": ((35.0, 0.8, 10.0), -106.0, -1.2, 12.0)}
view, mode = sys.argv[1], sys.argv[2]
W, H = 3840, 2160
pos, yaw, pitch, t = VIEWS[view]
cam = rrt.CameraState.from_angles(pos, yaw, pitch)
tex = rrt.SkyTexture(synthetic_sky()); fx = rrt.CameraEffects(); nt = rrt.NoiseTable(32.0)
ws = rrt.Workspace(8 << 30)
prm = rrt.RenderParams(spin=0.9, noise_table=nt.id, workspace=ws.id, path_policy=2, pass_chains=1, pool_rounds=1)
buf = torch.zeros(272 * W * 4, dtype=torch.uint8, device="cuda")
for _ in range(4):
    if mode == "band":
        rrt.launch_raymarch_rows(buf, W, H, 944, 944 + 272, t, cam, tex, fx, prm)
    else:
        rrt.launch_raymarch_tiles(buf, W, H, 16, 4, 8, t, cam, tex, fx, prm)
    torch.cuda.synchronize()
print(mode, ws.stats())
