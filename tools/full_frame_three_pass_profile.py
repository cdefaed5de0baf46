"""dev tool (GPU, under rocprofv3 --kernel-trace --stats): a FULL 4K frame of one view through the three-pass path (one chain, big pool)
and through the single kernel, 4 times each."""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import relativisticraytracer_amd as rrt
from relativisticraytracer_amd.sky import synthetic_sky
view = sys.argv[1] if len(sys.argv) > 1 else "key1"
V = {"default": ((0.0, 10.0, -60.0), 0.0, -10.0, 1.0), "skimmer": ((4.2, 0.6, 4.2), -90.0, -5.7, 14.0),
     "key1": ((15.0, 3.0, -30.0), -26.6, -5.1, 6.0)}[view]
W, H = 3840, 2160
cam = rrt.CameraState.from_angles(*V[:3]); t = V[3]
tex = rrt.SkyTexture(synthetic_sky()); fx = rrt.CameraEffects(); nt = rrt.NoiseTable(32.0)
ws = rrt.Workspace(48 << 30)
buf = torch.zeros(H * W * 4, dtype=torch.uint8, device="cuda")
for prm in (rrt.RenderParams(spin=0.9, noise_table=nt.id, workspace=ws.id, path_policy=2, pass_chains=1, pool_rounds=1),
            rrt.RenderParams(spin=0.9, noise_table=nt.id), rrt.RenderParams(spin=0.9, volumetrics=0)):
    for _ in range(4):
        rrt.launch_raymarch(buf, W, H, t, cam, tex, fx, prm)
    torch.cuda.synchronize()
print(ws.stats())
