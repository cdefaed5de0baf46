"""dev tool (GPU): 4K frame time of both arithmetic modes on several views (path keyframes)."""
import sys
import torch
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import relativisticraytracer_amd as rrt
from relativisticraytracer_amd.sky import synthetic_sky
w, h = 3840, 2160
tex = rrt.SkyTexture(synthetic_sky())
views = {"default (0,10,-60) t=1": ((0, 10, -60), 0, -10, 1.0), "path0 key1 (15,3,-30) t=6": ((15, 3, -30), -26.6, -5.1, 6.0),
         "grazing (35,0.8,10) t=12": ((35, 0.8, 10), -106, -1.2, 12.0), "path0 key3 (5,1.5,50) t=18": ((5, 1.5, 50), -174.3, -1.7, 18.0),
         "skimmer (4.2,0.6,4.2) t=14": ((4.2, 0.6, 4.2), -90.0, -5.7, 14.0), "orbit (40,2,0) t=0": ((40, 2, 0), -90, 0, 0.0)}
out = torch.zeros(h * w * 4, dtype=torch.uint8, device="cuda")
nt = rrt.NoiseTable(32.0)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for name, (pos, yaw, pitch, t) in views.items():
    cam = rrt.CameraState.from_angles(pos, yaw, pitch); fx = rrt.CameraEffects()
    res = []
    for mode in (0, 1):
        for vol in (1, 0, 2):
            prm = rrt.RenderParams(spin=0.9, arith_mode=mode, volumetrics=1 if vol else 0, noise_table=nt.id if vol == 2 else 0)
            rrt.launch_raymarch(out, w, h, t, cam, tex, fx, prm); torch.cuda.synchronize()
            e0.record(); rrt.launch_raymarch(out, w, h, t, cam, tex, fx, prm); e1.record(); torch.cuda.synchronize()
            res.append(e0.elapsed_time(e1))
    print(f"{name:32s} strict {res[0]:7.2f} ms (no-vol {res[1]:6.2f}, table {res[2]:6.2f})   fast {res[3]:7.2f} ms (no-vol {res[4]:6.2f}, table {res[5]:6.2f})", flush=True)
