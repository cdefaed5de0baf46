"""dev tool (GPU): three-pass path on one view, with or without the noise table -> run under
rocprofv3 --kernel-trace --stats to read the time of eval_sample_rows (pure media evaluation).
usage: eval_times.py <view> <table 0|1> [w h] [pool_gib]"""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import relativisticraytracer_amd as rrt
from relativisticraytracer_amd.sky import synthetic_sky
VIEWS = {"default": ((0, 10, -60), 0, -10, 1.0), "key1": ((15, 3, -30), -26.6, -5.1, 6.0),
         "grazing": ((35, 0.8, 10), -106, -1.2, 12.0), "key3": ((5, 1.5, 50), -174.3, -1.7, 18.0),
         "skimmer": ((4.2, 0.6, 4.2), -90.0, -5.7, 14.0), "orbit": ((40, 2, 0), -90, 0, 0.0)}
name = sys.argv[1]; tab = int(sys.argv[2])
w, h = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (3840, 2160)
gib = int(sys.argv[5]) if len(sys.argv) > 5 else 24
pos, yaw, pitch, t = VIEWS[name]
tex = rrt.SkyTexture(synthetic_sky())
cam = rrt.CameraState.from_angles(pos, yaw, pitch); fx = rrt.CameraEffects()
ws = rrt.Workspace(gib << 30)
nt = rrt.NoiseTable(32.0) if tab else None
prm = rrt.RenderParams(spin=0.9, workspace=ws.id, path_policy=2, noise_table=nt.id if nt else 0)
out = torch.zeros(h * w * 4, dtype=torch.uint8, device="cuda")
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for r in range(3):
    e0.record(); rrt.launch_raymarch(out, w, h, t, cam, tex, fx, prm); e1.record(); torch.cuda.synchronize()
    print(f"{name} {w}x{h} table={tab}: {e0.elapsed_time(e1):.2f} ms  {ws.stats()}", flush=True)
