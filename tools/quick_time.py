"""Quick kernel timing (dev tool): python tools/quick_time.py [w h spin vol reps]"""
import sys
import torch
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import relativisticraytracer_amd as rrt
from relativisticraytracer_amd.sky import synthetic_sky

w, h = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (3840, 2160)
spin = float(sys.argv[3]) if len(sys.argv) > 3 else 0.9
vol = int(sys.argv[4]) if len(sys.argv) > 4 else 1
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 3
mode = int(sys.argv[6]) if len(sys.argv) > 6 else 0
wsmb = int(sys.argv[7]) if len(sys.argv) > 7 else 0
ws = rrt.Workspace(wsmb << 20) if wsmb else None
tex = rrt.SkyTexture(synthetic_sky())
cam = rrt.CameraState.default(); fx = rrt.CameraEffects(); prm = rrt.RenderParams(spin=spin, volumetrics=vol, arith_mode=mode, workspace=ws.id if ws else 0, path_policy=2)
out = torch.zeros(h * w * 4, dtype=torch.uint8, device="cuda")
rrt.launch_raymarch(out, w, h, 1.0, cam, tex, fx, prm); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for r in range(reps):
    e0.record(); rrt.launch_raymarch(out, w, h, 1.0, cam, tex, fx, prm); e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    print(f"{w}x{h} a={spin} vol={vol} mode={mode} ws={wsmb}MB {ws.stats() if ws else ''}: {ms:.2f} ms  {w*h/ms/1e3:.1f} Mrays/s", flush=True)
