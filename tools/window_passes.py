"""dev tool (GPU): 1000x700 default view a=0.9 through the three-pass path, 10 launches (run under time_passes.sh)."""
import sys, os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import relativisticraytracer_amd as rrt
from relativisticraytracer_amd.sky import synthetic_sky
w, h = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1000, 700)
tex = rrt.SkyTexture(synthetic_sky()); ws = rrt.Workspace(2 << 30)
cam = rrt.CameraState.default(); fx = rrt.CameraEffects(); prm = rrt.RenderParams(spin=0.9, workspace=ws.id, path_policy=2)
out = torch.zeros(h * w * 4, dtype=torch.uint8, device="cuda")
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for rep in range(10):
    e0.record(); rrt.launch_raymarch(out, w, h, 1.0, cam, tex, fx, prm); e1.record(); torch.cuda.synchronize()
print(f"{w}x{h}: {e0.elapsed_time(e1):.3f} ms", ws.stats())
