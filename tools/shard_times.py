"""dev tool (GPU): kernel time of one rank's share of the 4K bench frame for N = 1, 2, 4, 8 (strong-scaling tail)."""
import sys
import torch
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import relativisticraytracer_amd as rrt
from relativisticraytracer_amd.sky import synthetic_sky
w, h = 3840, 2160
R = int(sys.argv[1]) if len(sys.argv) > 1 else 16
wsmb = int(sys.argv[2]) if len(sys.argv) > 2 else 0
ws = rrt.Workspace(wsmb << 20) if wsmb else None
tex = rrt.SkyTexture(synthetic_sky())
nt = rrt.NoiseTable(32.0) if os.environ.get('RRT_TOOL_TABLE', '1') == '1' else None
cam = rrt.CameraState.default(); fx = rrt.CameraEffects(); prm = rrt.RenderParams(spin=0.9, workspace=ws.id if ws else 0, path_policy=2, noise_table=nt.id if nt else 0)
buf = torch.zeros(h * w * 4, dtype=torch.uint8, device="cuda")
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
base = None
for n in (1, 2, 4, 8):
    ts = []
    for s in range(n):
        rrt.launch_raymarch_tiles(buf, w, h, R, s, n, 1.0, cam, tex, fx, prm); torch.cuda.synchronize()
        e0.record(); rrt.launch_raymarch_tiles(buf, w, h, R, s, n, 1.0, cam, tex, fx, prm); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    if n == 1: base = ts[0]
    print(f"R={R} ws={wsmb}MB N={n}: max shard {max(ts):.3f} ms  min {min(ts):.3f}  ideal {base / n:.3f}  kernel-level efficiency {base / n / max(ts):.3f}", flush=True)
