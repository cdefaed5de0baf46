"""dev tool (GPU): 4K, a = 0.9 kernel times of the named views -- strict with arithmetic noise / without media / with
the noise tables (static and cost-ordered dispatch), and the fast mode the same way (the table of DESIGN.md section 4)."""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import relativisticraytracer_amd as rrt
from relativisticraytracer_amd.sky import synthetic_sky
VIEWS = [("default (0,10,-60) t=1", (0, 10, -60), 0, -10, 1.0), ("path0 key1 (15,3,-30) t=6", (15, 3, -30), -26.6, -5.1, 6.0),
         ("grazing (35,0.8,10) t=12", (35, 0.8, 10), -106, -1.2, 12.0), ("path0 key3 (5,1.5,50) t=18", (5, 1.5, 50), -174.3, -1.7, 18.0),
         ("skimmer (4.2,0.6,4.2) t=14", (4.2, 0.6, 4.2), -90.0, -5.7, 14.0), ("orbit (40,2,0) t=0", (40, 2, 0), -90, 0, 0.0)]
w, h = 3840, 2160
tex = rrt.SkyTexture(synthetic_sky()); fx = rrt.CameraEffects(); nt = rrt.NoiseTable(32.0)
out = torch.zeros(h * w * 4, dtype=torch.uint8, device="cuda")
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)


def t(prm, cam, tt):
    rrt.launch_raymarch(out, w, h, tt, cam, tex, fx, prm); torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        e0.record(); rrt.launch_raymarch(out, w, h, tt, cam, tex, fx, prm); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return min(ts)


for name, pos, yaw, pitch, tt in VIEWS:
    cam = rrt.CameraState.from_angles(pos, yaw, pitch)
    r = {}
    for mode in (0, 1):
        r[mode] = (t(rrt.RenderParams(spin=0.9, arith_mode=mode), cam, tt),
                   t(rrt.RenderParams(spin=0.9, arith_mode=mode, volumetrics=0), cam, tt),
                   t(rrt.RenderParams(spin=0.9, arith_mode=mode, noise_table=nt.id), cam, tt))
    order = rrt.TileOrder()       # table + cost-ordered dispatch (rrt_tile_order; the sort behind every frame inside the time)
    prm = rrt.RenderParams(spin=0.9, noise_table=nt.id, tile_order=order.id)
    rrt.launch_raymarch(out, w, h, tt, cam, tex, fx, prm)
    ordered = t(prm, cam, tt)
    order.destroy()
    print(f"{name:32s} strict {r[0][0]:7.2f} ms (no-vol {r[0][1]:6.2f}, table {r[0][2]:6.2f}, table + cost-ordered {ordered:6.2f})   "
          f"fast {r[1][0]:7.2f} ms (no-vol {r[1][1]:6.2f}, table {r[1][2]:6.2f})", flush=True)
