"""dev tool (GPU): what the FIRST frame of a view costs -- static dispatch order | first launch through an rrt_tile_order
object (order from the coarse probe; probe + sort inside the time; buffers sized beforehand by a launch of another
geometry) | steady state (order from the previous frame's measured costs).  4K, a = 0.9, noise tables.
    python tools/first_frame.py   -> profiles/r04_first_frame_order.txt"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import relativisticraytracer_amd as rrt
from relativisticraytracer_amd.sky import synthetic_sky
W, H = 3840, 2160
VIEWS = {"default": ((0.0, 10.0, -60.0), 0.0, -10.0, 1.0), "key1": ((15.0, 3.0, -30.0), -26.6, -5.1, 6.0),
         "grazing": ((35.0, 0.8, 10.0), -106.0, -1.2, 12.0), "key3": ((5.0, 1.5, 50.0), -174.3, -1.7, 18.0),
         "skimmer": ((4.2, 0.6, 4.2), -90.0, -5.7, 14.0), "orbit": ((40.0, 2.0, 0.0), -90.0, 0.0, 3.0)}
tex = rrt.SkyTexture(synthetic_sky()); fx = rrt.CameraEffects(); nt = rrt.NoiseTable(32.0)
buf = torch.zeros((H + 8) * W * 4, dtype=torch.uint8, device="cuda")
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
def once(fn):
    torch.cuda.synchronize(); e0.record(); fn(); e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)
print("# 4K a=0.9, noise tables, single kernel; ms per frame (min of 3)")
print("# view      static order | first frame, probe-seeded order (probe + sort included) | steady state, measured order")
for name, (pos, yaw, pitch, t) in VIEWS.items():
    cam = rrt.CameraState.from_angles(pos, yaw, pitch)
    plain = rrt.RenderParams(spin=0.9, noise_table=nt.id)
    once(lambda: rrt.launch_raymarch(buf, W, H, t, cam, tex, fx, plain))
    static = min(once(lambda: rrt.launch_raymarch(buf, W, H, t, cam, tex, fx, plain)) for _ in range(3))
    first = []
    for _ in range(3):
        o = rrt.TileOrder()
        p = rrt.RenderParams(spin=0.9, noise_table=nt.id, tile_order=o.id)
        rrt.launch_raymarch(buf, W, H + 8, t, cam, tex, fx, p)       # another geometry: sizes the buffers, leaves no history for W x H
        first.append(once(lambda: rrt.launch_raymarch(buf, W, H, t, cam, tex, fx, p)))
        assert o.seeded_launches() == 2
        steady = min(once(lambda: rrt.launch_raymarch(buf, W, H, t, cam, tex, fx, p)) for _ in range(3))
        o.destroy()
    print(f"{name:8s}  {static:7.3f} | {min(first):7.3f} ({(min(first) / static - 1) * 100:+5.1f} %) | {steady:7.3f} ({(steady / static - 1) * 100:+5.1f} %)", flush=True)
