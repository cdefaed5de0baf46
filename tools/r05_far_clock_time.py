"""dev tool (GPU): the 4K frame from inside the disk (Horizon Skimmer key) at t = 14 s through the dense [0, 32] table and at
t = 500 s through what rrt_noise_table_fit_window(495, 505, 2 GiB) gives (round 5: the banded layout at FULL coverage), each also
with arithmetic noise; static and cost-ordered dispatch."""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import relativisticraytracer_amd as rrt
from relativisticraytracer_amd.sky import synthetic_sky
W, H = 3840, 2160
cam = rrt.CameraState.from_angles((4.2, 0.6, 4.2), -90.0, -5.7)
tex = rrt.SkyTexture(synthetic_sky()); fx = rrt.CameraEffects()
buf = torch.zeros(H * W * 4, dtype=torch.uint8, device="cuda")
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

def timed(prm, t, reps=6):
    ts = []
    for r in range(reps):
        e0.record(); rrt.launch_raymarch(buf, W, H, t, cam, tex, fx, prm); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return min(ts[2:]), sorted(ts[2:])[len(ts[2:]) // 2]

for t, (a, b) in ((14.0, (0.0, 32.0)), (500.0, (495.0, 505.0)), (1000.0, (995.0, 1005.0))):
    t1, cov, nbytes = rrt.NoiseTable.fit(a, b, 2 << 30)
    nt = rrt.NoiseTable.window(a, t1, cov)
    info = nt.info()
    order = rrt.TileOrder()
    res = {"arithmetic": timed(rrt.RenderParams(spin=0.9), t),
           "table": timed(rrt.RenderParams(spin=0.9, noise_table=nt.id), t),
           "table, cost-ordered": timed(rrt.RenderParams(spin=0.9, noise_table=nt.id, tile_order=order.id), t)}
    print(f"t = {t:6.1f}  window [{a:g}, {t1:g}] coverage {info['coverage'] & 15} {'banded' if info['coverage'] & 16 else 'dense'} {info['bytes'] / 1e6:.0f} MB: " +
          "  ".join(f"{k} {v[0]:.2f} (median {v[1]:.2f}) ms" for k, v in res.items()), flush=True)
    order.destroy(); nt.destroy()
