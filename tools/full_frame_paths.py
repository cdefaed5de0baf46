"""dev tool (GPU): FULL 4K frames through the single kernel (static order / cost-ordered steady state) and through the three-pass
path (two chains, a pool large enough for one round) -- is there a view where the three-pass path wins at full size?"""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import relativisticraytracer_amd as rrt
from relativisticraytracer_amd.sky import synthetic_sky
W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (3840, 2160)
VIEWS = {"default": ((0.0, 10.0, -60.0), 0.0, -10.0, 1.0), "key1": ((15.0, 3.0, -30.0), -26.6, -5.1, 6.0),
         "grazing": ((35.0, 0.8, 10.0), -106.0, -1.2, 12.0), "key3": ((5.0, 1.5, 50.0), -174.3, -1.7, 18.0),
         "skimmer": ((4.2, 0.6, 4.2), -90.0, -5.7, 14.0), "orbit": ((40.0, 2.0, 0.0), -90.0, 0.0, 3.0)}
tex = rrt.SkyTexture(synthetic_sky()); fx = rrt.CameraEffects(); nt = rrt.NoiseTable(32.0)
ws = rrt.Workspace(int(os.environ.get("RRT_POOL_GIB", "48")) << 30)
buf = torch.zeros(H * W * 4, dtype=torch.uint8, device="cuda")
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
def timed(prm, cam, t, reps=3):
    for _ in range(2):
        rrt.launch_raymarch(buf, W, H, t, cam, tex, fx, prm)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        e0.record(); rrt.launch_raymarch(buf, W, H, t, cam, tex, fx, prm); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best
print(f"# {W}x{H} a=0.9, noise tables; ms per frame: single kernel static | single kernel cost-ordered | three-pass two chains | three-pass one chain   (pool rows used, rounds)")
for name, (pos, yaw, pitch, t) in VIEWS.items():
    cam = rrt.CameraState.from_angles(pos, yaw, pitch)
    a = timed(rrt.RenderParams(spin=0.9, noise_table=nt.id), cam, t)
    o = rrt.TileOrder()
    b = timed(rrt.RenderParams(spin=0.9, noise_table=nt.id, tile_order=o.id), cam, t)
    o.destroy()
    c = timed(rrt.RenderParams(spin=0.9, noise_table=nt.id, workspace=ws.id, path_policy=2), cam, t)
    st = ws.stats()
    d = timed(rrt.RenderParams(spin=0.9, noise_table=nt.id, workspace=ws.id, path_policy=2, pass_chains=1), cam, t)
    print(f"{name:8s} {a:7.2f} | {b:7.2f} | {c:7.2f} | {d:7.2f}   ({st['rows_used']} rows, {st['rounds_with_work']} round(s), {st['overflow_waves']} fall-backs)", flush=True)
