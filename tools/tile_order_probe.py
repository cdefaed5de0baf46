"""dev tool (GPU): what cost-ordered dispatch (rrt_tile_order: longest wave tiles first, costs from the previous launch) buys a
single 4K frame, per view, and one rank's share of the bench frame.  usage: tile_order_probe.py [view ...]"""
import sys, os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import relativisticraytracer_amd as rrt
from relativisticraytracer_amd.sky import synthetic_sky
VIEWS = {"default": ((0, 10, -60), 0, -10, 1.0), "key1": ((15, 3, -30), -26.6, -5.1, 6.0),
         "grazing": ((35, 0.8, 10), -106, -1.2, 12.0), "key3": ((5, 1.5, 50), -174.3, -1.7, 18.0),
         "skimmer": ((4.2, 0.6, 4.2), -90.0, -5.7, 14.0), "orbit": ((40, 2, 0), -90, 0, 0.0)}
w, h = (int(os.environ.get("RRT_AB_W", 3840)), int(os.environ.get("RRT_AB_H", 2160)))
tex = rrt.SkyTexture(synthetic_sky()); fx = rrt.CameraEffects(); nt = rrt.NoiseTable(32.0)
out = torch.zeros(h * w * 4, dtype=torch.uint8, device="cuda"); ref = torch.zeros_like(out)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)


def timed(fn, n=4):
    fn(); torch.cuda.synchronize(); ts = []
    for _ in range(n):
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return min(ts)


for name in sys.argv[1:] or list(VIEWS):
    pos, yaw, pitch, t = VIEWS[name]
    cam = rrt.CameraState.from_angles(pos, yaw, pitch)
    for label, kw in (("table", dict(noise_table=nt.id)), ("no-vol", dict(volumetrics=0))):
        order = rrt.TileOrder()
        base = timed(lambda: rrt.launch_raymarch(ref, w, h, t, cam, tex, fx, rrt.RenderParams(spin=0.9, **kw)))
        prm = rrt.RenderParams(spin=0.9, tile_order=order.id, **kw)
        srt = timed(lambda: rrt.launch_raymarch(out, w, h, t, cam, tex, fx, prm))       # includes the sort behind every frame
        c = order.info(arrays=True)["cost"].astype("float64")
        print(f"{name:8s} {label:6s}: static order {base:.2f} ms, cost-ordered {srt:.2f} ms ({100 * (srt / base - 1):+.1f} %), same bytes {bool(torch.equal(out, ref))}; "
              f"wave cost max/mean {c.max() / c.mean():.1f}", flush=True)
        order.destroy()
# one rank's share of the default 4K frame (interleaved 16-row tiles), single kernel static / ordered, three-pass for comparison
cam = rrt.CameraState.default(); t = 1.0
ws = rrt.Workspace(3 << 30)
for n in (8, 4, 2):
    rows = rrt.tile_shard_rows(h, 16, 0, n)
    order = rrt.TileOrder()
    base = timed(lambda: rrt.launch_raymarch_tiles(ref, w, h, 16, 0, n, t, cam, tex, fx, rrt.RenderParams(spin=0.9, noise_table=nt.id, path_policy=1)))
    three = timed(lambda: rrt.launch_raymarch_tiles(out, w, h, 16, 0, n, t, cam, tex, fx, rrt.RenderParams(spin=0.9, noise_table=nt.id, workspace=ws.id, path_policy=2)))
    prm = rrt.RenderParams(spin=0.9, noise_table=nt.id, path_policy=1, tile_order=order.id)
    srt = timed(lambda: rrt.launch_raymarch_tiles(out, w, h, 16, 0, n, t, cam, tex, fx, prm))
    print(f"shard 0 of {n}: single kernel static {base:.3f} ms, cost-ordered {srt:.3f} ms ({100 * (srt / base - 1):+.1f} %), three-pass {three:.3f} ms; "
          f"same bytes {bool(torch.equal(out[: rows * w * 4], ref[: rows * w * 4]))}", flush=True)
    order.destroy()
# the reference's own window size, default view: single kernel static / ordered, three-pass
w2, h2 = 1000, 700
o2 = torch.zeros(h2 * w2 * 4, dtype=torch.uint8, device="cuda"); r2 = torch.zeros_like(o2)
order = rrt.TileOrder()
base = timed(lambda: rrt.launch_raymarch(r2, w2, h2, t, cam, tex, fx, rrt.RenderParams(spin=0.9, noise_table=nt.id)))
three = timed(lambda: rrt.launch_raymarch(o2, w2, h2, t, cam, tex, fx, rrt.RenderParams(spin=0.9, noise_table=nt.id, workspace=ws.id, path_policy=2)))
prm = rrt.RenderParams(spin=0.9, noise_table=nt.id, tile_order=order.id)
srt = timed(lambda: rrt.launch_raymarch(o2, w2, h2, t, cam, tex, fx, prm))
print(f"1000x700 default view: single kernel static {base:.3f} ms, cost-ordered {srt:.3f} ms, three-pass {three:.3f} ms; same bytes {bool(torch.equal(o2, r2))}", flush=True)
