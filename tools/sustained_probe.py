"""dev tool (GPU): sustained time per frame of ONE rank's share (shard 0 of 8, 16-row tiles, three-pass, two chains) of a 4K view with
1 ... 6 frames in flight (own stream + own share of a 16 GiB pool each), against the rank's fair share of the best single-GPU frame.
RRT_POLICY=1: the single kernel (media in line) instead of the three-pass path; RRT_CHAINS=1: one chain per launch; RRT_ORDER=1: cost-ordered dispatch, one rrt_tile_order object per frame in flight.
usage: sustained_probe.py [view] [shard]"""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import relativisticraytracer_amd as rrt
from relativisticraytracer_amd.sky import synthetic_sky
VIEWS = {"default": ((0.0, 10.0, -60.0), 0.0, -10.0, 1.0), "skimmer": ((4.2, 0.6, 4.2), -90.0, -5.7, 14.0),
         "key1": ((15.0, 3.0, -30.0), -26.6, -5.1, 6.0), "grazing": ((35.0, 0.8, 10.0), -106.0, -1.2, 12.0)}
view = sys.argv[1] if len(sys.argv) > 1 else "skimmer"
sh = int(sys.argv[2]) if len(sys.argv) > 2 else 0
W, H, R, N = 3840, 2160, 16, int(os.environ.get("RRT_N", "8"))
pos, yaw, pitch, t = VIEWS[view]
cam = rrt.CameraState.from_angles(pos, yaw, pitch)
tex = rrt.SkyTexture(synthetic_sky()); fx = rrt.CameraEffects(); nt = rrt.NoiseTable(32.0)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
full = torch.zeros(H * W * 4, dtype=torch.uint8, device="cuda")


def timed(fn, reps=3):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best


o = rrt.TileOrder()
single = min(timed(lambda: rrt.launch_raymarch(full, W, H, t, cam, tex, fx, rrt.RenderParams(spin=0.9, noise_table=nt.id))),
             timed(lambda: rrt.launch_raymarch(full, W, H, t, cam, tex, fx, rrt.RenderParams(spin=0.9, noise_table=nt.id, tile_order=o.id)), reps=4))
print(f"{view}: best single-GPU frame {single:.3f} ms -> a rank's fair share {single / N:.3f} ms", flush=True)
rows = rrt.tile_shard_rows(H, R, sh, N)
for slots in (1, 2, 3, 4, 6, 8):
    pools = [rrt.Workspace((16 << 30) // slots) for _ in range(slots)]
    streams = [torch.cuda.Stream() for _ in range(slots)]
    bufs = [torch.zeros(rows * W * 4, dtype=torch.uint8, device="cuda") for _ in range(slots)]
    orders = [rrt.TileOrder() for _ in range(slots)] if os.environ.get("RRT_ORDER", "0") == "1" else None
    prms = [rrt.RenderParams(spin=0.9, noise_table=nt.id, workspace=pools[j].id, path_policy=int(os.environ.get("RRT_POLICY", "2")),
                             tile_order=orders[j].id if orders else 0,
                             pass_chains=int(os.environ.get("RRT_CHAINS", "0")), pool_rounds=int(os.environ.get("RRT_ROUNDS", "0"))) for j in range(slots)]
    frames = 4 * slots if slots > 1 else 6

    def burst():
        cur = torch.cuda.current_stream()
        for s in streams:
            s.wait_stream(cur)
        for k in range(frames):
            rrt.launch_raymarch_tiles(bufs[k % slots], W, H, R, sh, N, t, cam, tex, fx, prms[k % slots], stream=streams[k % slots])
        for s in streams:
            cur.wait_stream(s)
    ms = timed(burst) / frames
    print(f"{view} shard {sh}: {slots} in flight: {ms:.3f} ms per frame = {single / ms:.2f}x  (efficiency {single / N / ms:.2f});  {pools[0].stats()}", flush=True)
    for p in pools + (orders or []):
        p.destroy()
