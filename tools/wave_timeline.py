"""dev tool (GPU): when did every wavefront of march_defer start and end?  Needs a build with -DRRT_WAVE_TIMELINE=1 (the
tile-cost word then holds start / end in microseconds).  Prints the number of resident waves over the launch in 0.25 ms
bins, the distribution of wave lifetimes by start time, and the last waves to finish.
    RRT_LIB_OVERRIDE=.../timeline.so python tools/wave_timeline.py [shard] [N] [view]"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import relativisticraytracer_amd as rrt
from relativisticraytracer_amd.sky import synthetic_sky
s = int(sys.argv[1]) if len(sys.argv) > 1 else 0
N = int(sys.argv[2]) if len(sys.argv) > 2 else 8
view = sys.argv[3] if len(sys.argv) > 3 else "default"
V = {"default": ((0.0, 10.0, -60.0), 0.0, -10.0, 1.0), "skimmer": ((4.2, 0.6, 4.2), -90.0, -5.7, 14.0)}[view]
W, H, R = 3840, 2160, 16
cam = rrt.CameraState.from_angles(*V[:3]); t = V[3]
tex = rrt.SkyTexture(synthetic_sky()); fx = rrt.CameraEffects(); nt = rrt.NoiseTable(32.0)
ws = rrt.Workspace(4096 << 20)
order = rrt.TileOrder(); order.set_seeding(False)
prm = rrt.RenderParams(spin=0.9, noise_table=nt.id, workspace=ws.id, path_policy=2, tile_order=order.id, pool_rounds=1)
buf = torch.zeros(H * W * 4 // N + W * 64 * 4, dtype=torch.uint8, device="cuda")
plain = rrt.RenderParams(spin=0.9, noise_table=nt.id, workspace=ws.id, path_policy=2, pool_rounds=1)
for _ in range(2):
    rrt.launch_raymarch_tiles(buf, W, H, R, s, N, t, cam, tex, fx, plain)
torch.cuda.synchronize()
rrt.launch_raymarch_tiles(buf, W, H, R, s, N, t, cam, tex, fx, prm)       # static order (first launch through the object, seeding off)
torch.cuda.synchronize()
word = order.info(arrays=True)["cost"]
st = (word & 0xffff).astype(np.int64); en = (word >> 16).astype(np.int64)
t0 = st.min() if st.max() - st.min() < 40000 else None
if t0 is None:      # wrapped: unwrap around the median
    st = np.where(st < 32768, st + 65536, st); en = np.where(en < 32768, en + 65536, en); t0 = st.min()
en = np.where(en < st, en + 65536, en)
st = (st - t0) / 1000.0; en = (en - t0) / 1000.0
life = en - st
print(f"# shard {s} of {N}, {view}: {len(word)} wave tiles; march_defer spans {en.max():.3f} ms; lifetimes mean {life.mean():.3f} p50 {np.median(life):.3f} p99 {np.percentile(life, 99):.3f} max {life.max():.3f} ms")
edges = np.arange(0.0, en.max() + 0.25, 0.25)
print("# t [ms]   resident waves (of 8192 slots at 8 per SIMD)   started in bin   mean lifetime of those")
for a, b in zip(edges[:-1], edges[1:]):
    mid = 0.5 * (a + b)
    res = int(((st <= mid) & (en > mid)).sum())
    started = (st >= a) & (st < b)
    print(f"{a:5.2f}-{b:5.2f}  {res:6d}   {int(started.sum()):6d}   {life[started].mean() if started.any() else 0:.3f}")
last = np.argsort(-en)[:8]
print("# last to finish: " + ", ".join(f"tile {i}: {st[i]:.2f}->{en[i]:.2f}" for i in last))
