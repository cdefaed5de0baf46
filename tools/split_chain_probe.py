"""dev experiment (GPU): does splitting one rank's share of a frame into a heavy and a light half, rendered as two
three-pass chains on two streams (high / low priority), hide the march's tail and the evaluation pass of the heavy half
under the light half's march?  Shard s of N (t mod N) of the 4K frame; heavy = the tiles nearest the image centre."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import relativisticraytracer_amd as rrt
from relativisticraytracer_amd.sky import synthetic_sky
s = int(sys.argv[1]) if len(sys.argv) > 1 else 0
N = int(sys.argv[2]) if len(sys.argv) > 2 else 8
view = sys.argv[3] if len(sys.argv) > 3 else "default"
frac = float(sys.argv[4]) if len(sys.argv) > 4 else 0.5
C = int(sys.argv[5]) if len(sys.argv) > 5 else 0          # > 0: C equal chunks by cost rank instead of the heavy / light split
V = {"default": ((0.0, 10.0, -60.0), 0.0, -10.0, 1.0), "skimmer": ((4.2, 0.6, 4.2), -90.0, -5.7, 14.0)}[view]
W, H, R = 3840, 2160, 16
cam = rrt.CameraState.from_angles(*V[:3]); t = V[3]
tex = rrt.SkyTexture(synthetic_sky()); fx = rrt.CameraEffects(); nt = rrt.NoiseTable(32.0)
n_tiles = (H + R - 1) // R
mine = [tt for tt in range(n_tiles) if tt % N == s]
cost = rrt.probe_tile_costs(W, H, R, t, cam, fx, rrt.RenderParams(spin=0.9))
by_cost = sorted(mine, key=lambda tt: -cost[tt])
heavy = set(by_cost[:max(1, int(round(len(mine) * frac)))])
assign = np.array([2 * (tt % N) + (0 if (tt % N != s or tt in heavy) else 1) for tt in range(n_tiles)], np.int32)
tm2 = rrt.TileMap(H, R, 2 * N, assign)
tm1 = rrt.TileMap(H, R, N, (np.arange(n_tiles) % N).astype(np.int32))
wsA, wsB, ws1 = rrt.Workspace(1024 << 20), rrt.Workspace(1024 << 20), rrt.Workspace(2048 << 20)
bufA = torch.zeros(tm2.max_shard_rows() * W * 4, dtype=torch.uint8, device="cuda"); bufB = torch.zeros_like(bufA)
buf1 = torch.zeros(tm1.max_shard_rows() * W * 4, dtype=torch.uint8, device="cuda")
def prm(ws): return rrt.RenderParams(spin=0.9, noise_table=nt.id, workspace=ws.id, path_policy=2)
hi, lo = torch.cuda.Stream(priority=-1), torch.cuda.Stream(priority=0)
eq1, eq2 = torch.cuda.Stream(), torch.cuda.Stream()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
def timed(fn, reps=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return min(ts), float(np.median(ts))
def one(): rrt.launch_raymarch_tilemap(buf1, W, H, tm1, s, t, cam, tex, fx, prm(ws1))
def two(sa, sb):
    cur = torch.cuda.current_stream()
    sa.wait_stream(cur); sb.wait_stream(cur)
    rrt.launch_raymarch_tilemap(bufA, W, H, tm2, 2 * s, t, cam, tex, fx, prm(wsA), stream=sa)
    rrt.launch_raymarch_tilemap(bufB, W, H, tm2, 2 * s + 1, t, cam, tex, fx, prm(wsB), stream=sb)
    cur.wait_stream(sa); cur.wait_stream(sb)
def seq():
    rrt.launch_raymarch_tilemap(bufA, W, H, tm2, 2 * s, t, cam, tex, fx, prm(wsA))
    rrt.launch_raymarch_tilemap(bufB, W, H, tm2, 2 * s + 1, t, cam, tex, fx, prm(wsB))
print(f"# shard {s} of {N}, {view}: {len(mine)} tiles, heavy half {len(heavy)} tiles ({frac:g})")
print("one launch                         min %.3f  median %.3f ms" % timed(one))
print("two chains, high / low priority    min %.3f  median %.3f ms" % timed(lambda: two(hi, lo)))
print("two chains, light high / heavy low min %.3f  median %.3f ms" % timed(lambda: two(lo, hi)))
print("two chains, equal priority         min %.3f  median %.3f ms" % timed(lambda: two(eq1, eq2)))
print("two chains, one stream (serial)    min %.3f  median %.3f ms" % timed(seq))

if C > 0:
    chunks = [by_cost[k::C] for k in range(C)] if os.environ.get("RRT_SPLIT_INTERLEAVE") else [by_cost[len(by_cost) * k // C:len(by_cost) * (k + 1) // C] for k in range(C)]
    where = {tt: k for k, ch in enumerate(chunks) for tt in ch}
    assignC = np.array([C * (tt % N) + where.get(tt, 0) for tt in range(n_tiles)], np.int32)
    tmC = rrt.TileMap(H, R, C * N, assignC)
    wsC = [rrt.Workspace((2048 // C) << 20) for _ in range(C)]
    bufC = [torch.zeros(tmC.max_shard_rows() * W * 4, dtype=torch.uint8, device="cuda") for _ in range(C)]
    stC = [torch.cuda.Stream() for _ in range(C)]
    def many():
        cur = torch.cuda.current_stream()
        for k in range(C):
            stC[k].wait_stream(cur)
            rrt.launch_raymarch_tilemap(bufC[k], W, H, tmC, C * s + k, t, cam, tex, fx, prm(wsC[k]), stream=stC[k])
        for k in range(C):
            cur.wait_stream(stC[k])
    print(f"{C} chains ({'interleaved' if os.environ.get('RRT_SPLIT_INTERLEAVE') else 'by cost rank'}), equal priority    min %.3f  median %.3f ms" % timed(many))
