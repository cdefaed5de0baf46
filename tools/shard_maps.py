"""dev tool (GPU): single-GPU projection of an N-way split of a 4K frame -- every shard rendered alone on this GPU
through the path a rank takes (three-pass pool, noise tables), timed with device events; MAX over the shards is what a
frame would take on N GPUs (gather / assemble excluded).  Compares tile -> shard assignments and dispatch orders:
  t mod N         tile t -> shard t mod N (rounds 1-3), with one chain (round 3's path) and with two (round 4)
  probe           rrt_probe_tile_costs + rrt_tile_map_balance (what a first frame can know)
  measured        the tiles' MEASURED costs (rrt_tile_order clocks of the one-chain run, summed per row tile), dealt the same way
    python tools/shard_maps.py [view] [N] [pool MiB] [spin]      -> profiles/r0N_shard_kernel_times_<view>.txt"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import relativisticraytracer_amd as rrt
from relativisticraytracer_amd.sky import synthetic_sky

VIEWS = {"default": ((0.0, 10.0, -60.0), 0.0, -10.0, 1.0), "key1": ((15.0, 3.0, -30.0), -26.6, -5.1, 6.0),
         "skimmer": ((4.2, 0.6, 4.2), -90.0, -5.7, 14.0), "grazing": ((35.0, 0.8, 10.0), -106.0, -1.2, 12.0)}
view = sys.argv[1] if len(sys.argv) > 1 else "default"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 8
pool_mib = int(sys.argv[3]) if len(sys.argv) > 3 else 2048
spin = float(sys.argv[4]) if len(sys.argv) > 4 else 0.9
W, H, R = 3840, 2160, int(os.environ.get("RRT_TILE_ROWS", "16"))
pos, yaw, pitch, t = VIEWS[view]
cam = rrt.CameraState.from_angles(pos, yaw, pitch)
tex = rrt.SkyTexture(synthetic_sky()); fx = rrt.CameraEffects(); nt = rrt.NoiseTable(32.0)
ws = rrt.Workspace(pool_mib << 20)
n_tiles = (H + R - 1) // R
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
buf = torch.zeros(H * W * 4, dtype=torch.uint8, device="cuda")

def timed(fn, reps=3):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best

full = timed(lambda: rrt.launch_raymarch(buf, W, H, t, cam, tex, fx, rrt.RenderParams(spin=spin, noise_table=nt.id)))
_o = rrt.TileOrder()
full_ord = timed(lambda: rrt.launch_raymarch(buf, W, H, t, cam, tex, fx, rrt.RenderParams(spin=spin, noise_table=nt.id, tile_order=_o.id)), reps=4)
_o.destroy()
best_single = min(full, full_ord)
print(f"# {view} 4K a={spin:g}, {N} shards of {R}-row tiles, three-pass through a {pool_mib} MiB pool, noise tables; single-GPU frame (single kernel) "
      f"{full:.3f} ms static order, {full_ord:.3f} ms cost-ordered: BEST {best_single:.3f} ms (what the ratios below are quoted against, round 5)")

def run(assignment, chains, collect=None, ordered=False):
    """every shard of `assignment` alone on this GPU; chains: rrt_params.pass_chains.  collect: also record the tiles' measured
    costs (one extra launch per shard through an rrt_tile_order object, seeding off).  ordered: time the launches through a
    tile-order object in its steady state (each dispatched by the previous one's measured costs)."""
    tm = rrt.TileMap(H, R, N, assignment)
    times, stats = [], []
    for sh in range(N):
        order = rrt.TileOrder() if ordered else None
        prm = rrt.RenderParams(spin=spin, noise_table=nt.id, workspace=ws.id, path_policy=2, pass_chains=chains, tile_order=order.id if order else 0)
        times.append(timed(lambda: rrt.launch_raymarch_tilemap(buf, W, H, tm, sh, t, cam, tex, fx, prm)))
        stats.append(ws.stats())
        if order is not None:
            order.destroy()
        if collect is not None:
            o2 = rrt.TileOrder(); o2.set_seeding(False)
            p2 = rrt.RenderParams(spin=spin, noise_table=nt.id, workspace=ws.id, path_policy=2, pass_chains=1, tile_order=o2.id)
            rrt.launch_raymarch_tilemap(buf, W, H, tm, sh, t, cam, tex, fx, p2); torch.cuda.synchronize()
            info = o2.info(arrays=True)
            rows = tm.shard_rows(sh)
            c = info["cost"].astype(np.float64).reshape(rows // 8, W // 8)
            mine = [tt for tt in range(n_tiles) if assignment[tt] == sh]
            per_tile = c.reshape(len(mine), R // 8, W // 8).sum(axis=(1, 2))
            for k, tt in enumerate(mine):
                collect[tt] = per_tile[k]
            o2.destroy()
    tm.destroy()
    return times, stats

modulo = (np.arange(n_tiles) % N).astype(np.int32)
measured = np.zeros(n_tiles)
can_collect = R % 8 == 0 and H % R == 0                       # the per-tile sums below assume whole wave tiles per row tile
one_t, one_s = run(modulo, 1, collect=measured if can_collect else None)
probe = rrt.probe_tile_costs(W, H, R, t, cam, fx, rrt.RenderParams(spin=spin))
cases = [("t mod N, one chain (round 3's path)", modulo, 1, False, (one_t, one_s)),
         ("t mod N, two chains", modulo, 0, False, None),
         ("t mod N, two chains, cost-ordered dispatch", modulo, 0, True, None),
         ("dealt by the probe's estimate, two chains", rrt.balance_tiles(probe, N), 0, False, None),
         ] + ([("dealt by measured costs, two chains", rrt.balance_tiles(measured.astype(np.float32), N), 0, False, None)] if can_collect else [])
for name, m, chains, ordered, done in cases:
    ts, st = done if done else run(m, chains, ordered=ordered)
    cnt = np.bincount(m, minlength=N)
    print(f"{name:44s}: max shard {max(ts):.3f} ms  min {min(ts):.3f}  mean {np.mean(ts):.3f}  balance min/max {min(ts) / max(ts):.3f}  "
          f"-> {best_single / max(ts):.2f}x of the best single-GPU frame ({full / max(ts):.2f}x of the static-order one);  tiles per shard {cnt.min()}-{cnt.max()}, rounds with work "
          f"{max(s['rounds_with_work'] for s in st)}, in-line fall-backs {sum(s['overflow_waves'] for s in st)}", flush=True)
    print("    per shard [ms]: " + " ".join(f"{v:.3f}" for v in ts), flush=True)

# Sustained: what bench.py / the headless drivers actually run at N > 1 -- --frames-in-flight 3 frames of a rank's share on three
# streams, each with a third of the rank's pool (--workspace-gib 16), so that the next frames fill this one's drain.  Per shard:
# wall time of 12 frames / 12; the frame rate of an N-GPU run is bounded by the slowest shard (gather / assemble excluded, as above).
if os.environ.get("RRT_SUSTAINED", "1") == "1":
    slots = 3
    pools = [rrt.Workspace((16 << 30) // slots) for _ in range(slots)]
    streams = [torch.cuda.Stream() for _ in range(slots)]
    bufs = [torch.zeros(rrt.tile_shard_rows(H, R, 0, N) * W * 4, dtype=torch.uint8, device="cuda") for _ in range(slots)]
    tm = rrt.TileMap(H, R, N, modulo)
    for label, chains, policy in (("t mod N, two chains, 3 frames in flight", 0, 2), ("t mod N, ONE chain, 3 frames in flight (the drivers)", 1, 2),
                                  ("t mod N, SINGLE KERNEL in line, 3 frames in flight", 1, 1)):
        ts = []
        for sh in range(N):
            prms = [rrt.RenderParams(spin=spin, noise_table=nt.id, workspace=pools[j].id, path_policy=policy, pass_chains=chains) for j in range(slots)]
            def burst(frames):
                cur = torch.cuda.current_stream()
                for s in streams:
                    s.wait_stream(cur)
                for k in range(frames):
                    rrt.launch_raymarch_tilemap(bufs[k % slots], W, H, tm, sh, t, cam, tex, fx, prms[k % slots], stream=streams[k % slots])
                for s in streams:
                    cur.wait_stream(s)
            ts.append(timed(lambda: burst(12)) / 12)
        print(f"{label:66s}: max shard {max(ts):.3f} ms per frame  min {min(ts):.3f}  mean {np.mean(ts):.3f}  "
              f"-> {best_single / max(ts):.2f}x of the best single-GPU frame ({full / max(ts):.2f}x of the static-order one), sustained", flush=True)
        print("    per shard [ms]: " + " ".join(f"{v:.3f}" for v in ts), flush=True)
    tm.destroy()
