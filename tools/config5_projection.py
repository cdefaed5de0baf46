"""dev tool (GPU): BASELINE config 5 (7680x4320, a = 0.9, path 0 "Gargantua Fly-By", all effects) -- the 8-rank PROJECTION on one GPU
for frames 1 / 75 / 150 / 225 / 300 of the recording: every rank's share (16-row tiles t mod 8) rendered alone on this GPU the way a
rank of `rrt_headless --gpus 8` / `bench.py --gpus 8` renders it (three frames in flight on three streams, the library's automatic
path: a share of 4.1 M rays is above the three-pass threshold, i.e. the single kernel), sustained ms per frame; the slowest share
bounds the frame rate.  Against the best single-GPU time of the same frame (static and cost-ordered dispatch).  Gather / assemble not
included (16.6 MB per rank on another stream).  A projection, not a multi-GPU measurement.
usage: config5_projection.py [arith: 0 strict | 2 fmad]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import relativisticraytracer_amd as rrt
from relativisticraytracer_amd import camera_paths as cp, sharding
from relativisticraytracer_amd.sky import synthetic_sky
arith = int(sys.argv[1]) if len(sys.argv) > 1 else 0
w, h, R, N, SLOTS = 7680, 4320, 16, 8, 3
tex = rrt.SkyTexture(synthetic_sky()); fx = rrt.CameraEffects(useChromaticAberration=True)
path = cp.CameraPath(0)
nt = rrt.NoiseTable(14.0)
streams = [torch.cuda.Stream() for _ in range(SLOTS)]
rows = sharding.shard_rows(h, R, 0, N)
bufs = [torch.zeros(rows * w * 4, dtype=torch.uint8, device="cuda") for _ in range(SLOTS)]
full = torch.zeros(h * w * 4, dtype=torch.uint8, device="cuda")
print(f"config 5 projection, arith_mode {arith} ({'strict' if arith == 0 else 'fmad' if arith == 2 else 'fast'}), {w}x{h}, {N} shards of {rows} rows, {SLOTS} frames in flight", flush=True)
tot_single, tot_proj = 0.0, 0.0
for frame in (1, 75, 150, 225, 300):
    st, pt = cp.recording_clock(frame)
    cam = path.camera_at(pt)
    best = 1e9
    order = rrt.TileOrder()
    singles = {}
    for tag, oid in (("static", 0), ("cost-ordered", order.id)):
        prm = rrt.RenderParams(spin=0.9, arith_mode=arith, noise_table=nt.id, tile_order=oid)
        for _ in range(2):
            rrt.launch_raymarch(full, w, h, st, cam, tex, fx, prm)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(3):
            rrt.launch_raymarch(full, w, h, st, cam, tex, fx, prm)
        torch.cuda.synchronize()
        singles[tag] = (time.perf_counter() - t0) / 3 * 1e3
    order.destroy()
    best = min(singles.values())
    per = []
    for s in range(N):
        prms = [rrt.RenderParams(spin=0.9, arith_mode=arith, noise_table=nt.id, pass_chains=1) for _ in range(SLOTS)]
        b = 1e9
        for rep in range(2):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for k in range(3 * SLOTS):
                rrt.launch_raymarch_tiles(bufs[k % SLOTS], w, h, R, s, N, st, cam, tex, fx, prms[k % SLOTS], stream=streams[k % SLOTS])
            torch.cuda.synchronize()
            b = min(b, (time.perf_counter() - t0) / (3 * SLOTS) * 1e3)
        per.append(b)
    tot_single += best; tot_proj += max(per)
    print(f"frame {frame:3d} (t = {st:6.3f} s): single GPU static {singles['static']:7.2f} / cost-ordered {singles['cost-ordered']:7.2f} ms | 8 shards sustained "
          f"{' '.join('%.2f' % v for v in per)} | MAX {max(per):6.2f} ms = {best / max(per):.2f}x of the best single-GPU frame, balance min/max {min(per) / max(per):.2f}", flush=True)
print(f"five frames together: single GPU {tot_single:.1f} ms, projected 8 ranks {tot_proj:.1f} ms = {tot_single / tot_proj:.2f}x ({5e3 / tot_proj:.1f} fps against {5e3 / tot_single:.2f})", flush=True)
