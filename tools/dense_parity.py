"""dev tool (GPU): dense full-size parity soak -- every `stride`-th pixel in x and y of a full-size frame from the
portable-math oracle (all host cores) against the HIP frame's bytes, for several views, with the noise tables.
usage: dense_parity.py [width height stride [view ...]]      (default 3840 2160 3: 921 600 rays per view, four views; RRT_DENSE_SPIN: the spin, default 0.9)"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
import relativisticraytracer_amd as rrt
from relativisticraytracer_amd.sky import synthetic_sky
from oracle import pyoracle as po

w, h, stride = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (3840, 2160, 3)
VIEWS = {"default": ((0, 10, -60), 0, -10, 1.0), "key1": ((15, 3, -30), -26.6, -5.1, 6.0),
         "grazing": ((35, 0.8, 10), -106, -1.2, 12.0), "skimmer": ((4.2, 0.6, 4.2), -90.0, -5.7, 14.0)}
SPIN = float(os.environ.get("RRT_DENSE_SPIN", "0.9"))
po.build(); po.use_native_build()
sky = synthetic_sky(2048, 1024, seed=1)
tex = rrt.SkyTexture(sky); fx = rrt.CameraEffects(); nt = rrt.NoiseTable(32.0)
out = torch.zeros(h * w * 4, dtype=torch.uint8, device="cuda")
ys = np.arange(0, h, stride); xs = np.arange(0, w, stride); rows = h - 1 - ys
bad_total = 0
for name in (sys.argv[4:] or list(VIEWS)):
    pos, yaw, pitch, t = VIEWS[name]
    cam = rrt.CameraState.from_angles(pos, yaw, pitch)
    rrt.launch_raymarch(out, w, h, t, cam, tex, fx, rrt.RenderParams(spin=SPIN, noise_table=nt.id))
    torch.cuda.synchronize()
    got = out.cpu().numpy().reshape(h, w, 4)[np.ix_(rows, xs)]
    if os.environ.get("RRT_DENSE_SHARDS"):
        # the same frame the way N ranks render it (round 4): tiles dealt by the probe's costs (rrt_tile_map), every shard
        # through the three-pass path -- two chains, a 2 GiB pool used in rounds, cost-ordered dispatch seeded by the probe --
        # gathered layout, ONE rrt_assemble_all_tilemap: must be the single launch's bytes, every pixel
        N, R = int(os.environ["RRT_DENSE_SHARDS"]), 16
        cost = rrt.probe_tile_costs(w, h, R, t, cam, fx, rrt.RenderParams(spin=SPIN))
        tm = rrt.TileMap(h, R, N, rrt.balance_tiles(cost, N))
        ws = rrt.Workspace(2048 << 20)
        stride_b = tm.max_shard_rows() * w * 4
        allbuf = torch.zeros(N * stride_b, dtype=torch.uint8, device="cuda")
        fallbacks = rounds = 0
        for sh in range(N):
            order = rrt.TileOrder()
            rrt.launch_raymarch_tilemap(allbuf[sh * stride_b:], w, h, tm, sh, t, cam, tex, fx,
                                        rrt.RenderParams(spin=SPIN, noise_table=nt.id, workspace=ws.id, path_policy=2, pool_rounds=24, tile_order=order.id))
            torch.cuda.synchronize()
            st = ws.stats(); fallbacks += st["overflow_waves"]; rounds = max(rounds, st["rounds_with_work"])
            order.destroy()
        frame = torch.zeros_like(out)
        rrt.assemble_all_tilemap(frame, allbuf, stride_b, w, h, tm)
        torch.cuda.synchronize()
        diff = int((frame != out).sum())
        bad_total += diff
        print(f"{name:8s} as {N} shards (probe-dealt tile map, three-pass, two chains, 2 GiB pool: {rounds} rounds with work, {fallbacks} in-line "
              f"fall-backs, probe-seeded dispatch order), one assemble: bytes different from the single launch: {diff} of {frame.numel()}", flush=True)
        ws.destroy(); tm.destroy(); del allbuf, frame
    a = cam.as_array()
    t0 = time.perf_counter()
    o = po.render(po.camera(a[0], a[1], a[2], a[3]), po.default_effects(),
                  po.default_params(spin=SPIN, math_mode=po.MATH_PORTABLE), t, w, h, sky, stride=(stride, stride))["rgba8"]
    dt = time.perf_counter() - t0
    want = o[np.ix_(rows, xs)]
    bad = int((got != want).any(axis=-1).sum())
    bad_total += bad
    print(f"{name:8s} {w}x{h} stride {stride}: {want.shape[0] * want.shape[1]} rays, oracle {dt:.1f} s, "
          f"pixels with different bytes: {bad}", flush=True)
    if os.environ.get("RRT_DENSE_REF") == "1" and po.ref_frames_available():
        # the same pixels from the REFERENCE's own kernel body (oracle/_ref, glibc math): step counts must be equal,
        # bytes within an LSB except at hard-gate flips
        steps = torch.zeros(h * w, dtype=torch.int32, device="cuda")
        rrt.launch_raymarch_debug(out, w, h, t, cam, tex, fx, rrt.RenderParams(spin=SPIN, noise_table=nt.id), steps=steps)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        rr = po.ref_render(a, po.default_effects(), SPIN, 1, t, w, h, sky, stride=(stride, stride))
        dt = time.perf_counter() - t0
        gs = steps.cpu().numpy().reshape(h, w)[np.ix_(ys, xs)]           # diagnostics are top-down
        rs = rr["steps"].reshape(h, w)[np.ix_(ys, xs)]
        d = np.abs(got.astype(int) - rr["rgba8"][np.ix_(rows, xs)].astype(int))
        step_bad = int((gs != rs).sum())
        bad_total += step_bad
        print(f"{name:8s} vs the reference kernel body ({dt:.1f} s): rays with a different step count: {step_bad}; "
              f"bytes differing {int((d > 0).sum())} of {d.size} ({(d > 0).mean():.2e}), by more than 1 LSB "
              f"{int((d > 1).sum())}, max {int(d.max())}", flush=True)
sys.exit(1 if bad_total else 0)
