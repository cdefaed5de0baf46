#!/usr/bin/env python3
"""dev tool (GPU): how well-conditioned is a frame, and what do the within-tolerance arithmetic modes do to it?

    python tools/conditioning_probe.py [--size 3840x2160] [--views default,key1,skimmer] [--ks 1,2,4,8] [--n 8] [--out f.json]

For every view: S = the strict frame (float RGB before the u8 cast), F = the same frame in RRT_ARITH_FMAD / RRT_ARITH_FAST, and
for every K the HULL [lo, hi] per pixel and channel of N strict frames whose primary directions were nudged by pseudo-random
<= K ulps (rrt_params.nudge_ulps / .nudge_seed).  tol(x) = 1e-4 |x| + 1e-5 (the bar of the libm-oracle tests).  Reported:
  outliers      pixels of F outside tol of S (any channel)
  ill(K)        pixels whose strict hull is wider than tol: the reference's own arithmetic does not pin them under a K-ulp nudge
  unexplained   outliers that are not ill(K);   outside_hull: pixels of F outside [lo - tol, hi + tol]
and the frame time of each mode (production kernels, noise tables).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)

VIEWS = {
    "default": ((0.0, 10.0, -60.0), 0.0, -10.0, 1.0),
    "key1": ((15.0, 3.0, -30.0), -20.0, -5.0, 3.0),
    "grazing": ((35.0, 0.8, 10.0), -106.0, -1.2, 5.0),
    "skimmer": ((4.2, 0.6, 4.2), -90.0, -5.7, 14.0),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", default="3840x2160")
    ap.add_argument("--views", default="default,skimmer")
    ap.add_argument("--ks", default="1,2,4,8")
    ap.add_argument("--n", type=int, default=8)
    ap.add_argument("--spin", type=float, default=0.9)
    ap.add_argument("--out", default="")
    ap.add_argument("--time-steps", type=int, default=5)
    args = ap.parse_args()
    import torch
    import relativisticraytracer_amd as rrt
    from relativisticraytracer_amd.sky import synthetic_sky
    w, h = [int(v) for v in args.size.split("x")]
    dev = torch.device("cuda", 0)
    tex = rrt.SkyTexture(synthetic_sky(2048, 1024, seed=1))
    fx = rrt.CameraEffects()
    ntab = rrt.NoiseTable(32.0)
    out8 = torch.zeros(h * w * 4, dtype=torch.uint8, device=dev)

    def frame(cam, t, **kw):
        ldr = torch.zeros(h * w * 4, device=dev)
        steps = torch.zeros(h * w, dtype=torch.int32, device=dev)
        prm = rrt.RenderParams(spin=args.spin, noise_table=ntab.id, **kw)
        rrt.launch_raymarch_debug(out8, w, h, t, cam, tex, fx, prm, ldr=ldr, steps=steps)
        torch.cuda.synchronize()
        return ldr.view(h, w, 4)[..., :3].clone(), steps, out8.clone()

    def timed(cam, t, mode):
        prm = rrt.RenderParams(spin=args.spin, noise_table=ntab.id, arith_mode=mode)
        for _ in range(2):
            rrt.launch_raymarch(out8, w, h, t, cam, tex, fx, prm)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.time_steps):
            rrt.launch_raymarch(out8, w, h, t, cam, tex, fx, prm)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / args.time_steps * 1e3

    res = {"size": [w, h], "spin": args.spin, "n_nudged_frames": args.n, "tol": "1e-4 |x| + 1e-5", "views": {}}
    for name in args.views.split(","):
        pos, yaw, pitch, t = VIEWS[name]
        cam = rrt.CameraState.from_angles(pos, yaw, pitch)
        S, s_steps, s8 = frame(cam, t)
        tol = 1e-4 * S.abs() + 1e-5
        v = {"ms": {m: round(timed(cam, t, k), 3) for m, k in (("strict", 0), ("fmad", 2), ("fast", 1))}}
        modes = {}
        for m, k in (("fmad", 2), ("fast", 1)):
            F, f_steps, f8 = frame(cam, t, arith_mode=k)
            modes[m] = (F, f_steps, f8)
        hulls = {}
        for K in [int(x) for x in args.ks.split(",")]:
            lo, hi = S.clone(), S.clone()
            step_moves = torch.zeros(h * w, dtype=torch.bool, device=dev)
            for sd in range(args.n):
                N, n_steps, _ = frame(cam, t, nudge_ulps=K, nudge_seed=1000 * K + sd)
                lo = torch.minimum(lo, N); hi = torch.maximum(hi, N)
                step_moves |= n_steps != s_steps
            hulls[K] = (lo, hi, step_moves)
        for m, (F, f_steps, f8) in modes.items():
            d = (F - S).abs()
            outl = (d > tol).any(dim=2)
            d8 = (f8.view(-1, 4)[:, :3].int() - s8.view(-1, 4)[:, :3].int()).abs()
            rec = {"outliers": int(outl.sum()), "max_rel": float((d / (S.abs() + 1e-5)).max()),
                   "steps_differ": int((f_steps != s_steps).sum()),
                   "bytes_differ": int((d8 > 0).sum()), "bytes_off_by_more_than_1": int((d8 > 1).sum()), "per_K": {}}
            for K, (lo, hi, step_moves) in hulls.items():
                ill = ((hi - lo) > tol).any(dim=2)
                inside = ((F >= lo - tol) & (F <= hi + tol)).all(dim=2)
                # a looser notion: the hull moved by a quarter of the tolerance or a step count moved
                ill_q = ((hi - lo) > 0.25 * tol).any(dim=2) | step_moves.view(h, w).flip(0)
                rec["per_K"][K] = {"ill": int(ill.sum()), "unexplained": int((outl & ~ill).sum()),
                                   "outside_hull": int((~inside).sum()), "outside_hull_and_not_ill": int((~inside & ~ill).sum()),
                                   "ill_quarter_or_steps": int(ill_q.sum()), "unexplained_quarter_or_steps": int((outl & ~ill_q).sum())}
            v[m] = rec
        res["views"][name] = v
        print(name, json.dumps(v), flush=True)
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        json.dump(res, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
