"""Conditioning map of a frame, and the account of a within-tolerance arithmetic mode against it.

The strict path (RRT_ARITH_STRICT) is bit-identical to the restatement of the reference.  RRT_ARITH_FMAD / RRT_ARITH_FAST
evaluate the geodesic integrator (integrators.h:23-59, geodesics.h:30-45) with fused multiply-adds, which moves every RK4
step by rounding noise; near-critical rays, zone boundaries (raymarcher.cu:56-58) and density gates (:71,76,91;
densities.h:85) turn such noise into finite jumps of a pixel.  Which pixels those are is a property of the REFERENCE's
arithmetic, and can be measured with it: render the strict frame again with every primary direction moved by a few ulps
(rrt_params.nudge_ulps) and see which pixels leave the tolerance.

    S      strict frame, float RGB before the u8 cast
    N_j    strict frames under pseudo-random nudges of <= K ulps, K cycling through KS
    ill    pixels where some N_j is outside tol(S) = rel |S| + floor in some channel, or takes another number of steps
    F      the mode's frame; DEVIANT pixels: outside tol(S), another step count, or a byte off by more than one LSB

`account()` always renders a FIXED set of nudged frames (FIXED_FRAMES, the same K / seed list for every view) and reports what
that set leaves uncovered; it then adds frames until every deviant pixel of every mode is ill (or the budget is spent) and returns
the masks and the counts.  The ill set only grows with the frames, so the "0 not ill" result depends on the stopping rule: the
bars that do not (outlier counts at the measured class, under what ONE 4-ulp nudge does to the strict frame) are the guard, and
tests/test_gpu_tolerance.py asserts them.  Everything runs on the GPU through the C ABI; torch is the device-memory plumbing.
"""
KS = (1, 2, 4, 8, 16)
FIXED_FRAMES = 20                        # the fixed nudge set: always rendered, the same (K, seed) list for every view; what it leaves
                                         # uncovered is reported (stats["fixed_set"]) next to the adaptive account
REL_TOL, ABS_FLOOR = 1e-4, 1e-5          # north_star: "within 1e-4 relative per channel"; floor for near-black pixels


def _frame(rrt, tex, fx, w, h, cam, t, prm_kw):
    """(float RGB (h, w, 3), steps (h, w), rgba8 (h, w, 4)), all bottom-up like the frame, on the device"""
    import torch
    out = torch.zeros(h * w * 4, dtype=torch.uint8, device="cuda")
    ldr = torch.zeros(h * w * 4, device="cuda")
    steps = torch.zeros(h * w, dtype=torch.int32, device="cuda")
    rrt.launch_raymarch_debug(out, w, h, t, cam, tex, fx, rrt.RenderParams(**prm_kw), ldr=ldr, steps=steps)
    torch.cuda.synchronize()
    return ldr.view(h, w, 4)[..., :3].clone(), steps.view(h, w).flip(0), out.view(h, w, 4)


def account(tex, w, h, cam, t, modes, fx=None, min_frames=10, budget=120, **prm_kw):
    """prm_kw: RenderParams fields shared by all frames (spin, noise_table ...).  Returns
    (frames: {"S", "steps", "rgba8", mode: {"F", "steps", "rgba8", "outliers"}}, ill mask, stats)."""
    import torch
    import relativisticraytracer_amd as rrt
    fx = fx or rrt.CameraEffects()
    S, s_steps, s8 = _frame(rrt, tex, fx, w, h, cam, t, prm_kw)
    tol = REL_TOL * S.abs() + ABS_FLOOR
    res = {"S": S, "steps": s_steps, "rgba8": s8}
    deviant = torch.zeros(h, w, dtype=torch.bool, device="cuda")
    for m in modes:
        F, f_steps, f8 = _frame(rrt, tex, fx, w, h, cam, t, dict(prm_kw, arith_mode=m))
        outl = ((F - S).abs() > tol).any(dim=2)
        deviant |= outl | (f_steps != s_steps) | ((f8[..., :3].int() - s8[..., :3].int()).abs() > 1).any(dim=2)
        res[m] = {"F": F, "steps": f_steps, "rgba8": f8, "outliers": outl}
    ill = torch.zeros(h, w, dtype=torch.bool, device="cuda")
    single = {}                  # K -> pixels ONE nudged strict frame moves (the first frame of each K)
    frames = 0
    fixed = None                 # the account after the FIXED set: FIXED_FRAMES frames (K, seed) = (KS[j % 5], 977 K + j), the same for every view
    while frames < budget:
        K = KS[frames % len(KS)]
        N, n_steps, _ = _frame(rrt, tex, fx, w, h, cam, t, dict(prm_kw, nudge_ulps=K, nudge_seed=977 * K + frames))
        moved = ((N - S).abs() > tol).any(dim=2) | (n_steps != s_steps)
        single.setdefault(K, int(moved.sum()))
        ill |= moved
        frames += 1
        if frames == min(FIXED_FRAMES, budget):
            fixed = {"frames": frames, "ill": int(ill.sum()), "deviant_not_ill": int((deviant & ~ill).sum()), "deviant": int(deviant.sum())}
        if frames >= max(min_frames, min(FIXED_FRAMES, budget)) and not bool((deviant & ~ill).any()):
            break
    stats = {"pixels": w * h, "nudged_frames": frames, "budget": budget, "ill": int(ill.sum()), "single_nudge_moves": single,
             "fixed_set": fixed,
             "tolerance": f"{REL_TOL:g} |x| + {ABS_FLOOR:g} per channel, float RGB before the u8 cast"}
    for m in modes:
        d8 = (res[m]["rgba8"][..., :3].int() - s8[..., :3].int()).abs()
        differ = res[m]["steps"] != s_steps
        stats[m] = {"outliers": int(res[m]["outliers"].sum()), "outliers_not_ill": int((res[m]["outliers"] & ~ill).sum()),
                    "steps_differ": int(differ.sum()), "steps_differ_not_ill": int((differ & ~ill).sum()),
                    "bytes_differ": int((d8 > 0).sum()), "bytes_off_by_more_than_1": int((d8 > 1).sum()),
                    "bytes_off_by_more_than_1_not_ill": int(((d8 > 1).any(dim=2) & ~ill).sum()),
                    "max_byte_diff": int(d8.max())}
    return res, ill, stats
