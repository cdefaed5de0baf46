"""dev tool (GPU): dense full-size parity soak -- every `stride`-th pixel in x and y of a full-size frame from the
portable-math oracle (all host cores) against the HIP frame's bytes, for several views, with the noise tables.
usage: dense_parity.py [width height stride [view ...]]      (default 3840 2160 3: 921 600 rays per view, four views)"""
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
po.build(); po.use_native_build()
sky = synthetic_sky(2048, 1024, seed=1)
tex = rrt.SkyTexture(sky); fx = rrt.CameraEffects(); nt = rrt.NoiseTable(32.0)
out = torch.zeros(h * w * 4, dtype=torch.uint8, device="cuda")
ys = np.arange(0, h, stride); xs = np.arange(0, w, stride); rows = h - 1 - ys
bad_total = 0
for name in (sys.argv[4:] or list(VIEWS)):
    pos, yaw, pitch, t = VIEWS[name]
    cam = rrt.CameraState.from_angles(pos, yaw, pitch)
    rrt.launch_raymarch(out, w, h, t, cam, tex, fx, rrt.RenderParams(spin=0.9, noise_table=nt.id))
    torch.cuda.synchronize()
    got = out.cpu().numpy().reshape(h, w, 4)[np.ix_(rows, xs)]
    a = cam.as_array()
    t0 = time.perf_counter()
    o = po.render(po.camera(a[0], a[1], a[2], a[3]), po.default_effects(),
                  po.default_params(spin=0.9, math_mode=po.MATH_PORTABLE), t, w, h, sky, stride=(stride, stride))["rgba8"]
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
        rrt.launch_raymarch_debug(out, w, h, t, cam, tex, fx, rrt.RenderParams(spin=0.9, noise_table=nt.id), steps=steps)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        rr = po.ref_render(a, po.default_effects(), 0.9, 1, t, w, h, sky, stride=(stride, stride))
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
