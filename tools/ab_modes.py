"""dev tool (GPU): the 4K bench frame timed in the three arithmetic modes (strict / FMAD / FAST), min and median of five frames, with a hash
of the bytes; RRT_LIB_OVERRIDE A/Bs a variant build (tools/ab_modes_variants.py loops over lib/variants)."""
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import relativisticraytracer_amd as rrt
from relativisticraytracer_amd.sky import synthetic_sky
w, h = 3840, 2160
tex = rrt.SkyTexture(synthetic_sky(2048, 1024, seed=1)); cam = rrt.CameraState.default(); fx = rrt.CameraEffects(); nt = rrt.NoiseTable(32.0)
out = torch.zeros(h * w * 4, dtype=torch.uint8, device="cuda")
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
import hashlib
for mode in (0, 2, 1):
    prm = rrt.RenderParams(spin=0.9, noise_table=nt.id, arith_mode=mode)
    ts = []
    for r in range(7):
        e0.record(); rrt.launch_raymarch(out, w, h, 1.0, cam, tex, fx, prm); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    print(f"mode {mode}: min {min(ts[2:]):.3f} median {sorted(ts[2:])[2]:.3f} ms  sha {hashlib.sha1(out.cpu().numpy().tobytes()).hexdigest()[:10]}", flush=True)
