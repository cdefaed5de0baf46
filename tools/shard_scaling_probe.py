"""dev tool (GPU): time of the bare march (volumetrics off, single kernel) and of the three-pass path for 1/N of the 4K bench
frame, N = 1 .. 64: T(N) against T(1)/N separates the per-launch tail from the work."""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import relativisticraytracer_amd as rrt
from relativisticraytracer_amd.sky import synthetic_sky
W, H, R = 3840, 2160, 16
cam = rrt.CameraState.default(); t = 1.0
tex = rrt.SkyTexture(synthetic_sky()); fx = rrt.CameraEffects(); nt = rrt.NoiseTable(32.0)
ws = rrt.Workspace(6144 << 20)
buf = torch.zeros(H * W * 4, dtype=torch.uint8, device="cuda")
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
def timed(fn, reps=4):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best
for label, prm in (("bare march (volumetrics off)", rrt.RenderParams(spin=0.9, volumetrics=0)),
                   ("single kernel, media + tables", rrt.RenderParams(spin=0.9, noise_table=nt.id)),
                   ("three-pass, media + tables", rrt.RenderParams(spin=0.9, noise_table=nt.id, workspace=ws.id, path_policy=2))):
    base = None
    for N in (1, 2, 4, 8, 16, 32, 64):
        ts = [timed(lambda: rrt.launch_raymarch_tiles(buf, W, H, R, s, N, t, cam, tex, fx, prm)) for s in range(min(N, 4))]
        if N == 1: base = ts[0]
        print(f"{label}: 1/{N:<2d} ({W * H // N // 64:6d} waves)  {min(ts):7.3f} - {max(ts):7.3f} ms   ideal {base / N:7.3f}   excess over ideal {max(ts) - base / N:6.3f} ms", flush=True)
