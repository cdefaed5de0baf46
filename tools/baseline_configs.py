"""dev tool (GPU): single-GPU frame time of every BASELINE.json config (default camera unless the config says otherwise), strict
arithmetic, noise tables where the config has volumetrics; configs[4] is one frame of the 8K path (its 300-frame run: rrt_headless)."""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import relativisticraytracer_amd as rrt
from relativisticraytracer_amd.sky import synthetic_sky
tex = rrt.SkyTexture(synthetic_sky(2048, 1024, seed=1)); cam = rrt.CameraState.default(); nt = rrt.NoiseTable(32.0)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
CASES = [("configs[0] Schwarzschild a=0 128x128 (the CPU plumbing case, on the GPU)", 128, 128, 0.0, 1, False),
         ("configs[1] Kerr a=0.9 1920x1080 skybox only", 1920, 1080, 0.9, 0, False),
         ("configs[2] Kerr a=0.9 1920x1080 full volumetrics", 1920, 1080, 0.9, 1, False),
         ("configs[3] Kerr a=0.99 3840x2160 full volumetrics (one GPU)", 3840, 2160, 0.99, 1, False),
         ("the bench line: Kerr a=0.9 3840x2160 full volumetrics", 3840, 2160, 0.9, 1, False),
         ("configs[4] 7680x4320 full volumetrics + post-FX, first frame of the path's camera (one GPU)", 7680, 4320, 0.9, 1, True)]
for name, w, h, spin, vol, allfx in CASES:
    fx = rrt.CameraEffects(useChromaticAberration=allfx)
    out = torch.zeros(h * w * 4, dtype=torch.uint8, device="cuda")
    for mode, label in ((0, "strict"), (2, "FMAD"), (1, "FAST")):
        prm = rrt.RenderParams(spin=spin, volumetrics=vol, noise_table=nt.id if vol else 0, arith_mode=mode)
        ts = []
        for r in range(6):
            e0.record(); rrt.launch_raymarch(out, w, h, 1.0, cam, tex, fx, prm); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
        ms = min(ts[2:])
        print(f"{name:92s} {label:6s} {ms:9.3f} ms  {1e3 / ms:8.1f} fps  {w * h / ms / 1e3:8.1f} Mrays/s", flush=True)
    del out
