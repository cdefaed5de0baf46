"""dev tool (GPU, RRT_WAVETIME variant): distribution of per-lane march durations on the 4K bench frame."""
import os, sys
os.environ["RRT_LIB_OVERRIDE"] = "relativisticraytracer_amd/lib/variants/wavetime.so"
import numpy as np, torch
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import relativisticraytracer_amd as rrt
from relativisticraytracer_amd.sky import synthetic_sky
w, h = 3840, 2160
tex = rrt.SkyTexture(synthetic_sky())
cam = rrt.CameraState.default(); fx = rrt.CameraEffects(); prm = rrt.RenderParams(spin=0.9)
out = torch.zeros(h * w * 4, dtype=torch.uint8, device="cuda")
hit = torch.zeros(h * w, dtype=torch.int32, device="cuda"); steps = torch.zeros(h * w, dtype=torch.int32, device="cuda")
rrt.launch_raymarch_debug(out, w, h, 1.0, cam, tex, fx, prm, hit=hit, steps=steps); torch.cuda.synchronize()
rrt.launch_raymarch_debug(out, w, h, 1.0, cam, tex, fx, prm, hit=hit, steps=steps); torch.cuda.synchronize()
d = hit.cpu().numpy().reshape(h, w).astype(np.float64) * 1024 / 100e6 * 1e3     # s_memtime ticks at 100 MHz -> ms
t = d.reshape(h // 8, 8, w // 8, 8).max(axis=(1, 3))                             # per 8x8 wave tile
print("per-wave duration ms: mean %.3f  p50 %.3f  p90 %.3f  p99 %.3f  p99.9 %.3f  max %.3f" % (t.mean(), np.percentile(t, 50), np.percentile(t, 90), np.percentile(t, 99), np.percentile(t, 99.9), t.max()))
rowmax = t.max(axis=1); rowmean = t.mean(axis=1)
top = np.argsort(rowmax)[-8:]
print("tile rows (of 270) with the longest waves:", [(int(r), round(float(rowmax[r]), 2), round(float(rowmean[r]), 2)) for r in top])
print("sum of wave durations (wave-ms):", t.sum())
