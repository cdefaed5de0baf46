"""dev tool: wave-level waste of the one-ray-per-lane kernel at 4K (needs GPU)."""
import sys
import numpy as np, torch
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import relativisticraytracer_amd as rrt
from relativisticraytracer_amd.sky import synthetic_sky
w, h = 3840, 2160
tex = rrt.SkyTexture(synthetic_sky())
cam = rrt.CameraState.default(); fx = rrt.CameraEffects(); prm = rrt.RenderParams(spin=0.9)
out = torch.zeros(h * w * 4, dtype=torch.uint8, device="cuda")
steps = torch.zeros(h * w, dtype=torch.int32, device="cuda")
rrt.launch_raymarch_debug(out, w, h, 1.0, cam, tex, fx, prm, steps=steps)
torch.cuda.synchronize()
s = steps.cpu().numpy().reshape(h, w).astype(np.int64)
print("mean steps", s.mean(), "sat frac", (s == 2000).mean())
for (th, tw) in ((8, 8), (4, 16), (2, 32), (1, 64), (16, 16)):
    t = s[: h // th * th, : w // tw * tw].reshape(h // th, th, w // tw, tw)
    mx = t.max(axis=(1, 3))
    print(f"tile {th}x{tw}: lane-steps executed / useful = {(mx.sum() * th * tw) / t.sum():.4f}")
# histogram of per-8x8-tile max
t = s.reshape(h // 8, 8, w // 8, 8); mx = t.max(axis=(1, 3)); mean = t.mean(axis=(1, 3))
print("tiles with max==2000:", (mx == 2000).mean(), " their mean steps:", mean[mx == 2000].mean())
np.save("gpurun_out/steps_4k.npy", s.astype(np.int16)[::4, ::4])
