"""dev tool (GPU): why is the first timed frame of a bench run its slowest?  20 frames of the 4K bench workload after a device synchronise,
(a) as bench.py times them (a clock probe launched on a side stream right before), (b) without the probe, (c) with the probe's kernel and the
side stream warmed up beforehand; per-frame kernel times from events."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import relativisticraytracer_amd as rrt
from relativisticraytracer_amd.sky import synthetic_sky
w, h = 3840, 2160
tex = rrt.SkyTexture(synthetic_sky(2048, 1024, seed=1)); cam = rrt.CameraState.default(); fx = rrt.CameraEffects(); nt = rrt.NoiseTable(32.0)
out = torch.zeros(h * w * 4, dtype=torch.uint8, device="cuda")
prm = rrt.RenderParams(spin=0.9, noise_table=nt.id)


def frames(n, probe, side=None, cbuf=None):
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    torch.cuda.synchronize()
    if probe:
        rrt.clock_probe(cbuf, 500_000, stream=side)
    for a, b in ev:
        a.record(); rrt.launch_raymarch(out, w, h, 1.0, cam, tex, fx, prm); b.record()
    torch.cuda.synchronize()
    return [round(a.elapsed_time(b), 2) for a, b in ev]


for _ in range(5):
    rrt.launch_raymarch(out, w, h, 1.0, cam, tex, fx, prm)
print("(b) no probe, after a synchronise:      ", frames(8, False))
print("(b) again:                              ", frames(8, False))
side = torch.cuda.Stream(); cbuf = torch.zeros(2, dtype=torch.int64, device="cuda")
print("(a) first use of the probe + new stream:", frames(8, True, side, cbuf))
print("(c) probe and stream already used once: ", frames(8, True, side, cbuf))
time.sleep(0.05)
print("(d) after 50 ms of idle, no probe:      ", frames(8, False))
