"""dev tool (GPU): per-shard cost of the 4K bench frame with two frames in flight (alternating streams), N = 8 and 4."""
import sys, os, time; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import relativisticraytracer_amd as rrt
from relativisticraytracer_amd.sky import synthetic_sky
w, h = 3840, 2160
R = int(sys.argv[1]) if len(sys.argv) > 1 else 16
K = 16
tex = rrt.SkyTexture(synthetic_sky()); cam = rrt.CameraState.default(); fx = rrt.CameraEffects()
pools = [rrt.Workspace(3 << 30), rrt.Workspace(3 << 30)]
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
bufs = [torch.zeros(h * w * 4, dtype=torch.uint8, device="cuda") for _ in range(2)]
prms = [rrt.RenderParams(spin=0.9, workspace=p.id) for p in pools]
for n in (8, 4):
    ts = []
    for s in range(n):
        for rep in range(2):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for k in range(K):
                j = k % 2
                with torch.cuda.stream(streams[j]):
                    rrt.launch_raymarch_tiles(bufs[j], w, h, R, s, n, 1.0, cam, tex, fx, prms[j])
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / K * 1e3
        ts.append(dt)
    print(f"R={R} N={n}: " + " ".join(f"{t:.3f}" for t in ts) + f" | max {max(ts):.3f} mean {sum(ts)/n:.3f} -> balance {sum(ts)/n/max(ts):.3f}", flush=True)
