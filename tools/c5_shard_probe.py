"""dev tool (GPU): config 5 (8K, path 0, all effects) as one of 8 ranks would see it: per-frame cost of shard 0 of 8
for a few frames of the path, single kernel vs three-pass (big pool), one stream vs two alternating streams,
against 1/8 of the full-frame time."""
import sys, os, time; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import relativisticraytracer_amd as rrt
from relativisticraytracer_amd import camera_paths as cp
from relativisticraytracer_amd.sky import synthetic_sky
w, h, R, N = 7680, 4320, 16, 8
POOL_GIB = int(sys.argv[1]) if len(sys.argv) > 1 else 48
tex = rrt.SkyTexture(synthetic_sky()); fx = rrt.CameraEffects(useChromaticAberration=True)
path = cp.CameraPath(0)
pools = [rrt.Workspace(POOL_GIB << 30), rrt.Workspace(POOL_GIB << 30)]
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
bufs = [torch.zeros(h * w * 4, dtype=torch.uint8, device="cuda") for _ in range(2)]
K = 6
for frame in (40, 110, 150, 200, 260):
    st, pt = cp.recording_clock(frame)
    cam = path.camera_at(pt)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for k in range(2):
        rrt.launch_raymarch(bufs[0], w, h, st, cam, tex, fx, rrt.RenderParams(spin=0.9))
    torch.cuda.synchronize(); full = (time.perf_counter() - t0) / 2 * 1e3
    line = f"frame {frame}: full frame {full:7.2f} ms (1/8 = {full / 8:6.2f}) | shard 0/8:"
    for pol, pname in ((1, "single"), (2, "three-pass")):
        prms = [rrt.RenderParams(spin=0.9, workspace=p.id, path_policy=pol) for p in pools]
        for mode in (1, 2):
            for rep in range(2):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                for k in range(K):
                    j = k % mode
                    with torch.cuda.stream(streams[j]):
                        rrt.launch_raymarch_tiles(bufs[j], w, h, R, 0, N, st, cam, tex, fx, prms[j])
                torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / K * 1e3
            line += f"  {pname} x{mode}: {dt:6.2f}"
        if pol == 2:
            line += f"  pool {pools[0].stats()}"
    print(line, flush=True)
