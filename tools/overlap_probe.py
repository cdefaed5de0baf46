"""dev tool (GPU): does rendering consecutive frames on two streams (two pools) fill the drain of one
rank's share?  Throughput per frame of shard 0 of N, one stream vs two alternating streams."""
import sys, os, time; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import relativisticraytracer_amd as rrt
from relativisticraytracer_amd.sky import synthetic_sky
w, h, R = 3840, 2160, 16
K = 24
tex = rrt.SkyTexture(synthetic_sky())
nt = rrt.NoiseTable(32.0) if os.environ.get('RRT_TOOL_TABLE', '1') == '1' else None; cam = rrt.CameraState.default(); fx = rrt.CameraEffects()
MAXS = 4
pools = [rrt.Workspace(3 << 30) for _ in range(MAXS)]
streams = [torch.cuda.Stream() for _ in range(MAXS)]
bufs = [torch.zeros(h * w * 4, dtype=torch.uint8, device="cuda") for _ in range(MAXS)]
NS = [int(a) for a in sys.argv[1:]] or [8, 4, 2, 1]
for n in NS:
    for policy, pname in ((0, "auto"),):
        prms = [rrt.RenderParams(spin=0.9, workspace=p.id, path_policy=policy, noise_table=nt.id if nt else 0) for p in pools]
        res = {}
        for mode, ns in (("one stream", 1), ("two streams", 2), ("three streams", 3), ("four streams", 4)):
            for rep in range(2):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                for k in range(K):
                    j = k % ns
                    with torch.cuda.stream(streams[j]):
                        rrt.launch_raymarch_tiles(bufs[j], w, h, R, 0, n, 1.0, cam, tex, fx, prms[j])
                torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / K * 1e3
            res[mode] = dt
        print(f"N={n} shard 0 ({pname}): one stream {res['one stream']:.3f} ms/frame, two streams {res['two streams']:.3f} ms/frame,"
              f" three {res['three streams']:.3f}, four {res['four streams']:.3f}, pool stats {pools[0].stats()}", flush=True)
