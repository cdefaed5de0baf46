"""dev tool (GPU): how much of a single 4K frame's time is drain (tail)?  The same view rendered back to back on one stream
against alternating on two / three streams (frames overlap, so one frame's tail is filled by the next frame's head)."""
import sys, os, time; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import relativisticraytracer_amd as rrt
from relativisticraytracer_amd.sky import synthetic_sky
VIEWS = {"default": ((0, 10, -60), 0, -10, 1.0), "key1": ((15, 3, -30), -26.6, -5.1, 6.0),
         "grazing": ((35, 0.8, 10), -106, -1.2, 12.0), "skimmer": ((4.2, 0.6, 4.2), -90.0, -5.7, 14.0)}
w, h = 3840, 2160
K = 12
tex = rrt.SkyTexture(synthetic_sky()); fx = rrt.CameraEffects(); nt = rrt.NoiseTable(32.0)
streams = [torch.cuda.Stream() for _ in range(3)]
bufs = [torch.zeros(h * w * 4, dtype=torch.uint8, device="cuda") for _ in range(3)]
for name in sys.argv[1:] or ["default", "skimmer", "key1"]:
    pos, yaw, pitch, t = VIEWS[name]
    cam = rrt.CameraState.from_angles(pos, yaw, pitch)
    for vol, nid in (("table", nt.id), ("no-vol", 0)):
        prm = rrt.RenderParams(spin=0.9, noise_table=nid, volumetrics=1 if nid else 0)
        res = []
        for ns in (1, 2, 3):
            for rep in range(2):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                for k in range(K):
                    with torch.cuda.stream(streams[k % ns]):
                        rrt.launch_raymarch(bufs[k % ns], w, h, t, cam, tex, fx, prm)
                torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / K * 1e3
            res.append(dt)
        print(f"{name:8s} {vol:6s}: one stream {res[0]:.2f} ms/frame, two {res[1]:.2f}, three {res[2]:.2f}  (drain share {100 * (1 - min(res[1:]) / res[0]):.1f} %)", flush=True)
