"""dev tool (GPU): frame time at the reference's window size (1000x700, config.h:7-8), single kernel vs three-pass."""
import sys, os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import relativisticraytracer_amd as rrt
from relativisticraytracer_amd.sky import synthetic_sky
tex = rrt.SkyTexture(synthetic_sky()); ws = rrt.Workspace(2 << 30)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for (w, h) in ((1000, 700), (1920, 1080)):
    out = torch.zeros(h * w * 4, dtype=torch.uint8, device="cuda")
    for name, (pos, yaw, pitch, t) in {"default": ((0, 10, -60), 0, -10, 1.0), "skimmer": ((4.2, 0.6, 4.2), -90.0, -5.7, 14.0)}.items():
        cam = rrt.CameraState.from_angles(pos, yaw, pitch); fx = rrt.CameraEffects()
        for spin in (0.0, 0.9):
            res = []
            for mode in (0, 1):
                for pol, wsid in ((1, 0), (2, ws.id)):
                    prm = rrt.RenderParams(spin=spin, arith_mode=mode, workspace=wsid, path_policy=pol)
                    rrt.launch_raymarch(out, w, h, t, cam, tex, fx, prm); torch.cuda.synchronize()
                    e0.record(); rrt.launch_raymarch(out, w, h, t, cam, tex, fx, prm); e1.record(); torch.cuda.synchronize()
                    res.append(e0.elapsed_time(e1))
            print(f"{w}x{h} {name:8s} a={spin}: strict single {res[0]:6.2f} ms three-pass {res[1]:6.2f} | fast single {res[2]:6.2f} three-pass {res[3]:6.2f}  overflow {ws.stats()['overflow_waves']}", flush=True)
