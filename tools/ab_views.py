"""dev tool (GPU): time every library under lib/variants (each in its own process via RRT_LIB_OVERRIDE) on a few 4K
views, arithmetic noise vs noise table, with frame checksums to confirm the bytes did not change.
usage: ab_views.py [view ...]   (views of tools/one_view.py; default: default key1 skimmer)"""
import glob
import os
import subprocess
import sys
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
code = r'''
import sys, os, hashlib, torch
sys.path.insert(0, sys.argv[1])
import relativisticraytracer_amd as rrt
from relativisticraytracer_amd.sky import synthetic_sky
VIEWS = {"default": ((0, 10, -60), 0, -10, 1.0), "key1": ((15, 3, -30), -26.6, -5.1, 6.0),
         "grazing": ((35, 0.8, 10), -106, -1.2, 12.0), "key3": ((5, 1.5, 50), -174.3, -1.7, 18.0),
         "skimmer": ((4.2, 0.6, 4.2), -90.0, -5.7, 14.0), "orbit": ((40, 2, 0), -90, 0, 0.0)}
w, h = (int(os.environ["RRT_AB_W"]), int(os.environ["RRT_AB_H"])) if "RRT_AB_W" in os.environ else (3840, 2160)
tex = rrt.SkyTexture(synthetic_sky()); fx = rrt.CameraEffects()
nt = rrt.NoiseTable(32.0) if hasattr(rrt, "NoiseTable") else None
out = torch.zeros(h * w * 4, dtype=torch.uint8, device="cuda")
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
msg = []
for name in sys.argv[2:]:
    pos, yaw, pitch, t = VIEWS[name]
    cam = rrt.CameraState.from_angles(pos, yaw, pitch)
    res = []
    for tab in (0, 1):
        if tab and nt is None: continue
        prm = rrt.RenderParams(spin=0.9, noise_table=nt.id if tab else 0) if nt else rrt.RenderParams(spin=0.9)
        rrt.launch_raymarch(out, w, h, t, cam, tex, fx, prm); torch.cuda.synchronize()
        ts = []
        for r in range(3):
            e0.record(); rrt.launch_raymarch(out, w, h, t, cam, tex, fx, prm); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        res.append("%.2f %s" % (min(ts), hashlib.sha1(out.cpu().numpy().tobytes()).hexdigest()[:6]))
    msg.append(name + " " + " / ".join(res))
print(" | ".join(msg))
'''
views = sys.argv[1:] or ["default", "key1", "skimmer"]
libs = sorted(glob.glob(os.path.join(R, "relativisticraytracer_amd", "lib", "variants", "*.so")))
for rnd in range(2):
    for lib in libs:
        env = dict(os.environ, RRT_LIB_OVERRIDE=lib)
        r = subprocess.run([sys.executable, "-c", code, R] + views, env=env, capture_output=True, text=True, timeout=600)
        print(f"round {rnd} {os.path.basename(lib):18s} {r.stdout.strip() or r.stderr.strip()[-400:]}", flush=True)
