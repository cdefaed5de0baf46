"""dev tool (GPU): time every library under lib/variants on the 4K bench frame (each in its own process via
RRT_LIB_OVERRIDE), with a frame checksum to confirm the bytes did not change."""
import os, subprocess, sys, glob
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
code = r'''
import sys, os, hashlib, torch
sys.path.insert(0, sys.argv[1])
import relativisticraytracer_amd as rrt
from relativisticraytracer_amd.sky import synthetic_sky
w, h = 3840, 2160
tex = rrt.SkyTexture(synthetic_sky()); cam = rrt.CameraState.default(); fx = rrt.CameraEffects()
out = torch.zeros(h * w * 4, dtype=torch.uint8, device="cuda")
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
res = []
for vol in (1, 0):
    prm = rrt.RenderParams(spin=0.9, volumetrics=vol)
    rrt.launch_raymarch(out, w, h, 1.0, cam, tex, fx, prm); torch.cuda.synchronize()
    ts = []
    for r in range(3):
        e0.record(); rrt.launch_raymarch(out, w, h, 1.0, cam, tex, fx, prm); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    res.append((min(ts), hashlib.sha1(out.cpu().numpy().tobytes()).hexdigest()[:10]))
msg = "vol %.2f ms %s | no-vol %.2f ms %s" % (res[0][0], res[0][1], res[1][0], res[1][1])
if os.environ.get("RRT_AB_MORE"):
    ws = rrt.Workspace(3 << 30)
    def t(fn):
        fn(); torch.cuda.synchronize(); ts = []
        for r in range(4):
            e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
        return min(ts)
    fast = t(lambda: rrt.launch_raymarch(out, w, h, 1.0, cam, tex, fx, rrt.RenderParams(spin=0.9, arith_mode=1)))
    p3 = rrt.RenderParams(spin=0.9, workspace=ws.id)
    sh8 = t(lambda: rrt.launch_raymarch_tiles(out, w, h, 16, 0, 8, 1.0, cam, tex, fx, p3))
    win = t(lambda: rrt.launch_raymarch(out, 1000, 700, 1.0, cam, tex, fx, p3))
    skim = rrt.CameraState.from_angles((4.2, 0.6, 4.2), -90.0, -5.7)
    heavy = t(lambda: rrt.launch_raymarch(out, w, h, 14.0, skim, tex, fx, rrt.RenderParams(spin=0.9)))
    msg += " | fast %.2f | 1/8 three-pass %.3f | 1000x700 three-pass %.3f | skimmer 4K %.2f" % (fast, sh8, win, heavy)
print(msg)
'''
libs = sorted(glob.glob(os.path.join(R, "relativisticraytracer_amd", "lib", "variants", "*.so")))
for rnd in range(2):
    for lib in libs:
        env = dict(os.environ, RRT_LIB_OVERRIDE=lib)
        r = subprocess.run([sys.executable, "-c", code, R], env=env, capture_output=True, text=True, timeout=300)
        print(f"round {rnd} {os.path.basename(lib):16s} {r.stdout.strip() or r.stderr.strip()[-300:]}", flush=True)
