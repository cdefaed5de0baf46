"""dev tool (GPU): every library under lib/variants (each in its own process via RRT_LIB_OVERRIDE) on the default view at 1080p / 4K / 8K and on the
first frame of path 0 at 8K, strict and FMAD, noise tables on; min of 4 frames + a hash of the bytes.  Same box, interleaved, two rounds."""
import glob, os, subprocess, sys
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
code = r'''
import sys, os, hashlib, torch
sys.path.insert(0, sys.argv[1])
import relativisticraytracer_amd as rrt
from relativisticraytracer_amd import camera_paths as cp
from relativisticraytracer_amd.sky import synthetic_sky
tex = rrt.SkyTexture(synthetic_sky()); nt = rrt.NoiseTable(14.0)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
path = cp.CameraPath(0)
st, pt = cp.recording_clock(1)
cases = [("1080p default", 1920, 1080, rrt.CameraState.default(), rrt.CameraEffects(), 1.0), ("4K default", 3840, 2160, rrt.CameraState.default(), rrt.CameraEffects(), 1.0),
         ("8K default", 7680, 4320, rrt.CameraState.default(), rrt.CameraEffects(), 1.0),
         ("8K path0 frame 1 all fx", 7680, 4320, path.camera_at(pt), rrt.CameraEffects(useChromaticAberration=True), st)]
msg = []
for name, w, h, cam, fx, t in cases:
    out = torch.zeros(h * w * 4, dtype=torch.uint8, device="cuda")
    for mode in (0, 2):
        prm = rrt.RenderParams(spin=0.9, noise_table=nt.id, arith_mode=mode)
        rrt.launch_raymarch(out, w, h, t, cam, tex, fx, prm); torch.cuda.synchronize()
        ts = []
        for r in range(4):
            e0.record(); rrt.launch_raymarch(out, w, h, t, cam, tex, fx, prm); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
        msg.append("%s %s %.2f ms %s" % (name, "strict" if mode == 0 else "fmad", min(ts), hashlib.sha1(out.cpu().numpy().tobytes()).hexdigest()[:8]))
    del out
print(" | ".join(msg))
'''
libs = sorted(glob.glob(os.path.join(R, "relativisticraytracer_amd", "lib", "variants", "*.so")))
for rnd in range(2):
    for lib in libs:
        r = subprocess.run([sys.executable, "-c", code, R], env=dict(os.environ, RRT_LIB_OVERRIDE=lib), capture_output=True, text=True, timeout=600)
        print(f"round {rnd} {os.path.basename(lib):18s} {r.stdout.strip() or r.stderr.strip()[-400:]}", flush=True)
