"""dev tool (GPU): does the three-pass path gain from FRAMES (or parts of a frame) in different phases on several streams?

Pass 2 (eval_sample_rows) idles at 0.62-0.68 of the VALU issue rate when it has the chip to itself (LABNOTES round 5, section 2); beside marching
waves its load latency would be hidden, as it is in the single kernel.  This probe renders the same 4K view n times
  seq      one stream, one chain per launch, back to back
  two      the library's two chains per launch (halves of the dispatch order on two streams), back to back
  s2 / s3  2 / 3 streams with their own workspaces, one chain per launch, the streams started a fraction of a frame apart
           (a part-frame launch first), so that one stream's pass 2 runs beside the others' pass 1
and prints wall time per full frame.  usage: overlap_probe.py [view] [frames per stream]"""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import relativisticraytracer_amd as rrt
from relativisticraytracer_amd.sky import synthetic_sky
VIEWS = {"default": ((0.0, 10.0, -60.0), 0.0, -10.0, 1.0), "skimmer": ((4.2, 0.6, 4.2), -90.0, -5.7, 14.0),
         "key1": ((15.0, 3.0, -30.0), -26.6, -5.1, 6.0), "grazing": ((35.0, 0.8, 10.0), -106.0, -1.2, 12.0)}
view = sys.argv[1] if len(sys.argv) > 1 else "key1"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 3
W, H = 3840, 2160
pos, yaw, pitch, t = VIEWS[view]
cam = rrt.CameraState.from_angles(pos, yaw, pitch)
tex = rrt.SkyTexture(synthetic_sky()); fx = rrt.CameraEffects(); nt = rrt.NoiseTable(32.0)
GIB = int(os.environ.get("RRT_POOL_GIB", "24"))
wss = [rrt.Workspace(GIB << 30) for _ in range(3)]
bufs = [torch.zeros(H * W * 4, dtype=torch.uint8, device="cuda") for _ in range(3)]
streams = [torch.cuda.Stream() for _ in range(3)]
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)


def prm(i, chains=1):
    return rrt.RenderParams(spin=0.9, noise_table=nt.id, workspace=wss[i].id, path_policy=2, pass_chains=chains, pool_rounds=1)


def timed(fn, reps=3):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize(); e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best


def seq(chains):
    def f():
        for _ in range(2 * n):
            rrt.launch_raymarch(bufs[0], W, H, t, cam, tex, fx, prm(0, chains))
    return timed(f) / (2 * n)


def staggered(k):
    """k streams, n frames each; stream i first renders rows [0, H * i / k) so that the streams end up i / k of a frame apart"""
    def f():
        cur = torch.cuda.current_stream()
        for i in range(k):
            streams[i].wait_stream(cur)
            lead = (H * i // k) // 16 * 16
            if lead:
                rrt.launch_raymarch_rows(bufs[i], W, H, 0, lead, t, cam, tex, fx, prm(i), stream=streams[i])
        for j in range(n):
            for i in range(k):
                rrt.launch_raymarch(bufs[i], W, H, t, cam, tex, fx, prm(i), stream=streams[i])
        for i in range(k):
            cur.wait_stream(streams[i])
    frames = k * n + sum(((H * i // k) // 16 * 16) / H for i in range(k))
    return timed(f) / frames


single = timed(lambda: rrt.launch_raymarch(bufs[0], W, H, t, cam, tex, fx, rrt.RenderParams(spin=0.9, noise_table=nt.id)))
print(f"{view}: single kernel (static order) {single:.3f} ms", flush=True)
print(f"{view}: seq  one chain   {seq(1):.3f} ms / frame", flush=True)
print(f"{view}: seq  two chains  {seq(2):.3f} ms / frame", flush=True)
print(f"{view}: 2 streams, half a frame apart      {staggered(2):.3f} ms / frame", flush=True)
print(f"{view}: 3 streams, thirds of a frame apart {staggered(3):.3f} ms / frame", flush=True)
ref = bufs[0].clone()
for i in (1, 2):
    assert torch.equal(ref, bufs[i]), "streams disagree"
print("same bytes on every stream")
