"""dev tool (GPU): the per-window path choice (rrt_path_chooser, what both headless drivers run) on ONE rank's share of a 4K view, the way a rank
runs it: 3 frames in flight on 3 streams with a pool each, one chain per launch, sustained over FRAMES frames; the chooser is fed the intervals
between consecutive frames' render-end events as the drivers feed it (three frames late).  Against the two fixed paths on the same box.
usage: chooser_probe.py [frames]      -> for (view, shard) in bench view 0 / 3, key 1 0, grazing 0 / 3 / 4, skimmer 0: ms per frame three-pass | single | chooser"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import relativisticraytracer_amd as rrt
from relativisticraytracer_amd.sharding import PathChooser
from relativisticraytracer_amd.sky import synthetic_sky
VIEWS = {"default": ((0.0, 10.0, -60.0), 0.0, -10.0, 1.0), "skimmer": ((4.2, 0.6, 4.2), -90.0, -5.7, 14.0),
         "key1": ((15.0, 3.0, -30.0), -26.6, -5.1, 6.0), "grazing": ((35.0, 0.8, 10.0), -106.0, -1.2, 12.0)}
FRAMES = int(sys.argv[1]) if len(sys.argv) > 1 else 300
W, H, R, N, SLOTS = 3840, 2160, 16, 8, 3
tex = rrt.SkyTexture(synthetic_sky()); fx = rrt.CameraEffects(); nt = rrt.NoiseTable(32.0)
pools = [rrt.Workspace(2 << 30) for _ in range(SLOTS)]
streams = [torch.cuda.Stream() for _ in range(SLOTS)]


def run(view, sh, mode):
    pos, yaw, pitch, t = VIEWS[view]
    cam = rrt.CameraState.from_angles(pos, yaw, pitch)
    rows = rrt.tile_shard_rows(H, R, sh, N)
    bufs = [torch.zeros(rows * W * 4, dtype=torch.uint8, device="cuda") for _ in range(SLOTS)]
    prms = [rrt.RenderParams(spin=0.9, noise_table=nt.id, workspace=pools[j].id, pass_chains=1) for j in range(SLOTS)]
    pc = PathChooser(SLOTS) if mode == "chooser" else None
    ends = {}
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for k in range(1, FRAMES + 1):
        j = k % SLOTS
        prms[j].path_policy = pc.policy(k) if pc else (1 if mode == "single" else 0)
        rrt.launch_raymarch_tiles(bufs[j], W, H, R, sh, N, t, cam, tex, fx, prms[j], stream=streams[j])
        e = torch.cuda.Event(enable_timing=True); e.record(streams[j]); ends[k] = e
        if k > SLOTS:                                   # the driver waits for frame k - SLOTS before it reuses the slot
            ends[k - SLOTS].synchronize()
            if pc and k - SLOTS - 1 in ends:
                pc.report(k - SLOTS, max(0.0, ends[k - SLOTS - 1].elapsed_time(ends[k - SLOTS])))
            ends.pop(k - SLOTS - 1, None)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / FRAMES * 1e3
    st = pc.stats() if pc else None
    if pc: pc.destroy()
    return ms, st


print(f"{FRAMES} frames of one rank's share (of {N}) at {W}x{H}, {SLOTS} in flight, one chain per launch; ms per frame", flush=True)
for view, sh in (("default", 0), ("default", 3), ("key1", 0), ("grazing", 0), ("grazing", 3), ("grazing", 4), ("skimmer", 0)):
    run(view, sh, "three-pass")                          # warm both paths' code objects
    res = {m: run(view, sh, m) for m in ("three-pass", "single", "chooser")}
    st = res["chooser"][1]
    best = min(res["three-pass"][0], res["single"][0])
    print(f"{view:8s} shard {sh}: three-pass {res['three-pass'][0]:6.3f} | single kernel {res['single'][0]:6.3f} | chooser {res['chooser'][0]:6.3f} "
          f"(= {res['chooser'][0] / best:.3f} of the better fixed path; {st['frames_single_kernel']} single-kernel frames of {FRAMES}, {st['trials']} trials, "
          f"{st['trials_aborted']} aborted, {st['switches']} switches, ends on {st['incumbent']})", flush=True)
