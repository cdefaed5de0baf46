"""dev tool (GPU): concurrency soak -- a rank's share of a 4K view rendered over and over on 3 streams (own workspace each), through the
three-pass path in two chains / one chain and through the single kernel, every buffer compared with the reference bytes after every burst.
usage: stream_soak.py [view] [bursts]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import relativisticraytracer_amd as rrt
from relativisticraytracer_amd.sky import synthetic_sky
VIEWS = {"default": ((0.0, 10.0, -60.0), 0.0, -10.0, 1.0), "skimmer": ((4.2, 0.6, 4.2), -90.0, -5.7, 14.0),
         "key1": ((15.0, 3.0, -30.0), -26.6, -5.1, 6.0), "grazing": ((35.0, 0.8, 10.0), -106.0, -1.2, 12.0)}
view = sys.argv[1] if len(sys.argv) > 1 else "skimmer"
bursts = int(sys.argv[2]) if len(sys.argv) > 2 else 100
W, H, R, N, slots = 3840, 2160, 16, 8, 3
pos, yaw, pitch, t = VIEWS[view]
cam = rrt.CameraState.from_angles(pos, yaw, pitch)
tex = rrt.SkyTexture(synthetic_sky()); fx = rrt.CameraEffects(); nt = rrt.NoiseTable(32.0)
for sh in (0, 4):
    rows = rrt.tile_shard_rows(H, R, sh, N)
    ref = torch.zeros(rows * W * 4, dtype=torch.uint8, device="cuda")
    rrt.launch_raymarch_tiles(ref, W, H, R, sh, N, t, cam, tex, fx, rrt.RenderParams(spin=0.9, noise_table=nt.id)); torch.cuda.synchronize()
    for label, chains, policy, pool_mib in (("three-pass, two chains", 0, 2, 2048), ("three-pass, one chain", 1, 2, 2048), ("three-pass, one chain, starved pool (rounds)", 1, 2, 96),
                                            ("single kernel", 1, 1, 2048)):
        pools = [rrt.Workspace(pool_mib << 20) for _ in range(slots)]
        streams = [torch.cuda.Stream() for _ in range(slots)]
        bufs = [torch.zeros_like(ref) for _ in range(slots)]
        prms = [rrt.RenderParams(spin=0.9, noise_table=nt.id, workspace=pools[j].id, path_policy=policy, pass_chains=chains) for j in range(slots)]
        t0 = time.perf_counter(); bad = 0
        for b in range(bursts):
            cur = torch.cuda.current_stream()
            for s in streams: s.wait_stream(cur)
            for k in range(4 * slots):
                rrt.launch_raymarch_tiles(bufs[k % slots], W, H, R, sh, N, t, cam, tex, fx, prms[k % slots], stream=streams[k % slots])
            for s in streams: cur.wait_stream(s)
            torch.cuda.synchronize()
            for j in range(slots):
                if not torch.equal(bufs[j], ref): bad += 1
                bufs[j].zero_()
        print(f"{view} shard {sh}: {label}: {bursts * 4 * slots} frames on {slots} streams, {bad} buffers differed, {time.perf_counter() - t0:.1f} s;  {pools[0].stats()}", flush=True)
        assert bad == 0
        for p in pools: p.destroy()
print("ok")
