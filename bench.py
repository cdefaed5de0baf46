#!/usr/bin/env python3
"""bench.py -- throughput of the geodesic ray-march hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]

A "step" is one full frame of the BASELINE.json workload: 3840x2160, Kerr a = 0.9,
full volumetric disk + dust + Doppler/redshift, the reference's start-up camera
(src/main.cpp:128-130), time = 1.0, default CameraEffects, synthetic 2048x1024 sky
(seed 1) already resident in HBM.  With N > 1 (launched by torch.distributed.run, one
rank per GPU) the SAME frame is split into interleaved 16-row tiles across the ranks
and assembled on rank 0 by one RCCL gather: strong scaling.

Rank 0 prints one JSON line.  `value` = Mrays/s (= Mpixels/s) of the whole job, strict arithmetic.
`roofline` prices the dominant kernel (raymarch_pixels) against the FP32 vector-ALU
issue rate, which is what bounds it, with every timed frame's kernel time,
the clock the chip held, the PMC view (issue-slot utilisation, HBM traffic: only from records
measured on this build's sources); the HBM view that the north star asks for is reported
alongside.  `cpu_baseline` is the reference's own kernel body (OpenMP, all host cores; the
oracle port beside it) on a strided sample of the same frame, on every N -- a reported
baseline, not a target.  `within_tolerance_mode` = RRT_ARITH_FMAD (the arithmetic class of the
reference's own build) with its account; never `value`.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# FP32 VALU issue peak: 256 CU x 4 SIMD x 32 lanes x 2.4 GHz, one unfused op per lane per
# clock (MI355X_MICROARCH.md: SIMD-32, 2-cycle wave64 issue; measured 58-64 T/s at the
# clock the chip holds under load: profiles/r01_valu_microbench.txt).
VALU_PEAK_TOPS = 256 * 4 * 32 * 2.4e9 / 1e12
HBM_PEAK_GBS = 8000.0
ALGO_BYTES_PER_RAY = 52.0          # 4 B RGBA8 store + 3 bilinear fetches x 4 texels x 4 B


def ops_per_ray(steps, noise, dens, samples):
    """source-level IEEE ops, FMA not assumed."""
    return 297.0 * steps + 223.0 * noise + 60.0 * dens + 40.0 * samples + 150.0


def source_hash():
    """sha256 over the kernel sources: ties a measured-traffic record (profiles/hbm_traffic.json) to the build
    it was measured on."""
    import hashlib
    h = hashlib.sha256()
    for f in ("rrt_hip.hip", "rrt_kernels.h", "rrt_device.h", "rrt_math.h", "rrt_tile_sort.h", "rrt_noise_plan.h", "rrt_tile_objects.h"):
        h.update(open(os.path.join(ROOT, "relativisticraytracer_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


CPU_TARGET_S = 12.0       # wall seconds the timed CPU leg should last (>= 10 s: start-up and imbalance no longer show)
CPU_MAX_RAYS = 2_200_000  # ... and never more rays than this (every 2nd pixel of the 4K frame), whatever the rate probe says


def cpu_threads(world):
    """Threads of the CPU leg.  N = 1: OpenMP's own default (all cores the process may use, or OMP_NUM_THREADS).  N > 1: launchers
    pin OMP_NUM_THREADS to 1 (torch.distributed.run) or to cores / N, but the leg runs on rank 0 while every other rank waits at
    a barrier, so it takes all the cores this process is allowed on -- and says how many it had."""
    from oracle import pyoracle as po
    if world > 1:
        try:
            return max(1, len(os.sched_getaffinity(0)))
        except AttributeError:
            return max(1, os.cpu_count() or 1)
    return po.max_threads()


def cpu_sample_stride(width, height, spin, sky, nthreads=0):
    """Stride of the CPU baseline's pixel sample, sized so that the leg runs about CPU_TARGET_S on THIS host: a short
    probe (every 16th pixel, a few tenths of a second on a 128-thread host) measures the rate, then
    stride = floor(sqrt(pixels / (rate * target))), at least 1 (= the whole frame).  Round 2 sampled every 8th pixel:
    0.9 s on 128 threads, i.e. two rows per thread -- start-up and imbalance dominated."""
    import math
    from oracle import pyoracle as po
    import relativisticraytracer_amd as rrt
    po.build()
    po.use_native_build()
    a = rrt.CameraState.default().as_array()
    cam = po.camera(a[0], a[1], a[2], a[3])
    probe = 16
    t0 = time.perf_counter()
    if po.ref_frames_available():
        po.ref_render(a, po.default_effects(), spin, 1, 1.0, width, height, sky, n_threads=nthreads or po.max_threads(), stride=(probe, probe))
    else:
        po.render(cam, po.default_effects(), po.default_params(spin=spin), 1.0, width, height, sky, stride=(probe, probe),
                  want=("diag",), n_threads=nthreads or po.max_threads())
    dt = max(time.perf_counter() - t0, 1e-3)
    rate = (math.ceil(width / probe) * math.ceil(height / probe)) / dt
    stride = max(1, int(math.floor(math.sqrt(width * height / max(rate * CPU_TARGET_S, 1.0)))))
    # The probe is short (a few hundredths of a second on 256 threads) and has been seen to overestimate the sustained rate five-fold
    # (a 4-rank rehearsal: stride 1, 69 s of CPU leg): the sample is also capped at CPU_MAX_RAYS rays, 15-20 s at the rates seen so far.
    return max(stride, int(math.ceil(math.sqrt(width * height / float(CPU_MAX_RAYS)))))


def cpu_baseline(width, height, spin, stride, sky, cam_arr=None, time_=1.0, nthreads=0):
    """Oracle (OpenMP, libm) on pixels (x, y) with x % stride == y % stride == 0 of the same frame."""
    import numpy as np
    from oracle import pyoracle as po
    import relativisticraytracer_amd as rrt
    po.build()
    native = po.use_native_build()      # -O3 -march=native build of the same source, made on this host
    cpu_model = "unknown"
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                cpu_model = ln.split(":", 1)[1].strip(); break
    except OSError:
        pass
    a = cam_arr if cam_arr is not None else rrt.CameraState.default().as_array()
    cam = po.camera(a[0], a[1], a[2], a[3])
    prm = po.default_params(spin=spin)
    nthreads = nthreads or po.max_threads()
    t0 = time.perf_counter()
    r = po.render(cam, po.default_effects(), prm, time_, width, height, sky, stride=(stride, stride),
                  want=("diag",), n_threads=nthreads)
    dt = time.perf_counter() - t0
    sel = np.zeros((height, width), bool); sel[::stride, ::stride] = True
    sel = sel.reshape(-1)
    n = int(sel.sum())
    means = {k: float(r[k][sel].mean()) for k in ("steps", "n_noise", "n_dens", "n_samples")}
    port = {"value": n / dt / 1e6, "unit": "Mrays/s", "cores": nthreads, "kind": "port",
            "rays_per_s_per_thread": round(n / dt / nthreads, 1), "sample_rays": n, "wall_s": round(dt, 2),
            "cpu_model": cpu_model,
            "build": "gcc -O3 -march=native -ffp-contract=off -fopenmp" if native else
                     "gcc -O2 -mfma -ffp-contract=off -fopenmp",
            "sample": f"every {stride}th pixel in x and y of the same {width}x{height} frame "
                      f"({n} rays, {dt:.1f} s wall, oracle/rrt_oracle.c, libm math, OpenMP dynamic over row segments of 128 samples)"}
    if not po.ref_frames_available():
        return port, means
    # oracle/_ref/libref_frames.so = the REFERENCE's own raymarch_kernel body (and headers), compiled by g++ in the
    # build container where /root/reference lies (oracle/Makefile), one call per pixel under the same OpenMP loop:
    # the reference's CPU path on the same sample.  Its step counts must equal the port's (both libm).
    fx = po.default_effects()
    t0 = time.perf_counter()
    rr = po.ref_render(a, fx, spin, 1, time_, width, height, sky, n_threads=nthreads, stride=(stride, stride))
    dtr = time.perf_counter() - t0
    same_steps = bool(np.array_equal(rr["steps"][sel], r["steps"][sel]))
    return {"value": n / dtr / 1e6, "unit": "Mrays/s", "cores": nthreads, "kind": "reference",
            "rays_per_s_per_thread": round(n / dtr / nthreads, 1), "sample_rays": n, "wall_s": round(dtr, 2),
            "cpu_model": cpu_model,
            "build": "g++ -O2 -ffp-contract=off -fopenmp on /root/reference/src/raymarcher.cu:15-174 + include/*.h "
                     "(oracle/Makefile ref; harness: launch indices, tex2D = the documented sky filter)",
            "sample": f"every {stride}th pixel in x and y of the same {width}x{height} frame "
                      f"({n} rays, {dtr:.1f} s wall, the reference's raymarch_kernel called per pixel, glibc math, "
                      f"OpenMP dynamic over row segments of 128 samples; stride sized for ~{CPU_TARGET_S:.0f} s on this host)",
            "step_counts_equal_port": same_steps, "port": port}, means


def self_launch(n, argv):
    """`python bench.py --gpus N` without a launcher: start the N ranks as CHILD processes (one per GPU,
    torch.distributed.run, rendezvous on 127.0.0.1) before this process has touched torch or the GPU,
    relay their output, and return the children's exit status.  Nothing is exec'ed."""
    import socket
    import subprocess
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env["MASTER_ADDR"] = "127.0.0.1"
    from relativisticraytracer_amd.sharding import single_node_environment      # plain Python: does not touch torch or the GPU
    single_node_environment(env)                           # RCCL warnings on, loopback bootstrap, dmabuf IPC
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or n) // n)))
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=None, text=True, bufsize=1)
    for ln in proc.stdout:
        sys.stdout.write(ln)
        sys.stdout.flush()
    return proc.wait()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--spin", type=float, default=0.9)
    ap.add_argument("--tile-rows", type=int, default=16)
    ap.add_argument("--cpu-stride", type=int, default=-1,
                    help="CPU baseline sample stride: -1 (default) = sized so that the leg runs ~12 s on this host, 0 = skip")
    ap.add_argument("--no-fast", action="store_true", help="skip the within-tolerance arithmetic modes (RRT_ARITH_FMAD / _FAST) and their account")
    ap.add_argument("--no-noise-table", action="store_true", help="hash every noise3D corner arithmetically (no lattice tables)")
    ap.add_argument("--no-heavy", action="store_true", help="skip the informational heavy-view leg")
    ap.add_argument("--frames-in-flight", type=int, default=3,
                    help="N > 1: frames rendered / gathered / assembled concurrently per rank (>= 2; 1 = no pipelining)")
    ap.add_argument("--workspace-gib", type=int, default=16,
                    help="per-rank pool for the three-pass path (N > 1), split between the frames in flight")
    ap.add_argument("--init-timeout", type=float, default=300.0,
                    help="N > 1: seconds the process-group / communicator bring-up may take before the run exits non-zero with "
                         "the tracebacks of all threads (instead of hanging)")
    ap.add_argument("--run-timeout", type=float, default=900.0,
                    help="seconds everything after the bring-up may take before the same happens (0: no limit)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(self_launch(args.gpus, sys.argv[1:]))

    import numpy as np
    import torch
    import relativisticraytracer_amd as rrt
    from relativisticraytracer_amd import sharding
    from relativisticraytracer_amd.sky import synthetic_sky

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    args.gpus = world                                          # under a launcher the launcher's world size rules
    backend = os.environ.get("RRT_DIST_BACKEND", "nccl")        # "gloo": rehearsal on fewer GPUs than ranks
    n_dev = torch.cuda.device_count()                           # does not initialise the GPU
    if n_dev == 0:
        sys.exit("bench.py needs a GPU (there is no CPU fallback in the product path)")
    if backend == "gloo":
        local_rank %= n_dev
    elif world > n_dev:
        sys.exit(f"bench.py: {world} ranks but {n_dev} GPU(s) visible; RCCL needs one GPU per rank "
                 "(RRT_DIST_BACKEND=gloo rehearses several ranks on one card)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dog = sharding.Watchdog(f"bench.py rank {rank}")
    if world > 1:
        # A multi-GPU run must explain itself even when it fails: RCCL warnings on, loopback bootstrap on one
        # node, RCCL's 0.3 GB library read into the page cache up front (its first collective loads the gfx950 code object out
        # of it: seconds on a cold cache), a deadline on the bring-up, and a watchdog that ends a stuck run with every thread's
        # traceback and a non-zero status.
        sharding.single_node_environment()
        warm = sharding.warm_library_pages(sharding.torch_rccl_library()) if backend == "nccl" and sharding.torch_rccl_library() else None
        dog.arm(args.init_timeout, "process group / communicator bring-up")
        dist = sharding.init_process_group(backend, rank, world, dev, timeout_s=args.init_timeout)
        if warm is not None:
            warm.join()

    w, h, R = args.width, args.height, args.tile_rows
    sky_np = synthetic_sky(2048, 1024, seed=1)
    tex = rrt.SkyTexture(sky_np)
    cam, fx = rrt.CameraState.default(), rrt.CameraEffects()
    # With several ranks every launch is a fraction of the frame; the library then prefers its three-pass
    # path for small launches (rrt_params.path_policy = auto), which needs a caller-owned pool.
    # N > 1 keeps --frames-in-flight frames in flight (FrameSharder pipeline mode), each with its own share of the
    # pool: one rank's share of a 4K frame is only a few rounds of wavefronts, and the next frames fill its drain
    # (profiles/r02_frames_in_flight.txt: 6.41 / 5.64 / 5.39 / 5.67 ms per frame with 1 / 2 / 3 / 4 at N = 8).
    pipeline = world > 1 and args.frames_in_flight >= 2 and os.environ.get("RRT_NO_PIPELINE", "0") != "1"
    n_slots = args.frames_in_flight if pipeline else 1
    pools = ([rrt.Workspace((args.workspace_gib << 30) // n_slots) for _ in range(n_slots)]
             if (world > 1 and args.workspace_gib > 0) else [])
    ws = pools[0] if pools else None
    # Lattice-hash tables for the volumetric noise: built once (like the sky upload: a resident input, outside the
    # timed region), camera- and time-independent within [0, t_max]; same bytes with or without them.
    t_build0 = time.perf_counter()
    ntab = None if args.no_noise_table else rrt.NoiseTable(32.0)
    torch.cuda.synchronize()
    table_build_ms = (time.perf_counter() - t_build0) * 1e3
    prms = [rrt.RenderParams(spin=args.spin, volumetrics=1, workspace=pools[j].id if pools else 0,
                             noise_table=ntab.id if ntab else 0,
                             # several frames in flight: ONE chain per launch -- the other frames fill a launch's tails and the second
                             # chain's streams only compete with them (profiles/r05_sustained_chains.txt: a rank's share of the 4K
                             # frame from inside the disk 7.4 -> 6.8 ms per frame, of the bench frame 4.9 -> 4.8); the PATH is chosen
                             # per rank below (tune_path)
                             pass_chains=1 if pipeline else 0,
                             path_policy=int(os.environ.get("RRT_PATH_POLICY", "0"))) for j in range(n_slots)]
    # the one-frame-at-a-time leg: the library's own choice (two chains)
    prm_one = rrt.RenderParams(spin=args.spin, volumetrics=1, workspace=pools[0].id if pools else 0, noise_table=ntab.id if ntab else 0,
                               path_policy=int(os.environ.get("RRT_PATH_POLICY", "0")))

    kernel_ms = []
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
          for _ in range(args.steps + args.warmup + 2)]          # + 2: the untimed frames of the clock measurement
    it = {"i": 0}

    def render(buf, slot):
        e0, e1 = ev[it["i"]]
        e0.record()                      # torch's current stream == the stream the launch goes to
        rrt.launch_raymarch_tiles(buf, w, h, R, rank, world, 1.0, cam, tex, fx, prms[slot])
        e1.record()

    def assemble(frame, buf, shard):
        rrt.assemble_tiles(frame, buf, w, h, R, shard, world)

    def assemble_all(frame, bufs, stride):
        rrt.assemble_all_tiles(frame, bufs, stride, w, h, R, world)

    # N > 1: the following frames are rendered (on their own streams) while frame k is gathered and assembled; the
    # frames still in flight are flushed (gathered + assembled) inside the timed region, so K timed steps deliver K frames.
    fs = sharding.FrameSharder(w, h, R, rank, world, dev, render, assemble, assemble_all=assemble_all,
                               pipeline=n_slots if pipeline else False, timing=world > 1)

    # Untimed one-off setup, so that even --warmup 0 times steady-state steps: load the code object with a
    # tiny launch, and bring up the RCCL communicator / its peer-to-peer channels with one small collective
    # of each kind the step uses.
    tiny = torch.zeros(16 * 16 * 4, dtype=torch.uint8, device=dev)
    rrt.launch_raymarch(tiny, 16, 16, 1.0, cam, tex, fx, rrt.RenderParams(spin=args.spin, max_steps=4))
    if world > 1:
        probe = torch.zeros(4096, dtype=torch.uint8, device=dev)
        if fs.stage_cpu:
            probe = probe.cpu()
        dist.gather(probe, [torch.zeros_like(probe) for _ in range(world)] if rank == 0 else None, dst=0)
        dist.barrier()
    torch.cuda.synchronize()
    # Rank-local choice of the path a rank's share takes while several frames are in flight (same bytes either way; untimed set-up, like
    # the table build).  The other frames fill a launch's drain, which is all the three-pass path is for, so the plain single kernel
    # (4 % less work, no pool traffic) is usually the fastest way through: 4.56 / 6.21 / 6.56 ms per frame on the bench / key-1 / skimmer
    # views against 4.74 / 6.61 / 6.85 three-pass with three in flight -- UNLESS the share holds a wavefront that takes longer than the
    # frames in flight together: a slot's next frame waits for it (the disk-grazing view's middle shards: 19 ms, 7.5 against 6.2 ms per
    # frame).  The camera is fixed here, so each rank times both on its own share (two bursts of 4 x slots frames each) and keeps the
    # faster; profiles/r05_sustained_chains.txt.  RRT_PATH_POLICY pins the path instead.
    path_tuning = None
    # Which branch is taken must not depend on THIS rank's share (the tuning branch ends in a collective): shares differ
    # by one tile between ranks, so the decision is made on the LARGEST share of the launch, which every rank computes alike.
    max_rays = w * max(sharding.shard_rows(h, R, r_, world) for r_ in range(world))
    auto_is_single = not pools or max_rays > rrt._lib.load().rrt_path_auto_max_rays()     # RRT_PATH_AUTO's own threshold
    if pipeline and auto_is_single and "RRT_PATH_POLICY" not in os.environ:
        path_tuning = {"all_ranks": "automatic per launch: single kernel, media in line, for shares above RRT_PATH_AUTO's three-pass threshold "
                                    "(the largest share has %d rays)" % max_rays}
    elif pipeline and fs.streams is not None and "RRT_PATH_POLICY" not in os.environ:
        def burst_ms(policy):
            for p in prms:
                p.path_policy = policy
            best = 1e9
            for _ in range(3):                      # the first burst also warms the path's code objects
                torch.cuda.synchronize(dev)
                t0b = time.perf_counter()
                for k in range(4 * n_slots):
                    with torch.cuda.stream(fs.streams[k % n_slots]):
                        rrt.launch_raymarch_tiles(fs.locals[k % n_slots], w, h, R, rank, world, 1.0, cam, tex, fx, prms[k % n_slots])
                torch.cuda.synchronize(dev)
                best = min(best, (time.perf_counter() - t0b) * 1e3 / (4 * n_slots))
            return best
        cand = {0: burst_ms(0), 1: burst_ms(1)}
        pick = min(cand, key=cand.get)
        for p in prms:
            p.path_policy = pick
        mine = {"rank": rank, "picked": "single kernel, media in line" if pick == 1 else "automatic (three-pass, one chain, for a small share)",
                "ms_per_frame_three_pass_auto": round(cand[0], 3), "ms_per_frame_single_kernel": round(cand[1], 3)}
        allt = [None] * world
        dist.all_gather_object(allt, mine)
        path_tuning = {"per_rank": allt, "note": "rank-local, untimed set-up: each rank's share sustained through both paths with the "
                                                 "configured frames in flight, the faster one kept (same bytes)"}
    elif pipeline:
        path_tuning = {"pinned": "RRT_PATH_POLICY=" + os.environ.get("RRT_PATH_POLICY", "0")}

    dog.disarm()
    dog.arm(args.run_timeout, "timed frames and their legs")

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        fs.step(); it["i"] += 1
    fs.flush()
    barrier()
    # The shader clock the chip HOLDS during the timed frames: rrt_clock_probe -- ONE wavefront that sleeps and
    # reads the shader-clock counter (s_memtime) and the constant 100 MHz counter (s_memrealtime) at both ends -- runs on a
    # second stream BESIDE the timed frames, for about three quarters of their expected duration (from the warm-up frames'
    # kernel times, so that it has ended before they have; without warm-up frames it runs beside two extra frames afterwards).
    # One sleeping wave of 8 192 slots does not move the measurement.
    clock_ghz, clock_when = None, None
    cbuf = torch.zeros(2, dtype=torch.int64, device=dev) if world == 1 else None
    side = torch.cuda.Stream() if world == 1 else None
    probe_us = 0
    if world == 1 and args.warmup > 0:
        warm_ms = min(a.elapsed_time(b) for a, b in ev[:args.warmup])
        probe_us = int(min(2_000_000, max(1_000, 0.75 * args.steps * warm_ms * 1000.0)))
        clock_when = "beside the timed frames"
    t0 = time.perf_counter()
    for s_ in range(args.steps):
        fs.step(); it["i"] += 1
        if s_ == 0 and probe_us:
            # The probe goes out right BEHIND the first timed frame's launch, not in front of it (round 6): a chip whose only work is
            # one sleeping wavefront drops its clocks, and the frame that starts next pays ~2.8 ms for the ramp -- measured: 38.4 ms
            # for the first of eight frames with the probe in front, 35.7 without a probe (tools/first_frame_probe.py).  That was the
            # probe's own artefact in `value`; the frames are the same K frames, none is dropped or warmed up in hiding.
            rrt.clock_probe(cbuf, probe_us, stream=side)
    fs.flush()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    kernel_ms = [a.elapsed_time(b) for a, b in ev[args.warmup:args.warmup + args.steps]]

    if world == 1:
        if clock_when is None:                 # --warmup 0: beside two untimed frames of the same workload, afterwards
            est_ms = float(np.mean(kernel_ms)) if kernel_ms else 40.0
            side.wait_stream(torch.cuda.current_stream())
            rrt.clock_probe(cbuf, int(min(2_000_000, max(2_000, 1.7 * est_ms * 1000.0))), stream=side)
            for _ in range(2):
                fs.step(); it["i"] += 1
            clock_when = "beside two untimed frames after the timed ones"
        torch.cuda.synchronize()
        cc = cbuf.cpu().numpy()
        if cc[1] > 0:
            clock_ghz = float(cc[0]) / float(cc[1]) * 0.1

    # N > 1, so that the first real multi-GPU run explains its own efficiency:
    #  - every rank's own phase latencies (render | gather = queueing + waiting for the slowest rank + transfer |
    #    assemble), from device events on the stream each step ran on, all-gathered to rank 0;
    #  - the same K frames ONE AT A TIME (no frames in flight): latency-bound strong scaling, reported next to the
    #    pipelined `value` (throughput with frames in flight, frames arrive n-1 steps late).
    phases, one_at_a_time = None, None
    if world > 1:
        mine = fs.phase_times(skip=args.warmup) or {"render": 0.0, "gather": 0.0, "assemble": 0.0, "steps": 0}
        allp = [None] * world
        dist.all_gather_object(allp, {k: round(float(v), 4) for k, v in mine.items()})
        phases = {"per_rank_render_ms": [p["render"] for p in allp], "per_rank_gather_ms": [p["gather"] for p in allp],
                  "rank0_assemble_ms": allp[0]["assemble"],
                  "render_balance_min_over_max": round(min(p["render"] for p in allp) / max(max(p["render"] for p in allp), 1e-9), 4),
                  "note": "device-event latencies of each frame's own phases on its stream; with several frames in flight "
                          "consecutive frames overlap, so these do not add up to ms_per_step"}
        if pipeline:
            it1 = {"i": 0}
            ev1 = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps + 1)]

            def render1(buf, slot):
                e0, e1 = ev1[it1["i"]]
                e0.record()
                rrt.launch_raymarch_tiles(buf, w, h, R, rank, world, 1.0, cam, tex, fx, prm_one)
                e1.record()

            fs1 = sharding.FrameSharder(w, h, R, rank, world, dev, render1, assemble, assemble_all=assemble_all,
                                        pipeline=False, timing=True)
            fs1.step(); it1["i"] += 1
            barrier()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                fs1.step(); it1["i"] += 1
            barrier()
            dt1 = time.perf_counter() - t1
            tt = torch.tensor([dt1], device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt1 = float(tt.item())
            p1 = fs1.phase_times(skip=1) or {"render": 0.0, "gather": 0.0, "assemble": 0.0}
            allp1 = [None] * world
            dist.all_gather_object(allp1, {k: round(float(v), 4) for k, v in p1.items()})
            one_at_a_time = {"ms_per_step": round(dt1 / args.steps * 1e3, 3),
                             "value": round(w * h * args.steps / dt1 / 1e6, 3), "unit": "Mrays/s",
                             "per_rank_render_ms": [p["render"] for p in allp1],
                             "per_rank_gather_ms": [p["gather"] for p in allp1], "rank0_assemble_ms": allp1[0]["assemble"],
                             "note": "the same frames with --frames-in-flight 1: each frame rendered, gathered and assembled "
                                     "before the next starts (latency-bound strong scaling); step = max over ranks of render "
                                     "+ gather + assemble"}

    # N > 1: what ONE GPU of this node needs for the same frame in the same process, the denominator a scaling figure should
    # be quoted against: rank 0 renders the whole frame alone with the single kernel, in the static order
    # and cost-ordered (rrt_tile_order), the other ranks wait at the barrier; best of the two is the reference.
    single_ref = None
    if world > 1:
        if rank == 0:
            full = torch.zeros(h * w * 4, dtype=torch.uint8, device=dev)
            order = rrt.TileOrder()
            single_ref = {}
            for tag, oid in (("static_order", 0), ("cost_ordered", order.id)):
                prm_s = rrt.RenderParams(spin=args.spin, volumetrics=1, noise_table=ntab.id if ntab else 0, tile_order=oid)
                for _ in range(2):
                    rrt.launch_raymarch(full, w, h, 1.0, cam, tex, fx, prm_s)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(5):
                    rrt.launch_raymarch(full, w, h, 1.0, cam, tex, fx, prm_s)
                torch.cuda.synchronize()
                single_ref[tag + "_ms"] = round((time.perf_counter() - t1) / 5 * 1e3, 3)
            order.destroy()
            single_ref["same_bytes_as_the_gathered_frame"] = bool(torch.equal(full, fs.frame.view(-1)))
            del full
        barrier()

    # Second leg (single GPU only): the same frame in the WITHIN-TOLERANCE arithmetic modes -- RRT_ARITH_FMAD (multiply-adds
    # fused, roots and divisions still correctly rounded: the arithmetic class of the reference's own nvcc build) and
    # RRT_ARITH_FAST (also 1-ulp rsq, no correctly rounded divide).  Never `value`.  Their account is made right here on the
    # full-size frame (relativisticraytracer_amd/conditioning.py, the same function tests/test_gpu_tolerance.py asserts on):
    # every pixel of a mode's frame that is outside 1e-4 of the strict frame, takes another number of steps or has a byte off
    # by more than one LSB must be ILL-CONDITIONED -- a pixel the strict arithmetic itself moves out of the tolerance when its
    # primary direction is nudged by a few ulps.  `within_tolerance_mode` = the fastest mode whose account is clean.
    wtol = None
    if world == 1 and not args.no_fast:
        from relativisticraytracer_amd import conditioning
        strict_frame = fs.frame.clone()
        buf = torch.zeros_like(strict_frame)
        legs = {}
        for name, mode in (("fmad", 2), ("fast", 1)):
            prm_f = rrt.RenderParams(spin=args.spin, volumetrics=1, arith_mode=mode, noise_table=ntab.id if ntab else 0)
            for _ in range(max(1, args.warmup)):
                rrt.launch_raymarch(buf, w, h, 1.0, cam, tex, fx, prm_f)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                rrt.launch_raymarch(buf, w, h, 1.0, cam, tex, fx, prm_f)
            torch.cuda.synchronize()
            dtf = time.perf_counter() - t1
            legs[name] = {"arith_mode": mode, "value": round(w * h * args.steps / dtf / 1e6, 3), "unit": "Mrays/s",
                          "ms_per_step": round(dtf / args.steps * 1e3, 3), "fps": round(args.steps / dtf, 3)}
        _, ill, st = conditioning.account(tex, w, h, cam, 1.0, (2, 1), fx=fx, budget=240, spin=args.spin, volumetrics=1,
                                          noise_table=ntab.id if ntab else 0)
        for name, mode in (("fmad", 2), ("fast", 1)):
            q = st[mode]
            legs[name]["vs_strict_frame"] = q
            legs[name]["account_clean"] = (q["outliers_not_ill"] == 0 and q["steps_differ_not_ill"] == 0 and
                                           q["bytes_off_by_more_than_1_not_ill"] == 0)
        legs["fmad"]["note"] = ("RRT_ARITH_FMAD: the RK4 step with multiply-adds fused, sqrt and divide correctly rounded "
                                "(nvcc's defaults for the reference: -fmad=true, IEEE div/sqrt); media, sky, post-FX unchanged")
        legs["fast"]["note"] = "RRT_ARITH_FAST: fused multiply-adds, 1-ulp v_rsq, no correctly rounded divide in the RK4 step"
        # Which mode the 30 fps statement may rest on: one whose arithmetic is NOT NARROWER than the reference's
        # own build.  CMakeLists.txt: nvcc defaults = -fmad=true, IEEE divide and square root -> RRT_ARITH_FMAD is that class.
        # RRT_ARITH_FAST (1-ulp v_rsq, no correctly rounded divide) is narrower: timed and accounted, listed under `modes`, never
        # the credited mode however fast or clean it is.
        legs["fmad"]["credited"] = True
        legs["fast"]["credited"] = False
        legs["fast"]["not_credited_because"] = "narrower arithmetic than the reference's build (1-ulp rsq, no correctly rounded divide)"
        clean = [n for n in ("fmad",) if legs[n]["account_clean"]]
        best = clean[0] if clean else None
        ill_frac = st["ill"] / float(st["pixels"])
        wtol = {"mode": best, "fps": legs[best]["fps"] if best else None, "ms_per_step": legs[best]["ms_per_step"] if best else None,
                "value": legs[best]["value"] if best else None, "unit": "Mrays/s",
                "meets_30_fps": bool(best and legs[best]["fps"] >= 30.0),
                "credit_rule": "the fastest account-clean mode whose arithmetic is not narrower than the reference's nvcc-default build: FMAD only",
                "conditioning": {"pixels": st["pixels"], "ill_conditioned_pixels": st["ill"], "ill_fraction": round(ill_frac, 6),
                                 "nudged_strict_frames": st["nudged_frames"], "nudged_frames_budget": st["budget"],
                                 "uncovered_after_fixed_set": st["fixed_set"],
                                 "pixels_one_nudged_frame_moves_by_K_ulps": st["single_nudge_moves"], "tolerance": st["tolerance"],
                                 "statement": "every pixel of the mode's frame that is outside the tolerance of the strict frame, takes "
                                              "another number of steps or has a byte off by more than one LSB is a pixel the STRICT "
                                              "arithmetic itself moves out of the tolerance (or to another step count) when its primary "
                                              "direction is nudged by <= 16 ulps (rrt_params.nudge_ulps; *_not_ill counts are 0).  The map "
                                              "grows with the frames added (stopping rule: all deviant pixels covered, budget "
                                              "`nudged_frames_budget`); `uncovered_after_fixed_set` is the same count after a FIXED set of "
                                              "nudged frames chosen up front (the same for every view), and the bars that do not depend on "
                                              "the stopping rule -- outliers at the measured class and under ONE 4-ulp nudge -- are "
                                              "asserted by tests/test_gpu_tolerance.py (2)"},
                "modes": legs,
                "note": "never `value`: the headline, its roofline and the bit-parity claims are the strict path's"}

    # The headline frame WITHOUT the lattice-hash tables (every noise3D hashed arithmetically): the tables are a resident
    # input built outside the timed region, so the line carries the table-free time beside `value`.
    arith_noise = None
    if world == 1 and ntab is not None:
        prm_a = rrt.RenderParams(spin=args.spin, volumetrics=1)
        abuf = torch.zeros(h * w * 4, dtype=torch.uint8, device=dev)
        rrt.launch_raymarch(abuf, w, h, 1.0, cam, tex, fx, prm_a)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            rrt.launch_raymarch(abuf, w, h, 1.0, cam, tex, fx, prm_a)
        torch.cuda.synchronize()
        dta = time.perf_counter() - t1
        arith_noise = {"value": round(w * h * args.steps / dta / 1e6, 3), "unit": "Mrays/s", "ms_per_step": round(dta / args.steps * 1e3, 3),
                       "fps": round(args.steps / dta, 3), "same_bytes_as_headline": bool(torch.equal(abuf, fs.frame.view(-1))),
                       "note": "rrt_params.noise_table = 0: the same frame with no precomputed input but the sky"}
        del abuf

    # Informational third leg (single GPU): the same launch on a disk-heavy view -- the "Horizon Skimmer" keyframe
    # (camera_paths.cpp:62) from inside the disk, where media sampling is ~40 % of the work -- strict arithmetic,
    # with and without the noise tables.  The headline `value` is the BASELINE view above.
    heavy = None
    if world == 1 and not args.no_heavy:
        hcam = rrt.CameraState.from_angles((4.2, 0.6, 4.2), -90.0, -5.7)
        ht = 14.0
        hbuf = torch.zeros(h * w * 4, dtype=torch.uint8, device=dev)
        res = {}
        horder = rrt.TileOrder()        # "noise_table_cost_ordered": rrt_tile_order, each frame dispatched longest-first by the previous one's costs
        for tag, tab, oid in (("arithmetic_noise", 0, 0), ("noise_table", ntab.id if ntab else 0, 0),
                              ("noise_table_cost_ordered", ntab.id if ntab else 0, horder.id)):
            if tag != "arithmetic_noise" and not ntab:
                continue
            hp = rrt.RenderParams(spin=args.spin, volumetrics=1, noise_table=tab, tile_order=oid)
            rrt.launch_raymarch(hbuf, w, h, ht, hcam, tex, fx, hp)
            rrt.launch_raymarch(hbuf, w, h, ht, hcam, tex, fx, hp)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(max(2, args.steps // 2)):
                rrt.launch_raymarch(hbuf, w, h, ht, hcam, tex, fx, hp)
            torch.cuda.synchronize()
            hms = (time.perf_counter() - t1) / max(2, args.steps // 2) * 1e3
            res[tag] = {"ms_per_step": round(hms, 3), "Mrays_per_s": round(w * h / hms / 1e3, 3), "fps": round(1e3 / hms, 3)}
        horder.destroy()
        # the FIRST frame of that view (a still image, frame 1 of a sequence): an rrt_tile_order object without history takes
        # its order from a coarse march-only probe of the view itself (one ray per 16x16 pixels, probe + sort inside the time);
        # the object's buffers are sized beforehand by a launch of another geometry, as a host that reuses one object would
        if ntab:
            firsts = []
            for _ in range(3):
                fo = rrt.TileOrder()
                hp = rrt.RenderParams(spin=args.spin, volumetrics=1, noise_table=ntab.id, tile_order=fo.id)
                big = torch.zeros((h + 8) * w * 4, dtype=torch.uint8, device=dev)
                rrt.launch_raymarch(big, w, h + 8, ht, hcam, tex, fx, hp)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); rrt.launch_raymarch(hbuf, w, h, ht, hcam, tex, fx, hp); e1.record()
                torch.cuda.synchronize()
                firsts.append(e0.elapsed_time(e1))
                fo.destroy(); del big
            res["noise_table_first_frame_probe_ordered"] = {"ms_per_step": round(min(firsts), 3), "Mrays_per_s": round(w * h / min(firsts) / 1e3, 3),
                                                            "fps": round(1e3 / min(firsts), 3)}
        heavy = {"view": "Horizon Skimmer key (4.2, 0.6, 4.2) yaw -90 pitch -5.7, t=14.0, same size / spin / effects", **res}
        if "noise_table_first_frame_probe_ordered" in heavy:
            heavy["noise_table_first_frame_probe_ordered"]["note"] = ("first launch through a fresh rrt_tile_order object: order from the coarse probe of "
                                                                      "the view (probe + sort inside the time); same bytes")
        if "noise_table_cost_ordered" in heavy:
            heavy["noise_table_cost_ordered"]["note"] = ("rrt_params.tile_order: wave tiles dispatched longest-first, costs measured by the previous "
                                                         "launch (the sort behind every frame is inside the time); same bytes")

    # Informational fourth leg (single GPU): what 8 ranks would do with THIS frame, projected on this GPU -- every rank's share (16-row
    # tiles t mod 8) rendered alone the way a rank of `bench.py --gpus 8` renders it (three frames in flight on three streams; the
    # three-pass path in one chain per launch, and the plain single kernel), sustained ms per frame; the slowest shard bounds the frame
    # rate.  Gather / assemble are not in it (4.15 MB per rank on another stream).  A projection, not a measurement of 8 GPUs.
    projection = None
    if world == 1 and not args.no_heavy and ntab and h >= 8 * R:
        NP, slots = 8, 3
        p_pools = [rrt.Workspace(2 << 30) for _ in range(slots)]
        p_streams = [torch.cuda.Stream(dev) for _ in range(slots)]
        p_bufs = [torch.zeros(sharding.shard_rows(h, R, 0, NP) * w * 4, dtype=torch.uint8, device=dev) for _ in range(slots)]

        def sustained(shard, policy):
            prm3 = [rrt.RenderParams(spin=args.spin, volumetrics=1, noise_table=ntab.id, workspace=p_pools[j].id, path_policy=policy, pass_chains=1)
                    for j in range(slots)]
            best = 1e9
            for _ in range(3):
                torch.cuda.synchronize(dev)
                t1 = time.perf_counter()
                for k in range(4 * slots):
                    rrt.launch_raymarch_tiles(p_bufs[k % slots], w, h, R, shard, NP, 1.0, cam, tex, fx, prm3[k % slots], stream=p_streams[k % slots])
                torch.cuda.synchronize(dev)
                best = min(best, (time.perf_counter() - t1) * 1e3 / (4 * slots))
            return best
        tp = [sustained(sh, 2) for sh in range(NP)]
        il = [sustained(sh, 1) for sh in range(NP)]
        best_rank = [min(a_, b_) for a_, b_ in zip(tp, il)]
        one = dt / args.steps * 1e3
        projection = {"ranks": NP, "frames_in_flight": slots, "tile_rows": R, "single_gpu_ms": round(one, 3),
                      "three_pass_one_chain": {"per_shard_ms": [round(v, 3) for v in tp], "max_ms": round(max(tp), 3), "x_of_single_gpu": round(one / max(tp), 2)},
                      "single_kernel": {"per_shard_ms": [round(v, 3) for v in il], "max_ms": round(max(il), 3), "x_of_single_gpu": round(one / max(il), 2)},
                      "faster_path_per_rank": {"max_ms": round(max(best_rank), 3), "x_of_single_gpu": round(one / max(best_rank), 2)},
                      "note": "PROJECTION on one GPU, not a multi-GPU measurement: each of the 8 shares of this frame alone on this GPU, sustained with "
                              "three frames in flight as a rank of `bench.py --gpus 8` keeps them (which also picks the faster path per rank at start-up); "
                              "the slowest share bounds the frame rate; the RCCL gather (4.15 MB per rank) and the assemble kernel are not included"}
        for pw in p_pools:
            pw.destroy()
        del p_bufs

    if rank == 0:
        rays = w * h
        ms_per_step = dt / args.steps * 1e3
        value = rays * args.steps / dt / 1e6
        k_ms = float(np.mean(kernel_ms))
        k_note = "HIP-event kernel time"
        # what the timed frames did, not only their mean: a slow box shows in min and median alike, a slow
        # first frame in max / slowest_frame only.  No hidden warm-up: these are exactly the K timed frames.
        k_stats = {"kernel_ms_min": round(float(np.min(kernel_ms)), 3), "kernel_ms_median": round(float(np.median(kernel_ms)), 3),
                   "kernel_ms_max": round(float(np.max(kernel_ms)), 3), "kernel_ms_slowest_frame": int(np.argmax(kernel_ms)),
                   "kernel_ms_per_frame": [round(float(v), 3) for v in kernel_ms]}
        if fs.pipeline:
            # several frames in flight: consecutive launches overlap on the device, so a launch's own start-to-end
            # time is not its cost; price the rank's share against the whole step instead (gather included)
            k_ms, k_note = ms_per_step, "step time (launches of consecutive frames overlap; gather included)"
        my_rays = sharding.shard_rows(h, R, 0, world) * w

        # The CPU leg runs on rank 0 for every N (north_star: "alongside a host-compiled OpenMP loop ... in the same run"); with
        # N > 1 the other ranks wait at the barrier below, still under the RUN watchdog (not the 60 s shutdown one).
        cpu, means = (None, None)
        if args.cpu_stride != 0:
            nthr = cpu_threads(world)
            stride = args.cpu_stride if args.cpu_stride > 0 else cpu_sample_stride(w, h, args.spin, sky_np, nthr)
            cpu, means = cpu_baseline(w, h, args.spin, stride, sky_np, nthreads=nthr)
            if world > 1:
                cpu["note"] = ("timed on rank 0 while the other %d ranks waited at a barrier; cores = the threads it actually ran on "
                               "(all cores this process may use: the launcher's OMP_NUM_THREADS is per rank)" % (world - 1))
        if means is None:   # per-ray work from a small oracle sample even when the baseline is skipped
            _, means = cpu_baseline(w, h, args.spin, 48, sky_np)
        opr = ops_per_ray(means["steps"], means["n_noise"], means["n_dens"], means["n_samples"])
        tops = opr * my_rays / (k_ms * 1e-3) / 1e12
        hbm_gbs = ALGO_BYTES_PER_RAY * my_rays / (k_ms * 1e-3) / 1e9
        # HBM traffic per launch comes from rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE, tools/profile_round.sh),
        # which cannot run inside this process.  The record is only used when it was measured on THIS build
        # (hash of the kernel sources) and workload; otherwise `traffic` is null rather than stale.
        traffic, traffic_note = None, "not measured for this build: run tools/profile_round.sh + tools/summarize_profile.py"
        tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(tpath) and world == 1:
            try:
                tj = json.load(open(tpath))
                if tj.get("workload") == f"{w}x{h}_a{args.spin:g}_vol" and tj.get("source_hash") == source_hash():
                    traffic = tj.get("bytes_per_launch")
                    traffic_note = "rocprofv3 PMC, " + str(tj.get("from", "profiles/"))
                else:
                    traffic_note = "profiles/hbm_traffic.json is from another build or workload; not used"
            except Exception:
                traffic = None
        # The counter view of the same kernel (rocprofv3 PMC, tools/pass_counters.sh): SQ_INSTS_VALU x 2 clocks / (1024 SIMDs x busy
        # clocks) -- the share of all VALU issue slots the kernel filled; quoted only if measured on THIS build's kernel sources.
        issue_util, issue_note = None, "not measured for this build: run tools/final_set.sh + tools/summarize_pass_counters.py"
        ppath0 = os.path.join(ROOT, "profiles", "pass_counters.json")
        if os.path.exists(ppath0) and world == 1 and (w, h) == (3840, 2160):
            try:
                pj0 = json.load(open(ppath0))
                if pj0.get("source_hash") == source_hash():
                    k0 = pj0.get("runs", {}).get("default/single", {}).get("raymarch_pixels")
                    if k0:
                        issue_util = k0.get("issue_slot_util")
                        issue_note = ("rocprofv3 PMC, " + str(pj0.get("from", "profiles/")) + ": SQ_INSTS_VALU %.4g per launch x 2 clocks / (1024 SIMDs x "
                                      "GRBM_GUI_ACTIVE / 8), %.2f ms under the profiler" % (k0.get("valu_insts") or 0.0, k0.get("ms_per_frame") or 0.0))
                else:
                    issue_note = "profiles/pass_counters.json is from another build; not used"
            except Exception:
                issue_util = None
        if heavy is not None and args.cpu_stride != 0:
            # per-ray work of the heavy view from a small oracle sample (exact counts, like the headline's)
            _, hm = cpu_baseline(w, h, args.spin, 96, sky_np, cam_arr=rrt.CameraState.from_angles((4.2, 0.6, 4.2), -90.0, -5.7).as_array(),
                                 time_=14.0)
            heavy["per_ray_means"] = {k: round(v, 2) for k, v in hm.items()}
            heavy["ops_per_ray"] = round(ops_per_ray(hm["steps"], hm["n_noise"], hm["n_dens"], hm["n_samples"]), 1)
            # counters of this view's kernels: issue-slot utilisation and HBM bytes from rocprofv3 PMC passes
            # (tools/pass_counters.sh), quoted only if the record was measured on THIS build's kernel sources
            pc_note = "not measured for this build: run tools/pass_counters.sh + tools/summarize_pass_counters.py"
            pcs = {}
            ppath = os.path.join(ROOT, "profiles", "pass_counters.json")
            if os.path.exists(ppath):
                try:
                    pj = json.load(open(ppath))
                    if pj.get("source_hash") == source_hash():
                        pcs = pj.get("runs", {})
                        pc_note = "rocprofv3 PMC, " + str(pj.get("from", "profiles/")) + "; " + str(pj.get("note", ""))
                    else:
                        pc_note = "profiles/pass_counters.json is from another build; not used"
                except Exception:
                    pcs = {}
            for tag, run in (("arithmetic_noise", None), ("noise_table", "skimmer/single"), ("noise_table_cost_ordered", "skimmer/single_ordered"),
                             ("noise_table_first_frame_probe_ordered", None)):
                if tag in heavy:
                    k = pcs.get(run, {}).get("raymarch_pixels") if run else None
                    heavy[tag]["issue_slot_util"] = k.get("issue_slot_util") if k else None
                    heavy[tag]["traffic_bytes"] = k.get("hbm_bytes_upper") if k else None
                    heavy[tag]["source_ops_per_s_over_2p4ghz_peak"] = round(heavy["ops_per_ray"] * rays / (heavy[tag]["ms_per_step"] * 1e-3) / 1e12
                                                                            / VALU_PEAK_TOPS, 4)
            heavy["counters_note"] = pc_note + ("  (source_ops_per_s_over_2p4ghz_peak is a FORMULA -- SURVEY 8d's source operations, which price 223 per "
                                                "noise3D that the tables replace by loads -- not a utilisation; it can exceed 1)")
            if "skimmer/shard0of8" in pcs:
                heavy["rank_share_three_pass_kernels"] = pcs["skimmer/shard0of8"]
        line = {
            "metric": "Mrays/s", "value": round(value, 3), "unit": "Mrays/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
            "fps": round(1e3 / ms_per_step, 3),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"{w}x{h} Kerr a={args.spin:g} full volumetric disk+dust, default camera "
                                   f"(0,10,-60) yaw 0 pitch -10, t=1.0, default effects, synthetic 2048x1024 sky seed 1",
                       "rays_per_frame": rays, "max_steps": 2000, "arith_mode": "strict (byte-identical to the CPU restatement of the reference; vs the reference's own kernel body compiled by g++: step counts identical, <= 1 LSB on <= 1e-4 of the bytes)",
                       "parallelism": (f"rowtiles{R}x{world}" + (", %d frames in flight (the next renders overlap gather/assemble of frame k)" % fs.n_slots if fs.pipeline else ""))
                                      if world > 1 else "single",
                       "path": ("auto: three-pass below 1.5 M rays per launch, %d GiB pool" % args.workspace_gib) if ws else "single kernel",
                       "noise_table": ("lattice-hash tables, t_max 32 s, %.0f MB, built once in %.1f ms (outside the timed region)"
                                       % (ntab.info()["bytes"] / 1e6, table_build_ms)) if ntab else "none (arithmetic hash)",
                       "dist_backend": (backend + (" (RCCL)" if backend == "nccl" else " (rehearsal: ranks share a card)")) if world > 1 else None,
                       "comm_ranks": dist.get_world_size() if world > 1 else 1},
            "roofline": {"bound": "valu", "achieved": round(tops, 3), "peak": round(VALU_PEAK_TOPS, 2),
                         "unit": "TFLOP/s", "frac": round(tops / VALU_PEAK_TOPS, 4), "traffic": traffic,
                         "clock_ghz": round(clock_ghz, 4) if clock_ghz else None,
                         "frac_at_held_clock": round(tops / (256 * 4 * 32 * clock_ghz * 1e9 / 1e12), 4) if clock_ghz else None,
                         "clock_note": "shader clock held under this workload: s_memtime / s_memrealtime of a one-wave probe (rrt_clock_probe) "
                                       "running " + str(clock_when) + "; `peak` prices 2.4 GHz, frac_at_held_clock the clock the chip actually ran at "
                                       "(boxes of this pool hold 2.2-2.4 GHz); > 1 is possible: the peak counts one SOURCE operation per lane and clock "
                                       "and the kernel needs 0.955 instructions per source operation",
                         "traffic_note": traffic_note,
                         "issue_slot_util": issue_util, "issue_slot_util_note": issue_note,
                         "frac_of_fma_peak_157.3": round(tops / 157.3, 4),
                         "kernel": "raymarch_pixels", "kernel_ms": round(k_ms, 3), **k_stats,
                         "ops_per_ray": round(opr, 1), "per_ray_means": {k: round(v, 2) for k, v in means.items()},
                         "note": "source-level unfused FP32 ops (SURVEY 8d formula) / " + k_note + "; "
                                 "peak = 256CU x 4SIMD x 32 lanes x 2.4 GHz = one unfused FP32 op per lane per clock "
                                 "(the datasheet's 157.3 TFLOP/s counts an FMA as two)",
                         "hbm": {"bound": "hbm", "achieved": round(hbm_gbs, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                 "frac": round(hbm_gbs / HBM_PEAK_GBS, 6),
                                 "note": "algorithmic 52 B/ray; the path is not HBM-bound"}},
            "cpu_baseline": cpu,
            "within_tolerance_mode": wtol,
            "headline_arithmetic_noise": arith_noise,
            "projection_8_ranks": projection,
            "heavy_view": heavy,
        }
        if world > 1:
            best_single = min(single_ref["static_order_ms"], single_ref["cost_ordered_ms"])
            line["multi_gpu"] = {"single_gpu_reference_ms": best_single, "single_gpu_reference": single_ref,
                                 "speedup_vs_single_gpu_reference": round(best_single / ms_per_step, 3),
                                 "single_gpu_reference_note": "the same frame on rank 0's GPU alone, single kernel, noise tables, best of static and "
                                                              "cost-ordered dispatch, measured in this run while the other ranks waited",
                                 "phases": phases, "one_frame_at_a_time": one_at_a_time,
                                 "frames_in_flight": fs.n_slots,
                                 "path_per_launch": path_tuning if fs.pipeline else "automatic (three-pass in two chains for a small share)",
                                 "frames_in_flight_note": "default 3: chosen on ONE GPU rendering a single rank's share "
                                                          "(profiles/r02_frames_in_flight.txt); provisional until a run on >= 2 GPUs"}
        
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()          # ranks > 0 wait here for rank 0's CPU leg and line, under the run watchdog
    dog.disarm()
    if world > 1:
        dog.arm(60.0, "process group shutdown")
        dist.barrier()
        dist.destroy_process_group()
        dog.disarm()
    for p in pools:
        p.destroy()
    if ntab:
        ntab.destroy()
    tex.destroy()


if __name__ == "__main__":
    main()
