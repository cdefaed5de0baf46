"""Headless frame driver: the reference's main loop (src/main.cpp:505-529) without a window.

    python -m relativisticraytracer_amd.headless --width 1920 --height 1080 --spin 0.9 \\
           --path 0 --frames 300 [--out frames.rgba | --out ppm_dir/] [--all-effects]
    python -m torch.distributed.run --nproc-per-node 8 -m relativisticraytracer_amd.headless ...

Per frame k = 1..N it does what `main()` does while recording: advance the fixed 1/24 s clock
(float accumulators, main.cpp:511-516), take the camera from the active path
(getInterpolatedState, :176-203) or the fixed start-up camera, render (launch_raymarch, :467),
hand the pixels to the sink (captureFrame, :85-97).  With several ranks every frame is
row-tile sharded and gathered to rank 0 (sharding.FrameSharder).  Prints one JSON line of metrics.
"""
import argparse
import json
import os
import sys
import time


def main(argv=None):
    ap = argparse.ArgumentParser(prog="relativisticraytracer_amd.headless")
    ap.add_argument("--width", type=int, default=1000)          # WINDOW_WIDTH, config.h:7
    ap.add_argument("--height", type=int, default=700)          # WINDOW_HEIGHT, config.h:8
    ap.add_argument("--frames", type=int, default=24)
    ap.add_argument("--fps", type=int, default=24)              # RECORDING_FPS, config.h:9
    ap.add_argument("--spin", type=float, default=0.0)          # SPIN_A, config.h:21
    ap.add_argument("--path", type=int, default=-1, help="built-in camera path 0..2; -1 = fixed start-up camera")
    ap.add_argument("--no-volumetrics", action="store_true")
    ap.add_argument("--fast", action="store_true", help="RRT_ARITH_FAST (not the parity path); = --arith fast")
    ap.add_argument("--arith", choices=("strict", "fmad", "fast"), default=None,
                    help="arithmetic of the RK4 step: strict (default; bit-identical to the oracle), fmad (multiply-adds fused, roots and "
                         "divisions correctly rounded: the class of the reference's nvcc-default build), fast (also 1-ulp rsq)")
    ap.add_argument("--path-window", type=int, default=0,
                    help="several frames in flight, a share under the three-pass threshold: frames per window of the per-rank path choice "
                         "(rrt_path_chooser; 0 = 48); -1: no choice, the three-pass path throughout")
    ap.add_argument("--all-effects", action="store_true", help="also enable chromatic aberration (key C)")
    ap.add_argument("--sky", default=None, help="equirectangular image file; default: synthetic sky, seed 1")
    ap.add_argument("--tile-rows", type=int, default=16)
    ap.add_argument("--frames-in-flight", type=int, default=3,
                    help="several ranks: frames rendered / gathered / assembled concurrently per rank (>= 2)")
    ap.add_argument("--workspace-gib", type=int, default=8,
                    help="per-rank pool for the three-pass path, used by launches of <= 1.5 M rays (0 = single kernel only)")
    ap.add_argument("--tile-order", choices=("auto", "on", "off"), default="auto",
                    help="cost-ordered dispatch (rrt_tile_order): every frame's wave tiles go out longest-first by the costs the "
                         "previous frame (of the same slot) measured.  auto: on when frames are rendered one at a time, off when "
                         "several are in flight (their drains already overlap)")
    ap.add_argument("--no-noise-table", action="store_true",
                    help="hash every noise3D corner arithmetically (default: lattice-hash tables over a sliding window of the clock)")
    ap.add_argument("--noise-table-gib", type=float, default=2.0,
                    help="per-GPU byte budget of the noise tables: the window of sim time one table covers (and, for long "
                         "sequences, its coverage) is chosen to fit; the table is rebuilt when the clock leaves the window")
    ap.add_argument("--out", default=None, help="x.rgba (raw, bottom-up) | dir/ (PPM per frame) | x.mp4 (needs ffmpeg)")
    ap.add_argument("--init-timeout", type=float, default=300.0,
                    help="several ranks: seconds the process-group bring-up may take before the run exits non-zero with "
                         "the tracebacks of all threads, instead of hanging")
    ap.add_argument("--frame-timeout", type=float, default=120.0, help="the same for any single frame (0: no limit)")
    args = ap.parse_args(argv)

    t_start = time.perf_counter()

    def trace(what):        # RRT_HEADLESS_TRACE=1: where the start-up time goes (stderr), as in csrc/rrt_headless.cpp
        dest = os.environ.get("RRT_HEADLESS_TRACE")        # "1": stderr; anything else: a file to append to
        if dest:
            line = f"[headless.py pid {os.getpid()} +{time.perf_counter() - t_start:7.2f}s] {what}"
            if dest == "1":
                print(line, file=sys.stderr, flush=True)
            else:
                with open(dest, "a") as fh:
                    print(line, file=fh)

    trace("importing torch")
    import torch
    trace("torch imported")
    import relativisticraytracer_amd as rrt
    from relativisticraytracer_amd import camera_paths, sharding, sinks
    from relativisticraytracer_amd.sky import load_sky, synthetic_sky

    world = int(os.environ.get("WORLD_SIZE", "1")); rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        sys.exit("the headless driver needs a GPU (no CPU fallback)")
    backend = os.environ.get("RRT_DIST_BACKEND", "nccl")        # "gloo": rehearsal on fewer GPUs than ranks
    if backend == "gloo":
        local_rank %= max(1, torch.cuda.device_count())
    trace("GPU visible")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dog = sharding.Watchdog(f"headless.py rank {rank}")
    if world > 1:
        sharding.single_node_environment()
        lib = sharding.torch_rccl_library() if backend == "nccl" else None
        warm = sharding.warm_library_pages(lib) if lib else None
        dog.arm(args.init_timeout, "process group / communicator bring-up")
        dist = sharding.init_process_group(backend, rank, world, dev, timeout_s=args.init_timeout)
        if warm is not None:
            warm.join()

    trace("process group ready" if world > 1 else "single rank")
    w, h = args.width, args.height
    tex = rrt.SkyTexture(load_sky(args.sky) if args.sky else synthetic_sky())
    fx = rrt.CameraEffects(useChromaticAberration=bool(args.all_effects))
    # with several ranks --frames-in-flight frames are in flight (FrameSharder pipeline mode), each with its own
    # share of the pool
    n_slots = max(2, args.frames_in_flight) if world > 1 else 1
    pools = [rrt.Workspace((args.workspace_gib << 30) // n_slots) for _ in range(n_slots)] if args.workspace_gib > 0 else []
    # lattice-hash tables for the volumetric noise over a sliding window of the recording clock (main.cpp:511-516 lets
    # simTime grow without bound; the table's size grows with it): one table within the byte budget, rebuilt when the
    # clock leaves its window -- never a silent fall-back: frames rendered without a table are counted in the summary
    t_end, _ = camera_paths.recording_clock(max(args.frames, 1), args.fps)
    nwin = rrt.NoiseWindows(float(t_end) + 1.0, int(args.noise_table_gib * (1 << 30)), sync=torch.cuda.synchronize,
                            enabled=not args.no_noise_table and not args.no_volumetrics)
    # cost-ordered dispatch pays on single-kernel launches whose frames do not overlap; on the three-pass path (small launches
    # with a pool) it piles the expensive tiles into the first of the two chains and was measured slower: auto leaves it off there
    my_rays = w * sharding.shard_rows(h, args.tile_rows, rank, world)
    three_pass_likely = bool(pools) and my_rays <= rrt._lib.load().rrt_path_auto_max_rays()      # RRT_PATH_AUTO's own threshold
    use_order = args.tile_order == "on" or (args.tile_order == "auto" and n_slots == 1 and not three_pass_likely)
    orders = [rrt.TileOrder() for _ in range(n_slots)] if use_order else []
    arith_name = args.arith or ("fast" if args.fast else "strict")
    arith_mode = {"strict": 0, "fast": 1, "fmad": 2}[arith_name]
    # the path of a small share under frames in flight is chosen per window by measurement (sharding.PathChooser; the rule and the
    # numbers behind it: csrc/rrt_path_chooser.cpp); RRT_PATH_POLICY pins it
    chooser = (sharding.PathChooser(n_slots, args.path_window)
               if (args.path_window >= 0 and n_slots >= 2 and three_pass_likely and "RRT_PATH_POLICY" not in os.environ) else None)
    ends = {}                      # frame -> the event at the end of its render on this rank
    prms = [rrt.RenderParams(spin=args.spin, volumetrics=0 if args.no_volumetrics else 1,
                             noise_table=0, tile_order=orders[j].id if orders else 0,
                             arith_mode=arith_mode, workspace=pools[j].id if pools else 0,
                             # frames in flight fill each other's drains: ONE chain per launch (the second chain's streams only compete with
                             # the other frames: 2-7 % per frame, profiles/r05_sustained_chains.txt).  Which PATH a small share takes --
                             # the three-pass path, or the plain single kernel, which is 6-8 % faster unless the share holds a wavefront that
                             # outlasts the frames in flight -- is chosen per window by measurement (chooser, above; round 6)
                             pass_chains=1 if n_slots >= 2 else 0,
                             path_policy=int(os.environ.get("RRT_PATH_POLICY", "0"))) for j in range(n_slots)]
    path = camera_paths.CameraPath(args.path) if args.path >= 0 else None
    state = {"t": 0.0, "cam": rrt.CameraState.default(), "table": 0, "k": 0}

    def render(buf, slot):
        prms[slot].noise_table = state["table"]
        k = state["k"]
        if chooser is not None:
            prms[slot].path_policy = chooser.policy(k)
        rrt.launch_raymarch_tiles(buf, w, h, args.tile_rows, rank, world, state["t"], state["cam"], tex, fx, prms[slot])
        if chooser is not None:
            e = torch.cuda.Event(enable_timing=True)
            e.record()                                        # on the slot's stream (FrameSharder runs the callback inside it)
            ends[k] = e
            # sustained time of the frames that have finished since: the interval between consecutive frames' render ends
            for j in sorted(ends):
                if j - 1 in ends and ends[j].query() and ends[j - 1].query():
                    # (a frame that ends BEFORE its predecessor -- the predecessor holds a long wavefront -- reports 0; the
                    # predecessor's own interval is then the long one, which is what the outlier rule looks for)
                    chooser.report(j, max(0.0, ends[j - 1].elapsed_time(ends[j])))
                    del ends[j - 1]

    def assemble(frame, buf, shard):
        rrt.assemble_tiles(frame, buf, w, h, args.tile_rows, shard, world)

    def assemble_all(frame, bufs, stride):
        rrt.assemble_all_tiles(frame, bufs, stride, w, h, args.tile_rows, world)

    # with several ranks the next frames render while frame k is gathered and assembled (frames arrive late, in order)
    fs = sharding.FrameSharder(w, h, args.tile_rows, rank, world, dev, render, assemble, assemble_all=assemble_all,
                               pipeline=n_slots if world > 1 else False)
    sink = sinks.open_sink(args.out, w, h, args.fps) if rank == 0 else None
    host = torch.empty(h * w * 4, dtype=torch.uint8, pin_memory=True) if sink else None

    def deliver(frame):
        host.copy_(frame, non_blocking=False)
        sink.write(host.numpy().reshape(h, w, 4))

    torch.cuda.synchronize()
    trace("sky, pools, sink ready; first frame")
    t0 = time.perf_counter()
    for k in range(1, args.frames + 1):
        dog.arm(args.frame_timeout + (args.init_timeout if k == 1 else 0.0), f"frame {k}")    # frame 1 brings the communicator's channels up
        sim_t, path_t = camera_paths.recording_clock(k, args.fps)
        state["t"] = sim_t
        state["k"] = k
        state["table"] = nwin.table_id(sim_t)
        if path is not None:
            state["cam"] = path.camera_at(path_t)
        frame = fs.step()
        if sink and frame is not None:
            deliver(frame)
    for frame in fs.drain():                # the frames still in flight, in order
        if sink:
            deliver(frame)
    torch.cuda.synchronize()
    trace("frames done")
    dog.arm(args.frame_timeout, "final barrier")
    if world > 1:
        dist.barrier()
    dog.disarm()
    dt = time.perf_counter() - t0
    if rank == 0:
        if sink:
            sink.close()
        print(json.dumps({"frames": args.frames, "width": w, "height": h, "n_gpus": world, "seconds": round(dt, 4),
                          "fps": round(args.frames / dt, 3), "Mrays_per_s": round(args.frames * w * h / dt / 1e6, 3),
                          "path": path.name if path else None, "spin": args.spin,
                          "arith_mode": arith_name, "sink": args.out,
                          "path_choice": chooser.stats() if chooser else None,
                          "noise_tables": nwin.summary(),
                          "tile_order": orders[0].info() if orders else None}), flush=True)
    if world > 1:
        dist.destroy_process_group()
    nwin.close()
    if chooser is not None:
        chooser.destroy()
    for o in orders:
        o.destroy()
    tex.destroy()


if __name__ == "__main__":
    main()
