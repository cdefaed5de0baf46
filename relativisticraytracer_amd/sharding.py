"""Image-plane sharding for multi-GPU rendering (no counterpart in the
single-GPU reference).

The frame is cut into row tiles of `tile_rows` image rows; tile t belongs to shard
t mod n_shards (interleaved, because contiguous row blocks are badly load-balanced:
the rows through the disk and the shadow cost several times the sky rows).  Every
rank renders its tiles into one compact buffer (tile-major, each tile bottom-up --
the layout rrt_launch_raymarch_tiles writes), ONE gather moves the buffers to rank 0,
and rank 0 scatters them into the bottom-up frame (rrt_assemble_tiles).

This module is backend-agnostic plumbing over torch.distributed: with the "nccl"
backend (= RCCL over xGMI) the buffers are device tensors and `render`/`assemble`
are the HIP entry points; the CPU tests drive the same code over "gloo" with the
oracle as the renderer.
"""
import numpy as np


def tile_plan(height, tile_rows, shard, n_shards, shard_of_tile=None):
    """[(tile index t, first image row y0, rows in tile)] for `shard`, in buffer order (increasing t).
    shard_of_tile: an explicit assignment (rrt_tile_map; one entry per tile) instead of t mod n_shards."""
    if height <= 0 or tile_rows <= 0 or n_shards <= 0 or not (0 <= shard < n_shards):
        raise ValueError("bad tile plan arguments")
    n_tiles = (height + tile_rows - 1) // tile_rows
    if shard_of_tile is None:
        mine = range(shard, n_tiles, n_shards)
    else:
        if len(shard_of_tile) != n_tiles or any(not (0 <= int(v) < n_shards) for v in shard_of_tile):
            raise ValueError("shard_of_tile: one shard index per row tile")
        mine = [t for t in range(n_tiles) if int(shard_of_tile[t]) == shard]
    return [(t, t * tile_rows, min(tile_rows, height - t * tile_rows)) for t in mine]


def shard_rows(height, tile_rows, shard, n_shards, shard_of_tile=None):
    return sum(rows for _, _, rows in tile_plan(height, tile_rows, shard, n_shards, shard_of_tile))


def max_shard_rows(height, tile_rows, n_shards, shard_of_tile=None):
    return max(shard_rows(height, tile_rows, s, n_shards, shard_of_tile) for s in range(n_shards))


def assemble_numpy(frame, tiles, width, height, tile_rows, shard, n_shards, shard_of_tile=None):
    """Host restatement of rrt_assemble_tiles / rrt_assemble_all_tilemap (used by the CPU tests).
    frame: (height, width, 4) bottom-up; tiles: (>= shard_rows, width, 4)."""
    for k, (t, y0, rows) in enumerate(tile_plan(height, tile_rows, shard, n_shards, shard_of_tile)):
        src = tiles[k * tile_rows:k * tile_rows + rows]          # tile-major; tile stored bottom-up
        frame[height - (y0 + rows):height - y0] = src
    return frame


def extract_numpy(frame, width, height, tile_rows, shard, n_shards, pad_rows=None, shard_of_tile=None):
    """Inverse of assemble_numpy: the compact tile buffer of `shard` cut from a full frame."""
    n = shard_rows(height, tile_rows, shard, n_shards, shard_of_tile)
    out = np.zeros((pad_rows if pad_rows is not None else n, width, 4), frame.dtype)
    for k, (t, y0, rows) in enumerate(tile_plan(height, tile_rows, shard, n_shards, shard_of_tile)):
        out[k * tile_rows:k * tile_rows + rows] = frame[height - (y0 + rows):height - y0]
    return out


# ---------------------------------------------------------------- bring-up of a multi-rank run (bench.py, headless.py)
def single_node_environment(env=None):
    """Defaults for a one-node RCCL job, set only where the caller has not chosen: warnings from RCCL (it says nothing
    below WARN, and a failing bring-up should explain itself), bootstrap / RAS sockets on the loopback when the
    rendezvous address is the loopback (RCCL picks the first non-loopback interface by default -- in a container a veth
    whose state is not this program's business), dmabuf IPC (the only kind this driver stack supports)."""
    import os
    env = os.environ if env is None else env
    env.setdefault("NCCL_DEBUG", "WARN")
    if env.get("MASTER_ADDR", "127.0.0.1") in ("127.0.0.1", "localhost", "::1"):
        env.setdefault("NCCL_SOCKET_IFNAME", "lo")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return env


def warm_library_pages(path):
    """Read a shared library's file front to back in a background thread, so that its pages sit in the page cache before
    the loader and the HIP runtime fault them in one by one.  RCCL's library is a 0.3-0.6 GB fat binary; its first
    collective loads the gfx950 code object out of it -- 5.5 s of a 5.6 s communicator bring-up on a healthy box with a
    cold cache, minutes on a box whose storage is slow.  Returns the thread (join() is optional)."""
    import threading

    def read():
        try:
            with open(path, "rb", buffering=0) as fh:
                while fh.read(8 << 20):
                    pass
        except OSError:
            pass

    th = threading.Thread(target=read, name="warm_library_pages", daemon=True)
    th.start()
    return th


def torch_rccl_library():
    """Path of the RCCL library torch will load (its bundled one), or None."""
    import os
    try:
        import torch
    except Exception:
        return None
    p = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
    return p if os.path.exists(p) else None


class Watchdog:
    """A run that stops making progress ends with a diagnosis instead of hanging: `arm(seconds, what)` (re-)starts a
    countdown; when it expires the tracebacks of all threads go to stderr with `what` and the process exits with status 1
    (faulthandler: no Python-level cooperation needed, works inside a blocked collective).  Nothing is retried."""

    def __init__(self, label):
        self.label = label
        self.armed = False

    def arm(self, seconds, what):
        """(Re-)start the countdown.  The previous one is ALWAYS cancelled first, so arm(0) / arm(None) means "no limit from
        here on" and not "keep counting down the limit of the phase before" (headless.py re-arms every frame, and
        with --frame-timeout 0 frame 1's bring-up allowance went on to kill a healthy run)."""
        import faulthandler
        import sys
        faulthandler.cancel_dump_traceback_later()
        self.armed = False
        if seconds and seconds > 0:
            if self.verbose():
                print(f"[{self.label}] watchdog: {what} (limit {seconds:.0f} s)", file=sys.stderr, flush=True)
            faulthandler.dump_traceback_later(seconds, repeat=False, file=sys.stderr, exit=True)
            self.armed = True

    @staticmethod
    def verbose():
        import os
        return bool(os.environ.get("RRT_HEADLESS_TRACE"))

    def disarm(self):
        import faulthandler
        if self.armed:
            faulthandler.cancel_dump_traceback_later()
            self.armed = False


def init_process_group(backend, rank, world, device=None, timeout_s=300.0):
    """torch.distributed bring-up with a deadline: the store rendezvous and the first collectives fail after `timeout_s`
    instead of waiting for ever (torch's own default is 10 minutes for nccl, 30 for gloo)."""
    import datetime
    import torch.distributed as dist
    kw = {"rank": rank, "world_size": world, "timeout": datetime.timedelta(seconds=timeout_s)}
    if backend == "nccl" and device is not None:
        kw["device_id"] = device
    dist.init_process_group(backend, **kw)
    return dist


class PathChooser:
    """Which path this rank's share takes while several frames of a sequence are in flight (include/rrt.h: rrt_path_chooser_*;
    the rule lives in csrc/rrt_path_chooser.cpp, the C++ headless driver uses the same object).  Host logic only.

        pc = PathChooser(frames_in_flight)
        per frame k = 1, 2, ...:   prm.path_policy = pc.policy(k)  ...  later, when known:  pc.report(k, sustained_ms)
    """

    def __init__(self, frames_in_flight, window_frames=0):
        import ctypes as C
        from . import _lib
        self._lib, self._C = _lib, C
        out = C.c_int(0)
        _lib.check(_lib.load().rrt_path_chooser_create(int(frames_in_flight), int(window_frames), C.byref(out)), "rrt_path_chooser_create")
        self.id = out.value

    def policy(self, frame):
        p = self._C.c_int(0)
        self._lib.check(self._lib.load().rrt_path_chooser_policy(self.id, int(frame), self._C.byref(p)), "rrt_path_chooser_policy")
        return p.value

    def report(self, frame, sustained_ms):
        self._lib.check(self._lib.load().rrt_path_chooser_report(self.id, int(frame), float(sustained_ms)), "rrt_path_chooser_report")

    def stats(self):
        st = self._lib.rrt_path_chooser_stats()
        self._lib.check(self._lib.load().rrt_path_chooser_get_stats(self.id, self._C.byref(st)), "rrt_path_chooser_get_stats")
        return {"incumbent": "single kernel" if st.incumbent == 1 else "automatic (three-pass for a small share)", "windows": st.windows,
                "trials": st.trials, "trials_aborted": st.trials_aborted, "switches": st.switches, "outliers": st.outliers,
                "frames_three_pass_auto": st.frames[0], "frames_single_kernel": st.frames[1]}

    def destroy(self):
        if self.id:
            self._lib.load().rrt_path_chooser_destroy(self.id)
            self.id = 0


class FrameSharder:
    """One rank's view of a sharded frame.

    render(buf, slot)      fills this rank's tile buffer (a (pad_rows*width*4,) uint8 tensor); `slot`
                           (0, or 0..n-1 in pipeline mode) tells the callback which of its per-slot
                           resources (e.g. the three-pass pool) belongs to this frame
    assemble(frame, buf, shard)   scatters one shard's buffer into the full frame (rank 0 only)
    assemble_all(frame, all_bufs, stride_bytes)   optional: all shards in one launch

    pipeline=True (world > 1) keeps two frames in flight, pipeline=n (an int >= 2) n of them.  Tile, gather
    and frame buffers exist once per slot; frame k is rendered, gathered (async_op: on the communicator's
    stream) and assembled on stream k mod n while the following frames are rendered on the other streams,
    so that (i) the transfer and the collective's latency sit under the next renders and (ii) the drain of
    one frame's kernels -- a rank's share is only a few rounds of wavefronts -- is filled by the next
    frames'.  step() then returns frame k-(n-1) (None on the first n-1 calls), drain() yields the frames
    still in flight in order and flush() the last of them; a returned tensor is valid on the caller's
    current stream until n steps after the one that rendered it.
    Ordering: an async collective starts after the work already queued on the stream it is issued from
    (the render that filled its input); work.wait() makes that stream wait for it, before the assemble
    that reads its output and before the slot's buffers are rendered into again, two steps later.
    """

    def __init__(self, width, height, tile_rows, rank, world, device, render, assemble, group=None,
                 assemble_all=None, pipeline=False, collective_at_world1=False, timing=False, shard_of_tile=None):
        import torch
        self.torch = torch
        # timing=True (GPU buffers only): device events around the three phases of every step, on the stream the
        # step runs on -- render | gather (queueing + waiting for the slowest rank + the transfer) | assemble --
        # so that a multi-GPU run can explain its own efficiency; read them with phase_times()
        self.timing = bool(timing) and torch.device(device).type == "cuda"
        self.events = []
        self.width, self.height, self.tile_rows = width, height, tile_rows
        self.rank, self.world, self.group = rank, world, group
        # shard_of_tile: an explicit tile -> rank assignment (rrt_tile_map: cost-weighted) instead of t mod world; it only
        # sizes the buffers here -- the callbacks render / assemble with the same map
        self.shard_of_tile = shard_of_tile
        self.pad_rows = max_shard_rows(height, tile_rows, world, shard_of_tile)
        self.n_bytes = self.pad_rows * width * 4
        self.render, self.assemble, self.assemble_all = render, assemble, assemble_all
        # collective_at_world1: run the gather path even on a one-rank group (self-checks of the RCCL calls
        # on a single GPU); normally a single rank assembles its own buffer directly.
        self.collective = world > 1 or bool(collective_at_world1)
        depth = 2 if pipeline is True else int(pipeline or 0)
        if depth == 1 or depth < 0:
            raise ValueError("pipeline: False, True (two frames in flight) or the number of frames in flight (>= 2)")
        self.pipeline = depth >= 2 and self.collective
        self.n_slots = depth if self.pipeline else 1
        self.k = 0
        self.pending = []                   # (work, staged, slot) of the frames in flight, oldest first
        on_gpu = torch.device(device).type == "cuda"
        self.streams = [torch.cuda.Stream(device) for _ in range(self.n_slots)] if (self.pipeline and on_gpu) else None
        self.locals = [torch.zeros(self.n_bytes, dtype=torch.uint8, device=device) for _ in range(self.n_slots)]
        self.local = self.locals[0]
        self.frames = [torch.zeros(height * width * 4, dtype=torch.uint8, device=device)
                       for _ in range(self.n_slots)] if rank == 0 else None
        self.frame = self.frames[0] if rank == 0 else None           # the most recently completed frame
        # one allocation for all shards, so that a single assemble launch can read them (assemble_all)
        self.gathered_alls = [torch.zeros(world * self.n_bytes, dtype=torch.uint8, device=device)
                              for _ in range(self.n_slots)] if (rank == 0 and self.collective) else None
        # rehearsal mode: a gloo group driving GPU buffers (several ranks sharing one card on a 1-GPU
        # box) stages the gather through host memory; the production backend is nccl (= RCCL).
        self.stage_cpu = False
        self.use_allgather = False
        if self.collective:
            import torch.distributed as dist
            self.stage_cpu = dist.get_backend(group) == "gloo" and on_gpu
            # Bring the communicator up with the collectives the step uses (untimed, once).  Which collective the
            # step will issue is decided COLLECTIVELY: every rank reports whether its probe worked and the ranks
            # all-reduce (MIN) that flag, so that no rank can end up issuing a different collective from its
            # peers (which would hang the job instead of failing it).
            probe = torch.zeros(256, dtype=torch.uint8, device="cpu" if self.stage_cpu else device)

            def agree(ok):
                flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=probe.device)
                dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
                return bool(flag.item())

            # An explicit tile -> rank map must be THE SAME map on every rank: each rank derives it on its own (rrt_probe_tile_costs
            # runs in fast arithmetic on that rank's GPU, rrt_tile_map_balance on its host), and a map that differs in one tile
            # loses or duplicates that tile in the gathered frame without any error.  All ranks compare a digest.
            if shard_of_tile is not None and world > 1:
                import hashlib
                import numpy as np
                digest = hashlib.sha256(np.asarray(shard_of_tile, dtype=np.int64).tobytes()).digest()[:8]
                mine = torch.tensor(list(digest), dtype=torch.int32, device=probe.device)
                lo, hi = mine.clone(), mine.clone()
                dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=group)
                dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=group)
                if not bool(torch.equal(lo, hi)):
                    raise RuntimeError(f"FrameSharder: rank {rank}'s tile -> rank map differs from another rank's "
                                       "(every rank must be given the same shard_of_tile)")

            def try_collective(allgather, asyn):
                try:
                    if allgather:
                        wk = dist.all_gather_into_tensor(
                            torch.zeros(world * 256, dtype=torch.uint8, device=probe.device), probe, group=group,
                            async_op=asyn)
                    else:
                        wk = dist.gather(probe, [torch.zeros_like(probe) for _ in range(world)] if rank == 0 else None,
                                         dst=0, group=group, async_op=asyn)
                    if asyn:
                        wk.wait()
                        if on_gpu:
                            torch.cuda.synchronize()
                    return True
                except (RuntimeError, NotImplementedError):
                    return False

            # `gather` is what the path needs (only rank 0 assembles); a backend build without it falls back to
            # an all-gather of the same buffers -- on every rank or on none.
            if not agree(try_collective(False, False)):
                if not agree(try_collective(True, False)):
                    raise RuntimeError("FrameSharder: neither gather nor all_gather works on this process group")
                self.use_allgather = True
                if self.gathered_alls is None:
                    self.gathered_alls = [torch.zeros(world * self.n_bytes, dtype=torch.uint8, device=device)
                                          for _ in range(self.n_slots)]
            if self.pipeline and not agree(try_collective(self.use_allgather, True)):
                # no async collectives on this backend: one frame at a time, on every rank
                self.pipeline = False
                self.streams = None
                self.n_slots = 1
        self.gathered_all = self.gathered_alls[0] if self.gathered_alls is not None else None

    def _on(self, slot):
        """Context: the slot's stream, ordered after what the caller has queued on the current stream."""
        if self.streams is None:
            import contextlib
            return contextlib.nullcontext()
        st = self.streams[slot]
        st.wait_stream(self.torch.cuda.current_stream())
        return self.torch.cuda.stream(st)

    def _start_gather(self, slot):
        """Queue the collective for `slot`; returns (work or None, staged host tensors or None)."""
        import torch.distributed as dist
        local = self.locals[slot]
        asyn = self.pipeline
        if self.use_allgather:
            return dist.all_gather_into_tensor(self.gathered_alls[slot], local, group=self.group, async_op=asyn), None
        if self.stage_cpu:
            loc = local.cpu()
            got = [self.torch.empty_like(loc) for _ in range(self.world)] if self.rank == 0 else None
            return dist.gather(loc, got, dst=0, group=self.group, async_op=asyn), got
        outs = list(self.gathered_alls[slot].split(self.n_bytes)) if self.rank == 0 else None
        return dist.gather(local, outs, dst=0, group=self.group, async_op=asyn), None

    def _mark(self):
        e = self.torch.cuda.Event(enable_timing=True)
        e.record()                      # on the current stream (the slot's stream inside _on())
        return e

    def phase_times(self, skip=0):
        """Mean milliseconds per step of the recorded phases (after a device synchronise), ignoring the first `skip`
        steps: {"render", "gather", "assemble", "steps"}.  With several frames in flight the phases of consecutive
        frames overlap on the device, so these are each frame's own latencies, not shares of the step time."""
        self.torch.cuda.synchronize()
        ev = [e for e in self.events[skip:] if len(e) == 4]
        if not ev:
            return None
        mean = lambda i: sum(e[i].elapsed_time(e[i + 1]) for e in ev) / len(ev)
        return {"render": mean(0), "gather": mean(1), "assemble": mean(2), "steps": len(ev)}

    def _finish(self, work, staged, slot, marks=None):
        """Wait for the slot's collective and assemble its frame (on the slot's stream)."""
        with self._on(slot):
            if work is not None:
                work.wait()
            if marks is not None:
                marks.append(self._mark())          # the gathered shards are here
            if self.rank == 0:
                ga, frame = self.gathered_alls[slot], self.frames[slot]
                if staged is not None:
                    for s in range(self.world):
                        ga[s * self.n_bytes:(s + 1) * self.n_bytes].copy_(staged[s])
                if self.assemble_all is not None:
                    self.assemble_all(frame, ga, self.n_bytes)
                else:
                    for s in range(self.world):
                        self.assemble(frame, ga[s * self.n_bytes:(s + 1) * self.n_bytes], s)
            if marks is not None:
                marks.append(self._mark())          # the frame is assembled
        if self.streams is not None:        # the caller reads the frame on its own (current) stream
            self.torch.cuda.current_stream().wait_stream(self.streams[slot])
        if self.rank == 0:
            self.frame = self.frames[slot]
        return self.frame

    def step(self):
        """Render this rank's tiles, gather to rank 0, assemble there.  Returns the frame on rank 0
        (pipeline mode with n frames in flight: the frame of n-1 steps ago, None on the first n-1 calls;
        see drain() / flush())."""
        slot = self.k % self.n_slots
        self.k += 1
        if not self.collective:
            self.render(self.locals[slot], slot)
            self.assemble(self.frame, self.locals[slot], 0)
            return self.frame
        marks = None
        with self._on(slot):
            if self.timing:
                marks = [self._mark()]
                self.events.append(marks)
            self.render(self.locals[slot], slot)
            if marks is not None:
                marks.append(self._mark())          # this rank's shard is rendered
            work, staged = self._start_gather(slot)
        if not self.pipeline:
            return self._finish(work, staged, slot, marks)
        self.pending.append((work, staged, slot, marks))
        if len(self.pending) < self.n_slots:
            return None
        return self._finish(*self.pending.pop(0))

    def drain(self):
        """Complete the frames still in flight (pipeline mode), oldest first; yields each on rank 0 (None elsewhere)."""
        while self.pending:
            yield self._finish(*self.pending.pop(0))

    def flush(self):
        """Complete every frame still in flight; returns the last frame on rank 0."""
        for _ in self.drain():
            pass
        return self.frame
