"""Tile plan / assemble logic and the world_size-2 gather path (gloo, CPU).

The renderer in these tests is the CPU oracle: they check the host-side sharding
code, not the HIP kernels (those are covered by tests/test_gpu_frames.py).
"""
import os
import socket
import sys

import numpy as np
import pytest

from relativisticraytracer_amd import sharding as sh

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


@pytest.mark.parametrize("height,tile_rows,n", [(2160, 16, 8), (90, 8, 3), (90, 7, 2), (5, 16, 4), (64, 64, 1)])
def test_tile_plan_covers_every_row_once(height, tile_rows, n):
    seen = np.zeros(height, int)
    for s in range(n):
        k_prev = -1
        for t, y0, rows in sh.tile_plan(height, tile_rows, s, n):
            assert t % n == s and t > k_prev and rows >= 1 and y0 == t * tile_rows
            seen[y0:y0 + rows] += 1
            k_prev = t
    assert np.all(seen == 1)
    assert sum(sh.shard_rows(height, tile_rows, s, n) for s in range(n)) == height


def test_tile_plan_matches_c_abi():
    """The C ABI's row count (rrt_tile_shard_rows) and the Python plan agree (no GPU needed)."""
    import relativisticraytracer_amd as rrt
    for height, R, n in ((2160, 16, 8), (90, 8, 3), (1, 1, 1), (33, 5, 7)):
        for s in range(n):
            assert rrt.tile_shard_rows(height, R, s, n) == sh.shard_rows(height, R, s, n)
    with pytest.raises(rrt.RRTError):
        rrt.tile_shard_rows(10, 0, 0, 1)
    with pytest.raises(ValueError):
        sh.tile_plan(10, 4, 3, 3)


def test_extract_assemble_roundtrip():
    rng = np.random.default_rng(0)
    h, w = 45, 13
    frame = rng.integers(0, 255, (h, w, 4), dtype=np.uint8)
    for n, R in ((3, 8), (8, 16), (2, 7), (1, 45)):
        out = np.zeros_like(frame)
        pad = sh.max_shard_rows(h, R, n)
        for s in range(n):
            sh.assemble_numpy(out, sh.extract_numpy(frame, w, h, R, s, n, pad_rows=pad), w, h, R, s, n)
        assert np.array_equal(out, frame)


def test_explicit_tile_maps_cover_every_row_once_and_round_trip():
    """rrt_tile_map on the host side: any assignment of tiles to shards -- uneven tile counts, an empty shard, a ragged
    last tile -- covers every row once, and extract / assemble with the map are inverses."""
    rng = np.random.default_rng(4)
    h, w, R, n = 45, 13, 7, 4
    n_tiles = (h + R - 1) // R
    frame = rng.integers(0, 255, (h, w, 4), dtype=np.uint8)
    for m in (rng.integers(0, n, n_tiles), np.array([0, 0, 0, 3, 3, 1, 0]), np.arange(n_tiles) % n):
        seen = np.zeros(h, int)
        for s in range(n):
            for t, y0, rows in sh.tile_plan(h, R, s, n, m):
                assert m[t] == s
                seen[y0:y0 + rows] += 1
        assert np.all(seen == 1) and sum(sh.shard_rows(h, R, s, n, m) for s in range(n)) == h
        out = np.zeros_like(frame)
        pad = sh.max_shard_rows(h, R, n, m)
        for s in range(n):
            sh.assemble_numpy(out, sh.extract_numpy(frame, w, h, R, s, n, pad_rows=pad, shard_of_tile=m), w, h, R, s, n, m)
        assert np.array_equal(out, frame)
    with pytest.raises(ValueError):
        sh.tile_plan(h, R, 0, n, [0] * (n_tiles - 1))
    with pytest.raises(ValueError):
        sh.tile_plan(h, R, 0, n, [n] * n_tiles)


def test_watchdog_ends_a_stuck_run_with_tracebacks():
    """sharding.Watchdog: a run that stops making progress exits non-zero and says where every thread sat."""
    import subprocess
    code = ("import time, sys; sys.path.insert(0, %r)\n"
            "from relativisticraytracer_amd.sharding import Watchdog\n"
            "d = Watchdog('t'); d.arm(30, 'a'); d.disarm(); d.arm(1.0, 'stuck phase')\n"
            "def stuck():\n    time.sleep(60)\n"
            "stuck()\n" % ROOT)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=30)
    assert r.returncode != 0 and "Timeout (0:00:01)" in r.stderr and "stuck" in r.stderr


def test_watchdog_arm_zero_means_no_limit():
    """arm(0) used to do nothing AND leave the previous countdown running -- headless.py re-arms every frame, so with
    `--frame-timeout 0` the bring-up allowance armed before frame 1 killed a healthy run.  arm() now always cancels first."""
    import subprocess
    code = ("import time, sys; sys.path.insert(0, %r)\n"
            "from relativisticraytracer_amd.sharding import Watchdog\n"
            "d = Watchdog('t'); d.arm(0.3, 'bring-up + frame 1')\n"
            "for k in range(4):\n    d.arm(0, 'frame, no limit'); time.sleep(0.2)\n"
            "d.arm(None, 'no limit either'); time.sleep(0.2)\n"
            "print('alive')\n" % ROOT)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=30)
    assert r.returncode == 0 and "alive" in r.stdout and "Timeout" not in r.stderr, (r.returncode, r.stderr[-300:])


def test_single_node_environment_only_fills_what_is_unset():
    env = {"MASTER_ADDR": "127.0.0.1"}
    sh.single_node_environment(env)
    assert env["NCCL_DEBUG"] == "WARN" and env["NCCL_SOCKET_IFNAME"] == "lo" and env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    env = {"MASTER_ADDR": "10.0.0.7", "NCCL_DEBUG": "INFO"}
    sh.single_node_environment(env)
    assert env["NCCL_DEBUG"] == "INFO" and "NCCL_SOCKET_IFNAME" not in env
    th = sh.warm_library_pages(os.path.join(ROOT, "include", "rrt.h")); th.join(10); assert not th.is_alive()
    sh.warm_library_pages("/nonexistent/file").join(10)


def _worker(rank, world, port, q, pipeline, tile_map=None):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist
    from oracle import pyoracle as po
    from relativisticraytracer_amd.sky import synthetic_sky
    dist.init_process_group("gloo", rank=rank, world_size=world)
    w, h, R = 48, 27, 4
    sky = synthetic_sky(256, 128)
    cam = po.camera((0, 10, -60), (0, -0.17364804, 0.9848078), (1, 0, 0), (0, 0.9848078, 0.17364804))
    fx, prm = po.default_effects(), po.default_params(spin=0.9)
    times = [1.0, 7.5, 12.5]
    calls = {"n": 0}

    def render(buf, slot):      # oracle stands in for rrt_launch_raymarch_tiles; every frame has its own time
        t_sim = times[calls["n"]]; calls["n"] += 1
        full = np.zeros((h, w, 4), np.uint8)
        for t, y0, rows in sh.tile_plan(h, R, rank, world, tile_map):
            full = po.render(cam, fx, prm, t_sim, w, h, sky, rect=(0, y0, w, y0 + rows), n_threads=1)["rgba8"] | full
        tiles = sh.extract_numpy(full, w, h, R, rank, world, pad_rows=sh.max_shard_rows(h, R, world, tile_map), shard_of_tile=tile_map)
        buf.copy_(torch.from_numpy(tiles.reshape(-1)))

    def assemble(frame, buf, shard):
        f = frame.numpy().reshape(h, w, 4)
        sh.assemble_numpy(f, buf.numpy().reshape(-1, w, 4), w, h, R, shard, world, tile_map)

    fs = sh.FrameSharder(w, h, R, rank, world, "cpu", render, assemble, pipeline=pipeline, shard_of_tile=tile_map)
    depth = 2 if pipeline is True else int(pipeline or 1)
    got = []
    for k in range(len(times)):
        f = fs.step()
        if k < depth - 1:
            assert f is None                     # nothing assembled yet: the first frames are still in flight
        elif rank == 0:
            got.append(f.numpy().copy())
    for f in fs.drain():                         # the frames still in flight, oldest first
        if rank == 0:
            got.append(f.numpy().copy())
    assert fs.flush() is fs.frame
    if rank == 0:
        ok = len(got) == len(times)
        for k, t_sim in enumerate(times):
            ref = po.render(cam, fx, prm, t_sim, w, h, sky, n_threads=2)["rgba8"]
            ok = ok and bool(np.array_equal(got[k].reshape(h, w, 4), ref))
        q.put(ok)
    dist.barrier()
    dist.destroy_process_group()


# the last case: a NON-UNIFORM explicit map (rank 0 gets 5 of the 7 tiles, the ragged last one included): rrt_tile_map's host side
@pytest.mark.parametrize("pipeline,tile_map", [(False, None), (True, None), (3, None), (3, [0, 1, 0, 0, 1, 0, 0])])
def test_gather_world2_gloo_matches_single_render(pipeline, tile_map):
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, pipeline, tile_map)) for r in range(2)]
    for p in procs:
        p.start()
    ok = q.get(timeout=240)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert ok


def _worker_maps_differ(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    tile_map = [0, 1, 0, 0, 1, 0, 0] if rank == 0 else [0, 1, 0, 1, 1, 0, 0]        # tile 3: both ranks think the other / they own it
    try:
        sh.FrameSharder(48, 27, 4, rank, world, "cpu", lambda buf, slot: None, lambda frame, buf, shard: None, shard_of_tile=tile_map)
        q.put((rank, "accepted"))
    except RuntimeError as e:
        q.put((rank, "refused" if "differs from another rank" in str(e) else f"other: {e}"))
    dist.barrier()
    dist.destroy_process_group()


def test_ranks_with_different_tile_maps_are_refused_world2_gloo():
    """every rank derives the explicit tile -> rank map by itself (probe + balance); maps that differ in one tile
    would lose or duplicate that tile in the gathered frame silently.  The sharder compares a digest over all ranks."""
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_maps_differ, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert got == {0: "refused", 1: "refused"}, got


def test_pipelined_step_single_process_gloo():
    """The collective path on a one-rank gloo group (no subprocesses): frame order and flush()."""
    import torch
    import torch.distributed as dist
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1)
    try:
        w, h, R = 8, 21, 4
        n = {"i": 0}

        def render(buf, slot):
            n["i"] += 1
            buf.fill_(n["i"])

        def assemble(frame, buf, shard):
            sh.assemble_numpy(frame.numpy().reshape(h, w, 4), buf.numpy().reshape(-1, w, 4), w, h, R, shard, 1)

        for pipeline, want in ((False, [1, 2, 3, 4, 5]), (True, [None, 1, 2, 3, 4]), (3, [None, None, 1, 2, 3]),
                               (4, [None, None, None, 1, 2])):
            n["i"] = 0
            fs = sh.FrameSharder(w, h, R, 0, 1, "cpu", render, assemble, pipeline=pipeline, collective_at_world1=True)
            seen = []
            for _ in range(5):
                f = fs.step()
                seen.append(None if f is None else int(f[0]))
            assert seen == want
            if pipeline == 3:                   # drain() hands out every frame still in flight, in order
                assert [int(f[0]) for f in fs.drain()] == [4, 5]
            last = int(fs.flush()[0])
            assert last == 5
            assert fs.flush() is fs.frame and bool((fs.frame == 5).all())       # idempotent
        with pytest.raises(ValueError):
            sh.FrameSharder(w, h, R, 0, 1, "cpu", render, assemble, pipeline=1, collective_at_world1=True)
    finally:
        dist.destroy_process_group()
