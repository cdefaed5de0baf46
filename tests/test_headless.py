"""Headless driver + sinks (rows f3/f4 of SURVEY.md 8)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def test_sinks_roundtrip(tmp_path):
    from relativisticraytracer_amd import sinks
    rng = np.random.default_rng(0)
    frames = [rng.integers(0, 255, (5, 7, 4), dtype=np.uint8) for _ in range(3)]
    raw = sinks.open_sink(str(tmp_path / "f.rgba"), 7, 5)
    ppm = sinks.open_sink(str(tmp_path / "ppm") + "/", 7, 5)
    for f in frames:
        raw.write(f); ppm.write(f)
    raw.close(); ppm.close()
    data = np.fromfile(tmp_path / "f.rgba", np.uint8).reshape(3, 5, 7, 4)
    assert np.array_equal(data, np.stack(frames))                       # raw keeps the bottom-up bytes
    blob = open(tmp_path / "ppm" / "frame_00002.ppm", "rb").read()
    assert blob.startswith(b"P6\n7 5\n255\n")
    rgb = np.frombuffer(blob[len(b"P6\n7 5\n255\n"):], np.uint8).reshape(5, 7, 3)
    assert np.array_equal(rgb, frames[1][::-1, :, :3])                  # PPM is flipped to top-down
    with pytest.raises(ValueError):
        raw2 = sinks.RawSink(str(tmp_path / "g.rgba"), 7, 5); raw2.write(np.zeros((4, 7, 4), np.uint8))
    assert sinks.open_sink(None, 7, 5) is None


def test_load_sky_matches_pil(tmp_path):
    from PIL import Image
    from relativisticraytracer_amd.sky import load_sky, synthetic_sky
    s = synthetic_sky(64, 32)
    Image.fromarray(s[..., :3]).save(tmp_path / "s.png")
    got = load_sky(str(tmp_path / "s.png"))
    assert got.shape == (32, 64, 4) and np.array_equal(got[..., :3], s[..., :3]) and np.all(got[..., 3] == 255)


@pytest.mark.gpu
def test_headless_path_playback_equals_manual_frames(tmp_path):
    """3 frames of path 0: the driver's raw output == frames rendered by hand with the same clock/camera."""
    import torch
    import relativisticraytracer_amd as rrt
    from relativisticraytracer_amd import camera_paths as cp
    from relativisticraytracer_amd.sky import synthetic_sky
    out = tmp_path / "seq.rgba"
    r = subprocess.run([sys.executable, "-m", "relativisticraytracer_amd.headless", "--width", "96", "--height", "54",
                        "--frames", "3", "--path", "0", "--spin", "0.9", "--all-effects", "--out", str(out)],
                       cwd=ROOT, capture_output=True, text=True, check=True)
    meta = json.loads(r.stdout.strip().splitlines()[-1])
    assert meta["frames"] == 3 and meta["path"] == "Gargantua Fly-By"
    data = np.fromfile(out, np.uint8).reshape(3, 54, 96, 4)
    tex = rrt.SkyTexture(synthetic_sky())
    path = cp.CameraPath(0)
    fx = rrt.CameraEffects(useChromaticAberration=True)
    for k in (1, 2, 3):
        st, pt = cp.recording_clock(k)
        buf = torch.zeros(54 * 96 * 4, dtype=torch.uint8, device="cuda")
        rrt.launch_raymarch(buf, 96, 54, st, path.camera_at(pt), tex, fx, rrt.RenderParams(spin=0.9))
        torch.cuda.synchronize()
        assert np.array_equal(buf.cpu().numpy().reshape(54, 96, 4), data[k - 1]), k


def test_cpp_headless_driver_builds():
    from relativisticraytracer_amd import build
    exe = build.build_headless()
    assert os.path.exists(exe)
    r = subprocess.run([exe, "--bogus"], capture_output=True, text=True)
    assert r.returncode == 2


@pytest.mark.gpu
def test_cpp_and_python_drivers_write_the_same_frames(tmp_path):
    """rrt_headless (C++ over the C ABI) and headless.py render the same 3 frames of path 2."""
    from relativisticraytracer_amd import build
    exe = build.build_headless()
    a, b = tmp_path / "cpp.rgba", tmp_path / "py.rgba"
    args = ["--width", "128", "--height", "72", "--frames", "3", "--path", "2", "--spin", "0.9", "--all-effects"]
    subprocess.run([exe] + args + ["--out", str(a)], check=True, capture_output=True)
    subprocess.run([sys.executable, "-m", "relativisticraytracer_amd.headless"] + args + ["--out", str(b)],
                   cwd=ROOT, check=True, capture_output=True)
    assert open(a, "rb").read() == open(b, "rb").read()


@pytest.mark.gpu
def test_cpp_driver_rccl_gather_path_writes_the_same_frames(tmp_path):
    """rrt_headless is ONE process for N GPUs (ncclCommInitAll, grouped ncclSend/ncclRecv gather into device 0,
    rrt_assemble_all_tiles, three frames in flight by default).  On the one-GPU box the same code runs over a one-rank
    communicator (--force-collective): tiles -> RCCL exchange -> assemble must give the frames of the plain
    single-GPU path, with and without the noise tables and the three-pass pool."""
    from relativisticraytracer_amd import build
    exe = build.build_headless()
    base = ["--width", "160", "--height", "90", "--frames", "5", "--path", "0", "--spin", "0.9", "--all-effects"]
    ref = tmp_path / "plain.rgba"
    subprocess.run([exe] + base + ["--no-noise-table", "--workspace-gib", "0", "--out", str(ref)], check=True, capture_output=True)
    want = open(ref, "rb").read()
    assert len(want) == 5 * 160 * 90 * 4
    for extra in (["--force-collective"], ["--force-collective", "--tile-rows", "7", "--workspace-gib", "1"],
                  ["--force-collective", "--no-noise-table", "--workspace-gib", "0"], [],
                  ["--force-collective", "--frames-in-flight", "1"], ["--force-collective", "--frames-in-flight", "2"],
                  ["--force-collective", "--frames-in-flight", "4"], ["--frames-in-flight", "4"],
                  ["--tile-order", "--workspace-gib", "0"], ["--frames-in-flight", "1", "--workspace-gib", "0"],
                  ["--frames-in-flight", "1", "--no-tile-order"]):
        out = tmp_path / "v.rgba"
        try:
            r = subprocess.run([exe] + base + extra + ["--out", str(out)], capture_output=True, text=True, timeout=240,
                               env=dict(os.environ, RRT_HEADLESS_TRACE="1"))
        except subprocess.TimeoutExpired as e:       # say where the driver sat (its timestamped trace), then fail
            err = e.stderr.decode() if isinstance(e.stderr, bytes) else (e.stderr or "")
            pytest.fail(f"rrt_headless {extra} hung; its trace:\n{err[-3000:]}")
        assert r.returncode == 0, r.stderr[-2000:]
        meta = json.loads(r.stdout.strip().splitlines()[-1])
        assert meta["n_gpus"] == 1 and meta["frames"] == 5
        assert ("rccl" in meta["collective"]) == ("--force-collective" in extra)
        # cost-ordered dispatch: on request, and by itself when frames are rendered one at a time through the SINGLE kernel
        # (round 4: not on launches that take the three-pass path -- small launches with a pool --, where it was measured slower)
        one_at_a_time = "--frames-in-flight" in extra and extra[extra.index("--frames-in-flight") + 1] == "1"
        no_pool = "--workspace-gib" in extra and extra[extra.index("--workspace-gib") + 1] == "0"
        assert meta["tile_order"] == ("--tile-order" in extra or (one_at_a_time and no_pool and "--no-tile-order" not in extra))
        assert open(out, "rb").read() == want, extra
    r = subprocess.run([exe] + base + ["--gpus", "99"], capture_output=True, text=True)
    assert r.returncode == 2 and "device(s) visible" in r.stderr
    assert subprocess.run([exe] + base + ["--frames-in-flight", "5"], capture_output=True).returncode == 2


@pytest.mark.gpu
def test_cpp_driver_watchdog_ends_a_stuck_run_with_a_diagnosis():
    """a run that makes no progress must END -- status 3, where it sat, the last trace points, every
    communicator's asynchronous error -- instead of hanging until a harness kills it.  Provoked here by a bring-up limit
    shorter than any communicator bring-up (RCCL loads its code object for over a second even with a warm page cache)."""
    from relativisticraytracer_amd import build
    exe = build.build_headless()
    r = subprocess.run([exe, "--width", "64", "--height", "36", "--frames", "2", "--force-collective", "--init-timeout", "0.3"],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 3, (r.returncode, r.stderr[-1500:])
    assert "no progress for" in r.stderr and "Last trace points" in r.stderr and "start" in r.stderr
    assert "{" not in r.stdout                                   # no summary line: the run did not complete
    # the same run with the default limits completes
    r = subprocess.run([exe, "--width", "64", "--height", "36", "--frames", "2", "--force-collective"], capture_output=True, text=True, timeout=240)
    assert r.returncode == 0 and json.loads(r.stdout.strip().splitlines()[-1])["frames"] == 2


def _gpu_count():
    import torch
    return torch.cuda.device_count()


@pytest.mark.gpu
def test_two_gpus_over_rccl_equal_one_gpu(tmp_path):
    """Needs >= 2 GPUs (skipped on the one-GPU box): both hosts over real RCCL -- rrt_headless --gpus 2 (one process,
    ncclCommInitAll, grouped send/recv) and headless.py under torch.distributed.run (backend nccl, async gather) --
    must write the frames of the one-GPU run."""
    if _gpu_count() < 2:
        pytest.skip("needs two GPUs")
    import socket
    from relativisticraytracer_amd import build
    exe = build.build_headless()
    base = ["--width", "320", "--height", "180", "--frames", "6", "--path", "0", "--spin", "0.9", "--all-effects"]
    one, cpp2, py2 = tmp_path / "one.rgba", tmp_path / "cpp2.rgba", tmp_path / "py2.rgba"
    subprocess.run([exe] + base + ["--out", str(one)], check=True, capture_output=True)
    r = subprocess.run([exe] + base + ["--gpus", "2", "--out", str(cpp2)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert json.loads(r.stdout.strip().splitlines()[-1])["n_gpus"] == 2
    assert open(cpp2, "rb").read() == open(one, "rb").read()
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("RRT_DIST_BACKEND", None)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(port),
                        "-m", "relativisticraytracer_amd.headless"] + base + ["--workspace-gib", "2", "--out", str(py2)],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert open(py2, "rb").read() == open(one, "rb").read()
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--width", "640", "--height", "360", "--steps", "3",
                        "--warmup", "1", "--cpu-stride", "0"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["n_gpus"] == 2 and d["config"]["comm_ranks"] == 2 and "RCCL" in d["config"]["dist_backend"]


@pytest.mark.gpu
def test_bench_self_launches_its_ranks():
    """`python bench.py --gpus 2` as the driver invokes it (no launcher): bench.py starts the two ranks itself as
    child processes and relays rank 0's single JSON line.  On the one-GPU box the ranks share the card over gloo."""
    env = dict(os.environ, RRT_DIST_BACKEND="gloo")
    env.pop("WORLD_SIZE", None); env.pop("RANK", None)
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--width", "320", "--height", "180", "--steps", "2",
                        "--warmup", "1", "--cpu-stride", "4", "--workspace-gib", "2"], cwd=ROOT, env=env, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["config"]["comm_ranks"] == 2
    assert d["scaling"] == "strong" and d["value"] > 0
    # round 6: the N > 1 line carries the CPU leg too (rank 0 times it while the other rank waits at a barrier under the run
    # watchdog; no rank trips one: the run ended with status 0 above), on the threads it actually had
    cb = d["cpu_baseline"]
    assert cb is not None and cb["value"] > 0 and cb["cores"] >= 1 and cb["kind"] in ("port", "reference") and "barrier" in cb["note"]
    if os.path.exists(os.path.join(ROOT, "oracle", "_ref", "libref_frames.so")):
        assert cb["kind"] == "reference"
    # round 3: the N > 1 line explains its own efficiency -- every rank's phase latencies (render | gather | assemble)
    # and the same frames one at a time (latency-bound strong scaling) next to the pipelined value
    mg = d["multi_gpu"]
    ph, one = mg["phases"], mg["one_frame_at_a_time"]
    assert len(ph["per_rank_render_ms"]) == 2 and all(v > 0 for v in ph["per_rank_render_ms"])
    assert len(ph["per_rank_gather_ms"]) == 2 and ph["rank0_assemble_ms"] >= 0 and 0 < ph["render_balance_min_over_max"] <= 1
    assert one["ms_per_step"] > 0 and one["value"] > 0 and len(one["per_rank_render_ms"]) == 2
    assert mg["frames_in_flight"] == 3 and len(mg["path_per_launch"]["per_rank"]) == 2
    # round 5: the denominator of a scaling figure travels with the line -- rank 0 alone on the same frame, same run, best of the
    # static and the cost-ordered dispatch, and its bytes equal the gathered frame's
    ref = mg["single_gpu_reference"]
    assert mg["single_gpu_reference_ms"] == min(ref["static_order_ms"], ref["cost_ordered_ms"]) > 0
    assert ref["same_bytes_as_the_gathered_frame"] is True and mg["speedup_vs_single_gpu_reference"] > 0


@pytest.mark.gpu
def test_two_ranks_pipelined_equal_one_rank(tmp_path):
    """Two ranks sharing the card (gloo rehearsal of the N > 1 path: interleaved tiles, three -- the default -- and
    two frames in flight, gather, assemble) write the same 5 frames as one rank."""
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    a, b = tmp_path / "one.rgba", tmp_path / "two.rgba"
    args = ["--width", "160", "--height", "90", "--frames", "5", "--path", "0", "--spin", "0.9", "--workspace-gib", "1"]
    subprocess.run([sys.executable, "-m", "relativisticraytracer_amd.headless"] + args + ["--out", str(a)],
                   cwd=ROOT, check=True, capture_output=True)
    env = dict(os.environ, RRT_DIST_BACKEND="gloo")
    for extra in ([], ["--frames-in-flight", "2"]):
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
        r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                            "--master-addr", "127.0.0.1", "--master-port", str(port),
                            "-m", "relativisticraytracer_amd.headless"] + args + extra + ["--out", str(b)],
                           cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        meta = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
        assert meta["n_gpus"] == 2 and meta["frames"] == 5
        assert open(a, "rb").read() == open(b, "rb").read(), extra


@pytest.mark.gpu
def test_both_drivers_render_the_same_fmad_frames(tmp_path):
    """--arith fmad (round 6) through the C++ and the Python driver: the same bytes (FMAD is one function on every path), within a
    handful of bytes of the strict frames, and the summary line names the mode."""
    from relativisticraytracer_amd import build
    exe = build.build_headless()
    base = ["--width", "160", "--height", "90", "--frames", "3", "--path", "0", "--spin", "0.9"]
    outs = {}
    for tag, cmd in (("cpp_fmad", [exe] + base + ["--arith", "fmad"]), ("cpp_strict", [exe] + base),
                     ("py_fmad", [sys.executable, "-m", "relativisticraytracer_amd.headless"] + base + ["--arith", "fmad"])):
        out = tmp_path / (tag + ".rgba")
        r = subprocess.run(cmd + ["--out", str(out)], cwd=ROOT, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        meta = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
        assert meta["arith_mode"] == ("fmad" if "fmad" in tag else "strict")
        outs[tag] = open(out, "rb").read()
    assert outs["cpp_fmad"] == outs["py_fmad"]
    diff = sum(a != b for a, b in zip(outs["cpp_fmad"], outs["cpp_strict"]))
    assert diff <= 1e-3 * len(outs["cpp_strict"])         # (frames this small may not differ from the strict ones at all: ~1e-4 of the pixels do)
    assert subprocess.run([exe] + base + ["--arith", "nvcc"], capture_output=True).returncode == 2


@pytest.mark.gpu
def test_both_drivers_render_a_raw_sky(tmp_path):
    """SURVEY row f1 on the C++ side (round 6): `rrt_headless --sky file` reads the raw sky format (sky.save_sky_raw: the texels the
    reference's own decoder returned, shipped as they are) -- the same frames as the Python driver with the same file, other than
    with the synthetic sky; a file that is not a raw sky is refused (status 2), not guessed at."""
    from relativisticraytracer_amd import build
    from relativisticraytracer_amd.sky import save_sky_raw
    exe = build.build_headless()
    rng = np.random.default_rng(7)
    sky = rng.integers(0, 256, (96, 200, 4), dtype=np.uint8); sky[..., 3] = 255          # an odd-sized sky
    raw = tmp_path / "sky.rrtsky"
    save_sky_raw(str(raw), sky)
    base = ["--width", "160", "--height", "90", "--frames", "2", "--path", "1", "--spin", "0.9", "--all-effects"]
    outs = {}
    for tag, cmd in (("cpp", [exe] + base + ["--sky", str(raw)]), ("cpp_synthetic", [exe] + base),
                     ("py", [sys.executable, "-m", "relativisticraytracer_amd.headless"] + base + ["--sky", str(raw)])):
        out = tmp_path / (tag + ".rgba")
        r = subprocess.run(cmd + ["--out", str(out)], cwd=ROOT, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        outs[tag] = open(out, "rb").read()
    assert outs["cpp"] == outs["py"] and outs["cpp"] != outs["cpp_synthetic"]
    bad = tmp_path / "bad.rrtsky"
    bad.write_bytes(b"RRTSKY1\n200 96\n" + bytes(100))
    r = subprocess.run([exe] + base + ["--sky", str(bad)], capture_output=True, text=True)
    assert r.returncode == 2 and "not a raw sky" in r.stderr
    r = subprocess.run([exe] + base + ["--sky", str(tmp_path / "missing")], capture_output=True, text=True)
    assert r.returncode == 2 and "cannot open" in r.stderr


@pytest.mark.gpu
def test_path_choice_changes_no_byte(tmp_path):
    """Round 6: under frames in flight a small share's path is chosen per window by measurement (rrt_path_chooser).  40 frames
    of path 0 through the C++ driver -- a single device with the exchange, and without -- and through two ranks of the Python
    driver sharing the card: with trials of the single kernel (windows of 18 frames, so that three trials fit) the bytes are
    those of the three-pass-only run (--path-window -1), and the summary line says which path rendered how many frames."""
    # (a window holds at least two trials' worth: 2 * 3 * frames in flight + 2 * frames in flight = 24 frames; 18 is widened to that)
    import socket
    from relativisticraytracer_amd import build
    exe = build.build_headless()
    base = ["--width", "160", "--height", "90", "--frames", "48", "--path", "0", "--spin", "0.9", "--workspace-gib", "1"]
    want = tmp_path / "want.rgba"
    r = subprocess.run([exe] + base + ["--path-window", "-1", "--out", str(want)], capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stderr[-2000:]
    assert json.loads(r.stdout.strip().splitlines()[-1])["path_choice"] is None
    for extra in (["--path-window", "18"], ["--path-window", "18", "--force-collective"], ["--path-window", "18", "--frames-in-flight", "2"]):
        out = tmp_path / "got.rgba"
        r = subprocess.run([exe] + base + extra + ["--out", str(out)], capture_output=True, text=True, timeout=240)
        assert r.returncode == 0, r.stderr[-2000:]
        pc = json.loads(r.stdout.strip().splitlines()[-1])["path_choice"]
        assert len(pc) == 1 and pc[0]["frames_three_pass"] + pc[0]["frames_single_kernel"] == 48
        assert pc[0]["trials"] >= 1 and pc[0]["frames_single_kernel"] >= 1, pc      # (frames this small are noisy: a trial may end on an outlier)
        assert open(out, "rb").read() == open(want, "rb").read(), extra
    # one frame at a time: nothing to choose (the three-pass path's two chains are the answer there)
    r = subprocess.run([exe] + base + ["--frames-in-flight", "1"], capture_output=True, text=True, timeout=240)
    assert r.returncode == 0 and json.loads(r.stdout.strip().splitlines()[-1])["path_choice"] is None
    env = dict(os.environ, RRT_DIST_BACKEND="gloo")
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    out = tmp_path / "py2.rgba"
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(port),
                        "-m", "relativisticraytracer_amd.headless"] + base + ["--path-window", "18", "--out", str(out)],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    meta = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert meta["n_gpus"] == 2 and meta["path_choice"]["trials"] >= 1 and meta["path_choice"]["frames_single_kernel"] >= 1
    assert open(out, "rb").read() == open(want, "rb").read()


@pytest.mark.gpu
def test_bench_line_contract():
    """bench.py at a reduced frame size: one JSON line with the driver's keys, the roofline and the CPU baseline."""
    r = subprocess.run([sys.executable, "bench.py", "--width", "320", "--height", "180", "--steps", "2", "--warmup", "1",
                        "--cpu-stride", "4"], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["metric"] == "Mrays/s" and d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1
    assert d["higher_is_better"] is True and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - 320 * 180 / d["ms_per_step"] / 1e3) < 0.01 * d["value"]
    rf = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in rf, k
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3 and 0 < rf["frac"] < 1
    # round 6: what the K timed frames did, not only their mean
    assert len(rf["kernel_ms_per_frame"]) == 2 and rf["kernel_ms_min"] <= rf["kernel_ms_median"] <= rf["kernel_ms_max"]
    assert rf["kernel_ms_min"] <= rf["kernel_ms"] <= rf["kernel_ms_max"] and rf["kernel_ms_slowest_frame"] in (0, 1)
    assert rf["kernel_ms_per_frame"][rf["kernel_ms_slowest_frame"]] == rf["kernel_ms_max"]
    # ... and the 30 fps statement may only rest on a mode whose arithmetic is not narrower than the reference's build
    wt = d["within_tolerance_mode"]
    assert wt["mode"] in ("fmad", None) and wt["modes"]["fast"]["credited"] is False and wt["modes"]["fmad"]["credited"] is True
    assert wt["conditioning"]["uncovered_after_fixed_set"]["frames"] == 20 and 0 <= wt["conditioning"]["ill_fraction"] < 0.25
    # round 4: the clock the chip held over the timed frames, and the fraction priced at it
    assert 1.0 < rf["clock_ghz"] < 2.6 and "beside the timed frames" in rf["clock_note"]
    assert abs(rf["frac_at_held_clock"] - rf["achieved"] / (256 * 4 * 32 * rf["clock_ghz"] / 1e3)) < 2e-3
    hv = d["heavy_view"]
    assert hv["noise_table_first_frame_probe_ordered"]["ms_per_step"] > 0 and hv["noise_table_cost_ordered"]["ms_per_step"] > 0
    # round 5: the 8-rank projection of this very frame travels with the single-GPU line (every share alone on this GPU, sustained)
    pj = d["projection_8_ranks"]
    assert pj["ranks"] == 8 and len(pj["three_pass_one_chain"]["per_shard_ms"]) == 8 and len(pj["single_kernel"]["per_shard_ms"]) == 8
    assert pj["faster_path_per_rank"]["max_ms"] <= min(pj["three_pass_one_chain"]["max_ms"], pj["single_kernel"]["max_ms"]) and "PROJECTION" in pj["note"]
    cb = d["cpu_baseline"]
    assert cb["kind"] in ("port", "reference") and cb["cores"] >= 1 and cb["value"] > 0 and cb["unit"] == "Mrays/s" and cb["sample"]
    if cb["kind"] == "reference":       # oracle/_ref travelled with the tree: the reference's own kernel body was timed
        assert cb["step_counts_equal_port"] is True and cb["port"]["kind"] == "port" and cb["port"]["value"] > 0
