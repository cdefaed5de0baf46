"""Host-side camera path code against the reference's own src/camera_paths.cpp
(tests/golden/camera_ref.npz, produced from oracle/_ref/libref_camera.so)."""
import numpy as np

from conftest import same_bits


def test_catmull_rom_and_lerp_angle_bit_exact(camera_ref):
    from relativisticraytracer_amd import camera_paths as cp
    pts, ts = camera_ref["cr_pts"], camera_ref["cr_t"]
    got = np.stack([cp.catmull_rom(pts[i, 0], pts[i, 1], pts[i, 2], pts[i, 3], ts[i]) for i in range(len(ts))])
    assert same_bits(got, camera_ref["cr_out"])
    ab, tt = camera_ref["la_ab"], camera_ref["la_t"]
    got = np.float32([cp.lerp_angle(ab[i, 0], ab[i, 1], tt[i]) for i in range(len(tt))])
    assert same_bits(got, camera_ref["la_out"])


def test_builtin_paths_match_reference_tables(camera_ref):
    from relativisticraytracer_amd import camera_paths as cp
    ps = cp.paths()
    assert len(ps) == 3
    for i, p in enumerate(ps):
        assert np.array_equal(p.keyframes, camera_ref[f"path{i}_keys"])
        assert p.name == bytes(camera_ref[f"path{i}_name"]).decode()
        assert p.t_end == p.keyframes[-1, 0]


def test_path_interpolation_semantics():
    """getInterpolatedState (src/main.cpp:176-203): clamped ends, Catmull-Rom on pos with clamped
    neighbour indices, lerp_angle on yaw/pitch between the two bracketing keys."""
    import relativisticraytracer_amd as rrt
    from relativisticraytracer_amd import camera_paths as cp
    p = cp.paths()[0]
    k = p.keyframes
    first = rrt.CameraState.from_angles(k[0, 1:4], k[0, 4], k[0, 5]).as_array()
    assert np.array_equal(p.camera_at(-1.0).as_array(), first)
    assert np.array_equal(p.camera_at(0.0).as_array(), first)
    last = rrt.CameraState.from_angles(k[-1, 1:4], k[-1, 4], k[-1, 5]).as_array()
    assert np.array_equal(p.camera_at(1e9).as_array(), last)
    # a mid-segment time, recomputed from the pinned primitives
    t = np.float32(7.25)
    factor = np.float32((t - k[1, 0]) / (k[2, 0] - k[1, 0]))
    pos = cp.catmull_rom(k[0, 1:4], k[1, 1:4], k[2, 1:4], k[3, 1:4], factor)
    want = rrt.CameraState.from_angles(pos, cp.lerp_angle(k[1, 4], k[2, 4], factor),
                                       cp.lerp_angle(k[1, 5], k[2, 5], factor)).as_array()
    assert np.array_equal(p.camera_at(float(t)).as_array(), want)
    # last segment uses the clamped i+2 index
    t = np.float32(20.0)
    factor = np.float32((t - k[3, 0]) / (k[4, 0] - k[3, 0]))
    pos = cp.catmull_rom(k[2, 1:4], k[3, 1:4], k[4, 1:4], k[4, 1:4], factor)
    assert np.array_equal(p.camera_at(float(t)).as_array()[0], pos)
    # exactly on a key
    on = rrt.CameraState.from_angles(k[2, 1:4], k[2, 4], k[2, 5]).as_array()
    assert np.allclose(p.camera_at(float(k[2, 0])).as_array(), on, atol=1e-5)


def test_recording_clock_is_a_float_accumulator():
    from relativisticraytracer_amd import camera_paths as cp
    dt = np.float32(1.0) / np.float32(24)
    acc = np.float32(0)
    for k in range(1, 301):
        acc = np.float32(acc + dt)
        if k in (1, 75, 150, 225, 300):
            s, p = cp.recording_clock(k)
            assert np.float32(s) == acc and np.float32(p) == acc
    assert cp.recording_clock(0) == (0.0, 0.0)
    assert abs(cp.recording_clock(300)[0] - 12.5) < 1e-3 and cp.recording_clock(300)[0] != 12.5


def test_path_argument_errors():
    from relativisticraytracer_amd import _lib
    lib = _lib.load()
    assert lib.rrt_path_info(3, None, None, None) == 1
    assert lib.rrt_path_camera_at(-1, 0.0, None) == 1
    assert lib.rrt_recording_clock(-1, 24, None, None) == 1


def test_path_camera_states_equal_the_reference_main_cpp(camera_ref):
    """f2 pinned to the reference: rrt_path_camera_at / rrt_camera_from_angles / rrt_recording_clock against the
    CameraState outputs of the reference's OWN CameraController::getCUDAStateFrom (main.cpp:141-167) and
    PathController::getInterpolatedState under the recording clock (:176-212, :511-516) -- that line range of
    src/main.cpp piped into g++ (oracle/Makefile, oracle/ref_main_camera_pre.h) -- bit for bit:
    recording frames {1, 75, 150, 225, 300} of the three paths, a sweep of path times (before the
    first key, on every key, past the end), and 256 random (pos, yaw, pitch)."""
    import relativisticraytracer_amd as rrt
    from relativisticraytracer_amd import camera_paths as cp
    for idx, path in enumerate(cp.paths()):
        for k in (1, 75, 150, 225, 300):
            _, pt = cp.recording_clock(k)
            assert np.float32(pt) == camera_ref[f"path{idx}_frame{k}_time"], (idx, k)      # the controller's own accumulator
            got = path.camera_at(pt).as_array()
            assert same_bits(got, camera_ref[f"path{idx}_frame{k}"]), (idx, k)
        ts = camera_ref[f"path{idx}_sweep_t"]
        got = np.stack([path.camera_at(float(t)).as_array() for t in ts])
        assert same_bits(got, camera_ref[f"path{idx}_sweep_state"]), idx
    pos, yaw, pitch = camera_ref["cam_pos"], camera_ref["cam_yaw"], camera_ref["cam_pitch"]
    got = np.stack([rrt.CameraState.from_angles(pos[i], yaw[i], pitch[i]).as_array() for i in range(len(yaw))])
    assert same_bits(got, camera_ref["cam_state"])
    d = camera_ref["default_camera"]                 # main.cpp:127-130
    assert same_bits(rrt.CameraState.default().as_array(), rrt.CameraState.from_angles(d[:3], d[3], d[4]).as_array())
    assert np.array_equal(d, np.float32([0.0, 10.0, -60.0, 0.0, -10.0]))
