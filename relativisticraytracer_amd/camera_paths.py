"""Camera-path playback: Python mirror of the reference's host code
(include/camera_paths.h, src/camera_paths.cpp, PathController in src/main.cpp:171-220).
All arithmetic happens in the C++ side of librrt_hip.so; this only wraps the C ABI."""
import ctypes as C

import numpy as np

from . import CameraState, _lib


def _f3(v):
    return (C.c_float * 3)(*[float(x) for x in v])


def catmull_rom(p0, p1, p2, p3, t):
    out = (C.c_float * 3)()
    _lib.check(_lib.load().rrt_catmull_rom(C.byref(_f3(p0)), C.byref(_f3(p1)), C.byref(_f3(p2)), C.byref(_f3(p3)),
                                           float(t), C.byref(out)), "rrt_catmull_rom")
    return np.array(list(out), np.float32)


def lerp_angle(a, b, t):
    out = C.c_float(0)
    _lib.check(_lib.load().rrt_lerp_angle(float(a), float(b), float(t), C.byref(out)), "rrt_lerp_angle")
    return np.float32(out.value)


class CameraPath:
    """One of the reference's built-in paths (PathManager entry)."""

    def __init__(self, index):
        lib = _lib.load()
        name, n, t_end = C.c_char_p(), C.c_int(0), C.c_float(0)
        _lib.check(lib.rrt_path_info(index, C.byref(name), C.byref(n), C.byref(t_end)), "rrt_path_info")
        self.index, self.name, self.t_end = index, name.value.decode(), t_end.value
        keys = np.zeros((n.value, 6), np.float32)
        _lib.check(lib.rrt_path_keyframes(index, keys.ctypes.data_as(C.c_void_p), n.value), "rrt_path_keyframes")
        self.keyframes = keys          # rows: time, x, y, z, yaw, pitch

    def camera_at(self, path_time):
        """PathController::getInterpolatedState (src/main.cpp:176-203)."""
        out = CameraState()
        _lib.check(_lib.load().rrt_path_camera_at(self.index, float(path_time), C.byref(out)), "rrt_path_camera_at")
        return out


def paths():
    return [CameraPath(i) for i in range(_lib.load().rrt_path_count())]


def recording_clock(frame_k, fps=24):
    """(simTime, pathTime) seen by 1-based frame k under the fixed recording clock
    (src/main.cpp:511-516; RECORDING_FPS = 24, config.h:9), accumulated in binary32."""
    s, p = C.c_float(0), C.c_float(0)
    _lib.check(_lib.load().rrt_recording_clock(int(frame_k), int(fps), C.byref(s), C.byref(p)), "rrt_recording_clock")
    return s.value, p.value
