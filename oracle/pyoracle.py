"""ctypes binding of the CPU oracle (oracle/librrt_oracle.so) and, where it has
been built, of the reference-unit libraries under oracle/_ref/.

TEST INFRASTRUCTURE ONLY: imported by tests/, by __graft_entry__.smoke() and by
bench.py's cpu_baseline leg -- never by the product package.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "librrt_oracle.so")
REF_UNITS_PATH = os.path.join(HERE, "_ref", "libref_units.so")
REF_CAMERA_PATH = os.path.join(HERE, "_ref", "libref_camera.so")
REF_FRAMES_PATH = os.path.join(HERE, "_ref", "libref_frames.so")
REF_STB_PATH = os.path.join(HERE, "_ref", "libref_stb.so")
REF_FRAMES_FMA_PATH = os.path.join(HERE, "_ref", "libref_frames_fma.so")     # the same kernel text under -ffp-contract=fast -mfma

MATH_LIBM = 0
MATH_PORTABLE = 1
MATH_NUDGED_BASE = 16      # + seed: glibc with every transcendental result moved by <= 2-3 ulp (rrt_oracle.c: nudge)

_f = C.c_float
_i = C.c_int
_fp = C.POINTER(C.c_float)
_ip = C.POINTER(C.c_int32)
_u8p = C.POINTER(C.c_uint8)


class Camera(C.Structure):
    _fields_ = [("pos", _f * 3), ("forward", _f * 3), ("right", _f * 3), ("up", _f * 3)]


class Effects(C.Structure):
    _fields_ = [("use_bloom", C.c_int32), ("bloom_threshold", _f), ("bloom_intensity", _f),
                ("use_vignette", C.c_int32), ("vignette_intensity", _f),
                ("use_ca", C.c_int32), ("ca_amount", _f),
                ("use_lens", C.c_int32), ("distortion_amount", _f)]


class Params(C.Structure):
    _fields_ = [("spin", _f), ("volumetrics", C.c_int32), ("max_steps", C.c_int32),
                ("math_mode", C.c_int32), ("sky_frac_bits", C.c_int32),
                ("nudge_ulps", C.c_int32), ("nudge_seed", C.c_uint32)]


class Diag(C.Structure):
    _fields_ = [("steps", _ip), ("hit", _ip), ("pos", _fp), ("vel", _fp), ("rad", _fp),
                ("n_noise", _ip), ("n_samples", _ip), ("n_dens", _ip)]


def build(ref=False, quiet=True):
    """(Re)build the oracle; `ref=True` also builds oracle/_ref when /root/reference exists."""
    targets = ["all"] + (["ref", "ref-fma"] if ref else [])
    subprocess.run(["make", "-C", HERE] + targets, check=True,
                   stdout=subprocess.DEVNULL if quiet else None)


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        _lib = C.CDLL(LIB_PATH)
        _lib.rrto_render.restype = _i
        _lib.rrto_max_threads.restype = _i
    return _lib


def use_native_build():
    """Switch this process to a -O3 -march=native build of the same oracle source, compiled on this
    machine (bench.py's cpu_baseline leg).  Returns True if it could be built and loaded."""
    global _lib
    try:
        # -B: always recompile here -- a -march=native object made on another machine must never be reused
        subprocess.run(["make", "-B", "-C", HERE, "native"], check=True, stdout=subprocess.DEVNULL,
                       stderr=subprocess.DEVNULL)
        nl = C.CDLL(os.path.join(HERE, "librrt_oracle_native.so"))
        nl.rrto_render.restype = _i
        nl.rrto_max_threads.restype = _i
        _lib = nl
        return True
    except Exception:
        return False


def _fa(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a):
    return a.ctypes.data_as(_fp)


def default_params(**kw):
    p = Params()
    lib().rrto_default_params(C.byref(p))
    for k, v in kw.items():
        setattr(p, k, v)
    return p


def default_effects(**kw):
    e = Effects()
    lib().rrto_default_effects(C.byref(e))
    for k, v in kw.items():
        setattr(e, k, v)
    return e


def camera(pos, forward, right, up):
    c = Camera()
    for name, v in (("pos", pos), ("forward", forward), ("right", right), ("up", up)):
        arr = getattr(c, name)
        for k in range(3):
            arr[k] = float(v[k])
    return c


def max_threads():
    return lib().rrto_max_threads()


def render(cam, fx, prm, time, width, height, sky, rect=None, stride=(1, 1),
           want=("rgba8",), n_threads=0, gates=None, gate_cap=8192):
    """Render with the oracle.  Returns a dict of full-frame arrays (see rrt_oracle.h).
    gates="record": also returns "gate_log" (n_samples, gate_cap) uint8 and "gate_count" (n_samples,) int32, the
    hard-gate decisions of every rendered ray (row-major over the strided sample grid, top-down); gates=(gate_log, gate_count) replays recorded decisions instead of
    evaluating the gates (rrto_render_gates)."""
    sky = np.ascontiguousarray(sky, dtype=np.uint8)
    sh, sw = sky.shape[:2]
    x0, y0, x1, y1 = rect if rect else (0, 0, width, height)
    out = {}
    rgba8 = np.zeros((height, width, 4), np.uint8) if "rgba8" in want else None
    ldr = np.zeros((height, width, 4), np.float32) if "ldr" in want else None
    hdr = np.zeros((height, width, 4), np.float32) if "hdr" in want else None
    d = Diag()
    dptr = None
    if "diag" in want:
        n = width * height
        out["steps"] = np.zeros(n, np.int32); d.steps = out["steps"].ctypes.data_as(_ip)
        out["hit"] = np.zeros(n, np.int32); d.hit = out["hit"].ctypes.data_as(_ip)
        out["n_noise"] = np.zeros(n, np.int32); d.n_noise = out["n_noise"].ctypes.data_as(_ip)
        out["n_samples"] = np.zeros(n, np.int32); d.n_samples = out["n_samples"].ctypes.data_as(_ip)
        out["n_dens"] = np.zeros(n, np.int32); d.n_dens = out["n_dens"].ctypes.data_as(_ip)
        out["pos"] = np.zeros((n, 3), np.float32); d.pos = _p(out["pos"])
        out["vel"] = np.zeros((n, 3), np.float32); d.vel = _p(out["vel"])
        out["rad"] = np.zeros((n, 4), np.float32); d.rad = _p(out["rad"])
        dptr = C.byref(d)
    gmode, glog, gcnt = 0, None, None
    if gates == "record":
        gmode = 1
        n_s = len(range(y0, y1, stride[1])) * len(range(x0, x1, stride[0]))     # logs are per rendered sample
        glog = np.zeros((n_s, gate_cap), np.uint8); gcnt = np.zeros(n_s, np.int32)
    elif gates is not None:
        gmode = 2
        glog = np.ascontiguousarray(gates[0], np.uint8); gcnt = np.array(gates[1], np.int32)
        gate_cap = glog.shape[1]
    fn = lib().rrto_render_gates
    fn.restype = _i
    rc = fn(C.byref(cam), C.byref(fx), C.byref(prm), _f(time), width, height,
            x0, y0, x1, y1, stride[0], stride[1],
            sky.ctypes.data_as(_u8p), sw, sh,
            rgba8.ctypes.data_as(_u8p) if rgba8 is not None else None,
            _p(ldr) if ldr is not None else None,
            _p(hdr) if hdr is not None else None,
            dptr, n_threads, gmode, glog.ctypes.data_as(_u8p) if glog is not None else None, int(gate_cap),
            gcnt.ctypes.data_as(_ip) if gcnt is not None else None)
    if rc != 0:
        raise ValueError("rrto_render: bad arguments")
    if gmode:
        out["gate_log"], out["gate_count"] = glog, gcnt
    if rgba8 is not None:
        out["rgba8"] = rgba8
    if ldr is not None:
        out["ldr"] = ldr
    if hdr is not None:
        out["hdr"] = hdr
    return out


# ---------------------------------------------------------------- unit functions
class _Units:
    """Array-form unit functions; `prefix` is 'rrto_' (oracle) or 'ref_' (reference build)."""

    def __init__(self, dll, prefix):
        self.dll, self.pre = dll, prefix
        self.is_ref = prefix == "ref_"

    def _fn(self, name):
        return getattr(self.dll, self.pre + name)

    def hash31(self, p):
        p = _fa(p); out = np.empty(len(p), np.float32)
        self._fn("hash31")(len(p), _p(p), _p(out)); return out

    def noise3d(self, p):
        p = _fa(p); out = np.empty(len(p), np.float32)
        self._fn("noise3d")(len(p), _p(p), _p(out)); return out

    def fbm(self, p, octaves):
        p = _fa(p); out = np.empty(len(p), np.float32)
        self._fn("fbm")(len(p), _p(p), int(octaves), _p(out)); return out

    def geodesic_acc(self, p, v, spin):
        p = _fa(p); v = _fa(v); out = np.empty_like(p)
        self._fn("geodesic_acc")(len(p), _p(p), _p(v), _f(spin), _p(out)); return out

    def rk4(self, p, v, h, spin):
        p = _fa(p).copy(); v = _fa(v).copy(); h = _fa(h)
        self._fn("rk4")(len(p), _p(p), _p(v), _p(h), _f(spin)); return p, v

    def redshift(self, p, vel, spin, mode=MATH_LIBM):
        p = _fa(p); vel = _fa(vel); out = np.empty(len(p), np.float32)
        if self.is_ref:
            self._fn("redshift")(len(p), _p(p), _p(vel), _f(spin), _p(out))
        else:
            self._fn("redshift")(len(p), _p(p), _p(vel), _f(spin), int(mode), _p(out))
        return out

    def disk_temperature(self, r, mode=MATH_LIBM):
        r = _fa(r); out = np.empty(len(r), np.float32)
        if self.is_ref:
            self._fn("disk_temperature")(len(r), _p(r), _p(out))
        else:
            self._fn("disk_temperature")(len(r), _p(r), int(mode), _p(out))
        return out

    def accretion_density(self, p, time, mode=MATH_LIBM):
        p = _fa(p); out = np.empty(len(p), np.float32)
        if self.is_ref:
            self._fn("accretion_density")(len(p), _p(p), _f(time), _p(out))
        else:
            self._fn("accretion_density")(len(p), _p(p), _f(time), int(mode), _p(out))
        return out

    def dust_density(self, p, time, mode=MATH_LIBM):
        p = _fa(p); out = np.empty(len(p), np.float32)
        if self.is_ref:
            self._fn("dust_density")(len(p), _p(p), _f(time), _p(out))
        else:
            self._fn("dust_density")(len(p), _p(p), _f(time), int(mode), _p(out))
        return out

    def smoothstep(self, e0, e1, x):
        e0 = _fa(e0); e1 = _fa(e1); x = _fa(x); out = np.empty(len(x), np.float32)
        self._fn("smoothstep")(len(x), _p(e0), _p(e1), _p(x), _p(out)); return out

    def lens(self, uv, k):
        uv = _fa(uv); out = np.empty_like(uv)
        self._fn("lens")(len(uv), _p(uv), _f(k), _p(out)); return out

    def vignette(self, rgb, uv, intensity):
        rgb = _fa(rgb); uv = _fa(uv); out = np.empty_like(rgb)
        self._fn("vignette")(len(rgb), _p(rgb), _p(uv), _f(intensity), _p(out)); return out

    def bloom(self, rgb, threshold):
        rgb = _fa(rgb); out = np.empty_like(rgb)
        self._fn("bloom")(len(rgb), _p(rgb), _f(threshold), _p(out)); return out


def units():
    return _Units(lib(), "rrto_")


def ref_available():
    return os.path.exists(REF_UNITS_PATH)


def ref_stb_available():
    return os.path.exists(REF_STB_PATH)


def ref_stb_load(path):
    """stbi_load(path, ..., 4) of the reference's own stb_image (src/main.cpp:240) -> (H, W, 4) uint8, row 0 = top."""
    dll = C.CDLL(REF_STB_PATH)
    dll.ref_stbi_load_rgba.restype = C.c_void_p
    dll.ref_stbi_load_rgba.argtypes = [C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    dll.ref_stbi_free.argtypes = [C.c_void_p]
    dll.ref_stbi_failure_reason.restype = C.c_char_p
    w, h, n = C.c_int(0), C.c_int(0), C.c_int(0)
    ptr = dll.ref_stbi_load_rgba(os.fsencode(path), C.byref(w), C.byref(h), C.byref(n))
    if not ptr:
        raise RuntimeError(f"stbi_load({path}): {dll.ref_stbi_failure_reason().decode()}")
    try:
        out = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint8)), shape=(h.value, w.value, 4)).copy()
    finally:
        dll.ref_stbi_free(ptr)
    return out, n.value


def ref_units():
    """The reference's own functions (oracle/_ref); only where it has been built."""
    return _Units(C.CDLL(REF_UNITS_PATH), "ref_")


def ref_constants():
    dll = C.CDLL(REF_UNITS_PATH)
    out = np.zeros(18, np.float32)
    dll.ref_constants(_p(out))
    return out


def ref_frames_available():
    return os.path.exists(REF_FRAMES_PATH)


def ref_frames_fma_available():
    return os.path.exists(REF_FRAMES_FMA_PATH)


def ref_render(cam_arr, fx, spin, volumetrics, time, width, height, sky, frac_bits=8, n_threads=0, stride=(1, 1), fma=False):
    """One frame from the REFERENCE's own raymarch_kernel body (oracle/_ref/libref_frames.so, compiled in the
    build container from /root/reference; oracle/ref_frames.cpp).  `cam_arr` is the 4x3 basis, `fx` an Effects;
    `stride` = (sx, sy) renders every sx-th column of every sy-th row only.
    `fma=True`: the same kernel text compiled with floating-point contraction (oracle/Makefile ref-fma).
    Returns {"rgba8": (h, w, 4) uint8 bottom-up, "steps": (h*w,) int32 top-down}."""
    lib()                                   # librrt_oracle.so (the sampler) must be loaded first
    dll = C.CDLL(REF_FRAMES_FMA_PATH if fma else REF_FRAMES_PATH)
    dll.ref_render_strided.restype = _i
    cam12 = _fa(np.asarray(cam_arr).reshape(12))
    flags = np.array([fx.use_bloom, fx.use_vignette, fx.use_ca, fx.use_lens], np.int32)
    vals = np.array([fx.bloom_threshold, fx.bloom_intensity, fx.vignette_intensity, fx.ca_amount,
                     fx.distortion_amount], np.float32)
    sky = np.ascontiguousarray(sky, dtype=np.uint8)
    rgba8 = np.zeros((height, width, 4), np.uint8)
    steps = np.zeros(width * height, np.int32)
    rc = dll.ref_render_strided(_p(cam12), flags.ctypes.data_as(_ip), _p(vals), _f(spin), int(volumetrics), _f(time),
                                width, height, sky.ctypes.data_as(_u8p), sky.shape[1], sky.shape[0], int(frac_bits),
                                rgba8.ctypes.data_as(_u8p), steps.ctypes.data_as(_ip), int(n_threads),
                                int(stride[0]), int(stride[1]))
    if rc != 0:
        raise ValueError("ref_render: bad arguments")
    return {"rgba8": rgba8, "steps": steps}


def sky_sample(dirs, off, sky, frac_bits=8, mode=MATH_LIBM):
    dirs = _fa(dirs); sky = np.ascontiguousarray(sky, np.uint8)
    out = np.empty((len(dirs), 4), np.float32)
    lib().rrto_sky_sample(len(dirs), _p(dirs), _f(off), sky.ctypes.data_as(_u8p),
                          sky.shape[1], sky.shape[0], int(frac_bits), int(mode), _p(out))
    return out


def rt_sample(d_disk, d_cloud, p, vel, h, spin, rad, mode=MATH_LIBM):
    d_disk = _fa(d_disk); d_cloud = _fa(d_cloud); p = _fa(p); vel = _fa(vel); h = _fa(h); rad = _fa(rad).copy()
    lib().rrto_rt_sample(len(h), _p(d_disk), _p(d_cloud), _p(p), _p(vel), _p(h), _f(spin), int(mode), _p(rad))
    return rad


def math_fn(fn, mode, a, b=None):
    """fn: 0 exp, 1 pow(a,b), 2 sin, 3 cos, 4 atan2(a,b), 5 asin."""
    a = _fa(a); b = _fa(b if b is not None else np.zeros_like(a)); out = np.empty_like(a)
    lib().rrto_math(int(fn), int(mode), len(a), _p(a), _p(b), _p(out))
    return out


class RefCamera:
    def __init__(self):
        self.dll = C.CDLL(REF_CAMERA_PATH)
        self.dll.ref_lerp_angle.restype = _f

    def catmull_rom(self, p0, p1, p2, p3, t):
        a = [_fa(x) for x in (p0, p1, p2, p3)]
        out = np.zeros(3, np.float32)
        self.dll.ref_catmull_rom(_p(a[0]), _p(a[1]), _p(a[2]), _p(a[3]), _f(t), _p(out))
        return out

    def lerp_angle(self, a, b, t):
        return float(self.dll.ref_lerp_angle(_f(a), _f(b), _f(t)))

    # the reference's camera code of src/main.cpp:125-220 (piped into g++, oracle/ref_main_camera_pre.h)
    def state_from(self, pos, yaw, pitch):
        """CameraController::getCUDAStateFrom (main.cpp:141-167) -> (4, 3): pos, forward, right, up."""
        out = np.zeros((4, 3), np.float32)
        self.dll.ref_camera_state_from(_p(_fa(pos)), _f(yaw), _f(pitch), _p(out))
        return out

    def path_state_at(self, path, path_time):
        """PathController::getInterpolatedState (main.cpp:176-203) at an explicit path time."""
        out = np.zeros((4, 3), np.float32)
        if self.dll.ref_path_state_at(int(path), _f(path_time), _p(out)) != 0:
            raise IndexError(path)
        return out

    def path_state_at_frame(self, path, frame):
        """State of 1-based recording frame `frame`: start(), then update(1.0f / RECORDING_FPS) per frame
        (main.cpp:511-516) -> ((4, 3) state, the controller's own pathTime)."""
        out = np.zeros((4, 3), np.float32)
        pt = _f(0)
        if self.dll.ref_path_state_at_frame(int(path), int(frame), _p(out), C.byref(pt)) != 0:
            raise IndexError((path, frame))
        return out, np.float32(pt.value)

    def default_camera(self):
        out = np.zeros(5, np.float32)
        self.dll.ref_default_camera(_p(out))
        return out

    def paths(self):
        res = []
        for idx in range(self.dll.ref_path_count()):
            n = self.dll.ref_path_len(idx)
            keys = np.zeros((n, 6), np.float32)
            self.dll.ref_path_keys(idx, _p(keys))
            buf = C.create_string_buffer(128)
            self.dll.ref_path_name(idx, buf, 128)
            res.append((buf.value.decode(), keys))
        return res
