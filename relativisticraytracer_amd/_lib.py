"""ctypes loader for librrt_hip.so (C ABI: include/rrt.h).

There is deliberately no CPU fallback: if the HIP library is missing or a
launch fails, the call raises.

load()       the product library, what the package, bench.py and smoke() use;
load_test()  librrt_hip_test.so -- the same sources built with -DRRT_TEST_HOOKS: everything above plus the entry points of
             include/rrt_test.h (rrt_unit_*, rrt_selfcheck_*, rrt_debug_fake_device).  A separate library with its own
             handle registries: objects created through one are unknown to the other.  Tests only.
"""
import ctypes as C
import os

PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RRT_LIB_OVERRIDE") or os.path.join(PKG, "lib", "librrt_hip.so")   # override: dev A/B builds only
TEST_LIB_PATH = os.path.join(PKG, "lib", "librrt_hip_test.so")

RRT_OK = 0


class RRTError(RuntimeError):
    def __init__(self, status, where, detail=""):
        self.status = status
        super().__init__(f"{where}: {detail}" if detail else where)


class rrt_camera(C.Structure):
    """== reference struct CameraState, include/raymarcher.h:11-16 (48 bytes)."""
    _fields_ = [("pos", C.c_float * 3), ("forward", C.c_float * 3),
                ("right", C.c_float * 3), ("up", C.c_float * 3)]


class rrt_effects(C.Structure):
    """== reference struct CameraEffects, camera_settings.h:4-17 (36 bytes)."""
    _fields_ = [("use_bloom", C.c_uint8), ("_pad0", C.c_uint8 * 3),
                ("bloom_threshold", C.c_float), ("bloom_intensity", C.c_float),
                ("use_vignette", C.c_uint8), ("_pad1", C.c_uint8 * 3),
                ("vignette_intensity", C.c_float),
                ("use_chromatic_aberration", C.c_uint8), ("_pad2", C.c_uint8 * 3),
                ("ca_amount", C.c_float),
                ("use_lens_distortion", C.c_uint8), ("_pad3", C.c_uint8 * 3),
                ("distortion_amount", C.c_float)]


class rrt_params(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("spin", C.c_float), ("max_steps", C.c_int32), ("volumetrics", C.c_int32),
                ("sky_frac_bits", C.c_int32), ("arith_mode", C.c_int32), ("workspace", C.c_int32),
                ("path_policy", C.c_int32), ("noise_table", C.c_int32), ("tile_order", C.c_int32),
                ("pool_rounds", C.c_int32), ("pass_chains", C.c_int32),
                ("nudge_ulps", C.c_int32), ("nudge_seed", C.c_uint32)]


class rrt_debug_outputs(C.Structure):
    _fields_ = [("d_ldr", C.c_void_p), ("d_hdr", C.c_void_p), ("d_steps", C.c_void_p),
                ("d_hit", C.c_void_p), ("d_pos", C.c_void_p), ("d_vel", C.c_void_p),
                ("d_rad", C.c_void_p), ("d_lut_oob", C.c_void_p)]


class rrt_path_chooser_stats(C.Structure):
    _fields_ = [("incumbent", C.c_int32), ("windows", C.c_int32), ("trials", C.c_int32), ("trials_aborted", C.c_int32),
                ("switches", C.c_int32), ("outliers", C.c_int32), ("frames", C.c_int32 * 2), ("last_three_pass_mean_ms", C.c_float)]


# every symbol include/rrt.h declares: (name, restype, argtypes)
_vp, _i, _f, _ull = C.c_void_p, C.c_int, C.c_float, C.c_ulonglong
_cam, _fx, _prm = C.POINTER(rrt_camera), C.POINTER(rrt_effects), C.POINTER(rrt_params)
SYMBOLS = [
    ("rrt_abi_version", _i, []),
    ("rrt_status_string", C.c_char_p, [_i]),
    ("rrt_last_hip_error", C.c_char_p, []),
    ("rrt_device_count", _i, [C.POINTER(_i)]),
    ("rrt_path_auto_max_rays", _i, []),
    ("rrt_params_init", _i, [_vp, C.c_uint32]),
    ("rrt_effects_default", _i, [_fx]),
    ("rrt_sky_create", _i, [_vp, _i, _i, C.POINTER(_ull)]),
    ("rrt_sky_create_from_device", _i, [_vp, _i, _i, C.POINTER(_ull)]),
    ("rrt_sky_destroy", _i, [_ull]),
    ("rrt_workspace_create", _i, [C.c_size_t, C.POINTER(_i)]),
    ("rrt_workspace_destroy", _i, [_i]),
    ("rrt_tile_order_create", _i, [C.POINTER(_i)]),
    ("rrt_tile_order_destroy", _i, [_i]),
    ("rrt_tile_order_set_seeding", _i, [_i, _i]),
    ("rrt_tile_order_seeded", _i, [_i, C.POINTER(_ull)]),
    ("rrt_tile_map_create", _i, [_i, _i, _i, _vp, C.POINTER(_i)]),
    ("rrt_tile_map_destroy", _i, [_i]),
    ("rrt_tile_map_shard_rows", _i, [_i, _i, C.POINTER(_i), C.POINTER(_i)]),
    ("rrt_tile_map_balance", _i, [_i, _vp, _i, _i, _vp]),
    ("rrt_probe_tile_costs", _i, [_i, _i, _i, _f, _cam, _fx, _prm, _vp, _i, _vp]),
    ("rrt_launch_raymarch_tilemap", _i, [_vp, _i, _i, _i, _i, _f, _cam, _ull, _fx, _prm, _vp]),
    ("rrt_assemble_all_tilemap", _i, [_vp, _vp, C.c_size_t, _i, _i, _i, _vp]),
    ("rrt_clock_probe", _i, [_vp, C.c_uint, _vp]),
    ("rrt_workspace_rounds", _i, [_i, C.POINTER(C.c_uint), C.POINTER(C.c_uint), C.POINTER(C.c_uint), C.POINTER(C.c_uint)]),
    ("rrt_tile_order_info", _i, [_i, C.POINTER(_ull), C.POINTER(_ull), C.POINTER(C.c_uint), _vp, _vp, C.c_uint]),
    ("rrt_noise_table_create", _i, [_f, C.POINTER(_i)]),
    ("rrt_noise_table_destroy", _i, [_i]),
    ("rrt_noise_table_info", _i, [_i, C.POINTER(_f), C.POINTER(C.c_size_t), C.POINTER(_i * 12)]),
    ("rrt_noise_table_plan", _i, [_f, C.POINTER(C.c_size_t), C.POINTER(_i * 12)]),
    ("rrt_noise_table_create_window", _i, [_f, _f, _i, C.POINTER(_i)]),
    ("rrt_noise_table_window", _i, [_i, C.POINTER(_f), C.POINTER(_f), C.POINTER(_i), C.POINTER(_i)]),
    ("rrt_noise_table_plan_window", _i, [_f, _f, _i, C.POINTER(C.c_size_t), C.POINTER(_i * 12)]),
    ("rrt_noise_table_plan_layout", _i, [_f, _f, _i, C.POINTER(_i), C.POINTER(_i), C.POINTER(_f), C.POINTER(_f), _vp, _i, _vp]),
    ("rrt_noise_table_fit_window", _i, [_f, _f, C.c_size_t, C.POINTER(_f), C.POINTER(_i), C.POINTER(C.c_size_t)]),
    ("rrt_set_launch_defaults", _i, [_prm]),
    ("rrt_launch_auto_resources", _i, [_i, _prm, C.c_size_t, C.c_size_t]),
    ("rrt_launch_auto_resources_info", _i, [C.POINTER(_i), C.POINTER(_i), C.POINTER(_f), C.POINTER(_f), C.POINTER(C.c_size_t)]),
    ("rrt_get_launch_defaults_sized", _i, [_vp, C.c_uint32]),
    ("rrt_launch_raymarch_compat", _i, [_vp, _i, _i, _f, C.POINTER(C.c_float * 12), _ull, _vp]),
    ("rrt_workspace_stats", _i, [_i, C.POINTER(C.c_uint), C.POINTER(C.c_uint)]),
    ("rrt_workspace_read", _i, [_i, C.c_size_t, C.c_size_t, _vp]),
    ("rrt_launch_raymarch", _i, [_vp, _i, _i, _f, _cam, _ull, _fx, _prm, _vp]),
    ("rrt_launch_raymarch_rows", _i, [_vp, _i, _i, _i, _i, _f, _cam, _ull, _fx, _prm, _vp]),
    ("rrt_launch_raymarch_tiles", _i, [_vp, _i, _i, _i, _i, _i, _f, _cam, _ull, _fx, _prm, _vp]),
    ("rrt_tile_shard_rows", _i, [_i, _i, _i, _i, C.POINTER(_i)]),
    ("rrt_assemble_tiles", _i, [_vp, _vp, _i, _i, _i, _i, _i, _vp]),
    ("rrt_assemble_all_tiles", _i, [_vp, _vp, C.c_size_t, _i, _i, _i, _i, _vp]),
    ("rrt_launch_raymarch_ex", _i, [_vp, _i, _i, _f, _cam, _ull, _fx, _prm,
                                    C.POINTER(rrt_debug_outputs), _vp]),
    ("rrt_camera_from_angles", _i, [C.POINTER(C.c_float * 3), _f, _f, _cam]),
    ("rrt_catmull_rom", _i, [C.POINTER(C.c_float * 3)] * 4 + [_f, C.POINTER(C.c_float * 3)]),
    ("rrt_lerp_angle", _i, [_f, _f, _f, C.POINTER(_f)]),
    ("rrt_path_count", _i, []),
    ("rrt_path_info", _i, [_i, C.POINTER(C.c_char_p), C.POINTER(_i), C.POINTER(_f)]),
    ("rrt_path_keyframes", _i, [_i, _vp, _i]),
    ("rrt_path_camera_at", _i, [_i, _f, _cam]),
    ("rrt_recording_clock", _i, [_i, _i, C.POINTER(_f), C.POINTER(_f)]),
    ("rrt_path_chooser_create", _i, [_i, _i, C.POINTER(_i)]),
    ("rrt_path_chooser_destroy", _i, [_i]),
    ("rrt_path_chooser_policy", _i, [_i, _i, C.POINTER(_i)]),
    ("rrt_path_chooser_report", _i, [_i, _i, _f]),
    ("rrt_path_chooser_get_stats", _i, [_i, _vp]),
]

# include/rrt_test.h: librrt_hip_test.so only
TEST_SYMBOLS = [
    ("rrt_debug_fake_device", _i, [_i]),
    ("rrt_unit_geodesic_acc", _i, [_i, _vp, _vp, _f, _vp, _vp]),
    ("rrt_unit_rk4", _i, [_i, _vp, _vp, _vp, _f, _vp]),
    ("rrt_unit_rk4_lean", _i, [_i, _vp, _vp, _vp, _f, _i, _f, _vp, _vp]),
    ("rrt_unit_div_seeded", _i, [_i, _vp, _vp, _vp, _vp, _vp]),
    ("rrt_unit_hash31", _i, [_i, _vp, _vp, _vp]),
    ("rrt_unit_noise3d", _i, [_i, _vp, _vp, _vp]),
    ("rrt_unit_fbm", _i, [_i, _vp, _i, _vp, _vp]),
    ("rrt_unit_accretion_density", _i, [_i, _vp, _f, _vp, _vp]),
    ("rrt_unit_dust_density", _i, [_i, _vp, _f, _vp, _vp]),
    ("rrt_unit_redshift", _i, [_i, _vp, _vp, _f, _vp, _vp]),
    ("rrt_unit_math", _i, [_i, _i, _vp, _vp, _vp, _vp]),
    ("rrt_unit_sky_sample", _i, [_i, _vp, _f, _ull, _i, _vp, _vp]),
    ("rrt_unit_disk_temperature", _i, [_i, _vp, _vp, _vp]),
    ("rrt_unit_smoothstep", _i, [_i, _vp, _vp, _vp, _vp, _vp]),
    ("rrt_unit_postfx", _i, [_i, _i, _vp, _vp, _f, _vp, _vp]),
    ("rrt_unit_rt_sample", _i, [_i, _vp, _vp, _vp, _vp, _vp, _f, _vp, _vp]),
    ("rrt_unit_noise3d_lut", _i, [_i, _vp, _i, _i, _vp, _vp, _vp]),
    ("rrt_unit_media_lut", _i, [_i, _vp, _f, _i, _vp, _vp, _vp, _vp]),
    ("rrt_selfcheck_sqrt", _i, [C.c_uint32, C.c_uint32, _vp, _vp]),
    ("rrt_selfcheck_div", _i, [_ull, C.c_uint32, _vp, _vp]),
    ("rrt_selfcheck_div_march", _i, [_ull, C.c_uint32, _f, _f, _vp, _vp]),
    ("rrt_selfcheck_sqrt_boundaries", _i, [_i, _i, _i, C.c_uint, _f, _f, _vp, _vp]),
    ("rrt_selfcheck_div_tame", _i, [_ull, C.c_uint32, _vp, _vp]),
    ("rrt_selfcheck_div_const", _i, [C.c_uint32, C.c_uint32, _vp, _vp]),
    ("rrt_selfcheck_sqrt_seeded", _i, [C.c_uint32, C.c_uint32, _vp, _vp]),
]

_lib = None
_test_lib = None


def _bind(lib, symbols, tolerate_missing=False):
    for name, res, args in symbols:
        try:
            fn = getattr(lib, name)      # AttributeError if the export is missing
        except AttributeError:
            if tolerate_missing:
                continue
            raise
        fn.restype = res
        fn.argtypes = args


def load_test():
    """Load librrt_hip_test.so (product + test hooks, include/rrt_test.h).  Tests and dev tools only."""
    global _test_lib
    if _test_lib is not None:
        return _test_lib
    if not os.path.exists(TEST_LIB_PATH):
        raise RRTError(-1, "librrt_hip_test.so is missing",
                       f"expected {TEST_LIB_PATH}; run `python -m relativisticraytracer_amd.build`")
    try:
        import torch  # noqa: F401
    except Exception:
        pass
    lib = C.CDLL(TEST_LIB_PATH)
    _bind(lib, SYMBOLS)
    _bind(lib, TEST_SYMBOLS)
    _test_lib = lib
    return lib


class using_test_library:
    """Tests only: inside `with _lib.using_test_library() as lib:` the whole package talks to librrt_hip_test.so instead of
    the product library (load() returns it), so that the Python wrappers and the test hooks share one set of handle
    registries.  Objects created inside must be destroyed inside."""

    def __enter__(self):
        global _lib
        self.prev = _lib
        _lib = load_test()
        return _lib

    def __exit__(self, *exc):
        global _lib
        _lib = self.prev
        return False


def load():
    """Load librrt_hip.so.  Raises if it has not been built (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RRTError(-1, "librrt_hip.so is missing",
                       f"expected {LIB_PATH}; run `python -m relativisticraytracer_amd.build`")
    try:  # share torch's HIP runtime when torch is in the process (same SONAME libamdhip64.so.7)
        import torch  # noqa: F401
    except Exception:
        pass
    lib = C.CDLL(LIB_PATH)
    # A/B timing of an older build (tools/ab_*.py): tolerate exports it does not have yet; the shipped library must have all
    _bind(lib, SYMBOLS, tolerate_missing="RRT_LIB_OVERRIDE" in os.environ)
    _lib = lib
    return lib


def check(status, where):
    if status != RRT_OK:
        lib = load()
        msg = lib.rrt_status_string(status).decode()
        if status == 3:
            msg += " -- " + lib.rrt_last_hip_error().decode()
        raise RRTError(status, where, msg)
