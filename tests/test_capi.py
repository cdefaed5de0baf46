"""The C-ABI library loads on a CPU-only host and exports everything include/rrt.h declares.
No compute is launched here (that is tests/test_gpu_*.py)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def declared_functions():
    src = open(os.path.join(ROOT, "include", "rrt.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(rrt_[a-z0-9_]+)\s*\(", src)))


def test_header_and_binding_table_agree():
    from relativisticraytracer_amd import _lib
    bound = sorted(name for name, _, _ in _lib.SYMBOLS)
    assert declared_functions() == bound


def test_library_exports_every_declared_symbol():
    from relativisticraytracer_amd import _lib
    lib = _lib.load()
    for name in declared_functions():
        assert hasattr(lib, name), name
    assert lib.rrt_abi_version() == 1


def test_struct_layouts_match_the_reference_structs():
    from relativisticraytracer_amd import _lib
    # CameraState: 4 x float3 = 48 B (include/raymarcher.h:11-16)
    assert C.sizeof(_lib.rrt_camera) == 48
    # CameraEffects: 36 B, bools in 4-byte slots at 0,12,20,28 (camera_settings.h:4-17)
    assert C.sizeof(_lib.rrt_effects) == 36
    offs = {f[0]: getattr(_lib.rrt_effects, f[0]).offset for f in _lib.rrt_effects._fields_}
    assert [offs[k] for k in ("use_bloom", "bloom_threshold", "bloom_intensity", "use_vignette",
                              "vignette_intensity", "use_chromatic_aberration", "ca_amount",
                              "use_lens_distortion", "distortion_amount")] == [0, 4, 8, 12, 16, 20, 24, 28, 32]
    assert C.sizeof(_lib.rrt_params) == 32


def test_defaults_are_the_reference_defaults():
    import relativisticraytracer_amd as rrt
    fx = rrt.CameraEffects()
    assert (fx.useBloom, fx.useVignette, fx.useChromaticAberration, fx.useLensDistortion) == (1, 1, 0, 1)
    assert abs(fx.bloomThreshold - 0.8) < 1e-7 and abs(fx.bloomIntensity - 0.5) < 1e-7
    assert abs(fx.vignetteIntensity - 0.4) < 1e-7 and abs(fx.caAmount - 0.005) < 1e-9
    assert abs(fx.distortionAmount - 0.15) < 1e-7
    p = rrt.RenderParams()
    assert (p.spin, p.max_steps, p.volumetrics, p.sky_frac_bits, p.arith_mode, p.workspace, p.path_policy) == (0.0, 2000, 1, 8, 0, 0, 0)
    with pytest.raises(AttributeError):
        rrt.RenderParams(nonexistent=1)


def test_status_strings_and_host_side_errors():
    from relativisticraytracer_amd import _lib
    lib = _lib.load()
    assert lib.rrt_status_string(0) == b"ok"
    assert lib.rrt_status_string(1) == b"invalid argument"
    assert lib.rrt_params_default(None) == 1
    assert lib.rrt_effects_default(None) == 1
    assert lib.rrt_sky_destroy(0) == 4
    assert lib.rrt_sky_create(None, 4, 4, None) == 1
    rows = C.c_int(0)
    assert lib.rrt_tile_shard_rows(2160, 16, 7, 8, C.byref(rows)) == 0 and rows.value == 256
    assert lib.rrt_tile_shard_rows(2160, 16, 8, 8, C.byref(rows)) == 1


def test_camera_from_angles_matches_reference_formula():
    """getCUDAStateFrom (src/main.cpp:141-167) for the start-up camera: closed form."""
    import numpy as np
    import relativisticraytracer_amd as rrt
    a = rrt.CameraState.default().as_array()
    rp = np.float32(-10.0) * np.float32(3.14159) / np.float32(180.0)
    assert np.allclose(a[0], [0, 10, -60])
    assert np.allclose(a[1], [0, np.sin(rp), np.cos(rp)], atol=1e-7)
    assert np.allclose(a[2], [1, 0, 0], atol=1e-7)
    assert np.allclose(a[3], np.cross(a[1], a[2]), atol=1e-7)
    for v in a[1:]:
        assert abs(np.linalg.norm(v) - 1) < 1e-6
