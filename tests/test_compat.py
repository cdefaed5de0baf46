"""The C++ drop-in boundary: include/raymarcher.h + librrt_hip.so used the way the reference's
src/main.cpp:467 uses its own header -- and an object built against the REFERENCE's header linking
against the same library (the reference's own mangled launch_raymarch symbol)."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
EXE = os.path.join(ROOT, "tests", "compat", "dropin_main")
REF_EXE = os.path.join(ROOT, "tests", "compat", "ref_header_main")
REF_INC = "/root/reference/include"


def _build():
    from relativisticraytracer_amd import build
    build.build_lib()
    cmd = ["g++", "-std=c++17", "-O1", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
           "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "compat", "dropin_main.cpp"),
           "-L" + os.path.dirname(build.LIB), "-lrrt_hip", "-L/opt/rocm/lib", "-lamdhip64",
           "-Wl,-rpath," + os.path.dirname(build.LIB) + ":/opt/rocm/lib", "-o", EXE]
    subprocess.run(cmd, check=True)


def _cuda_inc():
    import importlib.util
    s = importlib.util.find_spec("triton")
    return os.path.join(os.path.dirname(s.origin), "backends", "nvidia", "include") if s else ""


def build_ref_header_main():
    """Build container only: compile against the reference's include/raymarcher.h, link librrt_hip.so."""
    from relativisticraytracer_amd import build
    build.build_lib()
    cmd = ["g++", "-std=c++17", "-O1", "-w", "-I" + _cuda_inc(), "-I" + REF_INC, "-I" + os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests", "compat", "ref_header_main.cpp"),
           "-L" + os.path.dirname(build.LIB), "-lrrt_hip", "-L/opt/rocm/lib", "-lamdhip64",
           "-Wl,-rpath," + os.path.dirname(build.LIB) + ":/opt/rocm/lib", "-o", REF_EXE]
    subprocess.run(cmd, check=True)


def test_dropin_translation_unit_compiles_and_links():
    _build()
    assert os.path.exists(EXE)


@pytest.mark.skipif(not os.path.isdir(REF_INC), reason="the reference is only present in the build container")
def test_object_built_against_the_reference_header_links():
    build_ref_header_main()
    und = subprocess.run(["nm", "-u", REF_EXE], check=True, capture_output=True, text=True).stdout
    assert "_Z15launch_raymarchP6uchar4iif11CameraStatey13CameraEffects" in und


def _python_checksum(w=64, h=36):
    import torch
    import relativisticraytracer_amd as rrt
    sw, sh = 256, 128
    j, i = np.meshgrid(np.arange(sh), np.arange(sw), indexing="ij")
    sky = np.stack([i & 255, (2 * j) & 255, (i ^ j) & 255, np.full_like(i, 255)], -1).astype(np.uint8)
    tex = rrt.SkyTexture(sky)
    buf = torch.zeros(w * h * 4, dtype=torch.uint8, device="cuda")
    rrt.launch_raymarch(buf, w, h, 1.0, rrt.CameraState.default(), tex, rrt.CameraEffects())
    torch.cuda.synchronize()
    s = 1469598103934665603
    for b in buf.cpu().numpy().tolist():
        s = ((s ^ b) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return f"fnv1a64={s:016x}"


@pytest.mark.gpu
def test_dropin_frame_equals_python_path():
    if not os.path.exists(EXE):
        _build()
    out = subprocess.run([EXE, "64", "36"], check=True, capture_output=True, text=True).stdout.split()
    assert out[-1] == _python_checksum()


@pytest.mark.gpu
def test_reference_header_object_renders_the_same_frame():
    """The binary built here against the reference's header (it travels with the repository snapshot)."""
    if not os.path.exists(REF_EXE):
        if not os.path.isdir(REF_INC):
            pytest.skip("ref_header_main was not built (needs the build container)")
        build_ref_header_main()
    out = subprocess.run([REF_EXE, "64", "36"], check=True, capture_output=True, text=True).stdout.split()
    assert out[-1] == _python_checksum()


@pytest.mark.gpu
def test_auto_resources_give_the_drop_in_entry_point_its_speed_and_the_same_bytes(sky):
    """Round 6: a host that keeps calling the reference-signature launch_raymarch() gets pool, tile order
    and noise tables from ONE call, rrt_launch_auto_resources -- library-owned, the tables sliding along `time` -- and renders the
    bytes of an explicit launch with caller-owned objects (and of a launch with none).  Checked through rrt_launch_raymarch_compat,
    the C body of both C++ launch_raymarch symbols."""
    import ctypes as C
    import torch
    import relativisticraytracer_amd as rrt
    from relativisticraytracer_amd import _lib
    lib = _lib.load()
    tex = rrt.SkyTexture(sky)
    w, h = 320, 180
    cam = rrt.CameraState.from_angles((15.0, 3.0, -30.0), -20.0, -5.0)
    fx = rrt.CameraEffects()
    cam12 = (C.c_float * 12)(*cam.as_array().reshape(-1).tolist())
    out = torch.zeros(h * w * 4, dtype=torch.uint8, device="cuda")
    want = torch.zeros_like(out)
    on, builds, t0, t1, nbytes = C.c_int(0), C.c_int(0), C.c_float(0), C.c_float(0), C.c_size_t(0)

    def info():
        assert lib.rrt_launch_auto_resources_info(C.byref(on), C.byref(builds), C.byref(t0), C.byref(t1), C.byref(nbytes)) == 0
        return on.value, builds.value, t0.value, t1.value, nbytes.value

    base = rrt.RenderParams(spin=0.9)
    try:
        assert info()[0] == 0
        _lib.check(lib.rrt_launch_auto_resources(1, C.byref(base), 96 << 20, 256 << 20), "rrt_launch_auto_resources")
        assert info()[:2] == (1, 0)                                   # tables are built lazily, at the first launch
        seen = []
        # 96 MB hold [t, t + 120] at the coarsest coverage near the origin of the clock and nothing at t = 400: the window slides
        # once (t = 130), and the last launch remembers "nothing fits" and hashes arithmetically -- same bytes throughout
        for t in (1.0, 1.5, 130.0, 131.0, 400.0):
            rrt.launch_raymarch(want, w, h, t, cam, tex, fx, rrt.RenderParams(spin=0.9))
            out.zero_()
            _lib.check(lib.rrt_launch_raymarch_compat(C.c_void_p(out.data_ptr()), w, h, C.c_float(t), C.byref(cam12), tex.handle, C.byref(fx)),
                       "rrt_launch_raymarch_compat")
            torch.cuda.synchronize()
            assert torch.equal(out, want), t
            seen.append(info())
            assert seen[-1][2] <= t <= seen[-1][3]
        assert [s[1] for s in seen] == [1, 1, 2, 2, 2] and seen[0][4] > 0 and seen[2][4] > 0 and seen[4][4] == 0      # one build per window
        # explicit launches are not affected, and switching off destroys everything
        _lib.check(lib.rrt_launch_auto_resources(0, None, 0, 0), "rrt_launch_auto_resources(0)")
        assert info()[0] == 0
        out.zero_()
        _lib.check(lib.rrt_launch_raymarch_compat(C.c_void_p(out.data_ptr()), w, h, C.c_float(1.0), C.byref(cam12), tex.handle, C.byref(fx)), "compat")
        rrt.launch_raymarch(want, w, h, 1.0, cam, tex, fx, rrt.RenderParams())         # config.h defaults again: spin 0
        torch.cuda.synchronize()
        assert torch.equal(out, want)
        bad = rrt.RenderParams(spin=0.9); bad.struct_size = 40
        assert lib.rrt_launch_auto_resources(1, C.byref(bad), 0, 0) == 6                # RRT_ERR_ABI_MISMATCH, nothing created
        assert info()[0] == 0
    finally:
        lib.rrt_launch_auto_resources(0, None, 0, 0)
        tex.destroy()
