"""The C++ drop-in boundary: include/raymarcher.h + librrt_hip.so used the way the reference's
src/main.cpp:467 uses its own header."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
EXE = os.path.join(ROOT, "tests", "compat", "dropin_main")


def _build():
    from relativisticraytracer_amd import build
    build.build_lib()
    cmd = ["g++", "-std=c++17", "-O1", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
           "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "compat", "dropin_main.cpp"),
           "-L" + os.path.dirname(build.LIB), "-lrrt_hip", "-L/opt/rocm/lib", "-lamdhip64",
           "-Wl,-rpath," + os.path.dirname(build.LIB) + ":/opt/rocm/lib", "-o", EXE]
    subprocess.run(cmd, check=True)


def test_dropin_translation_unit_compiles_and_links():
    _build()
    assert os.path.exists(EXE)


@pytest.mark.gpu
def test_dropin_frame_equals_python_path():
    import torch
    import relativisticraytracer_amd as rrt
    if not os.path.exists(EXE):
        _build()
    out = subprocess.run([EXE, "64", "36"], check=True, capture_output=True, text=True).stdout.split()
    sw, sh = 256, 128
    j, i = np.meshgrid(np.arange(sh), np.arange(sw), indexing="ij")
    sky = np.stack([i & 255, (2 * j) & 255, (i ^ j) & 255, np.full_like(i, 255)], -1).astype(np.uint8)
    tex = rrt.SkyTexture(sky)
    buf = torch.zeros(64 * 36 * 4, dtype=torch.uint8, device="cuda")
    rrt.launch_raymarch(buf, 64, 36, 1.0, rrt.CameraState.default(), tex, rrt.CameraEffects())
    torch.cuda.synchronize()
    s = 1469598103934665603
    for b in buf.cpu().numpy().tolist():
        s = ((s ^ b) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    assert out[-1] == f"fnv1a64={s:016x}"
