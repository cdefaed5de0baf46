"""Build librrt_hip.so (the gfx950 kernels + C ABI) in-tree with hipcc.

    python -m relativisticraytracer_amd.build [--force] [--save-temps]

The library is cross-compiled (no GPU needed) and lands in
relativisticraytracer_amd/lib/, from where the ctypes loader picks it up and
from where it travels to the GPU box.
"""
import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
LIBDIR = os.path.join(PKG, "lib")
LIB = os.path.join(LIBDIR, "librrt_hip.so")
TEST_LIB = os.path.join(LIBDIR, "librrt_hip_test.so")   # the same sources + -DRRT_TEST_HOOKS: rrt_unit_*, rrt_selfcheck_*, rrt_debug_fake_device
SOURCES = [os.path.join(CSRC, "rrt_hip.hip")]
COMPAT_SRC = os.path.join(CSRC, "rrt_compat.cpp")     # launch_raymarch under the reference's mangled name (host only, g++)
CAMERA_SRC = os.path.join(CSRC, "rrt_camera.cpp")     # camera basis / path playback (host only, g++)
CHOOSER_SRC = os.path.join(CSRC, "rrt_path_chooser.cpp")   # per-window path choice of the animation drivers (host only, g++)
HEADERS = [os.path.join(CSRC, f) for f in ("rrt_device.h", "rrt_math.h", "rrt_tile_sort.h", "rrt_kernels.h", "rrt_test_hooks.h", "rrt_noise_plan.h", "rrt_tile_objects.h")] + [
    COMPAT_SRC, CAMERA_SRC, CHOOSER_SRC, os.path.join(PKG, "..", "include", "rrt.h"), os.path.join(PKG, "..", "include", "rrt_test.h"),
    os.path.join(PKG, "..", "include", "raymarcher.h")]

# -ffp-contract=off: the kernels' arithmetic contract (csrc/rrt_device.h).
# -fno-slp-vectorize: the SLP vectoriser packs the 3-vector math into v_pk_*_f32 plus a pile of
# v_mov shuffles; measured 8 % slower than scalar VALU on the march loop (profiles/README.md).
# -mllvm -enable-post-misched=false: the post-RA machine scheduler's reordering costs the kernels that carry the
# volumetric code 3-4 % (4K bench frame 44.6 -> 43.3 ms, same bytes; profiles/README.md); the bare march is unaffected.
# -mllvm -amdgpu-sched-strategy=max-ilp (round 3): inside the short guarded blocks the step is now made of, the ILP-first
# pre-RA strategy costs registers the bare march can spare (55 -> 73 VGPRs; it keeps its rate down to 4 waves per SIMD) and buys
# 1.2-1.4 % on the kernels without media code, 3 % in fast mode, 0.3-0.9 % with media (profiles/r03_sched_strategy_ab.txt).
# Round 2 measured the same flag 25 % SLOWER on the then monolithic step -- it interleaved the four stages.
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
               "-fno-gpu-rdc", "-fno-slp-vectorize", "-mllvm", "-enable-post-misched=false",
               "-mllvm", "-amdgpu-sched-strategy=max-ilp",
               "-Wall", "-Wno-unused-function"]


def hipcc_path():
    p = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(p):
        raise RuntimeError("hipcc not found: the HIP extension cannot be built")
    return p


def is_stale():
    if not os.path.exists(LIB) or not os.path.exists(TEST_LIB):
        return True
    t = min(os.path.getmtime(LIB), os.path.getmtime(TEST_LIB))
    return any(os.path.getmtime(f) > t for f in SOURCES + HEADERS + [os.path.abspath(__file__)])


def host_objects(outdir):
    """the plain-C++ translation units (g++): the reference-mangled launch_raymarch, the camera code, the path chooser"""
    objs = []
    for src in (COMPAT_SRC, CAMERA_SRC, CHOOSER_SRC):
        obj = os.path.join(outdir, os.path.splitext(os.path.basename(src))[0] + ".o")
        subprocess.run(["g++", "-std=c++17", "-O2", "-fPIC", "-Wall", "-c", src, "-o", obj], check=True)
        objs.append(obj)
    return objs


def build_lib(force=False, extra_flags=(), verbose=False):
    """librrt_hip.so (the product) and librrt_hip_test.so (+ test hooks), from the same sources.  Each hipcc compile runs in a
    directory of its own (lib/_obj_product, lib/_obj_test: flags that write fixed-name side files -- -save-temps -- cannot make
    the two collide); they run side by side, except under -save-temps, where they run one after the other.  BOTH libraries are
    always built, so that is_stale() has one meaning."""
    if not force and not is_stale():
        return LIB
    os.makedirs(LIBDIR, exist_ok=True)
    host_objs = host_objects(LIBDIR)
    # two steps per library: with a .hip input hipcc compiles every input as HIP source, objects included
    base = [hipcc_path()] + [f for f in HIPCC_FLAGS if f != "-shared"] + list(extra_flags)
    sequential = "-save-temps" in extra_flags
    todo = ((LIB, "product", []), (TEST_LIB, "test", ["-DRRT_TEST_HOOKS"]))
    jobs = []
    for lib, tag, defs in todo:
        wd = os.path.join(LIBDIR, "_obj_" + tag)
        os.makedirs(wd, exist_ok=True)
        cmd = base + defs + ["-c"] + SOURCES + ["-o", os.path.join(wd, "rrt_hip.o")]
        if verbose:
            print(" ".join(cmd), flush=True)
        if sequential:
            subprocess.run(cmd, check=True, cwd=wd)
        else:
            jobs.append((cmd, subprocess.Popen(cmd, cwd=wd)))
    for cmd, pr in jobs:
        if pr.wait() != 0:
            raise subprocess.CalledProcessError(pr.returncode, cmd)
    for lib, tag, defs in todo:
        wd = os.path.join(LIBDIR, "_obj_" + tag)
        link = [hipcc_path(), "--offload-arch=gfx950", "-shared", "-fPIC", "-fno-gpu-rdc", os.path.join(wd, "rrt_hip.o")] + host_objs + ["-o", lib]
        if verbose:
            print(" ".join(link), flush=True)
        subprocess.run(link, check=True, cwd=wd)
    return LIB


def build_variant(name, extra_flags=()):
    """dev builds for A/B timing: lib/variants/<name>.so with extra hipcc flags (tools/ab_views.py)."""
    vdir = os.path.join(LIBDIR, "variants")
    hdir = os.path.join(vdir, name + "_host")          # its own directory: variant builds may run side by side
    os.makedirs(hdir, exist_ok=True)
    host_objs = host_objects(hdir)
    obj = os.path.join(vdir, name + ".o")
    base = [f for f in HIPCC_FLAGS if f != "-shared"]
    for d in [f[len("--drop="):] for f in extra_flags if f.startswith("--drop=")]:      # --drop=<flag>: build WITHOUT a shipped flag
        i = base.index(d)
        if i > 0 and base[i - 1] == "-mllvm":
            del base[i - 1:i + 1]
        else:
            del base[i]
    extra_flags = [f for f in extra_flags if not f.startswith("--drop=")]
    subprocess.run([hipcc_path()] + base + list(extra_flags) + ["-c"] + SOURCES + ["-o", obj], check=True, cwd=hdir)
    out = os.path.join(vdir, name + ".so")
    subprocess.run([hipcc_path(), "--offload-arch=gfx950", "-shared", "-fPIC", "-fno-gpu-rdc", obj] + host_objs + ["-o", out],
                   check=True, cwd=vdir)
    for o in [obj] + host_objs:
        os.remove(o)
    os.rmdir(hdir)
    return out


HEADLESS_SRC = os.path.join(CSRC, "rrt_headless.cpp")
HEADLESS_BIN = os.path.join(LIBDIR, "rrt_headless")


def build_headless(force=False):
    """The C++ headless driver (host code only: g++, links librrt_hip.so, the HIP runtime and RCCL)."""
    build_lib()
    if not force and os.path.exists(HEADLESS_BIN) and os.path.getmtime(HEADLESS_BIN) > max(
            os.path.getmtime(HEADLESS_SRC), os.path.getmtime(LIB)):
        return HEADLESS_BIN
    cmd = ["g++", "-std=c++17", "-O2", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", HEADLESS_SRC,
           "-L" + LIBDIR, "-lrrt_hip", "-L/opt/rocm/lib", "-lamdhip64", "-lrccl", "-Wl,-rpath," + LIBDIR + ":/opt/rocm/lib",
           "-o", HEADLESS_BIN]
    subprocess.run(cmd, check=True)
    return HEADLESS_BIN


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--variant":
        print(build_variant(sys.argv[2], sys.argv[3:]))
        sys.exit(0)
    extra = []
    if "--save-temps" in sys.argv:
        extra += ["-save-temps", "-Rpass-analysis=kernel-resource-usage"]
    print(build_lib(force="--force" in sys.argv or bool(extra), extra_flags=extra, verbose=True))
    print(build_headless(force="--force" in sys.argv))
