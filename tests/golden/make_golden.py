#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/.

Runs in the BUILD CONTAINER only (it needs /root/reference for the unit
vectors).  Fixtures of different provenance are kept in separate files so their provenance
stays clear:

  units_ref.npz      inputs + outputs of the REFERENCE's own device functions
                     (oracle/_ref/libref_units.so = /root/reference/include/*.h
                     compiled by g++, see oracle/ref_units.cpp).  These pin the
                     oracle restatement to the reference, bit for bit.
  camera_ref.npz     catmull_rom / lerp_angle samples and the three keyframe
                     tables from the reference's src/camera_paths.cpp, and CameraState
                     outputs of the reference's own CameraController::getCUDAStateFrom /
                     PathController::getInterpolatedState (src/main.cpp:125-220 piped into
                     g++, oracle/ref_main_camera_pre.h): recording frames 1/75/150/225/300
                     of each path, a sweep of path times, random (pos, yaw, pitch).
  frames_ref.npz     small frames rendered by the REFERENCE's own raymarch_kernel body
                     (/root/reference/src/raymarcher.cu:15-174 compiled by g++ where it lies,
                     oracle/ref_frames.cpp + ref_frames_pre.h -> oracle/_ref/libref_frames.so):
                     RGBA8 and per-ray RK4 step counts.  The only harness-defined arithmetic in
                     them is the sky texel filter (tex2D<float4>).  These pin
                     the per-pixel glue of the restatement (loop order, zones, RT block, sky
                     lookup, post-FX, tone map, row flip) to the reference, byte for byte.
  frames_ref_fma.npz the same scenes (and four 320x180 views, with their strict reference frames) from the reference's kernel
                     body compiled with floating-point contraction (g++ -ffp-contract=fast -mfma; oracle/Makefile ref-fma):
                     what a contracting compiler does to the reference itself -- the bound for RRT_ARITH_FMAD's deviation.
  frames_oracle.npz  the same small frames rendered by the oracle RESTATEMENT in both math
                     modes, with per-ray diagnostics the reference kernel does not output
                     (final p / vel / radiance, float RGB): regression pins + GPU comparison data.

  rk4_chain_ref.npz  50-step chains of the REFERENCE's own integrate_rk4 (integrators.h:23-59, libref_units.so)
                     under the march's zone rule for the step size and its horizon test (raymarcher.cu:42-64),
                     a in {0, 0.9, 0.99}: states after {1, 2, 3, 5, 10, 25, 50} steps.  Pins the PRODUCTION step of the
                     HIP kernels (integrate_rk4_lean: seed pairs carried from step to step, wave-uniform vacuum step,
                     extrapolated seeds, v_rsq fall-backs) at unit level: rrt_unit_rk4_lean, tests/test_gpu_units.py.

  sky_ref.npz        the reference's sky LOADER: its own assets decoded by its own stb_image
                     (stbi_load(...,4), src/main.cpp:240; oracle/ref_stb.c) -- sizes, sha256 of the full decodes, six 64x64
                     crops and one 256x128 crop of skybox2.jpg, the histogram of PIL-minus-stb differences (what this
                     package's fallback decoder does to the same file), and one frame rendered by the reference's own
                     kernel body with the 256x128 crop as its sky.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
sys.path.insert(0, ROOT)

from oracle import pyoracle as po  # noqa: E402
from relativisticraytracer_amd.sky import synthetic_sky  # noqa: E402

SPINS = (0.0, 0.9, 0.99)
TIMES = (0.0, 1.0, 12.5)


def unit_inputs():
    rng = np.random.default_rng(20260130)
    n = 1024
    d = {}
    # positions with r in [1.5, 300] (log-uniform) + a few inside r < 1 for the geodesics.h:33 branch
    dirs = rng.normal(size=(n, 3)); dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
    r = np.exp(rng.uniform(np.log(1.5), np.log(300.0), n))
    r[:16] = rng.uniform(0.05, 0.999, 16)
    d["geo_p"] = (dirs * r[:, None]).astype(np.float32)
    v = rng.normal(size=(n, 3)); v /= np.linalg.norm(v, axis=1, keepdims=True)
    d["geo_v"] = (v * rng.uniform(0.8, 1.25, (n, 1))).astype(np.float32)
    d["rk4_h"] = np.float32(0.3) * np.float32([1.0, 0.3, 0.1])[rng.integers(0, 3, n)]
    d["rk4_h"] = d["rk4_h"].astype(np.float32)
    # lattice points for hash31: small, negative, large, and the exact-integer products at +-10000
    lat = rng.integers(-400, 400, (n, 3)).astype(np.float32)
    lat[:64] = rng.integers(-20000, 20000, (64, 3))
    lat[64:70] = [[10000, 0, 0], [-10000, 0, 0], [0, 10000, -10000], [0, 0, 0], [-1, -1, -1], [10000, 10000, 10000]]
    d["lattice"] = lat.astype(np.float32)
    pts = rng.uniform(-60, 60, (n, 3)).astype(np.float32)
    pts[:32] = rng.uniform(-3000, 3000, (32, 3))
    pts[32:40] = np.round(pts[32:40])             # exactly on lattice planes
    d["noise_p"] = pts.astype(np.float32)
    # in-zone sample points
    ang = rng.uniform(-np.pi, np.pi, n)
    rc = rng.uniform(9.0, 26.0, n)
    y_disk = rng.uniform(-4.0, 4.0, n); y_disk[: n // 2] = rng.normal(0, 0.5, n // 2)
    d["disk_p"] = np.stack([rc * np.cos(ang), y_disk, rc * np.sin(ang)], 1).astype(np.float32)
    y_cloud = rng.uniform(-0.75, 0.75, n); y_cloud[: n // 2] = rng.normal(0, 0.15, n // 2)
    d["cloud_p"] = np.stack([rc * np.cos(ang), y_cloud, rc * np.sin(ang)], 1).astype(np.float32)
    d["temp_r"] = rng.uniform(2.0, 40.0, n).astype(np.float32)
    d["ss_e0"] = rng.uniform(-1, 1, n).astype(np.float32)
    d["ss_e1"] = (d["ss_e0"] + rng.choice([-1.0, 1.0], n) * rng.uniform(0.05, 2, n)).astype(np.float32)
    d["ss_x"] = rng.uniform(-2, 2, n).astype(np.float32)
    d["uv"] = rng.uniform(0, 1, (n, 2)).astype(np.float32)
    d["rgb"] = rng.exponential(0.6, (n, 3)).astype(np.float32)
    return d


def make_units():
    if not po.ref_available():
        po.build(ref=True)
    ref = po.ref_units()
    d = unit_inputs()
    out = dict(d)
    out["constants"] = po.ref_constants()
    for a in SPINS:
        tag = f"{a:g}"
        out[f"geodesic_acc_a{tag}"] = ref.geodesic_acc(d["geo_p"], d["geo_v"], a)
        p, v = ref.rk4(d["geo_p"], d["geo_v"], d["rk4_h"], a)
        out[f"rk4_p_a{tag}"], out[f"rk4_v_a{tag}"] = p, v
        out[f"redshift_disk_a{tag}"] = ref.redshift(d["disk_p"], d["geo_v"], a)
    out["hash31"] = ref.hash31(d["lattice"])
    out["noise3d"] = ref.noise3d(d["noise_p"])
    out["fbm2"] = ref.fbm(d["noise_p"], 2)
    out["fbm5"] = ref.fbm(d["noise_p"], 5)
    for t in TIMES:
        out[f"accretion_t{t:g}"] = ref.accretion_density(d["disk_p"], t)
        out[f"dust_t{t:g}"] = ref.dust_density(d["cloud_p"], t)
    out["disk_temperature"] = ref.disk_temperature(d["temp_r"])
    out["smoothstep"] = ref.smoothstep(d["ss_e0"], d["ss_e1"], d["ss_x"])
    out["lens_k0.15"] = ref.lens(d["uv"], 0.15)
    out["vignette_i0.4"] = ref.vignette(d["rgb"], d["uv"], 0.4)
    out["bloom_t0.8"] = ref.bloom(d["rgb"], 0.8)
    np.savez_compressed(os.path.join(HERE, "units_ref.npz"), **out)
    print("units_ref.npz:", len(out), "arrays")


CHAIN_MARKS = (1, 2, 3, 5, 10, 25, 50)


def chain_inputs():
    """512 start states in wave-aligned groups of 64 (the vacuum step is taken per WAVEFRONT): 4 waves far out (all
    lanes stay at r >= 30 for 50 steps: vacuum step, extrapolated seeds), 1 wave straddling r = 30, 1 in the disk zone,
    1 near the hole (plunging rays stop at the horizon test), 1 log-uniform mix."""
    rng = np.random.default_rng(20261004)
    n = 512
    dirs = rng.normal(size=(n, 3)); dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
    r = np.empty(n)
    r[:256] = rng.uniform(60.0, 240.0, 256)
    r[256:320] = rng.uniform(28.0, 36.0, 64)
    r[320:384] = rng.uniform(19.0, 29.0, 64)
    r[384:448] = rng.uniform(2.5, 17.0, 64)
    r[448:] = np.exp(rng.uniform(np.log(2.1), np.log(300.0), 64))
    p = dirs * r[:, None]
    p[320:384, 1] = rng.uniform(-3.9, 3.9, 64)                       # inside |y| < 4: the disk zone's step size
    v = rng.normal(size=(n, 3)); v /= np.linalg.norm(v, axis=1, keepdims=True)
    v[384:416] = -dirs[384:416] + 0.3 * v[384:416]                    # half of the near-hole wave heads inwards
    v *= rng.uniform(0.9, 1.1, (n, 1))
    return p.astype(np.float32), v.astype(np.float32)


def march_step_size(p):
    """raymarcher.cu:42-62 in binary32: (alive = passes the horizon test, h by zone)."""
    f = np.float32
    x, y, z = p[:, 0], p[:, 1], p[:, 2]
    r = np.sqrt((x * x + y * y) + z * z)                              # dot() of math_utils.h, sqrtf
    alive = ~(r < f(2.0) * f(1.01))
    near = r < f(18.0)
    disk = (np.abs(y) < f(0.8) * f(5.0)) & (r < f(25.0) + f(5.0))
    h = np.where(near, f(0.3) * f(0.1), np.where(disk, f(0.3) * f(0.3), f(0.3))).astype(np.float32)
    return alive, h


def make_rk4_chains():
    if not po.ref_available():
        po.build(ref=True)
    ref = po.ref_units()
    p0, v0 = chain_inputs()
    out = {"p0": p0, "v0": v0, "marks": np.int32(CHAIN_MARKS)}
    for a in SPINS:
        p, v = p0.copy(), v0.copy()
        steps = np.zeros(len(p), np.int32)
        live = np.ones(len(p), bool)
        for k in range(1, max(CHAIN_MARKS) + 1):
            alive, h = march_step_size(p)
            live &= alive                                             # a ray that met the horizon test stays where it is
            pn, vn = ref.rk4(p[live], v[live], h[live], a)
            p[live], v[live] = pn, vn
            steps[live] += 1
            if k in CHAIN_MARKS:
                out[f"p_a{a:g}_k{k}"], out[f"v_a{a:g}_k{k}"] = p.copy(), v.copy()
                out[f"steps_a{a:g}_k{k}"] = steps.copy()
        print(f"rk4 chain a={a:g}: {int((~live).sum())} rays stopped at the horizon test, "
              f"{int((np.linalg.norm(p[:256], axis=1) < 30).sum())} far-out rays came inside r = 30")
    np.savez_compressed(os.path.join(HERE, "rk4_chain_ref.npz"), **out)
    print("rk4_chain_ref.npz:", len(out), "arrays")


SKY_ASSETS = ("skybox2.jpg", "skybox.png")          # /root/reference/assets/skyboxes; main.cpp:497 loads the first
SKY_CROPS = ((0, 0), (1000, 2016), (512, 3000), (1984, 4032), (900, 100), (1500, 2500))      # (row, col) of the 64x64 crops
SKY_BIG_CROP = (960, 1920, 128, 256)                # row, col, rows, cols: the sky of the fixture frame below


def make_sky():
    import hashlib
    from PIL import Image
    if not po.ref_stb_available():
        po.build(ref=True)
    out = {}
    for name in SKY_ASSETS:
        path = os.path.join("/root/reference/assets/skyboxes", name)
        stb, channels = po.ref_stb_load(path)
        with Image.open(path) as im:
            pil = np.ascontiguousarray(np.asarray(im.convert("RGBA"), dtype=np.uint8))
        tag = name.replace(".", "_")
        out[f"{tag}_size"] = np.int32([stb.shape[1], stb.shape[0], channels])
        out[f"{tag}_stb_sha256"] = np.frombuffer(hashlib.sha256(stb.tobytes()).digest(), np.uint8)
        out[f"{tag}_file_sha256"] = np.frombuffer(hashlib.sha256(open(path, "rb").read()).digest(), np.uint8)
        d = pil[..., :3].astype(np.int16) - stb[..., :3].astype(np.int16)
        vals, counts = np.unique(d, return_counts=True)
        out[f"{tag}_pil_minus_stb_values"], out[f"{tag}_pil_minus_stb_counts"] = vals.astype(np.int16), counts.astype(np.int64)
        out[f"{tag}_alpha_all_255"] = np.bool_((stb[..., 3] == 255).all())
        print(f"{name}: {stb.shape[1]}x{stb.shape[0]}, PIL differs from stb_image on {100.0 * (d != 0).mean():.2f} % of the colour bytes "
              f"(max |d| {int(np.abs(d).max())})")
        if name == SKY_ASSETS[0]:
            for k, (r, c) in enumerate(SKY_CROPS):
                out[f"{tag}_crop{k}_at"] = np.int32([r, c]); out[f"{tag}_crop{k}"] = stb[r:r + 64, c:c + 64].copy()
            r, c, nr, nc = SKY_BIG_CROP
            big = stb[r:r + nr, c:c + nc].copy()
            out[f"{tag}_bigcrop_at"] = np.int32(SKY_BIG_CROP); out[f"{tag}_bigcrop"] = big
            # one frame of the reference's own kernel body with that crop as the sky: the default camera looks past the
            # hole at the sky, a = 0.9, all four effects (three sky fetches per ray)
            w, h, spin, vol, camspec, t, fxkw = 96, 54, 0.9, 1, ((0.0, 10.0, -60.0), 0.0, -10.0), 1.0, {"use_ca": 1}
            cam_arr, _ = frame_camera(camspec)
            fx = po.default_effects(**fxkw)
            rr = po.ref_render(cam_arr, fx, spin, vol, t, w, h, big)
            out["frame_scene"] = np.array([w, h, spin, vol, t], np.float64)
            out["frame_camera"] = cam_arr
            out["frame_rgba8"], out["frame_steps"] = rr["rgba8"], rr["steps"].astype(np.int16)
    np.savez_compressed(os.path.join(HERE, "sky_ref.npz"), **out)
    print("sky_ref.npz:", len(out), "arrays")


def make_camera():
    rc = po.RefCamera()
    rng = np.random.default_rng(7)
    pts = rng.uniform(-80, 80, (64, 4, 3)).astype(np.float32)
    ts = rng.uniform(0, 1, 64).astype(np.float32)
    cr = np.stack([rc.catmull_rom(pts[i, 0], pts[i, 1], pts[i, 2], pts[i, 3], ts[i]) for i in range(64)])
    ab = rng.uniform(-540, 540, (256, 2)).astype(np.float32)
    tt = rng.uniform(0, 1, 256).astype(np.float32)
    la = np.array([rc.lerp_angle(ab[i, 0], ab[i, 1], tt[i]) for i in range(256)], np.float32)
    out = {"cr_pts": pts, "cr_t": ts, "cr_out": cr, "la_ab": ab, "la_t": tt, "la_out": la}
    for idx, (name, keys) in enumerate(rc.paths()):
        out[f"path{idx}_keys"] = keys
        out[f"path{idx}_name"] = np.frombuffer(name.encode(), np.uint8)
    # CameraState of the reference's OWN main.cpp code (CameraController::getCUDAStateFrom :141-167,
    # PathController::getInterpolatedState / start / update :176-212, piped into g++: oracle/Makefile):
    #  - recording frames {1, 75, 150, 225, 300} of each path under the recording clock, with the
    #    controller's own float path time;
    #  - a dense sweep of explicit path times over each path (before the first key, on keys, past the end);
    #  - 256 random (pos, yaw, pitch) through getCUDAStateFrom, and the default free camera (main.cpp:127-130).
    for idx in range(len(rc.paths())):
        for k in (1, 75, 150, 225, 300):
            st, pt = rc.path_state_at_frame(idx, k)
            out[f"path{idx}_frame{k}"] = st
            out[f"path{idx}_frame{k}_time"] = np.float32(pt)
        keys = rc.paths()[idx][1]
        ts = np.concatenate([np.float32([-1.0, 0.0]), keys[:, 0], rng.uniform(0, float(keys[-1, 0]) + 2.0, 96).astype(np.float32)])
        out[f"path{idx}_sweep_t"] = ts.astype(np.float32)
        out[f"path{idx}_sweep_state"] = np.stack([rc.path_state_at(idx, t) for t in ts])
    cpos = rng.uniform(-80, 80, (256, 3)).astype(np.float32)
    cyaw = rng.uniform(-540, 540, 256).astype(np.float32)
    cpitch = rng.uniform(-89, 89, 256).astype(np.float32)
    out["cam_pos"], out["cam_yaw"], out["cam_pitch"] = cpos, cyaw, cpitch
    out["cam_state"] = np.stack([rc.state_from(cpos[i], cyaw[i], cpitch[i]) for i in range(256)])
    out["default_camera"] = rc.default_camera()
    np.savez_compressed(os.path.join(HERE, "camera_ref.npz"), **out)
    print("camera_ref.npz:", len(out), "arrays")


# frame cases: name -> (w, h, spin, volumetrics, camera(pos,yaw,pitch), time, effects overrides)
FRAME_CASES = {
    "G1": (128, 128, 0.0, 1, ((0.0, 10.0, -60.0), 0.0, -10.0), 1.0, {}),
    "G2": (64, 36, 0.9, 0, ((0.0, 10.0, -60.0), 0.0, -10.0), 1.0, {}),
    "G3": (64, 36, 0.9, 1, ((0.0, 10.0, -60.0), 0.0, -10.0), 1.0, {}),
    "G4": (64, 36, 0.99, 1, ((0.0, 10.0, -60.0), 0.0, -10.0), 1.0, {}),
    "G5": (64, 36, 0.9, 1, ((35.0, 0.8, 10.0), -106.0, -1.2), 12.5, {"use_ca": 1}),
}


# reference-kernel frames: G1-G5 plus two camera-path keyframes and one odd-sized random view
REF_FRAME_CASES = dict(FRAME_CASES)
REF_FRAME_CASES.update({
    # path 0 "Gargantua Fly-By" key 1 (camera_paths.cpp:37) and path 2 "Horizon Skimmer" key 2 (:62)
    "K1": (64, 36, 0.9, 1, ((15.0, 3.0, -30.0), -26.6, -5.1), 6.0, {"use_ca": 1}),
    "K2": (64, 36, 0.9, 1, ((4.2, 0.6, 4.2), -90.0, -5.7), 14.0, {"use_ca": 1}),
    "R1": (50, 30, 0.99, 1, ((-22.0, 2.5, 31.0), 143.0, -3.0), 3.25,
           {"use_bloom": 0, "use_vignette": 1, "vignette_intensity": 0.7, "use_ca": 1, "ca_amount": 0.011,
            "use_lens": 0}),
})


def frame_camera(spec):
    import relativisticraytracer_amd as rrt
    a = rrt.CameraState.from_angles(*spec).as_array()
    return a, po.camera(a[0], a[1], a[2], a[3])


def make_frames():
    sky = synthetic_sky()
    out = {}
    for name, (w, h, spin, vol, camspec, t, fxkw) in FRAME_CASES.items():
        cam_arr, cam = frame_camera(camspec)
        out[f"{name}_camera"] = cam_arr
        for mode, mtag in ((po.MATH_LIBM, "libm"), (po.MATH_PORTABLE, "portable")):
            prm = po.default_params(spin=spin, volumetrics=vol, math_mode=mode)
            r = po.render(cam, po.default_effects(**fxkw), prm, t, w, h, sky, want=("rgba8", "ldr", "diag"))
            out[f"{name}_{mtag}_rgba8"] = r["rgba8"]
            if name != "G1":
                out[f"{name}_{mtag}_ldr"] = r["ldr"]
            out[f"{name}_{mtag}_steps"] = r["steps"].astype(np.int16)
            out[f"{name}_{mtag}_hit"] = r["hit"].astype(np.uint8)
            if mtag == "portable" and name != "G1":
                out[f"{name}_{mtag}_pos"] = r["pos"]
                out[f"{name}_{mtag}_vel"] = r["vel"]
                out[f"{name}_{mtag}_rad"] = r["rad"]
        print(name, "done")
    np.savez_compressed(os.path.join(HERE, "frames_oracle.npz"), **out)
    print("frames_oracle.npz:", len(out), "arrays")


def make_frames_ref():
    """Frames from the reference's own kernel body (see the module docstring)."""
    if not po.ref_frames_available():
        po.build(ref=True)
    sky = synthetic_sky()
    out = {}
    for name, (w, h, spin, vol, camspec, t, fxkw) in REF_FRAME_CASES.items():
        cam_arr, _ = frame_camera(camspec)
        fx = po.default_effects(**fxkw)
        r = po.ref_render(cam_arr, fx, spin, vol, t, w, h, sky)
        out[f"{name}_camera"] = cam_arr
        out[f"{name}_scene"] = np.array([w, h, spin, vol, t], np.float64)
        out[f"{name}_fx_flags"] = np.array([fx.use_bloom, fx.use_vignette, fx.use_ca, fx.use_lens], np.int32)
        out[f"{name}_fx_vals"] = np.array([fx.bloom_threshold, fx.bloom_intensity, fx.vignette_intensity,
                                           fx.ca_amount, fx.distortion_amount], np.float32)
        out[f"{name}_rgba8"] = r["rgba8"]
        out[f"{name}_steps"] = r["steps"].astype(np.int16)
        print(name, "reference kernel: mean steps %.1f" % r["steps"].mean())
    np.savez_compressed(os.path.join(HERE, "frames_ref.npz"), **out)
    print("frames_ref.npz:", len(out), "arrays")


# Larger frames for the contraction comparison (round 6): the 64 x 36 fixtures hold too few pixels to say anything about a
# 1e-4 class.  The strict reference frames of these cases live in frames_ref_fma.npz too (frames_ref.npz keeps its set).
FMA_EXTRA_CASES = {
    "B1": (320, 180, 0.9, 1, ((0.0, 10.0, -60.0), 0.0, -10.0), 1.0, {}),                  # the bench view (main.cpp:128-130)
    "B2": (320, 180, 0.9, 0, ((0.0, 10.0, -60.0), 0.0, -10.0), 1.0, {}),                  # ... skybox only: contraction of the geodesic code alone
    "B3": (320, 180, 0.9, 1, ((15.0, 3.0, -30.0), -20.0, -5.0), 3.0, {}),                 # path 0 key 1 (camera_paths.cpp:35)
    "B4": (320, 180, 0.9, 1, ((4.2, 0.6, 4.2), -90.0, -5.7), 14.0, {}),                   # from inside the disk (camera_paths.cpp:62)
}


def make_frames_ref_fma():
    """frames_ref_fma.npz: the reference's own kernel body compiled with floating-point CONTRACTION (oracle/Makefile ref-fma:
    the same piped text and headers under g++ -ffp-contract=fast -mfma) on every REF_FRAME_CASES scene and on four 320 x 180
    views, for which the strict reference frames are stored alongside.  RGBA8 + per-ray step counts.  What a contracting
    compiler does to raymarcher.cu:15-174 itself: the yardstick RRT_ARITH_FMAD's deviation is held against
    (tests/test_oracle_frames.py, tests/test_gpu_tolerance.py)."""
    if not po.ref_frames_fma_available():
        po.build(ref=True)
    sky = synthetic_sky()
    out = {}
    cases = dict(REF_FRAME_CASES)
    cases.update(FMA_EXTRA_CASES)
    for name, (w, h, spin, vol, camspec, t, fxkw) in cases.items():
        cam_arr, _ = frame_camera(camspec)
        fx = po.default_effects(**fxkw)
        out[f"{name}_camera"] = cam_arr
        out[f"{name}_scene"] = np.array([w, h, spin, vol, t], np.float64)
        out[f"{name}_fx_flags"] = np.array([fx.use_bloom, fx.use_vignette, fx.use_ca, fx.use_lens], np.int32)
        out[f"{name}_fx_vals"] = np.array([fx.bloom_threshold, fx.bloom_intensity, fx.vignette_intensity,
                                           fx.ca_amount, fx.distortion_amount], np.float32)
        r = po.ref_render(cam_arr, fx, spin, vol, t, w, h, sky, fma=True)
        out[f"{name}_fma_rgba8"] = r["rgba8"]
        out[f"{name}_fma_steps"] = r["steps"].astype(np.int16)
        if name in FMA_EXTRA_CASES:
            rs = po.ref_render(cam_arr, fx, spin, vol, t, w, h, sky)
            out[f"{name}_rgba8"] = rs["rgba8"]
            out[f"{name}_steps"] = rs["steps"].astype(np.int16)
        print(name, "contracted reference kernel: mean steps %.1f" % r["steps"].mean())
    np.savez_compressed(os.path.join(HERE, "frames_ref_fma.npz"), **out)
    print("frames_ref_fma.npz:", len(out), "arrays")


if __name__ == "__main__":
    po.build(ref=True)
    if len(sys.argv) > 1 and sys.argv[1] == "frames_ref_fma":
        make_frames_ref_fma()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "frames_ref":
        make_frames_ref()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "rk4_chains":
        make_rk4_chains()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "sky":
        make_sky()
        sys.exit(0)
    make_units()
    make_camera()
    make_frames()
    make_frames_ref()
    make_rk4_chains()
    make_sky()
    make_frames_ref_fma()
