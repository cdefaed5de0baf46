/*
 * rrt_device.h -- gfx950 device functions of the geodesic ray-march path.
 *
 * Hand-written for CDNA4; not a translation of the reference's CUDA headers.
 * Every function cites the reference lines whose RESULT it must reproduce.
 * Arithmetic contract (what makes the output bit-identical to the CPU oracle):
 *   - IEEE binary32 only, every source-level operation rounded once, in the
 *     reference's association order; the translation unit is compiled with
 *     -ffp-contract=off, so the only fused operations are the explicit fmaf()
 *     calls inside rrt_math.h (the portable transcendentals);
 *   - `/` and sqrtf are correctly rounded (hipcc default
 *     -fhip-fp32-correctly-rounded-divide-sqrt);
 *   - algebraic shortcuts are taken only where they are exact in binary32
 *     (x - 0 == x, 1*x == x, (0,1,0) x p == (p.z, 0, -p.x), fmod(x,1) ==
 *     x - trunc(x) up to the sign of a zero result, which no consumer observes).
 */
#ifndef RRT_DEVICE_H
#define RRT_DEVICE_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rrt_math.h"

#define RRT_DEV __device__ __forceinline__

namespace rrt {

/* ---- scene constants, reference include/config.h:18-48 (values verbatim) ---- */
constexpr float kDiskTempRef = 1.5e7f;      /* DISK_TEMP_REF    :18 */
constexpr float kEventHorizon = 2.0f;       /* EVENT_HORIZON    :29 */
constexpr float kIsco = 10.0f;              /* ISCO_RADIUS      :33 */
constexpr float kDiskOut = 25.0f;           /* DISK_OUT_M       :34 */
constexpr float kDiskH = 0.8f;              /* DISK_H_M         :35 */
constexpr float kDiskLum = 6.0f;            /* DISK_LUMINOSITY  :36 */
constexpr float kDiskOpacity = 0.4f;        /* DISK_OPACITY     :37 */
constexpr float kExposure = 0.8f;           /* EXPOSURE         :38 */
constexpr float kCloudH = 0.5f;             /* CLOUD_H_M        :41 */
constexpr float kCloudOut = 25.0f;          /* CLOUD_OUT_M      :42 */
constexpr float kCloudOpacity = 0.3f;       /* CLOUD_OPACITY    :43 */
constexpr float kCloudLum = 0.4f;           /* CLOUD_LUMINOSITY :44 */
constexpr float kStepSize = 0.3f;           /* STEP_SIZE_M      :47 */
constexpr float kPi = 3.1415926535f;        /* PI, math_utils.h:7 */

struct v3 { float x, y, z; };

RRT_DEV v3 mk(float x, float y, float z) { v3 r; r.x = x; r.y = y; r.z = z; return r; }
RRT_DEV float dot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }          /* math_utils.h:11 */
RRT_DEV v3 cross(v3 a, v3 b) {                                                       /* math_utils.h:15 */
    return mk(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
RRT_DEV float length(v3 v) { return sqrtf(v.x * v.x + v.y * v.y + v.z * v.z); }      /* :19 */
RRT_DEV v3 normalize(v3 v) {                                                         /* :23-27 */
    float mag = length(v);
    if (mag < 1e-6f) return mk(0.f, 0.f, 0.f);
    return mk(v.x / mag, v.y / mag, v.z / mag);
}
RRT_DEV v3 add(v3 a, v3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
RRT_DEV v3 mul(v3 v, float s) { return mk(v.x * s, v.y * s, v.z * s); }
RRT_DEV float lerp(float a, float b, float t) { return a + t * (b - a); }            /* :41 */
RRT_DEV float fmin2(float a, float b) { return __builtin_fminf(a, b); }
RRT_DEV float fmax2(float a, float b) { return __builtin_fmaxf(a, b); }
RRT_DEV float smoothstep(float e0, float e1, float x) {                              /* :45-48 */
    float t = fmin2(fmax2((x - e0) / (e1 - e0), 0.0f), 1.0f);
    return t * t * (3.0f - 2.0f * t);
}

/* fmodf(x, 1.0f) for finite x: exact, differs from libm only in the sign of a zero. */
RRT_DEV float fmod1(float x) { return x - __builtin_truncf(x); }

/* hash31, math_utils.h:91-96 -- arithmetic hash, must stay unfused. */
RRT_DEV float hash31(float px, float py, float pz) {
    float x = fmod1(px * 0.1031f), y = fmod1(py * 0.1031f), z = fmod1(pz * 0.1031f);
    float d = x * (y + 33.33f) + y * (z + 33.33f) + z * (x + 33.33f);
    x += d; y += d; z += d;
    return fmod1((x + y) * z);
}

/* noise3D, math_utils.h:98-110.  i + 0 == i and i + 1 are exact (|i| < 2^24). */
RRT_DEV float noise3d(v3 p) {
    float ix = floorf(p.x), iy = floorf(p.y), iz = floorf(p.z);
    float fx = p.x - ix, fy = p.y - iy, fz = p.z - iz;
    float ux = fx * fx * (3.0f - 2.0f * fx);
    float uy = fy * fy * (3.0f - 2.0f * fy);
    float uz = fz * fz * (3.0f - 2.0f * fz);
    /* "+ 0.0f" reproduces add(i, (0,0,0)): it turns a -0 lattice coordinate into +0 */
    float x0 = ix + 0.0f, y0 = iy + 0.0f, z0 = iz + 0.0f;
    float x1 = ix + 1.0f, y1 = iy + 1.0f, z1 = iz + 1.0f;
    float a = lerp(hash31(x0, y0, z0), hash31(x1, y0, z0), ux);
    float b = lerp(hash31(x0, y1, z0), hash31(x1, y1, z0), ux);
    float c = lerp(hash31(x0, y0, z1), hash31(x1, y0, z1), ux);
    float d = lerp(hash31(x0, y1, z1), hash31(x1, y1, z1), ux);
    return lerp(lerp(a, b, uy), lerp(c, d, uy), uz);
}

/* fbm, math_utils.h:112-121 */
template <int OCT>
RRT_DEV float fbm(v3 p) {
    float v = 0.0f, a = 0.5f;
#pragma unroll 1
    for (int i = 0; i < OCT; ++i) {
        v += a * noise3d(p);
        p = mk(p.x * 2.05f + 10.0f, p.y * 2.05f + 10.0f, p.z * 2.05f + 10.0f);
        a *= 0.5f;
    }
    return v;
}

/*
 * Correctly rounded sqrt and divide for the march loop.
 *
 * hipcc's own expansions of sqrtf and `/` are correctly rounded too, but carry range
 * scaling (v_div_scale / v_div_fixup / denormal rescue) that this loop never needs and
 * that costs ~19-21 issue slots each on gfx950 (profiles/r01_valu_microbench.txt).  The
 * operands here are tame (r in [1, 1.6e4], quotients normal or exactly zero -- a denormal
 * quotient would need a ray parallel to its position vector to within 1e-13 rad), so the
 * bare Newton/Markstein cores are enough; both are checked bit-for-bit against the hardware-IEEE forms over the whole
 * operand range on the GPU (tests/test_gpu_units.py::test_fast_sqrt_div_*).
 *
 *   sqrt_rsq:  y0 = v_rsq_f32(x) (1 ulp); one coupled Goldschmidt step for g ~ sqrt(x),
 *              h ~ 1/(2 sqrt(x)); final residual correction g + (x - g*g)*h.
 *   div_seeded: the FMA core of LLVM's f32 fdiv (refine the reciprocal once, then three
 *              residual corrections), started from a seed accurate to ~2^-20.
 */
RRT_DEV void sqrt_rsq(float x, float& root, float& inv_root) {
    float y0 = __builtin_amdgcn_rsqf(x);
    float g = x * y0;
    float h = 0.5f * y0;
    float r = __builtin_fmaf(-h, g, 0.5f);
    g = __builtin_fmaf(g, r, g);
    h = __builtin_fmaf(h, r, h);
    float d = __builtin_fmaf(-g, g, x);
    root = __builtin_fmaf(d, h, g);
    inv_root = h + h;
}

RRT_DEV float div_seeded(float a, float b, float seed) {
    float e = __builtin_fmaf(-b, seed, 1.0f);
    float y = __builtin_fmaf(e, seed, seed);
    float q = a * y;
    float r = __builtin_fmaf(-b, q, a);
    q = __builtin_fmaf(r, y, q);
    r = __builtin_fmaf(-b, q, a);
    return __builtin_fmaf(r, y, q);
}

/*
 * getGeodesicAcc, geodesics.h:30-45, with SPIN_AXIS = (0,1,0), EVENT_HORIZON = 2:
 *   radial = (-1.5f*2.0f * L2 / (r2*r2*r)) * p        (-1.5f*2.0f folds to -3.0f)
 *   drag   = ((2.0f*a*2.0f) / (r2*r)) * (p.z, 0, -p.x)
 * drag_c = (2.0f*a)*2.0f is computed once on the host (exact scaling).
 * r2, r and y ~ 1/r come from the caller (the march loop already has them for the
 * stage-1 point); both reciprocal seeds are powers of y.
 */
template <bool SPIN>
RRT_DEV v3 geodesic_acc_r(v3 p, v3 v, float drag_c, float r2, float r, float y) {
    v3 L = cross(p, v);
    float L2 = dot(L, L);
    float y2 = y * y;
    float y3 = y2 * y;
    float d2 = r2 * r;
    float d1 = (r2 * r2) * r;
    float radial_mag = div_seeded(-3.0f * L2, d1, y3 * y2);
    v3 acc = mul(p, radial_mag);
    if (SPIN) {
        float ds = div_seeded(drag_c, d2, y3);
        acc.x = acc.x + p.z * ds;
        acc.z = acc.z + (-p.x) * ds;
    }
    if (__builtin_expect(__any(r2 < 1.0f), 0)) {       /* geodesics.h:33: sqrt(r2) < 1  <=>  r2 < 1 */
        if (r2 < 1.0f) acc = mk(0.f, 0.f, 0.f);
    }
    return acc;
}

template <bool SPIN>
RRT_DEV v3 geodesic_acc(v3 p, v3 v, float drag_c) {
    float r2 = dot(p, p);
    float r, y;
    sqrt_rsq(r2, r, y);
    if (__builtin_expect(__any(!(r2 >= 1.0f)), 0)) {   /* r2 == 0 / tiny: keep the seeds finite */
        if (!(r2 >= 1.0f)) { r = sqrtf(r2); y = 1.0f; }
    }
    return geodesic_acc_r<SPIN>(p, v, drag_c, r2, r, y);
}

/*
 * integrate_rk4, integrators.h:23-59 (MASS_POS = 0: the sub() calls are identities).
 * `hh` = h*0.5f and `h6` = h/6.0f are passed in (h takes three values in the march).
 * 2*a + b is evaluated as fma(2, a, b): identical bits, because 2*a is exact.
 */
template <bool SPIN>
RRT_DEV void integrate_rk4_r(v3& p, v3& v, float h, float hh, float h6, float drag_c,
                             float r2, float r, float y) {
    v3 p0 = p, v0 = v;
    v3 kv1 = geodesic_acc_r<SPIN>(p0, v0, drag_c, r2, r, y);
    v3 v2 = add(v0, mul(kv1, hh));
    v3 p2 = add(p0, mul(v0, hh));
    v3 kv2 = geodesic_acc<SPIN>(p2, v2, drag_c);
    v3 v3_ = add(v0, mul(kv2, hh));
    v3 p3 = add(p0, mul(v2, hh));
    v3 kv3 = geodesic_acc<SPIN>(p3, v3_, drag_c);
    v3 v4 = add(v0, mul(kv3, h));
    v3 p4 = add(p0, mul(v3_, h));
    v3 kv4 = geodesic_acc<SPIN>(p4, v4, drag_c);
    v3 kv_sum, kp_sum;
    kv_sum.x = kv1.x + __builtin_fmaf(2.0f, kv2.x, __builtin_fmaf(2.0f, kv3.x, kv4.x));
    kv_sum.y = kv1.y + __builtin_fmaf(2.0f, kv2.y, __builtin_fmaf(2.0f, kv3.y, kv4.y));
    kv_sum.z = kv1.z + __builtin_fmaf(2.0f, kv2.z, __builtin_fmaf(2.0f, kv3.z, kv4.z));
    kp_sum.x = v0.x + __builtin_fmaf(2.0f, v2.x, __builtin_fmaf(2.0f, v3_.x, v4.x));
    kp_sum.y = v0.y + __builtin_fmaf(2.0f, v2.y, __builtin_fmaf(2.0f, v3_.y, v4.y));
    kp_sum.z = v0.z + __builtin_fmaf(2.0f, v2.z, __builtin_fmaf(2.0f, v3_.z, v4.z));
    v = add(v0, mul(kv_sum, h6));
    p = add(p0, mul(kp_sum, h6));
}

template <bool SPIN>
RRT_DEV void integrate_rk4(v3& p, v3& v, float h, float drag_c) {
    float r2 = dot(p, p);
    float r, y;
    sqrt_rsq(r2, r, y);
    if (__builtin_expect(__any(!(r2 >= 1.0f)), 0)) {
        if (!(r2 >= 1.0f)) { r = sqrtf(r2); y = 1.0f; }
    }
    integrate_rk4_r<SPIN>(p, v, h, h * 0.5f, h / 6.0f, drag_c, r2, r, y);
}

/*
 * FAST arithmetic mode (rrt_params.arith_mode = RRT_ARITH_FAST; NOT the parity path).
 * Same equations (geodesics.h:30-45, integrators.h:23-59) evaluated the way a GPU compiler with
 * contraction would: fused multiply-adds, 1/r from v_rsq_f32 (1 ulp) instead of correctly rounded
 * sqrt and divides.  Every step is perturbed at the 1e-7 relative level, so frames differ from the
 * strict path / the oracle by rounding noise that near-critical rays amplify; the measured
 * deviation is reported by tests/test_gpu_frames.py::test_fast_mode_* and DESIGN.md.
 */
template <bool SPIN>
RRT_DEV v3 geodesic_acc_fast(v3 p, v3 v, float drag_c, float r2, float y) {
    v3 L = mk(__builtin_fmaf(p.y, v.z, -(p.z * v.y)), __builtin_fmaf(p.z, v.x, -(p.x * v.z)),
              __builtin_fmaf(p.x, v.y, -(p.y * v.x)));
    float L2 = __builtin_fmaf(L.z, L.z, __builtin_fmaf(L.y, L.y, L.x * L.x));
    float y2 = y * y;
    float y3 = y2 * y;
    float mag = (-3.0f * L2) * (y3 * y2);
    v3 acc = mul(p, mag);
    if (SPIN) {
        float ds = drag_c * y3;
        acc.x = __builtin_fmaf(p.z, ds, acc.x);
        acc.z = __builtin_fmaf(-p.x, ds, acc.z);
    }
    if (__builtin_expect(__any(r2 < 1.0f), 0)) {
        if (r2 < 1.0f) acc = mk(0.f, 0.f, 0.f);
    }
    return acc;
}

RRT_DEV float dot_fma(v3 a, v3 b) { return __builtin_fmaf(a.z, b.z, __builtin_fmaf(a.y, b.y, a.x * b.x)); }
RRT_DEV v3 axpy(v3 x, float a, v3 y) {       /* a*x + y, fused */
    return mk(__builtin_fmaf(x.x, a, y.x), __builtin_fmaf(x.y, a, y.y), __builtin_fmaf(x.z, a, y.z));
}

template <bool SPIN>
RRT_DEV void integrate_rk4_fast(v3& p, v3& v, float h, float hh, float h6, float drag_c, float r2, float y) {
    v3 p0 = p, v0 = v;
    v3 kv1 = geodesic_acc_fast<SPIN>(p0, v0, drag_c, r2, y);
    v3 v2 = axpy(kv1, hh, v0);
    v3 p2 = axpy(v0, hh, p0);
    float r2b = dot_fma(p2, p2);
    v3 kv2 = geodesic_acc_fast<SPIN>(p2, v2, drag_c, r2b, __builtin_amdgcn_rsqf(r2b));
    v3 v3_ = axpy(kv2, hh, v0);
    v3 p3 = axpy(v2, hh, p0);
    float r2c = dot_fma(p3, p3);
    v3 kv3 = geodesic_acc_fast<SPIN>(p3, v3_, drag_c, r2c, __builtin_amdgcn_rsqf(r2c));
    v3 v4 = axpy(kv3, h, v0);
    v3 p4 = axpy(v3_, h, p0);
    float r2d = dot_fma(p4, p4);
    v3 kv4 = geodesic_acc_fast<SPIN>(p4, v4, drag_c, r2d, __builtin_amdgcn_rsqf(r2d));
    v3 kv_sum = add(kv1, axpy(kv2, 2.0f, axpy(kv3, 2.0f, kv4)));
    v3 kp_sum = add(v0, axpy(v2, 2.0f, axpy(v3_, 2.0f, v4)));
    v = axpy(kv_sum, h6, v0);
    p = axpy(kp_sum, h6, p0);
}

/* calculateRedshiftFactor, geodesics.h:11-25 */
RRT_DEV float redshift_factor(v3 p, v3 ray_vel, float spin) {
    float r = length(p);
    if (r < kEventHorizon * 1.01f) return 0.0f;
    float g_gravity = sqrtf(1.0f - kEventHorizon / r);
    float v_mag = 1.0f / (rrt_powf(r, 1.5f) + spin);
    v3 gas_dir = normalize(mk(-p.z, 0.f, p.x));
    float cos_theta = dot(ray_vel, gas_dir);
    float gamma = 1.0f / sqrtf(1.0f - v_mag * v_mag);
    float g_doppler = 1.0f / (gamma * (1.0f - v_mag * cos_theta));
    return g_gravity * g_doppler;
}

/* getDiskTemperature, densities.h:12-15 */
RRT_DEV float disk_temperature(float r) {
    if (r < kIsco) return 0.0f;
    return kDiskTempRef * rrt_powf(r / kIsco, -0.75f);
}

/* getAccretionDensity, densities.h:20-62.  EARLY_OUT=false is the literal function
 * (unit tests); the render kernels use EARLY_OUT=true (see below). */
template <bool EARLY_OUT>
RRT_DEV float accretion_density(v3 p, float time) {
    float r = sqrtf(p.x * p.x + 0.0f * 0.0f + p.z * p.z);
    if (r < kIsco || r > kDiskOut) return 0.0f;

    float edge_falloff = 1.0f;
    const float edge_start = kDiskOut * 0.85f;
    if (r > edge_start) {
        edge_falloff = 1.0f - (r - edge_start) / (kDiskOut - edge_start);
        edge_falloff *= edge_falloff;
    }
    float q = kIsco / r;
    float local_h = kDiskH * rrt_powf(q, 0.5f);
    float vertical_density = rrt_expf(-(p.y * p.y) / (2.0f * local_h * local_h + 1e-7f));
    float radial_density = rrt_powf(q, 0.4f);
    float base_envelope = vertical_density * radial_density * edge_falloff;

    /*
     * Exact early-out (not in the reference): cloud <= 6 (densities.h:59), so the
     * result is <= base_envelope * (0.02f + 5.0f*6.0f); rounding is monotone, so
     * if that bound is <= 0.001f the caller's `d_disk > 0.001f` test
     * (raymarcher.cu:71,76) fails and the value is never used.
     */
    if (EARLY_OUT && base_envelope * (0.02f + 5.0f * 6.0f) <= 0.001f) return 0.0f;

    float phi = rrt_atan2f(p.z, p.x);
    float omega = 3.5f * rrt_powf(q, 1.5f);
    float angle_rotated = phi - time * omega;
    float sn, cs;
    rrt_sincosf(angle_rotated, &sn, &cs);
    v3 rot_p = mk(r * cs, p.y * 4.0f, r * sn);
    float evolution = time * 0.35f;
    v3 nc = mk(rot_p.x * 0.45f + 0.0f, rot_p.y * 0.45f + evolution, rot_p.z * 0.45f + 0.0f);

    float n = fbm<5>(nc);
    float cloud = fmax2(0.0f, n - 0.32f);
    cloud = rrt_powf(cloud * 2.8f, 1.6f);
    cloud = fmin2(6.0f, cloud);
    return base_envelope * (0.02f + 5.0f * cloud);
}

/* getDustCloudDensity, densities.h:69-132 */
RRT_DEV float dust_density(v3 p, float time) {
    float r = sqrtf(p.x * p.x + 0.0f * 0.0f + p.z * p.z);
    if (r < kIsco || r > kDiskOut) return 0.0f;

    float edge_falloff = smoothstep(kDiskOut, kDiskOut * 0.8f, r);
    float inner_taper = smoothstep(kIsco, kIsco + 5.0f, r);
    float q = kIsco / r;
    float local_h = kCloudH * 0.5f * rrt_powf(q, 0.2f);
    float vertical_profile = rrt_expf(-(p.y * p.y) / (2.0f * local_h * local_h + 1e-7f));
    float base = vertical_profile * edge_falloff * inner_taper;
    if (base < 0.001f) return 0.0f;

    float phi = rrt_atan2f(p.z, p.x);
    float omega = rrt_powf(q, 1.5f);                   /* 1.0f * pow(...) */
    float angle_rot = phi - time * omega;

    v3 coords = mk(r * 0.8f, p.y * 15.0f, angle_rot * 10.0f);
    v3 c15 = mul(coords, 0.15f);
    v3 w1 = mk(fbm<2>(c15),
               fbm<2>(mk(c15.x + 1.0f, c15.y + 2.0f, c15.z + 3.0f)),
               fbm<2>(mk(c15.x + 4.0f, c15.y + 5.0f, c15.z + 6.0f)));
    v3 w2c = add(coords, mul(w1, 3.0f));
    v3 c40 = mul(w2c, 0.4f);
    v3 w2 = mk(fbm<2>(c40),
               fbm<2>(mk(c40.x + 2.0f, c40.y + 1.0f, c40.z + 0.0f)),
               fbm<2>(mk(c40.x + 0.0f, c40.y + 3.0f, c40.z + 1.0f)));
    v3 fc = add(coords, mul(w2, 1.5f));

    float n = 0.0f, amp = 1.0f, freq = 1.0f;
#pragma unroll 1
    for (int i = 0; i < 5; ++i) {
        float nv = noise3d(mul(fc, freq));
        float wisp = 1.0f - fabsf(nv * 2.0f - 1.0f);
        n += wisp * amp;
        amp *= 0.5f;
        freq *= 2.1f;
    }
    float strands = smoothstep(0.4f, 0.8f, n * 0.55f);
    strands = rrt_powf(strands, 4.0f);
    v3 dc = mul(fc, 4.0f);
    float detail = fbm<2>(mk(dc.x + 0.0f, dc.y + time * 0.5f, dc.z + 0.0f));
    strands *= (0.6f + 0.4f * detail);
    return base * strands * 12.0f;
}

/* ---- radiative transfer of one in-zone sample, raymarcher.cu:67-117 ---- */
struct Radiance { float r, g, b, t; };

/* Emission (ex, ey, ez) and transmittance of one sample; false when the sample contributes nothing
 * (raymarcher.cu:71 not taken). */
RRT_DEV bool sample_emission(float d_disk, float d_cloud, v3 rel_p, float r, v3 vel, float h, float spin,
                             float& ex, float& ey, float& ez, float& step_trans) {
    if (!(d_disk > 0.001f || d_cloud > 0.001f)) return false;
    ex = 0.f; ey = 0.f; ez = 0.f;
    float opacity = 0.f;
    if (d_disk > 0.001f) {
        float g = redshift_factor(rel_p, vel, spin);
        float T = disk_temperature(r);
        float tn = T / kDiskTempRef;
        float T_norm = rrt_powf(tn, 0.5f);
        float bol_I = rrt_powf(g, 4.0f) * T_norm * d_disk * kDiskLum;
        float color_t = g * rrt_powf(tn, 0.4f) * 2.5f;
        ex += bol_I;                                             /* 1.0f * bol_I */
        ey += fmin2(0.25f, 0.12f * color_t) * bol_I;
        ez += fmax2(0.0f, 0.01f * (color_t - 2.0f)) * bol_I;
        opacity += d_disk * kDiskOpacity;
    }
    if (d_cloud > 0.001f) {
        float g = redshift_factor(rel_p, vel, spin);
        float lighting = 0.5f + 3.0f * rrt_powf(kIsco / fmax2(r, kIsco), 1.2f);
        float cloud_I = d_cloud * kCloudLum * lighting;
        float shift = smoothstep(0.7f, 1.3f, g);
        ex += 0.60f * cloud_I * lerp(1.2f, 0.8f, shift);
        ey += 0.65f * cloud_I * lerp(0.8f, 1.1f, shift);
        ez += 0.80f * cloud_I * lerp(0.6f, 1.4f, shift);
        opacity += d_cloud * kCloudOpacity;
    }
    float d_tau = opacity * h;
    step_trans = rrt_expf(-d_tau);
    return true;
}

/* Beer-Lambert accumulation of one sample, raymarcher.cu:109-115 */
RRT_DEV void accumulate_emission(Radiance& acc, float ex, float ey, float ez, float step_trans) {
    float factor = (1.0f - step_trans) * acc.t;
    acc.r += ex * factor;
    acc.g += ey * factor;
    acc.b += ez * factor;
    acc.t *= step_trans;
}

RRT_DEV void accumulate_sample(Radiance& acc, float d_disk, float d_cloud, v3 rel_p, float r, v3 vel,
                               float h, float spin) {
    float ex, ey, ez, s;
    if (!sample_emission(d_disk, d_cloud, rel_p, r, vel, h, spin, ex, ey, ez, s)) return;
    accumulate_emission(acc, ex, ey, ez, s);
}

/* ---- sky lookup (replaces tex2D<float4>, raymarcher.cu:134-146; the filter is the
 *      build's definition, identical to oracle/rrt_oracle.c:sky_fetch) ---- */
struct SkyTex { const uint8_t* texels; int w, h, frac_bits; };

RRT_DEV int wrapi(int i, int n) { int m = i % n; return m < 0 ? m + n : m; }
RRT_DEV int clampi(int i, int lo, int hi) { return i < lo ? lo : (i > hi ? hi : i); }

RRT_DEV void sky_fetch(const SkyTex& s, float tx, float ty, float out[4]) {
    float xb = tx * (float)s.w - 0.5f;
    float yb = ty * (float)s.h - 0.5f;
    float fi = floorf(xb), fj = floorf(yb);
    float a = xb - fi, b = yb - fj;
    if (s.frac_bits > 0) {
        float q = (float)(1 << s.frac_bits);
        a = floorf(a * q + 0.5f) / q;
        b = floorf(b * q + 0.5f) / q;
    }
    int i0 = wrapi((int)fi, s.w), i1 = wrapi((int)fi + 1, s.w);
    int j0 = clampi((int)fj, 0, s.h - 1), j1 = clampi((int)fj + 1, 0, s.h - 1);
    float w00 = (1.0f - a) * (1.0f - b);
    float w10 = a * (1.0f - b);
    float w01 = (1.0f - a) * b;
    float w11 = a * b;
    const uint32_t* t = reinterpret_cast<const uint32_t*>(s.texels);
    uint32_t t00 = t[(size_t)j0 * s.w + i0], t10 = t[(size_t)j0 * s.w + i1];
    uint32_t t01 = t[(size_t)j1 * s.w + i0], t11 = t[(size_t)j1 * s.w + i1];
#pragma unroll
    for (int ch = 0; ch < 4; ++ch) {
        float c00 = (float)((t00 >> (8 * ch)) & 255u) / 255.0f;
        float c10 = (float)((t10 >> (8 * ch)) & 255u) / 255.0f;
        float c01 = (float)((t01 >> (8 * ch)) & 255u) / 255.0f;
        float c11 = (float)((t11 >> (8 * ch)) & 255u) / 255.0f;
        out[ch] = w00 * c00 + w10 * c10 + w01 * c01 + w11 * c11;
    }
}

RRT_DEV void sample_sky(const SkyTex& s, v3 dir, float off, float out[4]) {
    float phi = rrt_atan2f(dir.z, dir.x) + off;
    float theta = rrt_asinf(dir.y);
    float tx = 0.5f + phi / (2.0f * kPi);
    float ty = 0.5f - theta / kPi;
    sky_fetch(s, tx, ty, out);
}

}  // namespace rrt
#endif
