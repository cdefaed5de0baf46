/*
 * ref_units.cpp -- C wrappers around the REFERENCE's own device headers.
 *
 * TEST INFRASTRUCTURE ONLY (see rrt_oracle.h).  Built by oracle/Makefile into
 * oracle/_ref/libref_units.so, in this container only (the reference does not
 * travel to the GPU box).  The reference headers are compiled where they lie,
 * -I/root/reference/include, with g++ and the CUDA runtime headers that ship
 * inside this image's triton wheel (triton/backends/nvidia/include) -- those
 * give `float3`, `make_float3`, `__device__`, `__forceinline__` their real
 * definitions; nothing is stubbed.  No reference text is copied here: this
 * file only *calls* the reference's inline functions.
 *
 * raymarch_kernel itself is built by the sibling harness oracle/ref_frames.cpp
 * (frame-level fixtures); src/main.cpp (GLFW/GLAD/CUDA-GL interop) is not
 * buildable here.
 *
 * SPIN_A is a literal macro in the reference (config.h:21, expanded at its use
 * sites geodesics.h:17,41).  To exercise a != 0 from one library the macro is
 * re-pointed at a thread-local variable after config.h has been read; the
 * arithmetic at the use sites is unchanged.
 */
#include <cuda_runtime.h>
#include <math.h>
#include <stdint.h>

#include "config.h"
static thread_local float ref_spin_value = 0.0f;
#undef SPIN_A
#define SPIN_A ref_spin_value

#include "math_utils.h"
#include "densities.h"
#include "geodesics.h"
#include "integrators.h"
#include "camera_effects/post_processing.h"

static inline float3 ld3(const float* a, int i) { return make_float3(a[3 * i], a[3 * i + 1], a[3 * i + 2]); }
static inline void st3(float* a, int i, float3 v) { a[3 * i] = v.x; a[3 * i + 1] = v.y; a[3 * i + 2] = v.z; }

extern "C" {

void ref_hash31(int n, const float* p, float* out) { for (int i = 0; i < n; ++i) out[i] = hash31(ld3(p, i)); }
void ref_noise3d(int n, const float* p, float* out) { for (int i = 0; i < n; ++i) out[i] = noise3D(ld3(p, i)); }
void ref_fbm(int n, const float* p, int oct, float* out) { for (int i = 0; i < n; ++i) out[i] = fbm(ld3(p, i), oct); }

void ref_geodesic_acc(int n, const float* p, const float* v, float spin, float* out) {
    ref_spin_value = spin;
    for (int i = 0; i < n; ++i) st3(out, i, getGeodesicAcc(ld3(p, i), ld3(v, i)));
}
void ref_rk4(int n, float* p, float* v, const float* h, float spin) {
    ref_spin_value = spin;
    for (int i = 0; i < n; ++i) {
        float3 pp = ld3(p, i), vv = ld3(v, i);
        integrate_rk4(pp, vv, h[i]);
        st3(p, i, pp); st3(v, i, vv);
    }
}
void ref_redshift(int n, const float* p, const float* vel, float spin, float* out) {
    ref_spin_value = spin;
    for (int i = 0; i < n; ++i) out[i] = calculateRedshiftFactor(ld3(p, i), ld3(vel, i));
}
void ref_disk_temperature(int n, const float* r, float* out) { for (int i = 0; i < n; ++i) out[i] = getDiskTemperature(r[i]); }
void ref_accretion_density(int n, const float* p, float time, float* out) {
    for (int i = 0; i < n; ++i) out[i] = getAccretionDensity(ld3(p, i), time);
}
void ref_dust_density(int n, const float* p, float time, float* out) {
    for (int i = 0; i < n; ++i) out[i] = getDustCloudDensity(ld3(p, i), time);
}
void ref_smoothstep(int n, const float* e0, const float* e1, const float* x, float* out) {
    for (int i = 0; i < n; ++i) out[i] = smoothstep(e0[i], e1[i], x[i]);
}
void ref_lens(int n, const float* uv, float k, float* out) {
    for (int i = 0; i < n; ++i) {
        float2 r = apply_lens_distortion(make_float2(uv[2 * i], uv[2 * i + 1]), k);
        out[2 * i] = r.x; out[2 * i + 1] = r.y;
    }
}
void ref_vignette(int n, const float* rgb, const float* uv, float intensity, float* out) {
    for (int i = 0; i < n; ++i) st3(out, i, apply_vignette(ld3(rgb, i), make_float2(uv[2 * i], uv[2 * i + 1]), intensity));
}
void ref_bloom(int n, const float* rgb, float threshold, float* out) {
    for (int i = 0; i < n; ++i) st3(out, i, get_bloom_contribution(ld3(rgb, i), threshold));
}

/* config.h constants as the reference's compiler folds them (for the defaults test) */
void ref_constants(float* out) {
    out[0] = EVENT_HORIZON; out[1] = ISCO_RADIUS; out[2] = DISK_OUT_M; out[3] = DISK_H_M;
    out[4] = DISK_LUMINOSITY; out[5] = DISK_OPACITY; out[6] = EXPOSURE; out[7] = CLOUD_H_M;
    out[8] = CLOUD_OUT_M; out[9] = CLOUD_OPACITY; out[10] = CLOUD_LUMINOSITY; out[11] = STEP_SIZE_M;
    out[12] = (float)MAX_STEPS; out[13] = DISK_TEMP_REF; out[14] = PI;
    out[15] = (float)WINDOW_WIDTH; out[16] = (float)WINDOW_HEIGHT; out[17] = (float)RECORDING_FPS;
}

}  // extern "C"
