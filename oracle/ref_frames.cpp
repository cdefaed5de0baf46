/*
 * ref_frames.cpp -- per-pixel driver around the REFERENCE's own raymarch_kernel
 * (/root/reference/src/raymarcher.cu:15-174), for oracle/_ref/libref_frames.so.
 *
 * TEST INFRASTRUCTURE ONLY, build container only (see ref_frames_pre.h for how the kernel body is
 * compiled and exactly which four things the harness supplies).  The library renders the golden
 * frames of tests/golden/frames_ref.npz (tests/golden/make_golden.py); the restatement in
 * oracle/rrt_oracle.c must reproduce them byte for byte (tests/test_oracle_frames.py).
 */
#include "ref_frames_pre.h"
#undef integrate_rk4
#undef getAccretionDensity
#undef getDustCloudDensity

#include <stdint.h>
#include <omp.h>

#include "rrt_oracle.h"

/* the kernel, compiled from the reference's text in its own object */
void raymarch_kernel(uchar4* output, int width, int height, float time, CameraState cam,
                     cudaTextureObject_t skyboxTex, CameraEffects effects);

extern "C" {
thread_local uint3 threadIdx = {0, 0, 0};
thread_local uint3 blockIdx = {0, 0, 0};
thread_local dim3 blockDim(1, 1, 1);
thread_local dim3 gridDim(1, 1, 1);
thread_local int warpSize = 32;
}
thread_local float ref_spin_value = 0.0f;
thread_local int ref_step_count = 0;
thread_local int ref_volumetrics = 1;

namespace {
struct SkyImage { const uint8_t* texels; int w, h, frac_bits; };
}

/* tex2D<float4> on the texture object main.cpp:255-261 sets up (wrap x, clamp y, linear filter,
 * normalized coordinates, normalized-float reads): the build's definition of the hardware filter */
template <> float4 tex2D<float4>(cudaTextureObject_t tex, float x, float y) {
    const SkyImage* s = reinterpret_cast<const SkyImage*>(static_cast<uintptr_t>(tex));
    float o[4];
    rrto_sky_fetch(s->texels, s->w, s->h, s->frac_bits, x, y, o);
    return make_float4(o[0], o[1], o[2], o[3]);
}

/* every `sy`-th row and `sx`-th column of the frame (the others are left untouched): bench.py's CPU-baseline sample */
extern "C" int ref_render_strided(const float* cam12, const int32_t* fx_flags4, const float* fx_vals5, float spin,
                                  int volumetrics, float time, int width, int height, const uint8_t* sky, int sw, int sh,
                                  int frac_bits, uint8_t* rgba8, int32_t* steps, int n_threads, int sx, int sy) {
    if (!cam12 || !fx_flags4 || !fx_vals5 || !sky || !rgba8 || width <= 0 || height <= 0 || sx <= 0 || sy <= 0) return -1;
    CameraState cam;
    cam.pos = make_float3(cam12[0], cam12[1], cam12[2]);
    cam.forward = make_float3(cam12[3], cam12[4], cam12[5]);
    cam.right = make_float3(cam12[6], cam12[7], cam12[8]);
    cam.up = make_float3(cam12[9], cam12[10], cam12[11]);
    CameraEffects fx;
    fx.useBloom = fx_flags4[0] != 0; fx.useVignette = fx_flags4[1] != 0;
    fx.useChromaticAberration = fx_flags4[2] != 0; fx.useLensDistortion = fx_flags4[3] != 0;
    fx.bloomThreshold = fx_vals5[0]; fx.bloomIntensity = fx_vals5[1]; fx.vignetteIntensity = fx_vals5[2];
    fx.caAmount = fx_vals5[3]; fx.distortionAmount = fx_vals5[4];
    SkyImage img = {sky, sw, sh, frac_bits};
    const cudaTextureObject_t tex = static_cast<cudaTextureObject_t>(reinterpret_cast<uintptr_t>(&img));
    if (n_threads <= 0) n_threads = omp_get_max_threads();
    const int rows = (height + sy - 1) / sy;
    /* tasks = (row, run of kTaskSamples samples), dynamic: the same granularity as the restatement's loop
     * (oracle/rrt_oracle.c: rrto_render), so that bench.py's two CPU legs are scheduled alike */
    const int kTaskSamples = 128;
    const int nxs = (width + sx - 1) / sx, nbx = (nxs + kTaskSamples - 1) / kTaskSamples;
#pragma omp parallel for collapse(2) schedule(dynamic, 1) num_threads(n_threads)
    for (int j = 0; j < rows; ++j) {
        for (int bx = 0; bx < nbx; ++bx) {
            const int y = j * sy;
            ref_spin_value = spin;                  /* thread-local harness state: set in every task */
            ref_volumetrics = volumetrics;
            blockDim = dim3(1, 1, 1);
            threadIdx = make_uint3(0, 0, 0);
            const int xa = bx * kTaskSamples * sx;
            const int xb = xa + kTaskSamples * sx < width ? xa + kTaskSamples * sx : width;
            for (int x = xa; x < xb; x += sx) {
                blockIdx = make_uint3((unsigned)x, (unsigned)y, 0);
                ref_step_count = 0;
                raymarch_kernel(reinterpret_cast<uchar4*>(rgba8), width, height, time, cam, tex, fx);
                if (steps) steps[(size_t)y * width + x] = ref_step_count;
            }
        }
    }
    return 0;
}

extern "C" int ref_render(const float* cam12, const int32_t* fx_flags4, const float* fx_vals5, float spin,
                          int volumetrics, float time, int width, int height, const uint8_t* sky, int sw, int sh,
                          int frac_bits, uint8_t* rgba8, int32_t* steps, int n_threads) {
    return ref_render_strided(cam12, fx_flags4, fx_vals5, spin, volumetrics, time, width, height, sky, sw, sh, frac_bits,
                              rgba8, steps, n_threads, 1, 1);
}
