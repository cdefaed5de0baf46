/*
 * raymarcher.h -- source-compatible replacement for the reference's
 * include/raymarcher.h (21 lines: struct CameraState :11-16 and the
 * launch_raymarch prototype :19).
 *
 * A translation unit that includes this header instead of the reference's, and
 * links librrt_hip.so instead of compiling src/raymarcher.cu, keeps calling
 *
 *     launch_raymarch(d_out, w, h, simTime, camState, skyboxTexObj, effects);
 *                                                   (reference src/main.cpp:467)
 *
 * unchanged.  The vector types come from <hip/hip_vector_types.h> (float3 and
 * uchar4 are layout-identical to CUDA's), the texture handle is the 64-bit
 * rrt_sky_t (cudaTextureObject_t is `unsigned long long` too), and the body is
 * an inline call into the C ABI of include/rrt.h.  Like the reference's
 * launcher (src/raymarcher.cu:176-180) it is asynchronous on the null stream,
 * returns void and reports nothing; use rrt_launch_raymarch() directly for the
 * status code, a stream, run-time scene parameters (spin, volumetrics) or your
 * own workspace.
 *
 * `CameraEffects` is taken from the reference's own
 * camera_effects/camera_settings.h when that header is on the include path
 * (the reference tree keeps it; it contains no CUDA), otherwise the layout-
 * identical definition below is used.
 */
#ifndef RRT_RAYMARCHER_COMPAT_H
#define RRT_RAYMARCHER_COMPAT_H

#include <hip/hip_vector_types.h>

#include "rrt.h"

#if defined(__has_include)
#if __has_include("camera_effects/camera_settings.h")
#include "camera_effects/camera_settings.h"
#define RRT_HAVE_REFERENCE_CAMERA_SETTINGS 1
#endif
#endif

#ifndef RRT_HAVE_REFERENCE_CAMERA_SETTINGS
#ifndef CAMERA_SETTINGS_H
#define CAMERA_SETTINGS_H
/* layout and defaults of the reference's CameraEffects (camera_settings.h:4-17) */
struct CameraEffects {
    bool useBloom = true;
    float bloomThreshold = 0.8f;
    float bloomIntensity = 0.5f;
    bool useVignette = true;
    float vignetteIntensity = 0.4f;
    bool useChromaticAberration = false;
    float caAmount = 0.005f;
    bool useLensDistortion = true;
    float distortionAmount = 0.15f;
};
#endif
#endif

/* Camera basis passed host -> device (reference include/raymarcher.h:11-16). */
struct CameraState {
    float3 pos;
    float3 forward;
    float3 right;
    float3 up;
};

typedef rrt_sky_t cudaTextureObject_t;   /* the name src/main.cpp uses for the sky handle */

static_assert(sizeof(CameraState) == sizeof(rrt_camera), "CameraState must stay 4 packed float3");
static_assert(sizeof(CameraEffects) == sizeof(rrt_effects), "CameraEffects must stay 36 bytes");

inline void launch_raymarch(uchar4* d_out, int w, int h, float time, CameraState cam,
                            cudaTextureObject_t skyboxTex, CameraEffects effects) {
    rrt_effects fx;
    rrt_effects_default(&fx);
    fx.use_bloom = effects.useBloom;
    fx.bloom_threshold = effects.bloomThreshold;
    fx.bloom_intensity = effects.bloomIntensity;
    fx.use_vignette = effects.useVignette;
    fx.vignette_intensity = effects.vignetteIntensity;
    fx.use_chromatic_aberration = effects.useChromaticAberration;
    fx.ca_amount = effects.caAmount;
    fx.use_lens_distortion = effects.useLensDistortion;
    fx.distortion_amount = effects.distortionAmount;
    rrt_camera c;
    c.pos[0] = cam.pos.x;         c.pos[1] = cam.pos.y;         c.pos[2] = cam.pos.z;
    c.forward[0] = cam.forward.x; c.forward[1] = cam.forward.y; c.forward[2] = cam.forward.z;
    c.right[0] = cam.right.x;     c.right[1] = cam.right.y;     c.right[2] = cam.right.z;
    c.up[0] = cam.up.x;           c.up[1] = cam.up.y;           c.up[2] = cam.up.z;
    /* The reference's operating point (a 1000x700 window, src/main.cpp:467) is a small launch, where the
     * three-pass path is ~3x faster; it needs a pool, which this wrapper takes from
     * the library's per-device default (2 GiB, allocated on the first call).  Without a pool the single
     * kernel is used -- same pixels either way. */
    rrt_params prm;
    rrt_params_default(&prm);
    int ws = 0;
    if (rrt_default_workspace((size_t)2 << 30, &ws) == RRT_OK) prm.workspace = ws;
    (void)rrt_launch_raymarch(d_out, w, h, time, &c, skyboxTex, &fx, &prm, /*stream=*/nullptr);
}

#endif /* RRT_RAYMARCHER_COMPAT_H */
