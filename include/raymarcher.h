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
 * rrt_sky_t (cudaTextureObject_t is `unsigned long long` too), and the function
 * lives in librrt_hip.so over the C ABI of include/rrt.h.  Like the reference's
 * launcher (src/raymarcher.cu:176-180) it is asynchronous on the null stream
 * and returns void (the first failing launch is reported once on stderr); use
 * rrt_launch_raymarch() directly for the status code or a stream, and
 * rrt_set_launch_defaults() for run-time scene parameters (spin, volumetrics,
 * a workspace, a noise table).
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

/*
 * launch_raymarch: by default the out-of-line function that librrt_hip.so exports -- like the reference's,
 * which is defined in src/raymarcher.cu:176-180.  The library carries it twice: under the name this header
 * produces (HIP's uchar4 is a class template instance) and under the reference's own mangled name
 * _Z15launch_raymarchP6uchar4iif11CameraStatey13CameraEffects (csrc/rrt_compat.cpp), so that an object file
 * compiled against the REFERENCE's header with CUDA's vector types links against librrt_hip.so as it is.
 * Scene parameters the signature has no room for come from rrt_set_launch_defaults(); nothing is allocated.
 * -DRRT_INLINE_LAUNCH_RAYMARCH gives a header-only inline version instead.
 */
#ifdef RRT_INLINE_LAUNCH_RAYMARCH
inline void launch_raymarch(uchar4* d_out, int w, int h, float time, CameraState cam,
                            cudaTextureObject_t skyboxTex, CameraEffects effects) {
    (void)rrt_launch_raymarch_compat(d_out, w, h, time, reinterpret_cast<const float*>(&cam), skyboxTex, &effects);
}
#else
void launch_raymarch(uchar4* d_out, int w, int h, float time, CameraState cam,
                     cudaTextureObject_t skyboxTex, CameraEffects effects);
#endif

#endif /* RRT_RAYMARCHER_COMPAT_H */
