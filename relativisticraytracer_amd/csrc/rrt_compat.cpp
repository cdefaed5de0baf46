/*
 * rrt_compat.cpp -- launch_raymarch under the REFERENCE's own mangled name.
 *
 * The reference declares `void launch_raymarch(uchar4*, int, int, float, CameraState, cudaTextureObject_t,
 * CameraEffects)` as an ordinary C++ function (include/raymarcher.h:19) defined in src/raymarcher.cu:176-180.
 * With CUDA's vector types (`struct uchar4`) its Itanium name is
 *     _Z15launch_raymarchP6uchar4iif11CameraStatey13CameraEffects
 * whereas a definition written against HIP's headers mangles uchar4 as HIP_vector_type<unsigned char, 4>.
 * This translation unit therefore includes NO HIP header: it declares the three class names the mangling
 * needs with the reference's layouts (CameraState = four float3 = 48 bytes, raymarcher.h:11-16; CameraEffects =
 * 36 bytes, camera_settings.h:4-17 -- both passed in memory under the SysV ABI, so only size and alignment
 * matter) and forwards to the C ABI.  Compiled by g++ and linked into librrt_hip.so
 * (relativisticraytracer_amd/build.py); tests/test_compat.py links an object built against the reference's
 * header to it.
 */
#include "../../include/rrt.h"

struct uchar4;                                   /* only ever used through a pointer */
struct CameraState { float pos[3], forward[3], right[3], up[3]; };
struct CameraEffects {
    bool useBloom; float bloomThreshold; float bloomIntensity;
    bool useVignette; float vignetteIntensity;
    bool useChromaticAberration; float caAmount;
    bool useLensDistortion; float distortionAmount;
};
static_assert(sizeof(CameraState) == 48 && sizeof(CameraEffects) == 36, "layouts of the reference's structs");

void launch_raymarch(uchar4* d_out, int w, int h, float time, CameraState cam, unsigned long long skyboxTex,
                     CameraEffects effects) {
    (void)rrt_launch_raymarch_compat(d_out, w, h, time, cam.pos, skyboxTex, &effects);
}
