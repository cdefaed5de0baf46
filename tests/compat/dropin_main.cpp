// dropin_main.cpp -- what the reference's renderFrame() (src/main.cpp:460-469) does with the
// boundary, minus OpenGL: get a device buffer, call launch_raymarch(...) exactly as main.cpp:467
// spells it, read the pixels back.  Compiled against include/raymarcher.h + librrt_hip.so by
// tests/test_compat.py; prints a checksum of the frame.
#include <hip/hip_runtime_api.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "raymarcher.h"

int main(int argc, char** argv) {
    const int w = argc > 1 ? atoi(argv[1]) : 64, h = argc > 2 ? atoi(argv[2]) : 36;
    const int sw = 256, sh = 128;
    std::vector<uint8_t> sky((size_t)sw * sh * 4);
    for (int j = 0; j < sh; ++j)
        for (int i = 0; i < sw; ++i) {
            uint8_t* t = &sky[4 * ((size_t)j * sw + i)];
            t[0] = (uint8_t)(i & 255); t[1] = (uint8_t)(2 * j & 255); t[2] = (uint8_t)((i ^ j) & 255); t[3] = 255;
        }
    cudaTextureObject_t skyTex = 0;
    if (rrt_sky_create(sky.data(), sw, sh, &skyTex) != RRT_OK) { fprintf(stderr, "sky: %s\n", rrt_last_hip_error()); return 2; }
    uchar4* d_out = nullptr;
    if (hipMalloc((void**)&d_out, (size_t)w * h * 4) != hipSuccess) return 3;

    CameraState camState;                       // the reference's start-up camera, main.cpp:128-130
    rrt_camera c; const float pos[3] = {0.0f, 10.0f, -60.0f};
    rrt_camera_from_angles(pos, 0.0f, -10.0f, &c);
    camState.pos = make_float3(c.pos[0], c.pos[1], c.pos[2]);
    camState.forward = make_float3(c.forward[0], c.forward[1], c.forward[2]);
    camState.right = make_float3(c.right[0], c.right[1], c.right[2]);
    camState.up = make_float3(c.up[0], c.up[1], c.up[2]);
    CameraEffects g_Effects;                    // default member initialisers
    float simTime = 1.0f;

    launch_raymarch(d_out, w, h, simTime, camState, skyTex, g_Effects);   // == src/main.cpp:467

    std::vector<uint8_t> out((size_t)w * h * 4);
    if (hipMemcpy(out.data(), d_out, out.size(), hipMemcpyDeviceToHost) != hipSuccess) return 4;  // syncs like :469
    uint64_t sum = 1469598103934665603ull;
    for (uint8_t b : out) { sum ^= b; sum *= 1099511628211ull; }
    printf("%dx%d fnv1a64=%016llx\n", w, h, (unsigned long long)sum);
    hipFree(d_out);
    rrt_sky_destroy(skyTex);
    return 0;
}
