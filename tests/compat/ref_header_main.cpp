// ref_header_main.cpp -- a translation unit built against the REFERENCE's own include/raymarcher.h (CUDA vector
// types from the CUDA runtime headers that ship in this image), linked against librrt_hip.so: the object
// refers to _Z15launch_raymarchP6uchar4iif11CameraStatey13CameraEffects, the reference's symbol, which the
// library exports (csrc/rrt_compat.cpp).  Built in the build container only (tests/test_compat.py: it needs
// /root/reference); the binary travels to the GPU box, where it renders one frame and prints its checksum.
// No HIP header is included here (HIP's and CUDA's vector types cannot share a translation unit); the three
// HIP runtime calls it needs are declared by hand.
#include "raymarcher.h"      // the reference's: -I/root/reference/include

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "rrt.h"             // plain C: sky creation, camera basis

extern "C" {
int hipMalloc(void** p, size_t n);
int hipFree(void* p);
int hipMemcpy(void* dst, const void* src, size_t n, int kind);   // kind 2 = device to host
}

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
    if (hipMalloc((void**)&d_out, (size_t)w * h * 4) != 0) return 3;
    rrt_camera c; const float pos[3] = {0.0f, 10.0f, -60.0f};
    rrt_camera_from_angles(pos, 0.0f, -10.0f, &c);
    CameraState camState;
    camState.pos = float3{c.pos[0], c.pos[1], c.pos[2]};
    camState.forward = float3{c.forward[0], c.forward[1], c.forward[2]};
    camState.right = float3{c.right[0], c.right[1], c.right[2]};
    camState.up = float3{c.up[0], c.up[1], c.up[2]};
    CameraEffects g_Effects;
    float simTime = 1.0f;
    launch_raymarch(d_out, w, h, simTime, camState, skyTex, g_Effects);          // == src/main.cpp:467
    std::vector<uint8_t> out((size_t)w * h * 4);
    if (hipMemcpy(out.data(), d_out, out.size(), 2) != 0) return 4;
    uint64_t sum = 1469598103934665603ull;
    for (uint8_t b : out) { sum ^= b; sum *= 1099511628211ull; }
    printf("%dx%d fnv1a64=%016llx\n", w, h, (unsigned long long)sum);
    hipFree(d_out);
    rrt_sky_destroy(skyTex);
    return 0;
}
