// clock_probe.hip -- what the shader clock counter counts on gfx950, and the clock the chip holds (dev tool).
//   hipcc --offload-arch=gfx950 -O3 tools/clock_probe.hip -o tools/clock_probe
// One wave spins on a dependent v_fma chain for a fixed number of iterations and reads s_memtime (__builtin_readcyclecounter)
// and s_memrealtime (constant 100 MHz) at both ends; the host times the same kernel with HIP events.  Printed: the ratio
// of the two counters (= shader clock in units of 100 MHz if s_memtime counts shader clocks), and both against the events.
// With `load` > 0 the same measurement is repeated while `load` workgroups per CU keep the VALUs of the whole chip busy.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__global__ void spin(unsigned long long* out, int iters, float seed) {
    const unsigned long long c0 = __builtin_readcyclecounter();
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    float a = seed + threadIdx.x;
    for (int i = 0; i < iters; ++i) a = __builtin_fmaf(a, 1.0000001f, 0.25f);
    const unsigned long long c1 = __builtin_readcyclecounter();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { out[0] = c1 - c0; out[1] = r1 - r0; out[2] = (unsigned long long)__float_as_uint(a); }
}
__global__ void burn(float* sink, int iters) {
    float a = threadIdx.x, b = blockIdx.x, c = 1.0f, d = 2.0f;
    for (int i = 0; i < iters; ++i) {
        a = __builtin_fmaf(a, 1.0000001f, 0.25f); b = __builtin_fmaf(b, 0.9999999f, 0.5f);
        c = __builtin_fmaf(c, 1.0000002f, 0.125f); d = __builtin_fmaf(d, 0.9999998f, 0.75f);
    }
    if (a + b + c + d == 12345.678f) sink[0] = a;
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 2000000;
    int rate_khz = 0, wall_khz = 0;
    hipDeviceGetAttribute(&rate_khz, hipDeviceAttributeClockRate, 0);
    hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, 0);
    printf("hipDeviceAttributeClockRate %d kHz, WallClockRate %d kHz\n", rate_khz, wall_khz);
    unsigned long long* d_out; float* d_sink;
    hipMalloc(&d_out, 64); hipMalloc(&d_sink, 64);
    hipStream_t s1, s2; hipStreamCreate(&s1); hipStreamCreate(&s2);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int load = 0; load <= 1; ++load) {
        for (int rep = 0; rep < 3; ++rep) {
            if (load) hipLaunchKernelGGL(burn, dim3(256 * 8), dim3(256), 0, s2, d_sink, iters * 2);
            hipEventRecord(e0, s1);
            hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s1, d_out, iters, 1.0f);
            hipEventRecord(e1, s1);
            hipDeviceSynchronize();
            float ms = 0; hipEventElapsedTime(&ms, e0, e1);
            unsigned long long h[3]; hipMemcpy(h, d_out, 24, hipMemcpyDeviceToHost);
            printf("%s: iters %d  s_memtime %llu  s_memrealtime %llu  event %.3f ms | memtime/memrealtime %.4f  memtime/event %.4f GHz  "
                   "memrealtime/event %.4f MHz  memtime/iter %.3f\n", load ? "loaded" : "idle  ", iters, h[0], h[1], ms,
                   (double)h[0] / (double)h[1], h[0] / (ms * 1e6), h[1] / (ms * 1e3), (double)h[0] / iters);
        }
    }
    return 0;
}
