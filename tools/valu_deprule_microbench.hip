// valu_deprule_microbench.hip -- does the extra cost of a VALU instruction whose producer sits 2-6 instructions back
// (valu_depdist_microbench.hip) apply when ANOTHER operand comes from the instruction right before it?
// x[k] = x[k-a] * x[k-b] over a ring of 12 registers, 8 waves per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 tools/valu_deprule_microbench.hip -o tools/valu_deprule_microbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int A, int B, int FMA>
__global__ __launch_bounds__(256) void bench(float* out, int iters) {
    float c[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) c[k] = 1.0f + 1e-7f * (float)((threadIdx.x & 3) + k);
    float m = 1.0000001f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int rep = 0; rep < 20; ++rep) {
#pragma unroll
            for (int k = 0; k < 12; ++k) {
                const int ia = (k + 12 - A) % 12, ib = (k + 12 - B) % 12;
                if (FMA) asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(c[k]) : "v"(c[ia]), "v"(c[ib]), "v"(m));
                else if (B == 0) asm volatile("v_mul_f32 %0, %1, %2" : "=v"(c[k]) : "v"(c[ia]), "v"(m));
                else asm volatile("v_mul_f32 %0, %1, %2" : "=v"(c[k]) : "v"(c[ia]), "v"(c[ib]));
            }
        }
    }
    float s = 0;
#pragma unroll
    for (int k = 0; k < 12; ++k) s += c[k];
    if (s == 12345.678f) out[0] = s;
}

template <int A, int B, int FMA>
double run(int wps, float* d_out) {
    const int iters = 200;
    dim3 grid(256 * wps), block(256);
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    hipLaunchKernelGGL((bench<A, B, FMA>), grid, block, 0, 0, d_out, 4);
    CHK(hipDeviceSynchronize());
    float best = 1e9f;
    for (int r = 0; r < 3; ++r) {
        CHK(hipEventRecord(e0));
        hipLaunchKernelGGL((bench<A, B, FMA>), grid, block, 0, 0, d_out, iters);
        CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
        float ms; CHK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    double n_instr = 240.0 * iters, waves = 256.0 * wps * 4;
    return 2.4e9 / (n_instr * waves / 1024.0 / (best * 1e-3));
}

int main() {
    float* d_out; CHK(hipMalloc(&d_out, 1024));
    for (int wps : {4, 5, 8}) {
        printf("%d waves/SIMD, cycles per instruction @2.4 GHz; x[k] = x[k-a] * x[k-b]  (b = 0: second operand is a constant)\n", wps);
        printf("  mul a=1 b=0 %5.2f | a=2 b=0 %5.2f | a=3 b=0 %5.2f | a=6 b=0 %5.2f | a=11 b=0 %5.2f\n", run<1, 0, 0>(wps, d_out), run<2, 0, 0>(wps, d_out),
               run<3, 0, 0>(wps, d_out), run<6, 0, 0>(wps, d_out), run<11, 0, 0>(wps, d_out));
        printf("  mul a=1 b=2 %5.2f | a=1 b=3 %5.2f | a=1 b=4 %5.2f | a=1 b=6 %5.2f | a=1 b=11 %5.2f\n", run<1, 2, 0>(wps, d_out), run<1, 3, 0>(wps, d_out),
               run<1, 4, 0>(wps, d_out), run<1, 6, 0>(wps, d_out), run<1, 11, 0>(wps, d_out));
        printf("  mul a=2 b=3 %5.2f | a=2 b=6 %5.2f | a=3 b=5 %5.2f | a=6 b=11 %5.2f | a=8 b=11 %5.2f\n", run<2, 3, 0>(wps, d_out), run<2, 6, 0>(wps, d_out),
               run<3, 5, 0>(wps, d_out), run<6, 11, 0>(wps, d_out), run<8, 11, 0>(wps, d_out));
        printf("  fma a=1 b=2 %5.2f | a=1 b=6 %5.2f | a=1 b=11 %5.2f | a=2 b=3 %5.2f | a=8 b=11 %5.2f\n", run<1, 2, 1>(wps, d_out), run<1, 6, 1>(wps, d_out),
               run<1, 11, 1>(wps, d_out), run<2, 3, 1>(wps, d_out), run<8, 11, 1>(wps, d_out));
    }
    return 0;
}
