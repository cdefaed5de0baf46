// valu_depdist_microbench.hip -- cost of a VALU instruction as a function of how many instructions back its
// producer sits (1 = back-to-back dependent), 8 waves per SIMD.  Companion of valu_issue_microbench.hip.
// Build: hipcc --offload-arch=gfx950 -O3 tools/valu_depdist_microbench.hip -o tools/valu_depdist_microbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
#define CLOB "v0","v1","v2","v3","v4","v5","v6","v7","v8","v9","v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23"

// D independent chains interleaved round-robin: each instruction depends on the one D instructions earlier.
template <int D, int KIND>
__global__ __launch_bounds__(256) void bench(float* out, int iters) {
    float c[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) c[k] = 1.0f + 0.0001f * (float)(threadIdx.x & 3) * (float)k;
    float m = 1.0000001f, a = 1e-9f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int rep = 0; rep < 240 / D; ++rep) {
#pragma unroll
            for (int k = 0; k < D; ++k) {
                if (KIND == 0) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(c[k]) : "v"(m));
                if (KIND == 1) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(c[k]) : "v"(m), "v"(a));
                if (KIND == 2) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(c[k]) : "v"(m), "v"(a));
                if (KIND == 3) asm volatile("v_add_f32 %0, %0, %1" : "+v"(c[k]) : "v"(a));
            }
        }
    }
    float s = 0;
#pragma unroll
    for (int k = 0; k < 12; ++k) s += c[k];
    if (s == 12345.678f) out[0] = s;
}

template <int D, int KIND>
double run(int wps, float* d_out) {
    const int iters = 200;
    dim3 grid(256 * wps), block(256);
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    hipLaunchKernelGGL((bench<D, KIND>), grid, block, 0, 0, d_out, 4);
    CHK(hipDeviceSynchronize());
    float best = 1e9f;
    for (int r = 0; r < 3; ++r) {
        CHK(hipEventRecord(e0));
        hipLaunchKernelGGL((bench<D, KIND>), grid, block, 0, 0, d_out, iters);
        CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
        float ms; CHK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    double n_instr = (240 / D) * D * (double)iters, waves = 256.0 * wps * 4;
    return 2.4e9 / (n_instr * waves / 1024.0 / (best * 1e-3));
}

int main() {
    float* d_out; CHK(hipMalloc(&d_out, 1024));
    const char* kinds[] = {"v_mul_f32", "v_fma_f32", "v_fmac_f32", "v_add_f32"};
    for (int wps : {1, 2, 4, 8}) {
        printf("%d wave(s)/SIMD: cycles per instruction @2.4 GHz by dependency distance 1..6,8,12\n", wps);
        double r[4][8];
#define ROW(K) r[K][0] = run<1, K>(wps, d_out); r[K][1] = run<2, K>(wps, d_out); r[K][2] = run<3, K>(wps, d_out); r[K][3] = run<4, K>(wps, d_out); \
               r[K][4] = run<5, K>(wps, d_out); r[K][5] = run<6, K>(wps, d_out); r[K][6] = run<8, K>(wps, d_out); r[K][7] = run<12, K>(wps, d_out);
        ROW(0) ROW(1) ROW(2) ROW(3)
        for (int k = 0; k < 4; ++k) {
            printf("  %-11s", kinds[k]);
            for (int j = 0; j < 8; ++j) printf(" %6.2f", r[k][j]);
            printf("\n");
        }
    }
    return 0;
}
