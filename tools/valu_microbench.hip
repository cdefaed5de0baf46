// valu_microbench.hip -- measures the FP32 VALU issue rates of the MI355X that bound the
// ray-march kernel (the roofline peak used in DESIGN.md / bench.py is taken from here).
// Build: hipcc --offload-arch=gfx950 -O3 tools/valu_microbench.hip -o tools/valu_microbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

enum Op { ADD, MUL, FMA, PK_ADD, PK_MUL, PK_FMA, RCP, SQRT, RSQ, TRUNC, CNDMASK, IEEE_DIV, IEEE_SQRT, MIX_ADD_MUL, N_OPS };
static const char* kNames[] = {"v_add_f32", "v_mul_f32", "v_fma_f32", "v_pk_add_f32", "v_pk_mul_f32", "v_pk_fma_f32",
                               "v_rcp_f32", "v_sqrt_f32", "v_rsq_f32", "v_trunc_f32", "v_cndmask_b32", "ieee_div(a/b)",
                               "ieee_sqrtf", "add+mul mix"};
static const int kLaneOps[] = {1, 1, 1, 2, 2, 2, 1, 1, 1, 1, 1, 1, 1, 1};

template <int OP>
__global__ __launch_bounds__(256) void bench(float* out, int iters, float seed) {
    constexpr int C = 8;   // independent chains
    float a[C]; f2 p[C];
    float b = seed + 1.0009765625f, c = seed * 0.5f + 0.25f;
    f2 pb = {b, b}, pc = {c, c};
#pragma unroll
    for (int k = 0; k < C; ++k) { a[k] = seed + 1.0f + 0.001f * (k + threadIdx.x % 7); p[k] = f2{a[k], a[k] + 0.5f}; }
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int rep = 0; rep < 4; ++rep) {
#pragma unroll
            for (int k = 0; k < C; ++k) {
                if (OP == ADD) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[k]) : "v"(c));
                if (OP == MUL) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[k]) : "v"(b));
                if (OP == FMA) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c));
                if (OP == PK_ADD) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[k]) : "v"(pc));
                if (OP == PK_MUL) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[k]) : "v"(pb));
                if (OP == PK_FMA) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[k]) : "v"(pb), "v"(pc));
                if (OP == RCP) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[k]));
                if (OP == SQRT) asm volatile("v_sqrt_f32 %0, %0" : "+v"(a[k]));
                if (OP == RSQ) asm volatile("v_rsq_f32 %0, %0" : "+v"(a[k]));
                if (OP == TRUNC) asm volatile("v_trunc_f32 %0, %0" : "+v"(a[k]));
                if (OP == CNDMASK) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[k]) : "v"(b));
                if (OP == IEEE_DIV) a[k] = b / a[k];
                if (OP == IEEE_SQRT) a[k] = sqrtf(a[k]) + 1.5f;
                if (OP == MIX_ADD_MUL) { if (k & 1) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[k]) : "v"(c)); else asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[k]) : "v"(b)); }
            }
        }
    }
    float s = 0;
#pragma unroll
    for (int k = 0; k < C; ++k) s += a[k] + p[k].x + p[k].y;
    if (s == 12345.678f) out[0] = s;
}

template <int OP>
double run(int waves_per_simd, int iters, float* d_out) {
    dim3 grid(256 * waves_per_simd), block(256);
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    hipLaunchKernelGGL(bench<OP>, grid, block, 0, 0, d_out, 16, 0.0f);
    CHK(hipDeviceSynchronize());
    double best = 1e30;
    for (int r = 0; r < 3; ++r) {
        CHK(hipEventRecord(e0));
        hipLaunchKernelGGL(bench<OP>, grid, block, 0, 0, d_out, iters, 0.0f);
        CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
        float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    double instr = (double)grid.x * 256 * (double)iters * 4 * 8;   // lane-instructions
    return instr / (best * 1e-3);
}

int main() {
    float* d_out; CHK(hipMalloc(&d_out, 4));
    hipDeviceProp_t prop; CHK(hipGetDeviceProperties(&prop, 0));
    printf("device %s, CUs %d, clock %d kHz\n", prop.name, prop.multiProcessorCount, prop.clockRate);
    const int iters = 20000;
    printf("%-16s %10s %10s %10s %10s   (T lane-instr/s; x lane-ops per instr)\n", "op", "1w/SIMD", "2w/SIMD", "4w/SIMD", "8w/SIMD");
#define ROW(OP) { printf("%-16s", kNames[OP]); for (int w : {1, 2, 4, 8}) { double r = run<OP>(w, (OP >= RCP && OP <= RSQ) || OP >= IEEE_DIV ? iters / 4 : iters, d_out); printf(" %10.2f", r * 1e-12); } printf("   x%d\n", kLaneOps[OP]); }
    ROW(ADD) ROW(MUL) ROW(FMA) ROW(PK_ADD) ROW(PK_MUL) ROW(PK_FMA) ROW(RCP) ROW(SQRT) ROW(RSQ) ROW(TRUNC) ROW(CNDMASK) ROW(IEEE_DIV) ROW(IEEE_SQRT) ROW(MIX_ADD_MUL)
    return 0;
}
