// l1_return_microbench.hip -- how many bytes per clock does a CU's vector L1 return to the registers?
// (the bound of the table-served noise lookups: DESIGN.md section 4).  Every wave re-reads a tiny L1-resident
// array with global_load_dwordx4 / dwordx2 / dword, (a) 64 lanes x consecutive elements, (b) all lanes one address,
// (c) lanes scattered over 8 lines; 8 waves per SIMD, 16 independent loads in flight per wave.
// Build: hipcc --offload-arch=gfx950 -O3 tools/l1_return_microbench.hip -o tools/l1_return_microbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <typename T, int PATTERN>
__global__ __launch_bounds__(256) void k_loads(const T* __restrict__ src, int iters, float* sink) {
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * 256 + threadIdx.x) >> 6;
    // PATTERN 0: consecutive elements; 1: one address; 2: 8 lanes per 128-B line; 100 + k: k distinct elements, each read
    // by 64 / k neighbouring lanes (k = 2 .. 32); 200 + k: the same with the k elements 128 B apart (k lines)
    int idx = PATTERN == 0 ? lane : (PATTERN == 1 ? 0 : (PATTERN == 2 ? (lane & 7) * (128 / (int)sizeof(T)) + (lane >> 3) :
              (PATTERN < 200 ? lane / (64 / (PATTERN - 100)) : (lane / (64 / (PATTERN - 200))) * (128 / (int)sizeof(T)))));
    (void)wave;                                               // every wave reads the same <= 16 KB: L1 hits after the first pass
    float acc = 0.0f;
    for (int i = 0; i < iters; ++i) {
        asm volatile("" ::: "memory");                        // the loads of an iteration are re-issued, not hoisted
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            T v = src[idx + u * 256];                         // 16 independent loads in flight
            acc += reinterpret_cast<const float*>(&v)[0];
        }
    }
    if (acc == 123456.789f) sink[0] = acc;
}

template <typename T, int PATTERN>
void run(const char* name, const void* d_src, float* d_sink) {
    const int blocks = 256 * 8, iters = 2000;                // 8 workgroups of 4 waves per CU: 8 waves per SIMD
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k_loads<T, PATTERN>), dim3(blocks), dim3(256), 0, 0, static_cast<const T*>(d_src), 10, d_sink);
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(e0));
    hipLaunchKernelGGL((k_loads<T, PATTERN>), dim3(blocks), dim3(256), 0, 0, static_cast<const T*>(d_src), iters, d_sink);
    CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
    float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
    const double loads = (double)blocks * 4 * iters * 16;     // wave-level load instructions
    const double bytes = loads * 64 * sizeof(T);
    const double per_cu = bytes / (ms * 1e-3) / 256;
    printf("%-44s %7.2f ms  %7.1f GB/s per CU = %5.1f B/clk at 2.3 GHz; %5.1f clk per wave-instruction and CU\n", name, ms,
           per_cu / 1e9, per_cu / 2.3e9, 2.3e9 / (loads / (ms * 1e-3) / 256));
}

int main() {
    void* d_src; float* d_sink;
    CHK(hipMalloc(&d_src, 1 << 20)); CHK(hipMemset(d_src, 0, 1 << 20)); CHK(hipMalloc(&d_sink, 16));
    run<float4, 0>("dwordx4, 64 consecutive 16-B elements", d_src, d_sink);
    run<float4, 1>("dwordx4, all lanes one address", d_src, d_sink);
    run<float4, 2>("dwordx4, lanes over 8 lines", d_src, d_sink);
    run<float4, 102>("dwordx4, 2 distinct elements x 32 lanes", d_src, d_sink);
    run<float4, 104>("dwordx4, 4 distinct elements x 16 lanes", d_src, d_sink);
    run<float4, 108>("dwordx4, 8 distinct elements x 8 lanes", d_src, d_sink);
    run<float4, 116>("dwordx4, 16 distinct elements x 4 lanes", d_src, d_sink);
    run<float4, 132>("dwordx4, 32 distinct elements x 2 lanes", d_src, d_sink);
    run<float4, 204>("dwordx4, 4 elements in 4 lines x 16 lanes", d_src, d_sink);
    run<float4, 208>("dwordx4, 8 elements in 8 lines x 8 lanes", d_src, d_sink);
    run<float4, 216>("dwordx4, 16 elements in 16 lines x 4 lanes", d_src, d_sink);
    run<float2, 0>("dwordx2, 64 consecutive 8-B elements", d_src, d_sink);
    run<float2, 1>("dwordx2, all lanes one address", d_src, d_sink);
    run<float, 0>("dword, 64 consecutive elements", d_src, d_sink);
    run<float, 1>("dword, all lanes one address", d_src, d_sink);
    return 0;
}
