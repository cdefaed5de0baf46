// valu_issue_microbench.hip -- what makes one ORDER of the same VALU instructions faster than another on gfx950?
// The march kernel's time moves by up to 30 % with the instruction scheduler's choices (profiles/README.md, r02),
// at an unchanged instruction count.  This measures the candidates in isolation: dependent vs independent
// neighbours, and the VGPR banks (register number mod 4) of an instruction's source operands.
// Build: hipcc --offload-arch=gfx950 -O3 tools/valu_issue_microbench.hip -o tools/valu_issue_microbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))

// every pattern is 16 instructions, repeated 16x per loop iteration = 256 instructions
#define CLOB "v0","v1","v2","v3","v4","v5","v6","v7","v8","v9","v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31"

template <int P>
__global__ __launch_bounds__(256) void bench(float* out, int iters) {
    asm volatile("v_mov_b32 v0, 1.0\n v_mov_b32 v1, 1.0\n v_mov_b32 v2, 0\n v_mov_b32 v3, 1.0\n v_mov_b32 v4, 1.0\n v_mov_b32 v5, 1.0\n v_mov_b32 v6, 0\n v_mov_b32 v7, 1.0\n"
                 "v_mov_b32 v8, 1.0\n v_mov_b32 v9, 1.0\n v_mov_b32 v10, 0\n v_mov_b32 v11, 1.0\n v_mov_b32 v12, 1.0\n v_mov_b32 v13, 1.0\n v_mov_b32 v14, 0\n v_mov_b32 v15, 1.0\n"
                 "v_mov_b32 v16, 1.0\n v_mov_b32 v17, 1.0\n v_mov_b32 v18, 0\n v_mov_b32 v19, 1.0\n v_mov_b32 v20, 1.0\n v_mov_b32 v21, 1.0\n v_mov_b32 v22, 0\n v_mov_b32 v23, 1.0\n"
                 "v_mov_b32 v24, 1.0\n v_mov_b32 v25, 1.0\n v_mov_b32 v26, 0\n v_mov_b32 v27, 1.0\n v_mov_b32 v28, 1.0\n v_mov_b32 v29, 1.0\n v_mov_b32 v30, 0\n v_mov_b32 v31, 1.0\n" ::: CLOB);
    for (int i = 0; i < iters; ++i) {
        if (P == 0) asm volatile(REP16(REP16("v_fma_f32 v0, v0, v1, v2\n")) ::: CLOB);                       // dependent chain, sources in banks 0,1,2
        if (P == 1) asm volatile(REP16(REP16("v_fma_f32 v0, v0, v4, v8\n")) ::: CLOB);                       // dependent chain, all sources bank 0
        if (P == 2) asm volatile(REP16(REP4("v_fma_f32 v0, v0, v1, v2\n v_fma_f32 v4, v4, v5, v6\n v_fma_f32 v8, v8, v9, v10\n v_fma_f32 v12, v12, v13, v14\n")) ::: CLOB);   // 4 independent chains, banks 0,1,2
        if (P == 3) asm volatile(REP16(REP4("v_fma_f32 v0, v0, v4, v8\n v_fma_f32 v1, v1, v5, v9\n v_fma_f32 v2, v2, v6, v10\n v_fma_f32 v3, v3, v7, v11\n")) ::: CLOB);      // 4 independent chains, each instr all-same-bank
        if (P == 4) asm volatile(REP16(REP16("v_mul_f32 v0, v0, v1\n")) ::: CLOB);                           // dependent mul, banks 0,1
        if (P == 5) asm volatile(REP16(REP16("v_mul_f32 v0, v0, v4\n")) ::: CLOB);                           // dependent mul, same bank
        if (P == 6) asm volatile(REP16(REP4("v_mul_f32 v0, v0, v1\n v_mul_f32 v4, v4, v5\n v_mul_f32 v8, v8, v9\n v_mul_f32 v12, v12, v13\n")) ::: CLOB);  // independent muls, distinct banks
        if (P == 7) asm volatile(REP16(REP4("v_mul_f32 v0, v0, v4\n v_mul_f32 v1, v1, v5\n v_mul_f32 v2, v2, v6\n v_mul_f32 v3, v3, v7\n")) ::: CLOB);     // independent muls, same-bank sources
        if (P == 8) asm volatile(REP16(REP4("v_mul_f32 v16, v0, v1\n v_mul_f32 v17, v2, v3\n v_mul_f32 v18, v4, v5\n v_mul_f32 v19, v6, v7\n")) ::: CLOB); // independent, results never reused (no forwarding)
        if (P == 9) asm volatile(REP16(REP4("v_mul_f32 v16, v0, v1\n v_mul_f32 v17, v16, v3\n v_mul_f32 v18, v17, v5\n v_mul_f32 v19, v18, v7\n")) ::: CLOB); // chain through fresh registers
        if (P == 10) asm volatile(REP16(REP4("v_fma_f32 v16, v0, v1, v2\n v_fma_f32 v17, v4, v5, v6\n v_fma_f32 v18, v8, v9, v10\n v_fma_f32 v19, v12, v13, v14\n")) ::: CLOB); // independent fma, 12 distinct sources
        if (P == 11) asm volatile(REP16(REP4("v_fma_f32 v16, v0, v4, v8\n v_fma_f32 v17, v12, v20, v24\n v_fma_f32 v18, v28, v0, v4\n v_fma_f32 v19, v8, v12, v20\n")) ::: CLOB); // independent fma, all sources bank 0
        if (P == 12) asm volatile(REP16(REP4("v_fmac_f32 v16, v0, v1\n v_fmac_f32 v17, v2, v3\n v_fmac_f32 v18, v4, v5\n v_fmac_f32 v19, v6, v7\n")) ::: CLOB);    // VOP2 fmac
        if (P == 13) asm volatile(REP16(REP4("v_rsq_f32 v16, v0\n v_mul_f32 v17, v1, v2\n v_mul_f32 v18, v3, v4\n v_mul_f32 v19, v5, v6\n")) ::: CLOB);          // trans + 3 independent
        if (P == 14) asm volatile(REP16(REP4("v_rsq_f32 v16, v0\n s_nop 0\n v_mul_f32 v17, v16, v2\n v_mul_f32 v18, v17, v4\n v_mul_f32 v19, v18, v6\n")) ::: CLOB);   // trans + dependent chain
        if (P == 15) asm volatile(REP16(REP4("v_cmp_gt_f32 vcc, v0, v1\n v_cndmask_b32 v16, v2, v3, vcc\n v_mul_f32 v17, v4, v5\n v_mul_f32 v18, v6, v7\n")) ::: CLOB);  // compare + select
        if (P == 16) asm volatile(".p2align 3\n" REP16(REP16("v_fma_f32 v16, v0, v1, v2\n")) ::: CLOB);                 // 8-byte instrs, 8-byte aligned
        if (P == 17) asm volatile(".p2align 3\n s_nop 0\n" REP16(REP16("v_fma_f32 v16, v0, v1, v2\n")) ::: CLOB);      // the same, every one straddling an 8-byte boundary
        if (P == 18) asm volatile(".p2align 3\n" REP16(REP4("v_mul_f32 v16, v0, v1\n v_mul_f32 v17, v2, v3\n v_fma_f32 v18, v4, v5, v6\n v_fma_f32 v19, v8, v9, v10\n")) ::: CLOB);   // 4,4,8,8: all aligned
        if (P == 19) asm volatile(".p2align 3\n" REP16(REP4("v_mul_f32 v16, v0, v1\n v_fma_f32 v18, v4, v5, v6\n v_mul_f32 v17, v2, v3\n v_fma_f32 v19, v8, v9, v10\n")) ::: CLOB);   // 4,8,4,8: every fma straddles
        if (P == 20) asm volatile(".p2align 3\n" REP16(REP4("v_mul_f32 v16, 0x40033333, v1\n v_mul_f32 v17, v2, v3\n v_add_f32 v18, 0x41200000, v5\n v_mul_f32 v19, v8, v9\n")) ::: CLOB); // 8(lit),4,8(lit),4
        if (P == 21) asm volatile(".p2align 3\n" REP16(REP4("v_mul_f32 v17, v2, v3\n v_mul_f32 v16, 0x40033333, v1\n v_mul_f32 v19, v8, v9\n v_add_f32 v18, 0x41200000, v5\n")) ::: CLOB); // 4,8(lit),4,8(lit): literals straddle
    }
    float r;
    asm volatile("v_add_f32 %0, v0, v16" : "=v"(r) :: CLOB);
    if (r == 12345.678f) out[0] = r;
}

static const char* kNames[] = {"fma dependent chain, src banks 0,1,2", "fma dependent chain, src all bank 0", "fma 4 indep chains, src banks 0,1,2",
                               "fma 4 indep chains, each src same bank", "mul dependent, banks 0,1", "mul dependent, same bank", "mul 4 indep, distinct banks",
                               "mul 4 indep, same-bank sources", "mul indep, fresh dst (no reuse)", "mul chain through fresh regs", "fma indep, 12 distinct srcs",
                               "fma indep, all srcs bank 0", "fmac (VOP2) indep", "rsq + 3 indep mul", "rsq + dependent mul chain (5 instrs w/ s_nop)", "cmp + cndmask + 2 mul",
                               "fma x256, 8-byte aligned", "fma x256, all straddling 8-byte boundaries", "mul mul fma fma (aligned)", "mul fma mul fma (fma straddles)",
                               "lit-mul mul lit-add mul (aligned)", "mul lit-mul mul lit-add (literals straddle)"};

template <int P>
void run(int wps, float* d_out) {
    const int iters = 200;
    dim3 grid(256 * wps), block(256);
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    hipLaunchKernelGGL(bench<P>, grid, block, 0, 0, d_out, 4);
    CHK(hipDeviceSynchronize());
    float best = 1e9f;
    for (int r = 0; r < 3; ++r) {
        CHK(hipEventRecord(e0));
        hipLaunchKernelGGL(bench<P>, grid, block, 0, 0, d_out, iters);
        CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
        float ms; CHK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    double n_instr = (P == 14 ? 256.0 * 5 / 4 : 256.0) * iters;          // per wave (s_nop counted)
    double waves = 256.0 * wps * 4;                                       // 4 waves per 256-thread block
    // wave-instructions per SIMD per second -> cycles per wave-instruction per SIMD at 2.4 GHz
    double per_simd = n_instr * waves / 1024.0 / (best * 1e-3);
    printf("  %-48s %d waves/SIMD: %8.3f ms  %6.2f G wave-instr/s/SIMD  = %5.2f cycles/instr @2.4GHz\n", kNames[P], wps, best, per_simd / 1e9, 2.4e9 / per_simd);
}

int main() {
    float* d_out; CHK(hipMalloc(&d_out, 1024));
    for (int wps : {2, 8}) {
        printf("%d wave(s) per SIMD\n", wps);
        run<0>(wps, d_out); run<1>(wps, d_out); run<2>(wps, d_out); run<3>(wps, d_out); run<4>(wps, d_out); run<5>(wps, d_out);
        run<6>(wps, d_out); run<7>(wps, d_out); run<8>(wps, d_out); run<9>(wps, d_out); run<10>(wps, d_out); run<11>(wps, d_out);
        run<12>(wps, d_out); run<13>(wps, d_out); run<14>(wps, d_out); run<15>(wps, d_out);
        run<16>(wps, d_out); run<17>(wps, d_out); run<18>(wps, d_out); run<19>(wps, d_out); run<20>(wps, d_out); run<21>(wps, d_out);
    }
    return 0;
}
