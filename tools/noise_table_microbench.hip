// noise_table_microbench.hip -- is a precomputed lattice-hash table faster than the arithmetic hash?
//
// noise3D (reference math_utils.h:98-110) spends ~108 of its ~150 VALU instructions hashing the 8 lattice
// corners (hash31, :91-96).  hash31 of a lattice point is a pure function of three integers, so the corner
// values can be READ from a table H[z][y][x] = hash31(x, y, z) built once with the same arithmetic (hence
// bit-identical), leaving floor/fract/fade/lerp on the VALU and moving the rest to the vector-memory pipe,
// which the march kernel leaves idle.  This program measures both forms on march-like access patterns.
//
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize tools/noise_table_microbench.hip
//               -o tools/noise_table_microbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cstring>
#include <vector>

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
#define DEV __device__ __forceinline__

DEV float fmod1(float x) { return x - __builtin_truncf(x); }
DEV float hash31(float px, float py, float pz) {
    float x = fmod1(px * 0.1031f), y = fmod1(py * 0.1031f), z = fmod1(pz * 0.1031f);
    float d = x * (y + 33.33f) + y * (z + 33.33f) + z * (x + 33.33f);
    x += d; y += d; z += d;
    return fmod1((x + y) * z);
}
DEV float lerp(float a, float b, float t) { return a + t * (b - a); }

DEV float noise_arith(float px, float py, float pz) {
    float ix = floorf(px), iy = floorf(py), iz = floorf(pz);
    float fx = px - ix, fy = py - iy, fz = pz - iz;
    float ux = fx * fx * (3.0f - 2.0f * fx), uy = fy * fy * (3.0f - 2.0f * fy), uz = fz * fz * (3.0f - 2.0f * fz);
    float x0 = ix + 0.0f, y0 = iy + 0.0f, z0 = iz + 0.0f, x1 = ix + 1.0f, y1 = iy + 1.0f, z1 = iz + 1.0f;
    float a = lerp(hash31(x0, y0, z0), hash31(x1, y0, z0), ux);
    float b = lerp(hash31(x0, y1, z0), hash31(x1, y1, z0), ux);
    float c = lerp(hash31(x0, y0, z1), hash31(x1, y0, z1), ux);
    float d = lerp(hash31(x0, y1, z1), hash31(x1, y1, z1), ux);
    return lerp(lerp(a, b, uy), lerp(c, d, uy), uz);
}

struct Table { const float* h; int x0, y0, z0, nx, ny, nz; unsigned max_idx; };
struct __attribute__((packed, aligned(4))) pair_t { float a, b; };

// layout A: H[z][y][x], 4 x 8-byte loads per call
DEV float noise_table(const Table& t, float px, float py, float pz) {
    float ix = floorf(px), iy = floorf(py), iz = floorf(pz);
    float fx = px - ix, fy = py - iy, fz = pz - iz;
    float ux = fx * fx * (3.0f - 2.0f * fx), uy = fy * fy * (3.0f - 2.0f * fy), uz = fz * fz * (3.0f - 2.0f * fz);
    int xi = (int)ix - t.x0, yi = (int)iy - t.y0, zi = (int)iz - t.z0;
    unsigned idx = min((unsigned)((zi * t.ny + yi) * t.nx + xi), t.max_idx);   /* never read outside the table */
    const float* base = t.h + idx;
    pair_t r00 = *reinterpret_cast<const pair_t*>(base);
    pair_t r10 = *reinterpret_cast<const pair_t*>(base + t.nx);
    pair_t r01 = *reinterpret_cast<const pair_t*>(base + (size_t)t.nx * t.ny);
    pair_t r11 = *reinterpret_cast<const pair_t*>(base + (size_t)t.nx * t.ny + t.nx);
    float a = lerp(r00.a, r00.b, ux), b = lerp(r10.a, r10.b, ux);
    float c = lerp(r01.a, r01.b, ux), d = lerp(r11.a, r11.b, ux);
    return lerp(lerp(a, b, uy), lerp(c, d, uy), uz);
}

// layout B: Q[z][y][x] = float4{H(x,y,z), H(x+1,y,z), H(x,y+1,z), H(x+1,y+1,z)}: 2 x 16-byte loads per call
DEV float noise_quad(const Table& t, float px, float py, float pz) {
    float ix = floorf(px), iy = floorf(py), iz = floorf(pz);
    float fx = px - ix, fy = py - iy, fz = pz - iz;
    float ux = fx * fx * (3.0f - 2.0f * fx), uy = fy * fy * (3.0f - 2.0f * fy), uz = fz * fz * (3.0f - 2.0f * fz);
    int xi = (int)ix - t.x0, yi = (int)iy - t.y0, zi = (int)iz - t.z0;
    unsigned idx = min((unsigned)((zi * t.ny + yi) * t.nx + xi), t.max_idx);   /* never read outside the table */
    const float4* base = reinterpret_cast<const float4*>(t.h) + idx;
    float4 q0 = base[0], q1 = base[(size_t)t.nx * t.ny];
    float a = lerp(q0.x, q0.y, ux), b = lerp(q0.z, q0.w, ux);
    float c = lerp(q1.x, q1.y, ux), d = lerp(q1.z, q1.w, ux);
    return lerp(lerp(a, b, uy), lerp(c, d, uy), uz);
}

// layout C ("dq"): D[z][y][x] = float4{H(x,y,z), H(x+1,y,z)-H(x,y,z), H(x,y+1,z), H(x+1,y+1,z)-H(x,y+1,z)}, read with
// SCALAR loads: a waterfall loop serves, per iteration, all lanes whose lattice cell equals the first active
// lane's (wave-uniform address -> s_load_dwordx4 x2, corner values in SGPRs); after MAX_IT distinct cells the
// remaining lanes use the arithmetic hash.  lerp(a, b, t) = a + t*(b - a) keeps its rounding: (b - a) is stored rounded.
struct f4_t { float x, y, z, w; };
typedef const __attribute__((address_space(4))) f4_t* const_f4_ptr;
template <int MAX_IT>
DEV float noise_waterfall(const Table& t, float px, float py, float pz) {
    float ix = floorf(px), iy = floorf(py), iz = floorf(pz);
    float fx = px - ix, fy = py - iy, fz = pz - iz;
    float ux = fx * fx * (3.0f - 2.0f * fx), uy = fy * fy * (3.0f - 2.0f * fy), uz = fz * fz * (3.0f - 2.0f * fz);
    int cx = (int)ix, cy = (int)iy, cz = (int)iz;
    float res = 0.f;
    bool done = false;
#pragma unroll 1
    for (int it = 0; it < MAX_IT; ++it) {
        if (!done) {
            const int sx = __builtin_amdgcn_readfirstlane(cx), sy = __builtin_amdgcn_readfirstlane(cy), sz = __builtin_amdgcn_readfirstlane(cz);
            const unsigned qx = (unsigned)(sx - t.x0), qy = (unsigned)(sy - t.y0), qz = (unsigned)(sz - t.z0);
            const bool ok = qx < (unsigned)(t.nx - 1) && qy < (unsigned)(t.ny - 1) && qz < (unsigned)(t.nz - 1);
            if (ok && cx == sx && cy == sy && cz == sz) {
                const unsigned idx = (qz * (unsigned)t.ny + qy) * (unsigned)t.nx + qx;
                const_f4_ptr c = (const_f4_ptr)(reinterpret_cast<const f4_t*>(t.h));
                const unsigned idx1 = idx + (unsigned)t.nx * (unsigned)t.ny;
                const f4_t q0 = {c[idx].x, c[idx].y, c[idx].z, c[idx].w}, q1 = {c[idx1].x, c[idx1].y, c[idx1].z, c[idx1].w};
                float a = q0.x + ux * q0.y, b = q0.z + ux * q0.w;
                float cc = q1.x + ux * q1.y, d = q1.z + ux * q1.w;
                res = lerp(lerp(a, b, uy), lerp(cc, d, uy), uz);
                done = true;
            }
        }
        if (__all(done)) break;
    }
    if (!done) {
        float x0 = ix + 0.0f, y0 = iy + 0.0f, z0 = iz + 0.0f, x1 = ix + 1.0f, y1 = iy + 1.0f, z1 = iz + 1.0f;
        float a = lerp(hash31(x0, y0, z0), hash31(x1, y0, z0), ux);
        float b = lerp(hash31(x0, y1, z0), hash31(x1, y1, z0), ux);
        float c = lerp(hash31(x0, y0, z1), hash31(x1, y0, z1), ux);
        float d = lerp(hash31(x0, y1, z1), hash31(x1, y1, z1), ux);
        res = lerp(lerp(a, b, uy), lerp(c, d, uy), uz);
    }
    return res;
}

__global__ void build_dq(float4* q, const float* h, int nx, int ny, int nz) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t n = (size_t)nx * ny * nz;
    if (i >= n) return;
    int x = (int)(i % nx), y = (int)((i / nx) % ny);
    float a = h[i], b = x + 1 < nx ? h[i + 1] : 0.f, c = y + 1 < ny ? h[i + nx] : 0.f, d = (x + 1 < nx && y + 1 < ny) ? h[i + nx + 1] : 0.f;
    q[i] = make_float4(a, b - a, c, d - c);
}

__global__ void build_table(float* h, int x0, int y0, int z0, int nx, int ny, int nz) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t n = (size_t)nx * ny * nz;
    if (i >= n) return;
    int x = (int)(i % nx), y = (int)((i / nx) % ny), z = (int)(i / ((size_t)nx * ny));
    h[i] = hash31((float)(x + x0) + 0.0f, (float)(y + y0) + 0.0f, (float)(z + z0) + 0.0f);
}
__global__ void build_quad(float4* q, const float* h, int nx, int ny, int nz) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t n = (size_t)nx * ny * nz;
    if (i >= n) return;
    int x = (int)(i % nx), y = (int)((i / nx) % ny);
    float a = h[i], b = x + 1 < nx ? h[i + 1] : 0.f, c = y + 1 < ny ? h[i + nx] : 0.f, d = (x + 1 < nx && y + 1 < ny) ? h[i + nx + 1] : 0.f;
    q[i] = make_float4(a, b, c, d);
}

// A wave = an 8x8 pixel tile; each lane walks a "ray" through noise space: start + lane offset (spread) and
// a per-step advance; 5 octaves of fbm per step (p = p*2.05 + 10), like getAccretionDensity's fbm(.,5).
template <int MODE>
__global__ __launch_bounds__(64) void walk(Table t, float* out, int steps, float spread, float adv, int octaves, float aniso) {
    const int lane = threadIdx.x & 63;
    const unsigned wave = blockIdx.x;
    float bx = -9.0f + 0.37f * (float)(wave % 47), by = -5.0f + 0.21f * (float)(wave % 31), bz = -9.0f + 0.29f * (float)(wave % 59);
    float px = bx + aniso * spread * (float)(lane & 7), py = by + spread * (float)(lane >> 3), pz = bz + 0.5f * aniso * spread * (float)(lane & 7);
    float acc = 0.f;
    for (int s = 0; s < steps; ++s) {
        float qx = px, qy = py, qz = pz, amp = 0.5f, v = 0.f;
#pragma unroll 1
        for (int o = 0; o < octaves; ++o) {
            float n = MODE == 0 ? noise_arith(qx, qy, qz) : (MODE == 1 ? noise_table(t, qx, qy, qz) : (MODE == 2 ? noise_quad(t, qx, qy, qz) : noise_waterfall<3>(t, qx, qy, qz)));
            v += amp * n;
            qx = qx * 2.05f + 10.0f; qy = qy * 2.05f + 10.0f; qz = qz * 2.05f + 10.0f;
            amp *= 0.5f;
        }
        acc += v;
        px += adv; py += 0.13f * adv; pz += 0.71f * adv;
    }
    out[(size_t)wave * 64 + lane] = acc;
}

int main(int argc, char** argv) {
    // box that covers 5 octaves of the walk: octave k coordinate = c*2.05^k + 10*(2.05^k - 1)/1.05
    const int x0 = -16, y0 = -16, z0 = -16, nx = 640, ny = 560, nz = 640;
    size_t n = (size_t)nx * ny * nz;
    float* h; float4* q; float4* dq;
    CHK(hipMalloc(&h, n * 4)); CHK(hipMalloc(&q, n * 16)); CHK(hipMalloc(&dq, n * 16));
    hipLaunchKernelGGL(build_table, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, h, x0, y0, z0, nx, ny, nz);
    hipLaunchKernelGGL(build_quad, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, q, h, nx, ny, nz);
    hipLaunchKernelGGL(build_dq, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, dq, h, nx, ny, nz);
    CHK(hipDeviceSynchronize());
    const int waves = 256 * 4 * 8 * 4;      // 4 rounds of full occupancy
    float* out; CHK(hipMalloc(&out, (size_t)waves * 64 * 4 * 4));
    std::vector<float> r0((size_t)waves * 64), r1(r0.size()), r2(r0.size()), r3(r0.size());
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    printf("table %dx%dx%d = %.2f GB (float) / %.2f GB (quad)\n", nx, ny, nz, n * 4 / 1e9, n * 16 / 1e9);
    printf("%-34s %10s %10s %10s %10s  (ms per launch; %d waves x 64 lanes x steps x octaves noise3D)\n", "pattern", "arith", "table", "quad", "waterfall3", waves);
    struct Pat { const char* name; float spread, adv; int steps, oct; float aniso; } pats[] = {
        {"tile 0.01/px, adv 0.01, 2 oct", 0.01f, 0.01f, 160, 2, 1.0f},
        {"tile 0.03/px, adv 0.02, 2 oct", 0.03f, 0.02f, 160, 2, 1.0f},
        {"tile 0.06/px, adv 0.04, 2 oct", 0.06f, 0.04f, 160, 2, 1.0f},
        {"tile 0.12/px, adv 0.04, 2 oct", 0.12f, 0.04f, 160, 2, 1.0f},
        {"y-only 0.05/px, 1 oct", 0.05f, 0.02f, 200, 1, 0.05f},
        {"y-only 0.10/px, 1 oct", 0.10f, 0.02f, 200, 1, 0.05f},
        {"y-only 0.20/px, 1 oct", 0.20f, 0.02f, 200, 1, 0.05f},
        {"y-only 0.30/px, 1 oct", 0.30f, 0.02f, 200, 1, 0.05f},
        {"y-only 0.50/px, 1 oct", 0.50f, 0.02f, 200, 1, 0.05f},
        {"y-only 0.80/px, 1 oct", 0.80f, 0.02f, 200, 1, 0.05f},
        {"y-only 1.20/px, 1 oct", 1.20f, 0.02f, 200, 1, 0.05f},
        {"iso 0.05/px, 1 oct", 0.05f, 0.02f, 200, 1, 1.0f},
        {"iso 0.10/px, 1 oct", 0.10f, 0.02f, 200, 1, 1.0f},
        {"iso 0.20/px, 1 oct", 0.20f, 0.02f, 200, 1, 1.0f},
        {"iso 0.30/px, 1 oct", 0.30f, 0.02f, 200, 1, 1.0f},
        {"iso 0.50/px, 1 oct", 0.50f, 0.02f, 200, 1, 1.0f},
        {"tile 0.03/px, adv 0.04, 5 oct", 0.03f, 0.04f, 64, 5, 1.0f},
        {"tile 0.10/px, adv 0.04, 5 oct", 0.10f, 0.04f, 64, 5, 1.0f},
        {"tile 0.30/px, adv 0.09, 5 oct", 0.30f, 0.09f, 64, 5, 1.0f},
        {"tile 0.03/px, adv 0.04, 2 oct", 0.03f, 0.04f, 160, 2, 1.0f},
        {"tile 1.00/px, adv 0.30, 5 oct", 1.00f, 0.30f, 32, 5, 1.0f},
    };
    for (const Pat& p : pats) {
        float ms[4];
        for (int mode = 0; mode < 4; ++mode) {
            Table t{mode == 3 ? reinterpret_cast<const float*>(dq) : mode == 2 ? reinterpret_cast<const float*>(q) : h, x0, y0, z0, nx, ny, nz, (unsigned)(n - (size_t)nx * ny - nx - 2)};
            float* o = out + (size_t)mode * waves * 64;
            for (int rep = 0; rep < 3; ++rep) {
                CHK(hipEventRecord(e0));
                if (mode == 0) hipLaunchKernelGGL(walk<0>, dim3(waves), dim3(64), 0, 0, t, o, p.steps, p.spread, p.adv, p.oct, p.aniso);
                if (mode == 1) hipLaunchKernelGGL(walk<1>, dim3(waves), dim3(64), 0, 0, t, o, p.steps, p.spread, p.adv, p.oct, p.aniso);
                if (mode == 2) hipLaunchKernelGGL(walk<2>, dim3(waves), dim3(64), 0, 0, t, o, p.steps, p.spread, p.adv, p.oct, p.aniso);
                if (mode == 3) hipLaunchKernelGGL(walk<3>, dim3(waves), dim3(64), 0, 0, t, o, p.steps, p.spread, p.adv, p.oct, p.aniso);
                CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
                CHK(hipEventElapsedTime(&ms[mode], e0, e1));
            }
        }
        CHK(hipMemcpy(r0.data(), out, r0.size() * 4, hipMemcpyDeviceToHost));
        CHK(hipMemcpy(r1.data(), out + (size_t)waves * 64, r0.size() * 4, hipMemcpyDeviceToHost));
        CHK(hipMemcpy(r2.data(), out + (size_t)2 * waves * 64, r0.size() * 4, hipMemcpyDeviceToHost));
        CHK(hipMemcpy(r3.data(), out + (size_t)3 * waves * 64, r0.size() * 4, hipMemcpyDeviceToHost));
        size_t bad1 = 0, bad2 = 0, bad3 = 0;
        for (size_t i = 0; i < r0.size(); ++i) {
            bad1 += memcmp(&r0[i], &r1[i], 4) != 0;
            bad2 += memcmp(&r0[i], &r2[i], 4) != 0;
            bad3 += memcmp(&r0[i], &r3[i], 4) != 0;
        }
        double calls = (double)waves * 64 * p.steps * p.oct;
        printf("%-34s %10.3f %10.3f %10.3f %10.3f  bit-mismatches %zu / %zu / %zu;  G noise3D/s: %.1f / %.1f / %.1f / %.1f\n", p.name, ms[0], ms[1], ms[2], ms[3],
               bad1, bad2, bad3, calls / ms[0] / 1e6, calls / ms[1] / 1e6, calls / ms[2] / 1e6, calls / ms[3] / 1e6);
    }
    return 0;
}
