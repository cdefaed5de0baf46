/*
 * rrt_test_hooks.h -- what only tests need: one-thread-per-element wrappers of the device functions (rrt_unit_*), self-checks
 * of the hand-rolled sqrt / divide cores against the hardware IEEE forms (rrt_selfcheck_*), and rrt_debug_fake_device.
 *
 * Compiled ONLY into librrt_hip_test.so (the same sources built with -DRRT_TEST_HOOKS; include/rrt_test.h declares the
 * entry points): the product library librrt_hip.so exports none of this.  Two sections of rrt_hip.hip:
 * RRT_TEST_HOOKS_PART 1 = kernels (inside its anonymous namespace), 2 = C ABI (inside its extern "C" block).
 */
#if RRT_TEST_HOOKS_PART == 1

/* ------------------------------------------------------------------ unit kernels */
__device__ __forceinline__ v3 ld3(const float* a, int i) { return mk(a[3 * i], a[3 * i + 1], a[3 * i + 2]); }
__device__ __forceinline__ void st3(float* a, int i, v3 v) { a[3 * i] = v.x; a[3 * i + 1] = v.y; a[3 * i + 2] = v.z; }

__global__ void k_geodesic_acc(int n, const float* p, const float* v, float spin, float* out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float drag_c = (2.0f * spin) * 2.0f;
    st3(out, i, geodesic_acc<true>(ld3(p, i), ld3(v, i), drag_c));      /* the march's own code path */
}
__global__ void k_rk4(int n, float* p, float* v, const float* h, float spin) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    v3 pp = ld3(p, i), vv = ld3(v, i);
    float drag_c = (2.0f * spin) * 2.0f;
    if (spin != 0.0f) integrate_rk4<true>(pp, vv, h[i], drag_c);
    else integrate_rk4<false>(pp, vv, h[i], drag_c);
    st3(p, i, pp); st3(v, i, vv);
}
/* The PRODUCTION step (round 3's integrate_rk4_lean, what every render kernel runs) as a chain of n_steps steps per
 * element, driven exactly as march_inline drives it: loop-top radius from the seed pair the previous step handed on
 * (seeded Goldschmidt root, v_rsq fall-back where the seed is rejected), horizon test r < 2.02 (the ray stops; the lean
 * step's stage 1 relies on it), then
 *   h == NULL: the march's own zone rule for the step size (raymarcher.cu:56-62) and the wave-uniform VACUUM step when all
 *              64 lanes of the wavefront hold an accepted radius >= 30 -- both template instances, the extrapolated
 *              seeds and the fall-backs are exercised by the inputs of tests/test_gpu_units.py;
 *   h != NULL: the generic step with the caller's step size on every step.
 * seed_scale: the first loop-top root's seed is seed_scale / r (0: none, as a ray's first step; 1.3: a bad seed that must be
 * rejected; 1.00005: an imperfect one that is accepted).  steps[i] = steps taken before the horizon test stopped the ray. */
template <bool SPIN>
__global__ __launch_bounds__(64) void k_rk4_lean(int n, float* p, float* v, const float* h_in, float drag_c, int n_steps,
                                                 float seed_scale, int* steps_out) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    const bool valid = i < n;
    v3 pp = valid ? ld3(p, i) : mk(1000.f, 0.f, 0.f), vv = valid ? ld3(v, i) : mk(0.f, 0.f, 0.f);
    float ys = 0.0f, hs = 0.0f, hcp = 0.0f;
    if (seed_scale != 0.0f) {
        float r0, y0;
        sqrt_rsq(dot(pp, pp), r0, y0);
        ys = seed_scale * y0; hs = 0.5f * ys;
    }
    int k = 0;
    for (; k < (valid ? n_steps : 0); ++k) {
        const v3 rel_p = pp;
        const float r2 = dot(rel_p, rel_p);
        float r, y, hy;
        const bool rejected = sqrt_seeded_yh<1>(r2, ys, hs, r, y, hy);
        const unsigned long long rej_mask = __builtin_amdgcn_ballot_w64(rejected);
        const bool vacuum = h_in == nullptr && RRT_VACUUM_PATH && (rej_mask | __builtin_amdgcn_ballot_w64(!(r >= kVacuumR))) == 0ull;
        if (!vacuum && rej_mask != 0ull) {
            bool small;
            if (rejected) radius_fallback(r2, r, y, hy, small);
        }
        if (r < kEventHorizon * 1.01f) break;
        if (vacuum) {
            integrate_rk4_lean<SPIN, true>(pp, vv, 0.f, 0.f, 0.f, drag_c, r2, r, y, hy, ys, hs, hcp);
        } else {
            float h, hh, h6;
            if (h_in) { h = h_in[i]; hh = 0.5f * h; h6 = h / 6.0f; }
            else {
                const bool near_bh = r < 18.0f;
                const bool in_disk = fabsf(rel_p.y) < kDiskH * 5.0f && r < kDiskOut + 5.0f;
                zone_step(near_bh, in_disk, h, hh, h6);
            }
            integrate_rk4_lean<SPIN, false>(pp, vv, h, hh, h6, drag_c, r2, r, y, hy, ys, hs, hcp);
        }
    }
    if (valid) { st3(p, i, pp); st3(v, i, vv); if (steps_out) steps_out[i] = k; }
}
/* the march's divide on explicit operands: out = div_seeded(a, b, seed) */
__global__ void k_div_seeded(int n, const float* a, const float* b, const float* seed, float* out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = div_seeded(a[i], b[i], seed[i]);
}
__global__ void k_hash31(int n, const float* p, float* out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = hash31(p[3 * i], p[3 * i + 1], p[3 * i + 2]);
}
__global__ void k_noise3d(int n, const float* p, float* out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = noise3d(ld3(p, i));
}
__global__ void k_fbm(int n, const float* p, int oct, float* out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    v3 q = ld3(p, i);
    float v = 0.0f, amp = 0.5f;
    for (int o = 0; o < oct; ++o) {
        v += amp * noise3d(q);
        q = mk(q.x * 2.05f + 10.0f, q.y * 2.05f + 10.0f, q.z * 2.05f + 10.0f);
        amp *= 0.5f;
    }
    out[i] = v;
}
__global__ void k_accretion(int n, const float* p, float time, float* out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = accretion_density<false, false>(ld3(p, i), time, NoiseLut{}, nullptr);
}
__global__ void k_dust(int n, const float* p, float time, float* out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = dust_density<false, false>(ld3(p, i), time, NoiseLut{}, nullptr);
}
__global__ void k_redshift(int n, const float* p, const float* vel, float spin, float* out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = redshift_factor(ld3(p, i), ld3(vel, i), spin);          /* the literal (IEEE-division) form */
}
__global__ void k_math(int fn, int n, const float* a, const float* b, float* out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float r = 0.0f;
    switch (fn) {
        case 0: r = rrt_expf(a[i]); break;
        case 1: r = rrt_powf(a[i], b[i]); break;
        case 2: r = rrt_sinf(a[i]); break;
        case 3: r = rrt_cosf(a[i]); break;
        case 4: r = rrt_atan2f(a[i], b[i]); break;
        case 5: r = rrt_asinf(a[i]); break;
        default: break;
    }
    out[i] = r;
}
__global__ void k_sky(int n, const float* dir, float off, SkyTex sky, float* out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float s[4];
    sample_sky(sky, ld3(dir, i), off, s);
    out[4 * i] = s[0]; out[4 * i + 1] = s[1]; out[4 * i + 2] = s[2]; out[4 * i + 3] = s[3];
}

__global__ void k_disk_temperature(int n, const float* r, float* out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = disk_temperature(r[i]);
}
__global__ void k_smoothstep(int n, const float* e0, const float* e1, const float* x, float* out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = smoothstep(e0[i], e1[i], x[i]);
}
/* what: 0 lens (uv -> uv), 1 vignette (rgb, uv -> rgb), 2 bloom contribution (rgb -> rgb) */
__global__ void k_postfx(int what, int n, const float* rgb, const float* uv, float param, float* out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (what == 0) {
        float ux = uv[2 * i], uy = uv[2 * i + 1];
        lens_distort(ux, uy, param);
        out[2 * i] = ux; out[2 * i + 1] = uy;
    } else if (what == 1) {
        st3(out, i, vignette(ld3(rgb, i), uv[2 * i], uv[2 * i + 1], param));
    } else {
        st3(out, i, bloom_part(ld3(rgb, i), param));
    }
}
/* the radiative-transfer block raymarcher.cu:71-116 on one sample per element; rad = (I_r, I_g, I_b, T) in/out.
 * r = length(p) exactly as the march holds it. */
__global__ void k_rt_sample(int n, const float* d_disk, const float* d_cloud, const float* p, const float* vel,
                            const float* h, float spin, float* rad) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const v3 rp = ld3(p, i);
    float r2, r, y;
    march_radius<kArithStrict>(rp, r2, r, y);
    Radiance acc = {rad[4 * i], rad[4 * i + 1], rad[4 * i + 2], rad[4 * i + 3]};
    accumulate_sample(acc, d_disk[i], d_cloud[i], rp, r, ld3(vel, i), h[i], spin);
    rad[4 * i] = acc.r; rad[4 * i + 1] = acc.g; rad[4 * i + 2] = acc.b; rad[4 * i + 3] = acc.t;
}
/* noise3D through the lattice-hash table (which: 0 accretion box, 1 dust box); counts[0] += reads the clamp had to move */
__global__ void k_noise3d_lut(int n, const float* p, NoiseLut L, float* out, unsigned* counts) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = noise3d_lut(L, ld3(p, i), counts);
}
/* the two density functions exactly as the render kernels call them (early-out, table switches) */
template <int MEDIA>
__global__ void k_media_lut(int n, const float* p, float time, NoiseLut la, NoiseLut ld, DustBands bands, float* out_disk, float* out_dust,
                            unsigned* counts) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const v3 q = ld3(p, i);
    const float r = length(q);                           /* the zone tests of raymarcher.cu:57-58 gate the calls */
    const bool in_disk = fabsf(q.y) < kDiskH * 5.0f && r < kDiskOut + 5.0f;
    const bool in_cloud = fabsf(q.y) < kCloudH * 1.5f && r < kCloudOut;
    media_densities<MEDIA>(q, time, in_disk, in_cloud, la, ld, bands, counts, out_disk[i], out_dust[i]);
}

/*
 * Self-checks of the march loop's sqrt/divide cores against the hardware-IEEE forms (sqrtf, `/`).
 * sqrt: every float whose bit pattern lies in [lo, hi).  div: `n` pseudo-random cases shaped
 * like the loop's operands: r2 log-uniform in [1, 2^28), seeds from sqrt_rsq(r2), numerators
 * log-uniform in 2^[-40, 40) with random sign.  counters[0] += mismatches; counters[1..3] keep
 * one failing case.
 */
__device__ __forceinline__ uint32_t mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
__global__ void k_selfcheck_sqrt(uint32_t lo, uint32_t hi, unsigned long long* counters) {
    uint64_t idx = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    unsigned bad = 0;
    for (uint64_t b = (uint64_t)lo + idx; b < hi; b += stride) {
        float x = rrt_u2f((uint32_t)b);
        float r, y;
        sqrt_rsq(x, r, y);
        float want = sqrtf(x);
        if (rrt_f2u(r) != rrt_f2u(want)) { ++bad; counters[1] = b; }
    }
    if (bad) atomicAdd(counters, (unsigned long long)bad);
}
__global__ void k_selfcheck_div(unsigned long long n, uint32_t seed, unsigned long long* counters) {
    uint64_t idx = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    unsigned bad = 0;
    for (uint64_t k = idx; k < n; k += stride) {
        uint32_t h1 = mix32((uint32_t)k * 2654435761u + seed), h2 = mix32(h1 ^ (uint32_t)(k >> 32) ^ 0x9e3779b9u);
        uint32_t h3 = mix32(h2 + 0x85ebca6bu);
        float r2 = rrt_u2f(0x3f800000u + (h1 % (28u << 23)));                 /* [1, 2^28) */
        float num = rrt_u2f(((87u << 23) + (h2 % (80u << 23))) | (h3 & 0x80000000u));   /* +-2^[-40,40) */
        float c = rrt_u2f(0x3f000000u + (h3 & 0x01ffffffu));                 /* [0.5, 8): drag constants */
        float r, y;
        sqrt_rsq(r2, r, y);
        float y2 = y * y, y3 = y2 * y;
        float d2 = r2 * r, d1 = (r2 * r2) * r;
        float q1 = div_seeded(num, d1, y3 * y2), q2 = div_seeded(c, d2, y3);
        float w1 = num / d1, w2 = c / d2;
        if (rrt_f2u(q1) != rrt_f2u(w1)) { ++bad; counters[1] = rrt_f2u(num); counters[2] = rrt_f2u(d1); }
        if (rrt_f2u(q2) != rrt_f2u(w2)) { ++bad; counters[1] = rrt_f2u(c); counters[2] = rrt_f2u(d2); counters[3] = 2; }
    }
    if (bad) atomicAdd(counters, (unsigned long long)bad);
}

/* The same two divides with the reciprocal-root seed AS THE MARCH PRODUCES IT (round 4): y comes out of
 * sqrt_seeded_yh<1> / <2> started from an estimate that is off by up to the acceptance tolerance of each form (uniform in
 * +-1.45e-4 for the one-iteration root -- which also covers the linearly extrapolated seeds of the vacuum step --, +-8.9e-3
 * for the two-iteration one), not out of the v_rsq-based sqrt_rsq that k_selfcheck_div uses: such a y carries up to 1.5 e^2 =
 * 3.4e-8 of its own error into y^3 and y^5, i.e. the Markstein cores start from a seed ~1.7x worse than k_selfcheck_div's.
 * Rejected roots are skipped (the march takes the sqrt_rsq fall-back there).  The seeded ROOT is checked first (against
 * the v_rsq-based correctly rounded one): counters (8 x uint64) [0] += accepted one-iteration roots that are not correctly
 * rounded, [1] += two-iteration ones, [2] += divide mismatches, [3] += divides checked, [4]/[5] one failing root (x bits,
 * seed bits), [6]/[7] one failing divide (numerator, denominator bits).  tol1 / tol2: half-width of the seed errors tried. */
__global__ void k_selfcheck_div_march(unsigned long long n, uint32_t seed, float tol1, float tol2, unsigned long long* counters) {
    uint64_t idx = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    unsigned bad_root1 = 0, bad_root2 = 0, bad_div = 0;
    unsigned long long checked = 0;
    for (uint64_t k = idx; k < n; k += stride) {
        uint32_t h1 = mix32((uint32_t)k * 2654435761u + seed), h2 = mix32(h1 ^ (uint32_t)(k >> 32) ^ 0x9e3779b9u);
        uint32_t h3 = mix32(h2 + 0x85ebca6bu), h4 = mix32(h3 ^ 0xc2b2ae35u);
        float r2 = rrt_u2f(0x3f800000u + (h1 % (28u << 23)));                 /* [1, 2^28) */
        float num = rrt_u2f(((87u << 23) + (h2 % (80u << 23))) | (h3 & 0x80000000u));   /* +-2^[-40,40) */
        float c = rrt_u2f(0x3f000000u + (h3 & 0x01ffffffu));                 /* [0.5, 8): drag constants */
        float r_ref, y_ref;
        sqrt_rsq(r2, r_ref, y_ref);
        const bool two = (h4 & 1u) != 0;
        const float u = (float)((h4 >> 8) & 0xffffffu) * (2.0f / 16777216.0f) - 1.0f;      /* [-1, 1) */
        const float y0 = y_ref * (1.0f + u * (two ? tol2 : tol1));
        float r, y, hy;
        const bool rejected = two ? sqrt_seeded_yh<2>(r2, y0, 0.5f * y0, r, y, hy) : sqrt_seeded_yh<1>(r2, y0, 0.5f * y0, r, y, hy);
        if (rejected) continue;
        if (rrt_f2u(r) != rrt_f2u(r_ref)) {            /* an ACCEPTED root that is not the correctly rounded one */
            if (two) ++bad_root2; else ++bad_root1;
            counters[4] = rrt_f2u(r2); counters[5] = rrt_f2u(y0);
            continue;
        }
        float y2 = y * y, y3 = y2 * y;
        float d2 = r2 * r, d1 = (r2 * r2) * r;
        float q1 = div_seeded(num, d1, y3 * y2), q2 = div_seeded(c, d2, y3);
        float w1 = num / d1, w2 = c / d2;
        checked += 2;
        if (rrt_f2u(q1) != rrt_f2u(w1)) { ++bad_div; counters[6] = rrt_f2u(num); counters[7] = rrt_f2u(d1); }
        if (rrt_f2u(q2) != rrt_f2u(w2)) { ++bad_div; counters[6] = rrt_f2u(c); counters[7] = rrt_f2u(d2); }
    }
    if (bad_root1) atomicAdd(counters, (unsigned long long)bad_root1);
    if (bad_root2) atomicAdd(counters + 1, (unsigned long long)bad_root2);
    if (bad_div) atomicAdd(counters + 2, (unsigned long long)bad_div);
    atomicAdd(counters + 3, checked);
}

/* sqrt_seeded against sqrtf: every float whose bits lie in [lo, hi), with estimates of 1/sqrt(x) that are off by
 * 0, +-1e-5 ... +-1.2e-2 relative (a fixed ladder plus 16 pseudo-random errors per x), one and two iterations.  Wherever sqrt_seeded ACCEPTS its result (returns true) the
 * root must be sqrtf(x) bit for bit.  counters[0] += mismatches, [1]/[2] one failing case (x bits, seed bits),
 * [3] += accepted cases (so that a test can see the check was not vacuous). */
__global__ void k_selfcheck_sqrt_seeded(uint32_t lo, uint32_t hi, unsigned long long* counters) {
    uint64_t idx = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    const float deltas[10] = {0.0f, 1e-5f, 5e-5f, 1e-4f, 1.4e-4f, 1.6e-4f, 1e-3f, 5e-3f, 1e-2f, 1.2e-2f};
    unsigned bad = 0;
    unsigned long long accepted = 0;
    for (uint64_t b = (uint64_t)lo + idx; b < hi; b += stride) {
        const float x = rrt_u2f((uint32_t)b);
        const float want = sqrtf(x);
        const float y_exact = (float)(1.0 / sqrt((double)x));
        for (int k = 0; k < 18; ++k) {
            for (int sgn = -1; sgn <= 1; sgn += 2) {
                /* the ladder, then 8 pseudo-random errors per x: 4 inside the one-iteration tolerance, 4 inside the two-iteration one */
                float delta;
                if (k < 10) delta = deltas[k];
                else {
                    const uint32_t hsh = mix32((uint32_t)b * 2654435761u + (uint32_t)(k * 2 + (sgn > 0)));
                    delta = (float)(hsh & 0xffffffu) * (1.0f / 16777216.0f) * (k < 14 ? 1.45e-4f : 9.5e-3f);
                }
                const float seed = y_exact * (1.0f + (float)sgn * delta);
                float r1, y1, r2, y2;
#if RRT_MARCH_V2
                /* the form the march uses since round 3: (y, y/2) handed on, acceptance on the FIRST residual */
                float h1, h2;
                if (!sqrt_seeded_yh<1>(x, seed, 0.5f * seed, r1, y1, h1)) { ++accepted; if (rrt_f2u(r1) != rrt_f2u(want) || y1 != h1 + h1) { ++bad; counters[1] = b; counters[2] = rrt_f2u(seed); } }
                if (!sqrt_seeded_yh<2>(x, seed, 0.5f * seed, r2, y2, h2)) { ++accepted; if (rrt_f2u(r2) != rrt_f2u(want) || y2 != h2 + h2) { ++bad; counters[1] = b; counters[2] = rrt_f2u(seed); } }
                /* a seed of twice the reciprocal root (x*y0^2 = 4) converges to MINUS the root: it must be rejected */
                if (k == 0 && sgn > 0) {
                    if (!sqrt_seeded_yh<2>(x, 2.0f * y_exact, y_exact, r2, y2, h2) || !sqrt_seeded_yh<1>(x, 2.0f * y_exact, y_exact, r1, y1, h1)) { ++bad; counters[1] = b; counters[2] = 4; }
                }
#else
                if (sqrt_seeded<1>(x, seed, r1, y1)) { ++accepted; if (rrt_f2u(r1) != rrt_f2u(want)) { ++bad; counters[1] = b; counters[2] = rrt_f2u(seed); } }
                if (sqrt_seeded<2>(x, seed, r2, y2)) { ++accepted; if (rrt_f2u(r2) != rrt_f2u(want)) { ++bad; counters[1] = b; counters[2] = rrt_f2u(seed); } }
#endif
            }
        }
    }
    if (bad) atomicAdd(counters, (unsigned long long)bad);
    atomicAdd(counters + 3, accepted);
}

/* The seeded roots on the floats AROUND every power of two (x = 2^e (1 + j 2^-23), |j| <= span, e in [e_lo, e_hi)) under a DENSE
 * sweep of seeds: n_seeds estimates per x and form, spread evenly over +-tol1 (one iteration) / +-tol2 (two).  That is where
 * sqrt(x) comes closest to a rounding tie (x = 4^k (1 + 2^-23): 2^-26 ulp) and where round 4 found -- and guarded -- the one
 * class of accepted roots that were not correctly rounded.  counters: [0] / [1] mismatching accepted one- / two-iteration
 * roots, [2] accepted roots checked, [3] rejected ones, [4]/[5] one failing case (x bits, seed bits). */
__global__ void k_selfcheck_sqrt_boundaries(int e_lo, int e_hi, int span, unsigned n_seeds, float tol1, float tol2,
                                            unsigned long long* counters) {
    const uint64_t n_x = (uint64_t)(e_hi - e_lo) * (2 * span + 1);
    const uint64_t total = n_x * n_seeds;
    unsigned bad1 = 0, bad2 = 0;
    unsigned long long ok = 0, rej = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t xi = i / n_seeds;
        const unsigned si = (unsigned)(i - xi * n_seeds);
        const int e = e_lo + (int)(xi / (2 * span + 1)), j = (int)(xi % (2 * span + 1)) - span;
        const uint32_t bits = (uint32_t)((e + 127) << 23) + (uint32_t)j;          /* j < 0 reaches into the binade below */
        const float x = rrt_u2f(bits);
        float r_ref, y_ref;
        sqrt_rsq(x, r_ref, y_ref);
        const float want = sqrtf(x);
        const float u = ((float)si + 0.5f) * (2.0f / (float)n_seeds) - 1.0f;     /* (-1, 1) */
        float r, y, hy;
        const float s1 = y_ref * (1.0f + u * tol1), s2 = y_ref * (1.0f + u * tol2);
        if (!sqrt_seeded_yh<1>(x, s1, 0.5f * s1, r, y, hy)) {
            ++ok;
            if (rrt_f2u(r) != rrt_f2u(want)) { ++bad1; counters[4] = bits; counters[5] = rrt_f2u(s1); }
        } else ++rej;
        if (!sqrt_seeded_yh<2>(x, s2, 0.5f * s2, r, y, hy)) {
            ++ok;
            if (rrt_f2u(r) != rrt_f2u(want)) { ++bad2; counters[4] = bits; counters[5] = rrt_f2u(s2); }
        } else ++rej;
    }
    if (bad1) atomicAdd(counters, (unsigned long long)bad1);
    if (bad2) atomicAdd(counters + 1, (unsigned long long)bad2);
    atomicAdd(counters + 2, ok);
    atomicAdd(counters + 3, rej);
}

/* rrt_div_tame against IEEE `/` on `n` pseudo-random tame operand pairs: |b| in 2^[-40, 40), |a| in 2^[-20, 20) times
 * |b| (so |a/b| in 2^[-20, 20)), random signs, plus a == 0 every 64th case. */
__global__ void k_selfcheck_div_tame(unsigned long long n, uint32_t seed, unsigned long long* counters) {
    uint64_t idx = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    unsigned bad = 0;
    for (uint64_t k = idx; k < n; k += stride) {
        uint32_t h1 = mix32((uint32_t)k * 2654435761u + seed), h2 = mix32(h1 ^ (uint32_t)(k >> 32) ^ 0x9e3779b9u);
        uint32_t h3 = mix32(h2 + 0x85ebca6bu);
        float b = rrt_u2f(((87u << 23) + (h1 % (80u << 23))) | (h3 & 0x80000000u));          /* +-2^[-40, 40) */
        float ratio = rrt_u2f(((107u << 23) + (h2 % (40u << 23))) | ((h3 << 1) & 0x80000000u));  /* +-2^[-20, 20) */
        float a = (k & 63) == 0 ? 0.0f : b * ratio;
        float q = rrt_div_tame(a, b), w = a / b;
        if (rrt_f2u(q) != rrt_f2u(w)) { ++bad; counters[1] = rrt_f2u(a); counters[2] = rrt_f2u(b); }
    }
    if (bad) atomicAdd(counters, (unsigned long long)bad);
}

/* rrt_div_const against IEEE `/` for the constants the media code divides by (smoothstep edges of densities.h:74-77,
 * :124 and raymarcher.cu:97, the rim taper :27, ISCO_RADIUS, DISK_TEMP_REF): EVERY dividend whose bit pattern lies in
 * [lo, hi), both signs, plus +0.  counters[0] += mismatches, [1]/[2] one failing case (dividend bits, constant index). */
__global__ void k_selfcheck_div_const(uint32_t lo, uint32_t hi, unsigned long long* counters) {
    uint64_t idx = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    unsigned bad = 0;
    for (uint64_t b = (uint64_t)lo + idx; b < hi; b += stride) {
        for (int sgn = 0; sgn < 2; ++sgn) {
            const float a = rrt_u2f((uint32_t)b | (sgn ? 0x80000000u : 0u));
#define RRT_CHK(K, B) do { const float q = rrt_div_const(a, (B)), w = a / (B); \
                           if (rrt_f2u(q) != rrt_f2u(w)) { ++bad; counters[1] = rrt_f2u(a); counters[2] = (K); } } while (0)
            RRT_CHK(0, kDiskOut * 0.8f - kDiskOut);          /* -5 */
            RRT_CHK(1, (kIsco + 5.0f) - kIsco);               /* 5 */
            RRT_CHK(2, 0.8f - 0.4f);
            RRT_CHK(3, kDiskOut - kDiskOut * 0.85f);          /* 3.75 */
            RRT_CHK(4, kIsco);                                /* 10 */
            RRT_CHK(5, kDiskTempRef);                         /* 1.5e7 */
            RRT_CHK(6, 1.3f - 0.7f);
#undef RRT_CHK
        }
    }
    if (idx == 0) {
        /* +0 dividends (x - e0 with x == e0): exact, sign included.  A -0 dividend would come back as +0 for a positive
         * constant (the last fma adds +0 to it); none of the use sites can produce one -- every dividend is a
         * difference with a non-zero literal, or a radius / temperature >= 1. */
        const float z = 0.0f;
        if (rrt_f2u(rrt_div_const(z, kDiskOut * 0.8f - kDiskOut)) != rrt_f2u(z / (kDiskOut * 0.8f - kDiskOut))) ++bad;
        if (rrt_f2u(rrt_div_const(z, kIsco)) != rrt_f2u(z / kIsco)) ++bad;
        if (rrt_f2u(rrt_div_const(z, 0.8f - 0.4f)) != rrt_f2u(z / (0.8f - 0.4f))) ++bad;
    }
    if (bad) atomicAdd(counters, (unsigned long long)bad);
}

template <class F>
int unit_launch(int n, void* stream, F f) {
    if (n < 0) return RRT_ERR_INVALID_ARGUMENT;
    if (n == 0) return RRT_OK;
    f(dim3((n + 255) / 256), dim3(256), static_cast<hipStream_t>(stream));
    RRT_HIP(hipGetLastError());
    return RRT_OK;
}

#elif RRT_TEST_HOOKS_PART == 2
/* test hook: make every device check see `device` as the current one (< 0: ask HIP again) */
int rrt_debug_fake_device(int device) {
    if (!test_hooks_enabled()) return RRT_ERR_INVALID_ARGUMENT;
    g_fake_device.store(device < 0 ? -1 : device);
    return RRT_OK;
}

/* ---- unit kernels ---- */
int rrt_unit_geodesic_acc(int n, const float* p, const float* v, float spin, float* out, void* st) {
    if (n > 0 && (!p || !v || !out)) return RRT_ERR_INVALID_ARGUMENT;
    return unit_launch(n, st, [&](dim3 g, dim3 b, hipStream_t s) { hipLaunchKernelGGL(k_geodesic_acc, g, b, 0, s, n, p, v, spin, out); });
}
int rrt_unit_rk4(int n, float* p, float* v, const float* h, float spin, void* st) {
    if (n > 0 && (!p || !v || !h)) return RRT_ERR_INVALID_ARGUMENT;
    return unit_launch(n, st, [&](dim3 g, dim3 b, hipStream_t s) { hipLaunchKernelGGL(k_rk4, g, b, 0, s, n, p, v, h, spin); });
}
int rrt_unit_rk4_lean(int n, float* p, float* v, const float* h, float spin, int n_steps, float seed_scale, int32_t* steps, void* st) {
    if (n < 0 || n_steps < 0 || (n > 0 && (!p || !v)) || !(seed_scale == seed_scale)) return RRT_ERR_INVALID_ARGUMENT;
    if (n == 0) return RRT_OK;
    const float drag_c = (2.0f * spin) * 2.0f;
    const dim3 g((n + 63) / 64), b(64);                     /* one wavefront per workgroup, like the render kernels */
    if (spin != 0.0f) hipLaunchKernelGGL((k_rk4_lean<true>), g, b, 0, static_cast<hipStream_t>(st), n, p, v, h, drag_c, n_steps, seed_scale, steps);
    else hipLaunchKernelGGL((k_rk4_lean<false>), g, b, 0, static_cast<hipStream_t>(st), n, p, v, h, drag_c, n_steps, seed_scale, steps);
    RRT_HIP(hipGetLastError());
    return RRT_OK;
}
int rrt_unit_div_seeded(int n, const float* a, const float* b, const float* seed, float* out, void* st) {
    if (n > 0 && (!a || !b || !seed || !out)) return RRT_ERR_INVALID_ARGUMENT;
    return unit_launch(n, st, [&](dim3 g, dim3 bl, hipStream_t s) { hipLaunchKernelGGL(k_div_seeded, g, bl, 0, s, n, a, b, seed, out); });
}
int rrt_unit_hash31(int n, const float* p, float* out, void* st) {
    if (n > 0 && (!p || !out)) return RRT_ERR_INVALID_ARGUMENT;
    return unit_launch(n, st, [&](dim3 g, dim3 b, hipStream_t s) { hipLaunchKernelGGL(k_hash31, g, b, 0, s, n, p, out); });
}
int rrt_unit_noise3d(int n, const float* p, float* out, void* st) {
    if (n > 0 && (!p || !out)) return RRT_ERR_INVALID_ARGUMENT;
    return unit_launch(n, st, [&](dim3 g, dim3 b, hipStream_t s) { hipLaunchKernelGGL(k_noise3d, g, b, 0, s, n, p, out); });
}
int rrt_unit_fbm(int n, const float* p, int oct, float* out, void* st) {
    if (n > 0 && (!p || !out)) return RRT_ERR_INVALID_ARGUMENT;
    if (oct < 0 || oct > 16) return RRT_ERR_INVALID_ARGUMENT;
    return unit_launch(n, st, [&](dim3 g, dim3 b, hipStream_t s) { hipLaunchKernelGGL(k_fbm, g, b, 0, s, n, p, oct, out); });
}
int rrt_unit_accretion_density(int n, const float* p, float time, float* out, void* st) {
    if (n > 0 && (!p || !out)) return RRT_ERR_INVALID_ARGUMENT;
    return unit_launch(n, st, [&](dim3 g, dim3 b, hipStream_t s) { hipLaunchKernelGGL(k_accretion, g, b, 0, s, n, p, time, out); });
}
int rrt_unit_dust_density(int n, const float* p, float time, float* out, void* st) {
    if (n > 0 && (!p || !out)) return RRT_ERR_INVALID_ARGUMENT;
    return unit_launch(n, st, [&](dim3 g, dim3 b, hipStream_t s) { hipLaunchKernelGGL(k_dust, g, b, 0, s, n, p, time, out); });
}
int rrt_unit_redshift(int n, const float* p, const float* vel, float spin, float* out, void* st) {
    if (n > 0 && (!p || !vel || !out)) return RRT_ERR_INVALID_ARGUMENT;
    return unit_launch(n, st, [&](dim3 g, dim3 b, hipStream_t s) { hipLaunchKernelGGL(k_redshift, g, b, 0, s, n, p, vel, spin, out); });
}
int rrt_unit_math(int fn, int n, const float* a, const float* b, float* out, void* st) {
    if (n > 0 && (!a || !b || !out)) return RRT_ERR_INVALID_ARGUMENT;
    if (fn < 0 || fn > 5) return RRT_ERR_INVALID_ARGUMENT;
    return unit_launch(n, st, [&](dim3 g, dim3 bl, hipStream_t s) { hipLaunchKernelGGL(k_math, g, bl, 0, s, fn, n, a, b, out); });
}
int rrt_unit_sky_sample(int n, const float* dir, float off, rrt_sky_t sky, int frac_bits, float* out, void* st) {
    if (n > 0 && (!dir || !out)) return RRT_ERR_INVALID_ARGUMENT;
    if (frac_bits < 0 || frac_bits > 16) return RRT_ERR_INVALID_ARGUMENT;
    SkyObject so;
    if (!sky_lookup(sky, so)) return RRT_ERR_BAD_HANDLE;
    SkyTex t{so.d_texels, so.w, so.h, frac_bits};
    return unit_launch(n, st, [&](dim3 g, dim3 b, hipStream_t s) { hipLaunchKernelGGL(k_sky, g, b, 0, s, n, dir, off, t, out); });
}

int rrt_unit_disk_temperature(int n, const float* r, float* out, void* st) {
    if (n > 0 && (!r || !out)) return RRT_ERR_INVALID_ARGUMENT;
    return unit_launch(n, st, [&](dim3 g, dim3 b, hipStream_t s) { hipLaunchKernelGGL(k_disk_temperature, g, b, 0, s, n, r, out); });
}
int rrt_unit_smoothstep(int n, const float* e0, const float* e1, const float* x, float* out, void* st) {
    if (n > 0 && (!e0 || !e1 || !x || !out)) return RRT_ERR_INVALID_ARGUMENT;
    return unit_launch(n, st, [&](dim3 g, dim3 b, hipStream_t s) { hipLaunchKernelGGL(k_smoothstep, g, b, 0, s, n, e0, e1, x, out); });
}
int rrt_unit_postfx(int what, int n, const float* rgb, const float* uv, float param, float* out, void* st) {
    if (what < 0 || what > 2) return RRT_ERR_INVALID_ARGUMENT;
    if (n > 0 && (!out || (what != 2 && !uv) || (what != 0 && !rgb))) return RRT_ERR_INVALID_ARGUMENT;
    return unit_launch(n, st, [&](dim3 g, dim3 b, hipStream_t s) { hipLaunchKernelGGL(k_postfx, g, b, 0, s, what, n, rgb, uv, param, out); });
}
int rrt_unit_rt_sample(int n, const float* d_disk, const float* d_cloud, const float* p, const float* vel, const float* h,
                       float spin, float* rad, void* st) {
    if (n > 0 && (!d_disk || !d_cloud || !p || !vel || !h || !rad)) return RRT_ERR_INVALID_ARGUMENT;
    return unit_launch(n, st, [&](dim3 g, dim3 b, hipStream_t s) { hipLaunchKernelGGL(k_rt_sample, g, b, 0, s, n, d_disk, d_cloud, p, vel, h, spin, rad); });
}
int rrt_unit_noise3d_lut(int n, const float* p, int table, int which, float* out, unsigned* d_counts, void* st) {
    if (n > 0 && (!p || !out)) return RRT_ERR_INVALID_ARGUMENT;
    if (which < 0 || which > 1) return RRT_ERR_INVALID_ARGUMENT;
    NoiseTableObject nt;
    {
        std::lock_guard<std::mutex> lk(g_nt_mu);
        auto it = g_nt.find(table);
        if (it == g_nt.end()) return RRT_ERR_BAD_HANDLE;
        nt = it->second;
    }
    if (!on_current_device(nt.device)) return RRT_ERR_BAD_HANDLE;
    const NoiseLut L = which == 0 ? make_lut(nt.d_cells, nt.acc, nt.acc_families)
                                  : make_lut(nt.d_cells + dust_cell0(nt), nt.dust, nt.dust_families);
    return unit_launch(n, st, [&](dim3 g, dim3 b, hipStream_t s) { hipLaunchKernelGGL(k_noise3d_lut, g, b, 0, s, n, p, L, out, d_counts); });
}
int rrt_unit_media_lut(int n, const float* p, float time, int table, float* out_disk, float* out_dust, unsigned* d_counts, void* st) {
    if (n > 0 && (!p || !out_disk || !out_dust)) return RRT_ERR_INVALID_ARGUMENT;
    NoiseTableObject nt;
    {
        std::lock_guard<std::mutex> lk(g_nt_mu);
        auto it = g_nt.find(table);
        if (it == g_nt.end()) return RRT_ERR_BAD_HANDLE;
        nt = it->second;
    }
    if (!on_current_device(nt.device)) return RRT_ERR_BAD_HANDLE;
    if (!(time >= nt.t0 && time <= nt.t1)) return RRT_ERR_INVALID_ARGUMENT;
    const NoiseLut la = make_lut(nt.d_cells, nt.acc, nt.acc_families);
    const NoiseLut ld = make_lut(nt.d_cells + dust_cell0(nt), nt.dust, nt.dust_families);
    const DustBands db = make_bands(nt);
    if (nt.banded) return unit_launch(n, st, [&](dim3 g, dim3 b, hipStream_t s) { hipLaunchKernelGGL(k_media_lut<3>, g, b, 0, s, n, p, time, la, ld, db, out_disk, out_dust, d_counts); });
    return unit_launch(n, st, [&](dim3 g, dim3 b, hipStream_t s) { hipLaunchKernelGGL(k_media_lut<2>, g, b, 0, s, n, p, time, la, ld, db, out_disk, out_dust, d_counts); });
}

int rrt_selfcheck_div_const(uint32_t lo_bits, uint32_t hi_bits, unsigned long long* d_counters, void* st) {
    if (!d_counters || lo_bits > hi_bits || hi_bits > 0x7f800000u) return RRT_ERR_INVALID_ARGUMENT;
    hipLaunchKernelGGL(k_selfcheck_div_const, dim3(4096), dim3(256), 0, static_cast<hipStream_t>(st), lo_bits, hi_bits, d_counters);
    RRT_HIP(hipGetLastError());
    return RRT_OK;
}

int rrt_selfcheck_sqrt(uint32_t lo_bits, uint32_t hi_bits, unsigned long long* d_counters, void* st) {
    if (!d_counters || lo_bits > hi_bits) return RRT_ERR_INVALID_ARGUMENT;
    hipLaunchKernelGGL(k_selfcheck_sqrt, dim3(2048), dim3(256), 0, static_cast<hipStream_t>(st), lo_bits, hi_bits, d_counters);
    RRT_HIP(hipGetLastError());
    return RRT_OK;
}
int rrt_selfcheck_sqrt_seeded(uint32_t lo_bits, uint32_t hi_bits, unsigned long long* d_counters, void* st) {
    if (!d_counters || lo_bits > hi_bits) return RRT_ERR_INVALID_ARGUMENT;
    hipLaunchKernelGGL(k_selfcheck_sqrt_seeded, dim3(2048), dim3(256), 0, static_cast<hipStream_t>(st), lo_bits, hi_bits, d_counters);
    RRT_HIP(hipGetLastError());
    return RRT_OK;
}
int rrt_selfcheck_sqrt_boundaries(int e_lo, int e_hi, int span, unsigned n_seeds, float tol1, float tol2, unsigned long long* d_counters, void* st) {
    if (!d_counters || e_lo >= e_hi || e_lo < -60 || e_hi > 100 || span < 0 || span > 4096 || n_seeds == 0 || !(tol1 >= 0.0f) || !(tol2 >= 0.0f))
        return RRT_ERR_INVALID_ARGUMENT;
    hipLaunchKernelGGL(k_selfcheck_sqrt_boundaries, dim3(2048), dim3(256), 0, static_cast<hipStream_t>(st), e_lo, e_hi, span, n_seeds, tol1, tol2, d_counters);
    RRT_HIP(hipGetLastError());
    return RRT_OK;
}
int rrt_selfcheck_div_tame(unsigned long long n, uint32_t seed, unsigned long long* d_counters, void* st) {
    if (!d_counters) return RRT_ERR_INVALID_ARGUMENT;
    hipLaunchKernelGGL(k_selfcheck_div_tame, dim3(2048), dim3(256), 0, static_cast<hipStream_t>(st), n, seed, d_counters);
    RRT_HIP(hipGetLastError());
    return RRT_OK;
}
int rrt_selfcheck_div_march(unsigned long long n, uint32_t seed, float tol1, float tol2, unsigned long long* d_counters, void* st) {
    if (!d_counters || !(tol1 >= 0.0f) || !(tol2 >= 0.0f)) return RRT_ERR_INVALID_ARGUMENT;
    hipLaunchKernelGGL(k_selfcheck_div_march, dim3(2048), dim3(256), 0, static_cast<hipStream_t>(st), n, seed, tol1, tol2, d_counters);
    RRT_HIP(hipGetLastError());
    return RRT_OK;
}
int rrt_selfcheck_div(unsigned long long n, uint32_t seed, unsigned long long* d_counters, void* st) {
    if (!d_counters) return RRT_ERR_INVALID_ARGUMENT;
    hipLaunchKernelGGL(k_selfcheck_div, dim3(2048), dim3(256), 0, static_cast<hipStream_t>(st), n, seed, d_counters);
    RRT_HIP(hipGetLastError());
    return RRT_OK;
}

#endif
