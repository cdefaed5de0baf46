/*
 * rrt_oracle.h -- CPU oracle for the per-pixel geodesic ray-march hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library, and only as the checker / reported CPU baseline.
 *
 * What it is: a plain-C restatement of the reference's algorithm
 * (/root/reference/src/raymarcher.cu:15-174 and include/{integrators,
 * geodesics,densities,math_utils}.h, include/camera_effects/post_processing.h),
 * same operations in the same order, strict IEEE binary32 (-ffp-contract=off).
 *
 * Pinning status
 *   - unit functions (hash31, noise3D, fbm, getGeodesicAcc, integrate_rk4,
 *     calculateRedshiftFactor, getDiskTemperature, getAccretionDensity,
 *     getDustCloudDensity, smoothstep, lens/vignette/bloom): PINNED bit-exact
 *     to the reference's own headers compiled by g++ (oracle/_ref, built from
 *     /root/reference/include by oracle/Makefile; vectors in tests/golden/).
 *   - the per-pixel pipeline (raymarcher.cu:15-174: loop order, zone / step logic,
 *     radiative transfer block, sky lookup, composition, post-FX, tone map, row
 *     flip): PINNED byte for byte -- RGBA8 and per-ray step counts -- to the
 *     reference's own raymarch_kernel body, compiled by g++ where it lies
 *     (oracle/ref_frames.cpp + ref_frames_pre.h -> oracle/_ref/libref_frames.so;
 *     fixtures tests/golden/frames_ref.npz, G1-G5 + two path keyframes + one odd
 *     view; tests/test_oracle_frames.py).  The harness supplies the launch
 *     indices and tex2D<float4>, nothing else.
 *   - the sky sampler replaces CUDA's hardware bilinear filter, which the
 *     reference source does not define: "parity unpinned" for that one function
 *     (DESIGN.md section 6); it is also what the reference-kernel harness uses
 *     as tex2D<float4>.
 *
 * math_mode selects the transcendental library:
 *   RRTO_MATH_LIBM     glibc powf/expf/sinf/cosf/atan2f/asinf (default)
 *   RRTO_MATH_PORTABLE relativisticraytracer_amd/csrc/rrt_math.h -- the exact
 *                      functions the HIP kernels use, for byte-level checks.
 *   RRTO_MATH_NUDGED_BASE + seed: glibc perturbed by <= 2-4 ulp per call (see below)
 */
#ifndef RRT_ORACLE_H
#define RRT_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RRTO_MATH_LIBM 0
#define RRTO_MATH_PORTABLE 1
/* RRTO_MATH_NUDGED_BASE + seed: glibc with every transcendental result moved by a pseudo-random number of ulps
 * inside a GPU math library's documented error class (conditioning probes, tests/test_density_conditioning.py) */
#define RRTO_MATH_NUDGED_BASE 16

typedef struct {
    float pos[3], forward[3], right[3], up[3];     /* include/raymarcher.h:11-16 */
} rrto_camera;

typedef struct {                                    /* camera_settings.h:4-17 */
    int32_t use_bloom;       float bloom_threshold;  float bloom_intensity;
    int32_t use_vignette;    float vignette_intensity;
    int32_t use_ca;          float ca_amount;
    int32_t use_lens;        float distortion_amount;
} rrto_effects;

typedef struct {
    float spin;              /* SPIN_A, config.h:21 */
    int32_t volumetrics;     /* 0: densities forced to 0 (zone step sizes kept) */
    int32_t max_steps;       /* MAX_STEPS, config.h:48 */
    int32_t math_mode;       /* RRTO_MATH_* */
    int32_t sky_frac_bits;   /* bilinear weight quantisation; 0 = none, 8 = CUDA-like */
    int32_t nudge_ulps;      /* conditioning probe (no counterpart in the reference): K > 0 moves every component of every
                                primary direction by a pseudo-random whole number of ulps in [-K, K]; the product's
                                rrt_params.nudge_ulps / .nudge_seed is the same function (rrto_nudge_component) */
    uint32_t nudge_seed;
} rrto_params;

/* v moved by k ulps, k in [-K, K] from a hash of (x, y, seed, component 0..2) */
float rrto_nudge_component(float v, int32_t K, uint32_t seed, int32_t x, int32_t y, uint32_t comp);

typedef struct {             /* per-ray diagnostics, all optional */
    int32_t* steps;          /* loop iterations executed (RK4 steps taken) */
    int32_t* hit;            /* 1 if horizon */
    float* pos;              /* 3 per ray: final p */
    float* vel;              /* 3 per ray: final vel */
    float* rad;              /* 4 per ray: intensity r,g,b, transmittance */
    int32_t* n_noise;        /* noise3D evaluations */
    int32_t* n_samples;      /* RT samples accumulated (d > 0.001 block entered) */
    int32_t* n_dens;         /* density-function calls that passed the radial gate */
} rrto_diag;

void rrto_default_params(rrto_params* p);
void rrto_default_effects(rrto_effects* e);

/* ---- unit functions (array form; n elements, xyz interleaved) ---- */
void rrto_hash31(int n, const float* p, float* out);
void rrto_noise3d(int n, const float* p, float* out);
void rrto_fbm(int n, const float* p, int octaves, float* out);
void rrto_geodesic_acc(int n, const float* p, const float* v, float spin, float* out);
void rrto_rk4(int n, float* p, float* v, const float* h, float spin);
void rrto_redshift(int n, const float* p, const float* vel, float spin, int math_mode, float* out);
void rrto_disk_temperature(int n, const float* r, int math_mode, float* out);
void rrto_accretion_density(int n, const float* p, float time, int math_mode, float* out);
void rrto_dust_density(int n, const float* p, float time, int math_mode, float* out);
void rrto_smoothstep(int n, const float* e0, const float* e1, const float* x, float* out);
void rrto_lens(int n, const float* uv, float k, float* out);
void rrto_vignette(int n, const float* rgb, const float* uv, float intensity, float* out);
void rrto_bloom(int n, const float* rgb, float threshold, float* out);
void rrto_sky_sample(int n, const float* dir, float off, const uint8_t* sky, int sw, int sh,
                     int frac_bits, int math_mode, float* out_rgba);
/* the sky texel filter alone (the build's definition of tex2D<float4>) */
void rrto_sky_fetch(const uint8_t* sky, int sw, int sh, int frac_bits, float tx, float ty, float* out_rgba);
/* radiative transfer of one sample per element (raymarcher.cu:71-116); rad = 4 floats (I_rgb, T) in/out */
void rrto_rt_sample(int n, const float* d_disk, const float* d_cloud, const float* p, const float* vel,
                    const float* h, float spin, int math_mode, float* rad);
/* portable-vs-libm probes: fn 0 exp, 1 pow(x,y), 2 sin, 3 cos, 4 atan2(x=y_arg,y=x_arg), 5 asin */
void rrto_math(int fn, int math_mode, int n, const float* a, const float* b, float* out);

/*
 * Render pixels (x, y) with x0 <= x < x1, y0 <= y < y1 stepping by (sx, sy), of
 * a width x height frame.  Outputs are full-frame buffers (may be NULL):
 *   rgba8  width*height*4 bytes, bottom-up rows as raymarcher.cu:168
 *   ldr    width*height*4 floats, tone-mapped r,g,b before quantisation, a=1,
 *          same bottom-up indexing
 *   hdr    width*height*4 floats, final_hdr after post-FX, same indexing
 *   diag   arrays indexed y*width + x (top-down pixel order, NOT flipped)
 * Untouched pixels are left as they are.  Returns 0, or -1 on bad arguments.
 */
int rrto_render(const rrto_camera* cam, const rrto_effects* fx, const rrto_params* prm,
                float time, int width, int height,
                int x0, int y0, int x1, int y1, int sx, int sy,
                const uint8_t* sky_rgba8, int sky_w, int sky_h,
                uint8_t* rgba8, float* ldr, float* hdr, const rrto_diag* diag,
                int n_threads);

/*
 * The same render with the ray's hard-gate decisions logged or imposed (see `gate` in rrt_oracle.c):
 *   gate_mode 1: every decision of the s-th rendered pixel (s = jy*nx + jx over the strided sample grid of the
 *                rectangle, row-major, top-down) is appended to gate_log[s*gate_cap ...], their number goes to
 *                gate_count[s] (-1 if more than gate_cap);
 *   gate_mode 2: the decisions are READ from gate_log instead of being evaluated (gate_count[..] becomes -2 if
 *                the ray asks for a different number of decisions than were recorded);
 *   gate_mode 0: plain rrto_render.
 * Used to separate gate flips from arithmetic differences when two math libraries are compared.
 */
int rrto_render_gates(const rrto_camera* cam, const rrto_effects* fx, const rrto_params* prm,
                      float time, int width, int height,
                      int x0, int y0, int x1, int y1, int sx, int sy,
                      const uint8_t* sky_rgba8, int sky_w, int sky_h,
                      uint8_t* rgba8, float* ldr, float* hdr, const rrto_diag* diag, int n_threads,
                      int gate_mode, uint8_t* gate_log, int gate_cap, int32_t* gate_count);

int rrto_max_threads(void);

#ifdef __cplusplus
}
#endif
#endif
