/*
 * rrt_test.h -- test hooks of the MI355X ray-march library: NOT part of the product.
 *
 * librrt_hip.so exports none of these.  librrt_hip_test.so is the same sources built with -DRRT_TEST_HOOKS
 * (relativisticraytracer_amd/build.py): everything include/rrt.h declares plus the entry points below -- the device
 * functions of the path on arrays (parity tests against the oracle and the reference's vectors), self-checks of the
 * hand-rolled square-root / division cores against the hardware IEEE forms, and a hook that lets a test pretend another
 * HIP device is current.  The reference has no counterpart (it has no tests: SURVEY.md section 4).
 */
#ifndef RRT_TEST_H
#define RRT_TEST_H

#include "rrt.h"

#ifdef __cplusplus
extern "C" {
#endif

/* pretend `device` is the current HIP device in every handle check (< 0: ask HIP again).  Works only in a process that
 * was started with RRT_ENABLE_TEST_HOOKS=1 in its environment (RRT_ERR_INVALID_ARGUMENT otherwise). */
int rrt_debug_fake_device(int device);

/* ---- unit kernels: the device functions of the path on arrays, for parity
 *      tests against the oracle (device pointers, n elements, xyz interleaved). ---- */
int rrt_unit_geodesic_acc(int n, const float* d_p, const float* d_v, float spin, float* d_out, void* stream);
int rrt_unit_rk4(int n, float* d_p, float* d_v, const float* d_h, float spin, void* stream);
/* the PRODUCTION RK4 step (csrc/rrt_device.h: integrate_rk4_lean -- what the render kernels run instead of the
 * literal integrators.h:23-59) as a chain of n_steps steps per element, driven as the march drives it: loop-top radius
 * from the seed pair the previous step handed on, v_rsq fall-back on a rejected seed, horizon test (a ray whose
 * loop-top radius is < 2.02 stops; d_steps, may be NULL, receives the steps taken).  d_h == NULL: step size by the
 * march's zone rule (raymarcher.cu:56-62) and the wave-uniform vacuum step wherever a whole wavefront (64 consecutive
 * elements) is at r >= 30; d_h != NULL: the generic step with h = d_h[i] on every step.  seed_scale: the first root is
 * seeded with seed_scale / r (0: no seed, like a ray's first step). */
int rrt_unit_rk4_lean(int n, float* d_p, float* d_v, const float* d_h, float spin, int n_steps, float seed_scale,
                      int32_t* d_steps, void* stream);
/* the march's divide (csrc/rrt_device.h: div_seeded, one Markstein correction) on explicit operands and seeds */
int rrt_unit_div_seeded(int n, const float* d_a, const float* d_b, const float* d_seed, float* d_out, void* stream);
int rrt_unit_hash31(int n, const float* d_p, float* d_out, void* stream);
int rrt_unit_noise3d(int n, const float* d_p, float* d_out, void* stream);
int rrt_unit_fbm(int n, const float* d_p, int octaves, float* d_out, void* stream);
int rrt_unit_accretion_density(int n, const float* d_p, float time, float* d_out, void* stream);
int rrt_unit_dust_density(int n, const float* d_p, float time, float* d_out, void* stream);
int rrt_unit_redshift(int n, const float* d_p, const float* d_vel, float spin, float* d_out, void* stream);
int rrt_unit_math(int fn, int n, const float* d_a, const float* d_b, float* d_out, void* stream);
int rrt_unit_sky_sample(int n, const float* d_dir, float off, rrt_sky_t sky, int frac_bits,
                        float* d_out_rgba, void* stream);
int rrt_unit_disk_temperature(int n, const float* d_r, float* d_out, void* stream);          /* densities.h:12-15 */
int rrt_unit_smoothstep(int n, const float* d_e0, const float* d_e1, const float* d_x, float* d_out, void* stream);
/* post_processing.h:13-31.  what = 0: apply_lens_distortion (uv[2n] -> out[2n], param = k);
 * 1: apply_vignette (rgb[3n], uv[2n] -> out[3n], param = intensity); 2: get_bloom_contribution (rgb -> out, param = threshold) */
int rrt_unit_postfx(int what, int n, const float* d_rgb, const float* d_uv, float param, float* d_out, void* stream);
/* the radiative-transfer block raymarcher.cu:71-116, one sample per element; d_rad = n x (I_r, I_g, I_b, T), in/out */
int rrt_unit_rt_sample(int n, const float* d_disk, const float* d_cloud, const float* d_p, const float* d_vel,
                       const float* d_h, float spin, float* d_rad, void* stream);
/* noise3D read through a noise table (which = 0 accretion box, 1 dust box); d_counts[0] (may be NULL) counts
 * reads outside the box.  And both density functions as the render kernels evaluate them (table switches on). */
int rrt_unit_noise3d_lut(int n, const float* d_p, int table, int which, float* d_out, unsigned* d_counts, void* stream);
int rrt_unit_media_lut(int n, const float* d_p, float time, int table, float* d_out_disk, float* d_out_dust,
                       unsigned* d_counts, void* stream);

/* Self-checks of the march loop's hand-rolled correctly-rounded sqrt / divide against the
 * hardware IEEE forms.  d_counters: 4 x uint64 on the device, zeroed by the caller;
 * [0] receives the number of mismatching cases, [1..3] one failing case. */
int rrt_selfcheck_sqrt(uint32_t lo_bits, uint32_t hi_bits, unsigned long long* d_counters, void* stream);
int rrt_selfcheck_div(unsigned long long n_cases, uint32_t seed, unsigned long long* d_counters, void* stream);
/* the march's seeded roots and the same two divides with the reciprocal-root seeds those roots hand on: roots out of
 * sqrt_seeded_yh<1> / <2> started from estimates off by up to +-tol1 / +-tol2 relative (the forms' acceptance tolerances are
 * 1.5e-4 / 9e-3; extrapolated seeds included); rejected roots are skipped.  d_counters: EIGHT uint64 -- [0] / [1] accepted
 * one- / two-iteration roots that are not the correctly rounded root, [2] divide mismatches, [3] divides checked, [4..7] one
 * failing root (x, seed bits) and one failing divide (numerator, denominator bits) */
int rrt_selfcheck_div_march(unsigned long long n_cases, uint32_t seed, float tol1, float tol2, unsigned long long* d_counters, void* stream);
/* the march's transcendental-free square root (csrc/rrt_device.h: sqrt_seeded) over a range of float bit patterns
 * and a ladder of seed errors; d_counters[3] receives the number of accepted (checked) cases */
int rrt_selfcheck_sqrt_seeded(uint32_t lo_bits, uint32_t hi_bits, unsigned long long* d_counters, void* stream);
/* the seeded roots on the floats within `span` ulps of every power of two 2^e, e in [e_lo, e_hi) -- where sqrt(x) comes closest
 * to a rounding tie -- under a dense sweep of n_seeds seeds per x and form over +-tol1 / +-tol2.  d_counters: SIX uint64 -- [0] / [1]
 * accepted one- / two-iteration roots that differ from sqrtf, [2] accepted roots checked, [3] rejected, [4]/[5] one failing case */
int rrt_selfcheck_sqrt_boundaries(int e_lo, int e_hi, int span, unsigned n_seeds, float tol1, float tol2,
                                  unsigned long long* d_counters, void* stream);
/* the media code's three-instruction division by a compile-time constant (csrc/rrt_device.h: rrt_div_const): every
 * dividend with bits in [lo_bits, hi_bits), both signs, for each constant the media code divides by */
int rrt_selfcheck_div_const(uint32_t lo_bits, uint32_t hi_bits, unsigned long long* d_counters, void* stream);
/* the media code's scaling-free division (csrc/rrt_device.h: rrt_div_tame) on random tame operand pairs */
int rrt_selfcheck_div_tame(unsigned long long n_cases, uint32_t seed, unsigned long long* d_counters, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* RRT_TEST_H */
