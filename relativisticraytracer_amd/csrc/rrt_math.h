/*
 * rrt_math.h -- portable, bit-reproducible single-precision transcendentals.
 *
 * Why this exists
 * ---------------
 * The reference hot path (/root/reference/src/raymarcher.cu:15-174 and the
 * headers it pulls in) calls powf / expf / sinf / cosf / atan2f / asinf from
 * CUDA libdevice.  Those implementations cannot be reproduced here (no CUDA),
 * glibc's differ from ROCm OCML's by an ulp here and there, and the path has
 * hard gates (`d > 0.001f`, raymarcher.cu:71,76,91; `base < 0.001f`,
 * densities.h:85) that turn 1-ulp differences into visible LSB flips.
 *
 * So the product defines ONE implementation of the six functions, written in
 * plain IEEE binary32 `+ - * /`, `sqrtf` and *explicit* `fmaf` only (no
 * compiler contraction: every translation unit that includes this header is
 * built with -ffp-contract=off).  The same source is compiled
 *   - by hipcc into the gfx950 kernels (v_fma_f32 is exact fused), and
 *   - by gcc into the CPU oracle's "portable" mode (vfmadd / libm fmaf, exact),
 * which makes the HIP output byte-identical to the oracle.  The oracle's
 * default mode keeps glibc libm so the two can be compared (tests/).
 *
 * Accuracy target: <= 4 ulp on the argument ranges the path produces, i.e. the
 * error class CUDA documents for its own single-precision library.  Measured
 * against glibc in tests/test_portable_math.py.
 *
 * Algorithms: argument reductions and minimax polynomials follow the published
 * Cephes single-precision library (S. Moshier); the general powf is exp(y*log x)
 * with the logarithm carried as an unevaluated hi+lo pair.  The five rational
 * exponents the path's call sites pass as literals (0.2f 0.4f 1.2f 1.6f -0.75f)
 * are served by division-free Newton roots instead (rrt_root5 / rrt_pow_m075
 * below: a fifth root of x, x^2, x^6 or x^8, the last two corrected to first
 * order for the literals not being 6/5 and 8/5) -- about a third of the
 * instructions and 0.9-1.9 ulp against float64 on the path's ranges; arguments
 * outside the windows those forms are proven on take the general route.
 */
#ifndef RRT_MATH_H
#define RRT_MATH_H

#include <stdint.h>

#if defined(__HIPCC__)
#define RRT_FN __host__ __device__ static __forceinline__
#else
#define RRT_FN static inline __attribute__((always_inline))
#endif

RRT_FN uint32_t rrt_f2u(float f) { uint32_t u; __builtin_memcpy(&u, &f, 4); return u; }
RRT_FN float rrt_u2f(uint32_t u) { float f; __builtin_memcpy(&f, &u, 4); return f; }
RRT_FN float rrt_fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
RRT_FN float rrt_sqrt(float x) { return __builtin_sqrtf(x); }
RRT_FN float rrt_abs(float x) { return __builtin_fabsf(x); }

/* Hook for the division of the general powf below, whose operands are tame by construction (the denominator lies
 * in [1.70, 2.42], the quotient in (0.41, 0.59)): the gfx950 kernels substitute a bare reciprocal + Markstein sequence that gives the
 * IEEE quotient bit for bit on such operands (csrc/rrt_device.h: rrt_div_tame) but skips hipcc's range scaling;
 * every other translation unit (the CPU oracle) uses the plain `/`. */
#ifndef RRT_MATH_TAME_DIV
#define RRT_MATH_TAME_DIV(a, b) ((a) / (b))
#endif

#define RRT_LN2_HI 0.693359375f        /* 10 significant bits: k*LN2_HI is exact */
#define RRT_LN2_LO (-2.12194440e-4f)   /* ln2 = LN2_HI + LN2_LO */
#define RRT_LOG2E 1.44269504088896341f
#define RRT_RND_MAGIC 12582912.0f      /* 1.5 * 2^23: x + M - M == rint(x) */

/* 2^k for k in [-126, 127] */
RRT_FN float rrt_pow2i(int k) { return rrt_u2f((uint32_t)(k + 127) << 23); }

/* y * 2^k for y in [0.5, 2), k in [-152, 129]: two exact-or-once-rounded multiplies (the first product is a normal
 * number, so only the second can round -- once, like ldexpf, into the subnormals or to infinity).  The gfx950 kernels
 * substitute the one-instruction v_ldexp_f32 (csrc/rrt_device.h), which is the same function. */
RRT_FN float rrt_scale2(float y, int k) {
    const int k1 = k >> 1, k2 = k - k1;
    return (y * rrt_pow2i(k1)) * rrt_pow2i(k2);
}
#ifndef RRT_MATH_SCALE2
#define RRT_MATH_SCALE2(y, k) rrt_scale2((y), (k))
#endif

/* exp(hi + lo) for |lo| << |hi|; shared tail of expf and powf. */
RRT_FN float rrt_exp_hl(float hi, float lo) {
    if (hi > 88.7228394f) return rrt_u2f(0x7f800000u);
    if (hi < -104.0f) return 0.0f;
    float t = rrt_fma(hi, RRT_LOG2E, RRT_RND_MAGIC);
    float k = t - RRT_RND_MAGIC;
    float r = rrt_fma(k, -RRT_LN2_HI, hi);
    r = rrt_fma(k, -RRT_LN2_LO, r);
    r = r + lo;
    float z = r * r;
    float p = 1.9875691500E-4f;
    p = rrt_fma(p, r, 1.3981999507E-3f);
    p = rrt_fma(p, r, 8.3334519073E-3f);
    p = rrt_fma(p, r, 4.1665795894E-2f);
    p = rrt_fma(p, r, 1.6666665459E-1f);
    p = rrt_fma(p, r, 5.0000001201E-1f);
    float y = rrt_fma(p, z, r) + 1.0f;
    return RRT_MATH_SCALE2(y, (int)k);
}

RRT_FN float rrt_expf(float x) {
    if (x != x) return x;
    return rrt_exp_hl(x, 0.0f);
}

/* x^y for finite x > 0 via exp(y*log(x)), log carried as hi+lo. */
RRT_FN float rrt_pow_pos(float x, float y) {
    int e0 = 0;
    uint32_t ix = rrt_f2u(x);
    if (ix < 0x00800000u) { x = x * 16777216.0f; e0 = -24; ix = rrt_f2u(x); }
    uint32_t adj = ix - 0x3f3504f3u;                 /* m in [sqrt(1/2), sqrt(2)) */
    int e = ((int32_t)adj >> 23) + e0;
    float m = rrt_u2f((adj & 0x007fffffu) + 0x3f3504f3u);
    float f = m - 1.0f;                              /* exact */
    float t = 2.0f + f;
    float t_lo = (2.0f - t) + f;                     /* exact rounding error of t */
    float rt = RRT_MATH_TAME_DIV(1.0f, t);             /* t in [1.70, 2.42] */
    float s = f * rt;
    float res = rrt_fma(-s, t, f);
    res = rrt_fma(-s, t_lo, res);
    float s_lo = res * rt;
    float z = s * s;
    /* log(m) = 2 atanh(s) = 2s + s*z*(2/3 + 2/5 z + 2/7 z^2 + 2/9 z^3 + 2/11 z^4) */
    float q = rrt_fma(z, 0.181818187f, 0.222222224f);
    q = rrt_fma(q, z, 0.285714298f);
    q = rrt_fma(q, z, 0.400000006f);
    q = rrt_fma(q, z, 0.666666687f);
    float R = (s * z) * q;
    float hi = 2.0f * s;
    float lo = rrt_fma(2.0f, s_lo, R);
    float ef = (float)e;
    float a = ef * RRT_LN2_HI;                       /* exact */
    float L_hi = a + hi;                             /* TwoSum */
    float bb = L_hi - a;
    float err = (a - (L_hi - bb)) + (hi - bb);
    float L_lo = err + lo;
    L_lo = rrt_fma(ef, RRT_LN2_LO, L_lo);
    float P_hi = y * L_hi;
    float P_lo = rrt_fma(y, L_hi, -P_hi);
    P_lo = rrt_fma(y, L_lo, P_lo);
    return rrt_exp_hl(P_hi, P_lo);
}

/*
 * Round 3: the path's other exponents are fifths and -3/4 (0.2, 0.4, 1.2, 1.6: densities.h:34,58,80, raymarcher.cu:82,93;
 * -0.75: densities.h:14) -- roots, not general powers.  x^(k/5) = fifth root of x^k, and a fifth root divides the
 * relative error of its argument by five, so a few multiplies plus ONE accurate root beat exp(y log x) on both counts:
 * ~30 instructions instead of ~85, and 1.0-1.9 ulp instead of 2 (measured against float64 over [1e-4, 1e3],
 * tests/test_portable_math.py).  Division-free, no table, integer seed:
 *   rrt_root5:     z ~ x^(-1/5) from the bit trick (3 % off), three Newton steps z <- z (1.2 - 0.2 x z^5) (quadratic),
 *                  w = x z^4, then one Newton step on w^5 = x whose residual w^4*w - x is formed in one fma:
 *                  w - (w^5 - x) z^4 / 5.  <= 1 ulp over the whole normal range.
 *   rrt_pow_m075:  y ~ x^(-3/4) from the bit trick (4 % off), two steps y <- y (1.25 - 0.25 x^3 y^4), then the same
 *                  kind of fma-residual correction.  <= 1.3 ulp.
 * Both need x normal and the intermediate powers in range; rrt_powf checks the exponent field and sends anything else
 * (and every other exponent) through rrt_pow_pos.
 */
RRT_FN float rrt_root5(float x) {                      /* x normal, > 0 */
    float z = rrt_u2f(0x4c2ba000u - rrt_f2u(x) / 5u);
    const float x02 = x * 0.2f;
    for (int k = 0; k < 3; ++k) {
        const float z2 = z * z, z4 = z2 * z2, z5 = z4 * z;
        z = z * rrt_fma(-x02, z5, 1.2f);
    }
    const float z2 = z * z, z4 = z2 * z2;
    const float w = x * z4;
    const float w2 = w * w, w4 = w2 * w2;
    const float e = rrt_fma(w4, w, -x);
    return rrt_fma(-e, z4 * 0.2f, w);
}

RRT_FN float rrt_pow_m075(float x) {                   /* x in 2^[-20, 20] */
    const uint32_t i = rrt_f2u(x);
    float y = rrt_u2f(0x6f150000u - (i - (i >> 2)));
    const float u = (x * x) * x;
    const float u025 = u * 0.25f;
    for (int k = 0; k < 2; ++k) {
        const float y2 = y * y, y4 = y2 * y2;
        y = y * rrt_fma(-u025, y4, 1.25f);
    }
    const float y2 = y * y, y4 = y2 * y2;
    const float r = rrt_fma(-u, y4, 1.0f);
    return rrt_fma(y * 0.25f, r, y);
}

/* is x a normal float with unbiased exponent in [lo, hi)?  (one subtract + one unsigned compare) */
RRT_FN int rrt_exp_in(float x, int lo, int hi) {
    return (rrt_f2u(x) - ((uint32_t)(lo + 127) << 23)) < ((uint32_t)(hi - lo) << 23);
}

/*
 * powf as the path uses it: base >= 0, finite, and the exponent is one of a
 * handful of literals (geodesics.h:17, densities.h:14,32,34,41,58,80,89,125,
 * raymarcher.cu:79,80,82,93).  Exponents with an exact radical form are
 * evaluated through sqrt/multiplies (<= 2 ulp, usually correctly rounded), fifths and
 * -3/4 through the roots above; everything else goes through rrt_pow_pos.  Negative bases do not occur.
 */
RRT_FN float rrt_powf(float x, float y) {
    if (x != x) return x;
    if (x == 0.0f) return (y > 0.0f) ? 0.0f : ((y == 0.0f) ? 1.0f : rrt_u2f(0x7f800000u));
    if (x == rrt_u2f(0x7f800000u)) return (y > 0.0f) ? x : ((y == 0.0f) ? 1.0f : 0.0f);
    if (y == 0.5f) return rrt_sqrt(x);
    if (y == 1.5f) return x * rrt_sqrt(x);
    if (y == 4.0f) { float x2 = x * x; return x2 * x2; }
    if (y == 0.2f && rrt_exp_in(x, -20, 20)) return rrt_root5(x);
    if (y == 0.4f && rrt_exp_in(x, -20, 20)) return rrt_root5(x * x);
    /* The literals 1.2f and 1.6f are not 6/5 and 8/5: they exceed them by d = 4.77e-8 and 2.38e-8, and the reference's
     * powf raises to the literal.  x^d = 1 + d ln x to first order, with ln x good to 0.06 from the exponent and
     * mantissa bits read as an integer; applied to the root's ARGUMENT (as 1 + 5 d ln x, so that its rounding is divided
     * by five too) it keeps the result within 2 ulp of powf(x, 1.6f) for small x as well (without it: up to 2 ulp
     * more at x = 1e-4).  0.2f and 0.4f are off by 3e-9 and 6e-9: nothing to correct inside 2^[-20, 20]. */
    if (y == 1.2f && rrt_exp_in(x, -20, 20)) {
        const float x3 = (x * x) * x, x6 = x3 * x3;
        const float lnx = (float)(int32_t)(rrt_f2u(x) - 0x3f800000u) * 8.26295829e-8f;      /* ln 2 / 2^23 */
        return rrt_root5(rrt_fma(x6 * 2.38418579e-7f, lnx, x6));                            /* 5 * 4.76837158e-8 */
    }
    if (y == 1.6f && rrt_exp_in(x, -15, 15)) {
        const float x2 = x * x, x4 = x2 * x2, x8 = x4 * x4;
        const float lnx = (float)(int32_t)(rrt_f2u(x) - 0x3f800000u) * 8.26295829e-8f;
        return rrt_root5(rrt_fma(x8 * 1.19209290e-7f, lnx, x8));                            /* 5 * 2.38418579e-8 */
    }
    if (y == -0.75f && rrt_exp_in(x, -20, 20)) return rrt_pow_m075(x);
    return rrt_pow_pos(x, y);
}

/* sin and cos of x together (|x| < ~8000 for full accuracy). */
RRT_FN void rrt_sincosf(float x, float* sn, float* cs) {
    float ax = rrt_abs(x);
    if (!(ax < 1.0e9f)) {                            /* NaN, infinity, or past the int conversion below (undefined in C, and
                                                        different on CPUs and the GPU): NaN for NaN / infinity, (0, 1) otherwise */
        const float nan_or_zero = x - x;
        *sn = nan_or_zero; *cs = nan_or_zero + 1.0f;
        return;
    }
    int j = (int)(ax * 1.27323954473516f);           /* 4/pi */
    j = (j + 1) & ~1;
    float y = (float)j;
    float r = rrt_fma(y, -0.78515625f, ax);
    r = rrt_fma(y, -2.4187564849853515625e-4f, r);
    r = rrt_fma(y, -3.77489497744594108e-8f, r);
    float z = r * r;
    float ps = -1.9515295891E-4f;
    ps = rrt_fma(ps, z, 8.3321608736E-3f);
    ps = rrt_fma(ps, z, -1.6666654611E-1f);
    ps = rrt_fma(ps * z, r, r);
    float pc = 2.443315711809948E-005f;
    pc = rrt_fma(pc, z, -1.388731625493765E-003f);
    pc = rrt_fma(pc, z, 4.166664568298827E-002f);
    pc = rrt_fma(pc * z, z, rrt_fma(-0.5f, z, 1.0f));
    int q = (j >> 1) & 3;
    float s_val = (q & 1) ? pc : ps;
    float c_val = (q & 1) ? ps : pc;
    int s_neg = ((q & 2) != 0) ^ (x < 0.0f);
    int c_neg = (q == 1) | (q == 2);
    *sn = s_neg ? -s_val : s_val;
    *cs = c_neg ? -c_val : c_val;
}
RRT_FN float rrt_sinf(float x) { float s, c; rrt_sincosf(x, &s, &c); return s; }
RRT_FN float rrt_cosf(float x) { float s, c; rrt_sincosf(x, &s, &c); return c; }

/*
 * atan2 with ONE division (round 3).  With mx = max(|x|, |y|), mn = min(|x|, |y|) the angle of the first-octant
 * point (mx, mn) is atan(mn / mx) in [0, pi/4]; Cephes' reduction at tan(pi/8) -- atan(s) = pi/4 + atan((s - 1)/(s + 1))
 * for s > 0.4142 -- is applied to numerator and denominator BEFORE they are divided ((mn - mx) / (mn + mx)), so the
 * quotient y/x, the second quotient of the reduction and the -1/s of the steep case (three IEEE divisions on three
 * divergent branches in the textbook form) collapse into a single t = num / den with |t| <= tan(pi/8), followed by
 * Cephes' degree-4 minimax in t^2.  The octant is put back by pi/2 - r (|y| > |x|), pi - r (x < 0) and the sign of y.
 * <= 3 ulp against float64 on the path's arguments (tests/test_portable_math.py; the form it replaces: <= 4).
 * Axis cases as before: atan2(+-0, x > 0) = 0, atan2(+-0, x < 0) = +pi, atan2(y, 0) = +-pi/2, atan2(0, 0) = 0.
 */
RRT_FN float rrt_atan2f(float y, float x) {
    if (x != x || y != y) return x + y;
    const float ax = rrt_abs(x), ay = rrt_abs(y);
    const int steep = ay > ax;
    const float mx = steep ? ay : ax, mn = steep ? ax : ay;
    if (mx == 0.0f) return 0.0f;
    const int fold = mn > 0.4142135623730950f * mx;
    const float num = fold ? mn - mx : mn;
    const float den = fold ? mn + mx : mx;
    const float t = num / den;
    const float z = t * t;
    float p = 8.05374449538e-2f;
    p = rrt_fma(p, z, -1.38776856032E-1f);
    p = rrt_fma(p, z, 1.99777106478E-1f);
    p = rrt_fma(p, z, -3.33329491539E-1f);
    float r = (fold ? 0.7853981633974483f : 0.0f) + rrt_fma(p * z, t, t);
    if (steep) r = 1.5707963267948966f - r;
    if (x < 0.0f) r = 3.14159265358979323846f - r;
    return (y < 0.0f) ? -r : r;
}

/* asin for |x| <= 1 */
RRT_FN float rrt_asinf(float x) {
    float a = rrt_abs(x);
    if (!(a <= 1.0f)) return rrt_u2f(0x7fc00000u);
    if (a < 1.0e-4f) return x;
    int flag = a > 0.5f;
    float z, t;
    if (flag) { z = 0.5f * (1.0f - a); t = rrt_sqrt(z); }
    else { t = a; z = t * t; }
    float p = 4.2163199048E-2f;
    p = rrt_fma(p, z, 2.4181311049E-2f);
    p = rrt_fma(p, z, 4.5470025998E-2f);
    p = rrt_fma(p, z, 7.4953002686E-2f);
    p = rrt_fma(p, z, 1.6666752422E-1f);
    float r = rrt_fma(p * z, t, t);
    if (flag) { r = r + r; r = 1.5707963267948966f - r; }
    return (x < 0.0f) ? -r : r;
}

#endif /* RRT_MATH_H */
