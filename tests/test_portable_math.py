"""csrc/rrt_math.h (the transcendentals the HIP kernels use) against glibc, on the CPU.

Bars: a few ulp on the argument ranges the path produces -- the accuracy class CUDA documents
for its own single-precision library (the reference's real math library, which cannot be
reproduced here).  sin/cos are held to an absolute bound near their zeros, as usual.
"""
import numpy as np

from conftest import ulp_diff


def _dense(lo, hi, n=400_000, log=False, seed=0):
    rng = np.random.default_rng(seed)
    if log:
        return np.exp(rng.uniform(np.log(lo), np.log(hi), n)).astype(np.float32)
    return rng.uniform(lo, hi, n).astype(np.float32)


def test_expf(po):
    x = np.concatenate([_dense(-32, 0.5), _dense(-104, -32, 100_000), np.float32([0, -0.0, -200, 1e-8])])
    got = po.math_fn(0, po.MATH_PORTABLE, x)
    ref = np.exp(x.astype(np.float64))
    big = ref > 1.2e-38
    assert ulp_diff(got[big], ref[big]).max() <= 1.5
    assert np.all(np.abs(got[~big] - ref[~big]) <= 2e-45 + 1e-6 * ref[~big])    # denormal results


def test_powf_path_exponents(po):
    for y in (0.4, 1.6, 0.2, 1.2, 0.5, 1.5, 4.0, -0.75):
        x = _dense(1e-4, 1e3, log=True, seed=int(y * 10) + 7)
        got = po.math_fn(1, po.MATH_PORTABLE, x, np.full_like(x, y))
        ref = x.astype(np.float64) ** np.float64(np.float32(y))
        assert ulp_diff(got, ref).max() <= 2.0, y
    z = np.float32([0.0, 0.0])
    assert np.array_equal(po.math_fn(1, po.MATH_PORTABLE, z, np.float32([1.6, 4.0])), z)


def test_root_based_powers_on_the_path_and_beyond_their_range(po):
    """Round 3: exponents 0.2 / 0.4 / 1.2 / 1.6 go through one division-free fifth root (of x, x^2, x^6, x^8, the
    last two corrected for the literals 1.2f / 1.6f not being 6/5 / 8/5), -0.75 through a Newton root: tighter than
    exp(y log x) on the arguments the path produces, and anything outside their exponent windows (or denormal) takes
    the general route with the same bar."""
    path = {0.4: (0.39, 1.0), 0.2: (0.39, 1.0), 1.6: (1e-6, 2.0), 1.2: (0.3, 1.0), -0.75: (1.0, 6.5)}
    worst = {0.4: 1.4, 0.2: 1.2, 1.6: 2.0, 1.2: 1.7, -0.75: 1.4}              # measured 1.13 / 0.94 / 1.87 / 1.54 / 1.25
    for y, (lo, hi) in path.items():
        x = _dense(lo, hi, 1_000_000, log=True, seed=11)
        got = po.math_fn(1, po.MATH_PORTABLE, x, np.full_like(x, y))
        ref = x.astype(np.float64) ** np.float64(np.float32(y))
        assert ulp_diff(got, ref).max() <= worst[y], (y, ulp_diff(got, ref).max())
    edge = np.float32([1.2e-38, 1e-30, 2.0 ** -21, 2.0 ** -20, 2.0 ** -16, 2.0 ** -15, 2.0 ** 15 * 1.99, 2.0 ** 16, 2.0 ** 19.9, 2.0 ** 20,
                       1e30, 3e38, 1e-40])
    for y in path:
        got = po.math_fn(1, po.MATH_PORTABLE, edge, np.full_like(edge, y))
        ref = edge.astype(np.float64) ** np.float64(np.float32(y))
        ok = np.isfinite(ref) & (ref < 3e38) & (ref > 1.2e-38)
        assert ulp_diff(got[ok], ref[ok]).max() <= 2.0, y


def test_sincos(po):
    x = _dense(-400, 400)
    s = po.math_fn(2, po.MATH_PORTABLE, x); c = po.math_fn(3, po.MATH_PORTABLE, x)
    rs, rc = np.sin(x.astype(np.float64)), np.cos(x.astype(np.float64))
    assert np.abs(s - rs).max() <= 1.2e-7 and np.abs(c - rc).max() <= 1.2e-7
    far = np.abs(rs) > 0.05
    assert ulp_diff(s[far], rs[far]).max() <= 2.5
    far = np.abs(rc) > 0.05
    assert ulp_diff(c[far], rc[far]).max() <= 2.5


def test_atan2_asin(po):
    rng = np.random.default_rng(2)
    y = rng.uniform(-300, 300, 400_000).astype(np.float32); x = rng.uniform(-300, 300, 400_000).astype(np.float32)
    got = po.math_fn(4, po.MATH_PORTABLE, y, x)
    assert ulp_diff(got, np.arctan2(y.astype(np.float64), x.astype(np.float64))).max() <= 3.2      # measured 2.9 (one-division form, round 3)
    th = rng.uniform(-np.pi, np.pi, 400_000); rr = rng.uniform(10, 25, 400_000)       # the disk's azimuths, incl. near the axes
    for squash in (1.0, 1e-4):
        y = (rr * np.sin(th) * squash).astype(np.float32); x = (rr * np.cos(th)).astype(np.float32)
        got = po.math_fn(4, po.MATH_PORTABLE, y, x)
        assert ulp_diff(got, np.arctan2(y.astype(np.float64), x.astype(np.float64))).max() <= 3.2
    a = np.concatenate([_dense(-1, 1), np.float32([1, -1, 0, 1e-5, -1e-5])])
    got = po.math_fn(5, po.MATH_PORTABLE, a)
    assert ulp_diff(got, np.arcsin(a.astype(np.float64))).max() <= 3.0
    # axis cases of atan2
    got = po.math_fn(4, po.MATH_PORTABLE, np.float32([0, 1, -1, 0]), np.float32([1, 0, 0, -1]))
    assert np.allclose(got, [0, np.pi / 2, -np.pi / 2, np.pi], atol=3e-7)
    got = po.math_fn(4, po.MATH_PORTABLE, np.float32([-0.0, 0, -0.0, 5, -5]), np.float32([-1, 0, 0, np.inf, np.inf]))
    assert np.array_equal(got, np.float32([np.pi, 0, 0, 0, 0]))
    assert np.isnan(po.math_fn(4, po.MATH_PORTABLE, np.float32([np.nan, 1]), np.float32([1, np.nan]))).all()


def test_libm_mode_is_glibc(po):
    x = _dense(0.01, 100, 10_000, log=True)
    import ctypes as C
    import ctypes.util
    libm = C.CDLL(ctypes.util.find_library("m"))
    libm.powf.restype = C.c_float; libm.powf.argtypes = [C.c_float, C.c_float]
    want = np.float32([libm.powf(float(v), 0.4) for v in x[:2000]])
    assert np.array_equal(po.math_fn(1, po.MATH_LIBM, x[:2000], np.full(2000, 0.4, np.float32)), want)
