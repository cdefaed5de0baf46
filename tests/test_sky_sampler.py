"""The sky sampler is the one piece of the path the reference leaves to hardware (tex2D with a
linear, wrap-x / clamp-y, normalized-coordinate texture object, src/main.cpp:255-261).  The build's
definition is checked here against an independent float64 numpy statement of
the same rule, and for the properties a bilinear filter must have."""
import numpy as np


def _numpy_sampler(dirs, off, sky, frac_bits):
    d = dirs.astype(np.float64)
    phi = np.arctan2(d[:, 2], d[:, 0]) + off
    theta = np.arcsin(np.clip(d[:, 1], -1, 1))
    pi = float(np.float32(3.1415926535))
    tx = 0.5 + phi / (2 * pi)
    ty = 0.5 - theta / pi
    h, w = sky.shape[:2]
    xb, yb = tx * w - 0.5, ty * h - 0.5
    i, j = np.floor(xb), np.floor(yb)
    a, b = xb - i, yb - j
    if frac_bits:
        q = float(1 << frac_bits)
        a, b = np.floor(a * q + 0.5) / q, np.floor(b * q + 0.5) / q
    i0, i1 = np.mod(i, w).astype(int), np.mod(i + 1, w).astype(int)
    j0, j1 = np.clip(j, 0, h - 1).astype(int), np.clip(j + 1, 0, h - 1).astype(int)
    t = sky.astype(np.float64) / 255.0
    return ((1 - a) * (1 - b))[:, None] * t[j0, i0] + (a * (1 - b))[:, None] * t[j0, i1] + \
           ((1 - a) * b)[:, None] * t[j1, i0] + (a * b)[:, None] * t[j1, i1]


def test_matches_float64_statement(po, sky):
    rng = np.random.default_rng(4)
    d = rng.normal(size=(20000, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
    d = d.astype(np.float32)
    for off in (0.0, 0.005, -0.005):
        for bits in (0, 8):
            got = po.sky_sample(d, off, sky, bits, po.MATH_LIBM)
            want = _numpy_sampler(d, off, sky, bits)
            err = np.abs(got - want)
            # float32 evaluation of the texture coordinate moves a sample by <= ~1e-3 texel; with 8-bit
            # weights a sample may land in the neighbouring weight bucket (1/256 of a texel difference)
            tol = 2e-5 if bits == 0 else 5e-3
            assert np.percentile(err, 99) <= tol and err.max() <= 0.05, (off, bits, err.max())


def test_filter_properties(po):
    flat = np.full((16, 32, 4), 77, np.uint8)
    d = np.float32([[1, 0, 0], [0, 1, 0], [0, -1, 0], [-1, 0, 1e-7], [-1, 0, -1e-7], [0.3, 0.5, -0.8]])
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    out = po.sky_sample(d, 0.0, flat, 8, po.MATH_LIBM)
    assert np.allclose(out, 77 / 255.0, atol=1e-6)            # constant image -> constant, also at the poles and seam
    # wrap in x: the two directions either side of the +-pi seam see the same two columns
    ramp = np.zeros((8, 64, 4), np.uint8); ramp[..., 0] = (np.arange(64) * 4)[None, :]
    o = po.sky_sample(d[3:5], 0.0, ramp, 0, po.MATH_LIBM)[:, 0]
    assert abs(o[0] - o[1]) < 1e-3 and abs(o[0] - 0.5 * (252 + 0) / 255.0) < 1e-3
    # clamp in y: straight up / down read the first / last row only
    rows = np.zeros((8, 16, 4), np.uint8); rows[0] = 10; rows[-1] = 200
    o = po.sky_sample(np.float32([[0, 1, 0], [0, -1, 0]]), 0.0, rows, 0, po.MATH_LIBM)[:, 0]
    assert abs(o[0] - 10 / 255.0) < 1e-6 and abs(o[1] - 200 / 255.0) < 1e-6


def test_portable_and_libm_samplers_agree(po, sky):
    rng = np.random.default_rng(9)
    d = rng.normal(size=(5000, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
    a = po.sky_sample(d.astype(np.float32), 0.004, sky, 8, po.MATH_LIBM)
    b = po.sky_sample(d.astype(np.float32), 0.004, sky, 8, po.MATH_PORTABLE)
    assert np.percentile(np.abs(a - b), 99.5) <= 5e-3 and (a == b).mean() > 0.97
