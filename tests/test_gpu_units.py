"""HIP device functions (through the C ABI's unit kernels) against the CPU oracle.

Arithmetic-only functions are compared with the oracle directly (they are also
pinned to the reference by tests/test_oracle_units.py); functions that contain
transcendentals are compared with the oracle in PORTABLE math mode, where both
sides run csrc/rrt_math.h -- the bar is bit equality (+-0 aside).
"""
import numpy as np
import pytest

from conftest import same_bits

pytestmark = pytest.mark.gpu

SPINS = (0.0, 0.9, 0.99)
TIMES = (0.0, 1.0, 12.5)


@pytest.fixture(scope="module")
def g():
    import torch
    assert torch.cuda.is_available(), "the -m gpu tests need a GPU"
    import gpu_util
    return gpu_util


@pytest.mark.parametrize("spin", SPINS)
def test_geodesic_acc_matches_reference_vectors(g, units_ref, spin):
    import torch
    p, v = g.dev(units_ref["geo_p"]), g.dev(units_ref["geo_v"])
    out = torch.empty_like(p)
    g.unit("geodesic_acc", len(units_ref["geo_p"]), p, v, float(spin), out)
    assert same_bits(g.host(out), units_ref[f"geodesic_acc_a{spin:g}"])


@pytest.mark.parametrize("spin", SPINS)
def test_rk4_matches_reference_vectors(g, units_ref, spin):
    p, v, h = g.dev(units_ref["geo_p"]), g.dev(units_ref["geo_v"]), g.dev(units_ref["rk4_h"])
    g.unit("rk4", len(units_ref["rk4_h"]), p, v, h, float(spin))
    assert same_bits(g.host(p), units_ref[f"rk4_p_a{spin:g}"])
    assert same_bits(g.host(v), units_ref[f"rk4_v_a{spin:g}"])


def test_hash_noise_fbm_match_reference_vectors(g, units_ref):
    import torch
    n = len(units_ref["lattice"])
    lat, pts = g.dev(units_ref["lattice"]), g.dev(units_ref["noise_p"])
    out = torch.empty(n, device="cuda")
    g.unit("hash31", n, lat, out);      assert same_bits(g.host(out), units_ref["hash31"])
    g.unit("noise3d", n, pts, out);     assert same_bits(g.host(out), units_ref["noise3d"])
    g.unit("fbm", n, pts, 2, out);      assert same_bits(g.host(out), units_ref["fbm2"])
    g.unit("fbm", n, pts, 5, out);      assert same_bits(g.host(out), units_ref["fbm5"])


def test_hash_noise_large_random(g, po):
    """1M random points, incl. large coordinates, against the oracle."""
    import torch
    rng = np.random.default_rng(3)
    pts = (rng.uniform(-1, 1, (1 << 20, 3)) * np.exp(rng.uniform(0, 9, (1 << 20, 1)))).astype(np.float32)
    out = torch.empty(len(pts), device="cuda")
    g.unit("noise3d", len(pts), g.dev(pts), out)
    assert same_bits(g.host(out), po.units().noise3d(pts))


def test_portable_math_bitexact(g, po):
    import torch
    rng = np.random.default_rng(5)
    n = 1 << 18
    cases = [
        (0, rng.uniform(-110, 5, n), None),
        (1, np.exp(rng.uniform(np.log(1e-6), np.log(1e4), n)), rng.choice([0.4, 1.6, 0.2, 1.2, 0.5, 1.5, 4.0, -0.75], n)),
        (2, rng.uniform(-400, 400, n), None),
        (3, rng.uniform(-400, 400, n), None),
        (4, rng.uniform(-300, 300, n), rng.uniform(-300, 300, n)),
        (5, rng.uniform(-1, 1, n), None),
    ]
    for fn, a, b in cases:
        a = a.astype(np.float32); b = (b if b is not None else np.zeros(n)).astype(np.float32)
        out = torch.empty(n, device="cuda")
        g.unit("math", fn, n, g.dev(a), g.dev(b), out)
        assert same_bits(g.host(out), po.math_fn(fn, po.MATH_PORTABLE, a, b)), f"fn {fn}"


@pytest.mark.parametrize("t", TIMES)
def test_densities_bitexact_vs_portable_oracle(g, po, units_ref, t):
    import torch
    n = len(units_ref["disk_p"])
    out = torch.empty(n, device="cuda")
    g.unit("accretion_density", n, g.dev(units_ref["disk_p"]), float(t), out)
    acc = g.host(out).copy()
    assert same_bits(acc, po.units().accretion_density(units_ref["disk_p"], t, po.MATH_PORTABLE))
    g.unit("dust_density", n, g.dev(units_ref["cloud_p"]), float(t), out)
    dust = g.host(out)
    assert same_bits(dust, po.units().dust_density(units_ref["cloud_p"], t, po.MATH_PORTABLE))
    # ... and close to the reference's own values (glibc transcendentals).  The density functions are
    # ill-conditioned in their transcendentals: a 1-ulp change of sinf/cosf at angle ~ time*omega moves
    # the noise coordinates by ~1e-6, the octaves scale that by up to 17.7, and pow(n - 0.32, 1.6)
    # amplifies it again when n ~ 0.32 -- so individual samples differ by up to ~1e-3 relative between ANY
    # two libms (the per-pixel integral damps this, see test_gpu_frames.py).  Bars: >= 97 % of points
    # within 1e-4 relative, all within 5e-3.
    ref_acc, ref_dust = units_ref[f"accretion_t{t:g}"], units_ref[f"dust_t{t:g}"]
    for got, ref in ((acc, ref_acc), (dust, ref_dust)):
        err = np.abs(got - ref)
        assert (err <= 1e-4 * np.abs(ref) + 1e-6).mean() >= 0.97
        assert np.all(err <= 5e-3 * np.abs(ref) + 1e-5)


@pytest.mark.parametrize("spin", SPINS)
def test_redshift_bitexact_vs_portable_oracle(g, po, units_ref, spin):
    import torch
    n = len(units_ref["disk_p"])
    out = torch.empty(n, device="cuda")
    g.unit("redshift", n, g.dev(units_ref["disk_p"]), g.dev(units_ref["geo_v"]), float(spin), out)
    got = g.host(out)
    assert same_bits(got, po.units().redshift(units_ref["disk_p"], units_ref["geo_v"], spin, po.MATH_PORTABLE))
    ref = units_ref[f"redshift_disk_a{spin:g}"]
    assert np.all(np.abs(got - ref) <= 1e-5 * np.abs(ref))


def test_sky_sampler_bitexact(g, po, sky):
    import torch
    import relativisticraytracer_amd as rrt
    rng = np.random.default_rng(11)
    d = rng.normal(size=(1 << 16, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
    d = d.astype(np.float32)
    d[:4] = [[0, 1, 0], [0, -1, 0], [-1, 0, 0], [1, 0, 0]]     # poles and the +-pi seam
    tex = rrt.SkyTexture(sky)
    for off in (0.0, 0.005, -0.005):
        for bits in (8, 0):
            out = torch.empty(len(d) * 4, device="cuda")
            g.unit("sky_sample", len(d), g.dev(d), float(off), tex.handle, bits, out)
            want = po.sky_sample(d, off, sky, bits, po.MATH_PORTABLE)
            assert same_bits(g.host(out).reshape(-1, 4), want), (off, bits)
    tex.destroy()


def test_fast_sqrt_is_correctly_rounded_everywhere_it_is_used(g):
    """sqrt_rsq (rsq + Newton + residual fix-up) == IEEE sqrtf for EVERY float in [1, 2^64)."""
    import ctypes as C
    import torch
    from relativisticraytracer_amd import _lib
    cnt = torch.zeros(4, dtype=torch.int64, device="cuda")
    lo, hi = 0x3f800000, 0x3f800000 + (64 << 23)
    _lib.check(_lib.load().rrt_selfcheck_sqrt(lo, hi, C.c_void_p(cnt.data_ptr()), None), "selfcheck_sqrt")
    torch.cuda.synchronize()
    assert int(cnt[0]) == 0, f"{int(cnt[0])} mismatches, e.g. bits {int(cnt[1]):#x}"


def test_fast_divide_is_correctly_rounded_on_march_operands(g):
    """div_seeded (Markstein core seeded from powers of 1/r) == IEEE `/` on 2^32 march-shaped cases."""
    import ctypes as C
    import torch
    from relativisticraytracer_amd import _lib
    cnt = torch.zeros(4, dtype=torch.int64, device="cuda")
    _lib.check(_lib.load().rrt_selfcheck_div(1 << 32, 12345, C.c_void_p(cnt.data_ptr()), None), "selfcheck_div")
    torch.cuda.synchronize()
    assert int(cnt[0]) == 0, f"{int(cnt[0])} mismatches, e.g. {int(cnt[1]):#x} / {int(cnt[2]):#x}"


def test_unit_kernels_empty_and_bad_args(g):
    import torch
    from relativisticraytracer_amd import _lib
    z = torch.empty(0, device="cuda")
    g.unit("hash31", 0, z, z)                         # n = 0 is a no-op
    assert _lib.load().rrt_unit_hash31(4, None, None, None) == 1        # RRT_ERR_INVALID_ARGUMENT
    assert _lib.load().rrt_unit_fbm(4, z.data_ptr(), 99, z.data_ptr(), None) == 1
