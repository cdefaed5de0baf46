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


@pytest.mark.parametrize("spin", SPINS)
def test_production_rk4_step_matches_reference_vectors(g, units_ref, spin):
    """integrate_rk4_lean -- the step every render kernel runs (seeded Goldschmidt roots, one-correction Markstein
    divides, guards behind the fall-back) -- on the reference's one-step vectors, with no seed, a good seed, an
    imperfect seed that is accepted and a bad one that must be rejected.  The lean step's precondition is the march's:
    the loop-top radius has passed the horizon test (r >= 2.02), so the vectors inside it are left to `rrt_unit_rk4`."""
    import torch
    r = np.linalg.norm(units_ref["geo_p"].astype(np.float64), axis=1)
    sel = r > 2.03
    assert sel.sum() > 900
    for seed_scale in (0.0, 1.0, 1.00005, 0.9999, 1.3, 0.5):
        p, v, h = g.dev(units_ref["geo_p"][sel]), g.dev(units_ref["geo_v"][sel]), g.dev(units_ref["rk4_h"][sel])
        steps = torch.zeros(int(sel.sum()), dtype=torch.int32, device="cuda")
        g.unit("rk4_lean", int(sel.sum()), p, v, h, float(spin), 1, float(seed_scale), steps)
        assert same_bits(g.host(p), units_ref[f"rk4_p_a{spin:g}"][sel]), seed_scale
        assert same_bits(g.host(v), units_ref[f"rk4_v_a{spin:g}"][sel]), seed_scale
        assert (g.host(steps) == 1).all()


@pytest.mark.parametrize("spin", SPINS)
def test_production_rk4_chain_matches_the_reference_chain(g, rk4_chain_ref, spin):
    """50 steps of the production step, driven as the march drives it (zone rule, wave-uniform vacuum step with
    extrapolated seeds on the four far-out wavefronts, generic step elsewhere, seeds carried from step to step,
    horizon test), against chains of the REFERENCE's integrate_rk4 (integrators.h:23-59): bit-exact at every mark."""
    import torch
    c = rk4_chain_ref
    n = len(c["p0"])
    for seed_scale in (0.0, 1.3):
        for k in (int(m) for m in c["marks"]):
            p, v = g.dev(c["p0"]), g.dev(c["v0"])
            steps = torch.zeros(n, dtype=torch.int32, device="cuda")
            g.unit("rk4_lean", n, p, v, None, float(spin), k, float(seed_scale), steps)
            assert same_bits(g.host(p), c[f"p_a{spin:g}_k{k}"]), (k, seed_scale)
            assert same_bits(g.host(v), c[f"v_a{spin:g}_k{k}"]), (k, seed_scale)
            assert np.array_equal(g.host(steps), c[f"steps_a{spin:g}_k{k}"]), (k, seed_scale)


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
        (0, rng.uniform(80, 92, n), None),                                        # up to and past the overflow of the 2^k tail (v_ldexp_f32)
        (1, np.exp(rng.uniform(np.log(1e-30), np.log(1e30), n)), rng.choice([0.4, 1.6, 0.2, 1.2, -0.75, 2.5, -3.0, 7.0], n)),
        (4, rng.choice([0.0, -0.0, 1e-30, -1e-30, 1.0, -3.0, 1e30], n), rng.choice([0.0, -0.0, 1e-30, -1e-30, 2.0, -5.0, 1e30], n)),   # axes, tiny, huge
        (4, rng.uniform(-25, 25, n) * rng.choice([1.0, 1e-5, 1e-12], n), rng.uniform(-25, 25, n)),      # the disk's azimuths, incl. next to the axes
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
    # within 1e-4 relative, all within 5e-3 -- and tests/test_density_conditioning.py (CPU, on the oracle values these
    # bits were just shown to equal) accounts for every single miss: it lies inside the cone the REFERENCE's own
    # expression spans when its libm results move by <= 2-3 ulp.
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
    tex = g.HookSky(sky)
    for off in (0.0, 0.005, -0.005):
        for bits in (8, 0):
            out = torch.empty(len(d) * 4, device="cuda")
            g.unit("sky_sample", len(d), g.dev(d), float(off), tex.handle, bits, out)
            want = po.sky_sample(d, off, sky, bits, po.MATH_PORTABLE)
            assert same_bits(g.host(out).reshape(-1, 4), want), (off, bits)
    tex.destroy()


def test_disk_temperature_smoothstep_postfx_match_reference_vectors(g, po, units_ref):
    """The small functions that round 1 only exercised inside frames, against the REFERENCE's own outputs
    (units_ref.npz: densities.h:12-15, math_utils.h:45-48, post_processing.h:13-31 compiled by g++)."""
    import torch
    n = len(units_ref["temp_r"])
    out = torch.empty(n, device="cuda")
    g.unit("disk_temperature", n, g.dev(units_ref["temp_r"]), out)
    got = g.host(out)
    assert same_bits(got, po.units().disk_temperature(units_ref["temp_r"], po.MATH_PORTABLE))
    ref = units_ref["disk_temperature"]                      # powf from glibc there, rrt_powf here
    assert np.all(np.abs(got - ref) <= 1e-5 * np.abs(ref)) and np.array_equal(got == 0, ref == 0)
    # smoothstep, lens, vignette, bloom: + - * / sqrt only -> bit-exact against the reference
    g.unit("smoothstep", n, g.dev(units_ref["ss_e0"]), g.dev(units_ref["ss_e1"]), g.dev(units_ref["ss_x"]), out)
    assert same_bits(g.host(out), units_ref["smoothstep"])
    uv, rgb = g.dev(units_ref["uv"]), g.dev(units_ref["rgb"])
    o2 = torch.empty(n * 2, device="cuda"); o3 = torch.empty(n * 3, device="cuda")
    g.unit("postfx", 0, n, None, uv, 0.15, o2)
    assert same_bits(g.host(o2).reshape(n, 2), units_ref["lens_k0.15"])
    g.unit("postfx", 1, n, rgb, uv, 0.4, o3)
    assert same_bits(g.host(o3).reshape(n, 3), units_ref["vignette_i0.4"])
    g.unit("postfx", 2, n, rgb, None, 0.8, o3)
    assert same_bits(g.host(o3).reshape(n, 3), units_ref["bloom_t0.8"])


@pytest.mark.parametrize("spin", SPINS)
def test_radiative_transfer_block_matches_oracle(g, po, units_ref, spin):
    """raymarcher.cu:71-116 on its own (rrt_unit_rt_sample <-> rrto_rt_sample): every gate combination
    (none / disk only / dust only / both), bit-exact in portable mode, <= 1e-5 relative vs libm."""
    import torch
    rng = np.random.default_rng(17)
    n = len(units_ref["disk_p"])
    p, vel = units_ref["disk_p"], units_ref["geo_v"]
    d_disk = np.where(rng.random(n) < 0.3, 0.0005, np.exp(rng.uniform(np.log(0.0011), np.log(8.0), n))).astype(np.float32)
    d_cloud = np.where(rng.random(n) < 0.3, 0.001, np.exp(rng.uniform(np.log(0.0011), np.log(12.0), n))).astype(np.float32)
    h = (np.float32(0.3) * np.float32([0.1, 0.3])[rng.integers(0, 2, n)]).astype(np.float32)
    rad0 = np.concatenate([rng.uniform(0, 2, (n, 3)), rng.uniform(0, 1, (n, 1))], 1).astype(np.float32)
    rad = g.dev(rad0.reshape(-1).copy())
    g.unit("rt_sample", n, g.dev(d_disk), g.dev(d_cloud), g.dev(p), g.dev(vel), g.dev(h), float(spin), rad)
    got = g.host(rad).reshape(n, 4)
    assert same_bits(got, po.rt_sample(d_disk, d_cloud, p, vel, h, spin, rad0, po.MATH_PORTABLE))
    ref = po.rt_sample(d_disk, d_cloud, p, vel, h, spin, rad0, po.MATH_LIBM)
    assert np.all(np.abs(got - ref) <= 1e-5 * np.abs(ref) + 1e-7)
    untouched = (d_disk <= 0.001) & (d_cloud <= 0.001)
    assert untouched.any() and np.array_equal(got[untouched], rad0[untouched])


def test_noise_table_reads_equal_the_arithmetic_hash(g):
    """noise3D through the lattice-hash table == noise3D computed, bit for bit, for points all over both boxes
    (incl. their edges); points outside a box are counted and clamped, never read out of bounds."""
    import torch
    import relativisticraytracer_amd as rrt
    nt = g.HookNoiseTable(8.0)
    try:
        info = nt.info()
        rng = np.random.default_rng(23)
        for which, key in ((0, "accretion_box"), (1, "dust_box")):
            x0, y0, z0, nx, ny, nz = info[key]
            n = 1 << 20
            lo = np.array([x0, y0, z0], np.float64); span = np.array([nx - 1, ny - 1, nz - 1], np.float64)
            pts = (lo + rng.uniform(0, 1, (n, 3)) * span).astype(np.float32)
            pts[:8] = lo.astype(np.float32)                              # first cell
            pts[8:16] = (lo + span - 1e-3).astype(np.float32)            # last cell
            pts[16:24] = np.floor(pts[16:24])                            # exactly on lattice planes
            pts = np.minimum(pts, (lo + span - 1e-3).astype(np.float32))
            want = torch.empty(n, device="cuda"); got = torch.empty(n, device="cuda")
            cnt = torch.zeros(1, dtype=torch.int32, device="cuda")
            g.unit("noise3d", n, g.dev(pts), want)
            g.unit("noise3d_lut", n, g.dev(pts), nt.id, which, got, cnt)
            assert same_bits(g.host(got), g.host(want)), key
            assert int(cnt.item()) == 0
            far = (pts + np.float32(1e5)).astype(np.float32)             # way outside: clamped + counted
            g.unit("noise3d_lut", 1024, g.dev(far[:1024]), nt.id, which, got, cnt)
            assert int(cnt.item()) == 1024
    finally:
        nt.destroy()


@pytest.mark.parametrize("t,window", [(0.0, (0.0, 8.0, 0)), (1.0, (0.0, 8.0, 0)), (7.5, (0.0, 8.0, 0)),
                                      (7.5, (0.0, 8.0, 16)), (500.0, (495.0, 505.0, 0)), (1003.0, (995.0, 1005.0, 1)), (-33.0, (-40.0, -30.0, 16))])
def test_density_functions_with_table_switches_equal_the_arithmetic_ones(g, po, t, window):
    """The densities exactly as the render kernels evaluate them (early-out + wave-uniform table switches) on
    wave-coherent sample points -- 64 neighbours a few hundredths of a unit apart, as the 8x8-pixel wavefronts
    of a 4K frame produce -- so that the switches really are on; bits must equal the oracle's.  Round 5: also through the
    BANDED table layout (coverage | 16, or chosen automatically far along the clock: [495, 505 s] at full coverage)."""
    import torch
    import relativisticraytracer_amd as rrt
    nt = g.HookNoiseTable(window[1], window[0], window[2])
    try:
        rng = np.random.default_rng(29)
        waves = 4096
        rc = rng.uniform(10.0, 25.0, waves); ang = rng.uniform(-np.pi, np.pi, waves)
        yc = np.where(rng.random(waves) < 0.6, rng.uniform(-0.7, 0.7, waves), rng.uniform(-3.9, 3.9, waves))
        centre = np.stack([rc * np.cos(ang), yc, rc * np.sin(ang)], 1)
        spread = np.exp(rng.uniform(np.log(1e-3), np.log(0.3), waves))[:, None, None]
        pts = (centre[:, None, :] + spread * rng.uniform(-0.5, 0.5, (waves, 64, 3))).reshape(-1, 3).astype(np.float32)
        n = len(pts)
        disk = torch.empty(n, device="cuda"); dust = torch.empty(n, device="cuda")
        cnt = torch.zeros(1, dtype=torch.int32, device="cuda")
        rr = np.sqrt((pts.astype(np.float64) ** 2).sum(1))
        in_cloud = (np.abs(pts[:, 1]) < 0.75) & (rr < 24.99)
        in_disk = (np.abs(pts[:, 1]) < 4.0) & (rr < 29.99)
        g.unit("media_lut", n, g.dev(pts), float(t), nt.id, disk, dust, cnt)
        assert int(cnt.item()) == 0
        want_disk = po.units().accretion_density(pts, t, po.MATH_PORTABLE)
        want_dust = po.units().dust_density(pts, t, po.MATH_PORTABLE)
        got_disk, got_dust = g.host(disk), g.host(dust)
        # the render path's exact early-out returns 0 where the literal function returns a value <= 0.001
        live = want_disk > 0.001
        assert same_bits(got_disk[live & in_disk], want_disk[live & in_disk])
        assert np.all(got_disk[~live & in_disk] <= 0.001)
        # likewise for the dust (round 3: exact early-outs once the ridge sum / the strand factor cannot lift the
        # density over the `d_cloud > 0.001f` gates of raymarcher.cu:71,91)
        live_d = want_dust > 0.001
        assert same_bits(got_dust[live_d & in_cloud], want_dust[live_d & in_cloud])
        assert np.all(got_dust[~live_d & in_cloud] <= 0.001) and np.all(got_dust[~live_d & in_cloud] >= 0.0)
        assert (want_dust[in_cloud] > 0.001).mean() > 0.2 and live.mean() > 0.2
    finally:
        nt.destroy()


def test_early_outs_agree_with_the_literal_densities_through_the_gate(g, po):
    """The render kernels' density functions return 0 early wherever they can PROVE the literal value stays at or under
    the `d > 0.001f` gates (rrt_device.h: media_densities' invariant).  Points chosen to STRADDLE each threshold -- the slab
    cut y^2 rc = 135 (+-3 %), the accretion slab exponent -10.5, the rims where the envelopes cross 0.001 / 30.02, thin dust
    slabs around the base < 0.001 cut of densities.h:85 -- plus random in-zone points: wherever the literal density
    exceeds the gate the early-out variant has the same bits; elsewhere it is some value <= the gate (and >= 0)."""
    import torch
    import relativisticraytracer_amd as rrt
    rng = np.random.default_rng(77)
    n = 1 << 16
    rc = rng.uniform(10.0, 25.0, n); ang = rng.uniform(-np.pi, np.pi, n)
    y = np.empty(n)
    q = n // 4
    y[:q] = np.sqrt(135.0 / rc[:q]) * rng.uniform(0.97, 1.03, q) * rng.choice([-1.0, 1.0], q)            # y^2 rc ~ 135
    # accretion slab exponent -(y / (0.8 (0.5 + 0.5 (rc - 10) / 15)))^2 ... around -10.5: |y| ~ 3.24 x the local scale height
    hgt = 0.8 * (0.5 + 0.5 * (rc[q:2 * q] - 10.0) / 15.0)
    y[q:2 * q] = hgt * np.sqrt(10.5) * rng.uniform(0.9, 1.1, q) * rng.choice([-1.0, 1.0], q)
    y[2 * q:3 * q] = rng.uniform(-0.75, 0.75, q)                                                             # the dust slab
    rc[2 * q:3 * q] = np.concatenate([rng.uniform(10.0, 10.6, q // 2), rng.uniform(24.0, 25.0, q - q // 2)])  # rims: envelopes -> 0
    y[3 * q:] = rng.uniform(-4.0, 4.0, n - 3 * q)
    pts = np.stack([rc * np.cos(ang), y, rc * np.sin(ang)], 1).astype(np.float32)
    for t in (0.0, 3.0):
        nt = g.HookNoiseTable(8.0)
        try:
            d = g.dev(pts)
            got_disk, got_dust = torch.empty(n, device="cuda"), torch.empty(n, device="cuda")
            cnt = torch.zeros(1, dtype=torch.int32, device="cuda")
            g.unit("media_lut", n, d, float(t), nt.id, got_disk, got_dust, cnt)
            lit_disk, lit_dust = torch.empty(n, device="cuda"), torch.empty(n, device="cuda")
            g.unit("accretion_density", n, d, float(t), lit_disk)
            g.unit("dust_density", n, d, float(t), lit_dust)
            gd, gu, ld, lu = g.host(got_disk), g.host(got_dust), g.host(lit_disk), g.host(lit_dust)
            r = np.linalg.norm(pts.astype(np.float64), axis=1)
            in_disk = (np.abs(pts[:, 1]) < 4.0) & (r < 30.0)
            in_cloud = (np.abs(pts[:, 1]) < 0.75) & (r < 25.0)
            for got, lit, zone, name in ((gd, ld, in_disk, "disk"), (gu, lu, in_cloud, "dust")):
                live = (lit > 0.001) & zone
                assert same_bits(got[live], lit[live]), name
                dead = ~(lit > 0.001) & zone
                assert np.all(got[dead] <= 0.001) and np.all(got[dead] >= 0.0), name
                assert live.sum() > 500 and dead.sum() > 500, (name, int(live.sum()), int(dead.sum()))
            assert np.all(gd[~in_disk] == 0.0) and np.all(gu[~in_cloud] == 0.0)       # outside the zone the kernels never call
            # the thresholds really are straddled: early-outs fired on a good share of the sub-gate points
            assert ((gd == 0.0) & (ld > 0.0) & in_disk).sum() > 1000 and ((gu == 0.0) & (lu > 0.0) & in_cloud).sum() > 100
            assert int(cnt[0]) == 0
        finally:
            nt.destroy()


def test_fast_sqrt_is_correctly_rounded_everywhere_it_is_used(g):
    """sqrt_rsq (rsq + Newton + residual fix-up) == IEEE sqrtf for EVERY float in [1, 2^64)."""
    import ctypes as C
    import torch
    from relativisticraytracer_amd import _lib
    cnt = torch.zeros(4, dtype=torch.int64, device="cuda")
    lo, hi = 0x3f800000, 0x3f800000 + (64 << 23)
    _lib.check(_lib.load_test().rrt_selfcheck_sqrt(lo, hi, C.c_void_p(cnt.data_ptr()), None), "selfcheck_sqrt")
    torch.cuda.synchronize()
    assert int(cnt[0]) == 0, f"{int(cnt[0])} mismatches, e.g. bits {int(cnt[1]):#x}"


def test_fast_divide_is_correctly_rounded_on_march_operands(g):
    """div_seeded (Markstein core seeded from powers of 1/r) == IEEE `/` on 2^32 march-shaped cases."""
    import ctypes as C
    import torch
    from relativisticraytracer_amd import _lib
    cnt = torch.zeros(4, dtype=torch.int64, device="cuda")
    _lib.check(_lib.load_test().rrt_selfcheck_div(1 << 35, 12345, C.c_void_p(cnt.data_ptr()), None), "selfcheck_div")
    torch.cuda.synchronize()
    assert int(cnt[0]) == 0, f"{int(cnt[0])} mismatches, e.g. {int(cnt[1]):#x} / {int(cnt[2]):#x}"


def test_divide_with_the_marchs_own_seeds(g):
    """The march's seeded roots and its divides with the reciprocal roots THOSE hand on (round 4): roots out of
    sqrt_seeded_yh<1> / <2> started from estimates off by up to each form's acceptance tolerance, instead of the v_rsq-based
    root of the test above -- the one-correction Markstein core then starts from a seed about 1.7x worse.  2^33 operand
    sets here: every accepted root is the correctly rounded one and no divide differs from IEEE.  The one-off runs behind
    it (profiles/r04_div_march_seeds_probe.txt): before round 4's guard 176 of 1.1e12 accepted two-iteration roots were one
    ulp off, all on x = 4^k (1 + 2^-23); with the guard (such an x takes the v_rsq fall-back) 0 of 1.1e13 roots and 0 of
    2.2e13 divides over five seed-error ranges."""
    import ctypes as C
    import torch
    from relativisticraytracer_amd import _lib
    cnt = torch.zeros(8, dtype=torch.int64, device="cuda")
    _lib.check(_lib.load_test().rrt_selfcheck_div_march(1 << 33, 2026, 1.45e-4, 8.9e-3, C.c_void_p(cnt.data_ptr()), None), "selfcheck_div_march")
    torch.cuda.synchronize()
    assert int(cnt[2]) == 0, f"{int(cnt[2])} divide mismatches, e.g. {int(cnt[6]):#x} / {int(cnt[7]):#x}"
    assert int(cnt[3]) > 1.5 * (1 << 33), int(cnt[3])              # most roots were accepted, two divides each
    assert int(cnt[0]) + int(cnt[1]) == 0, (int(cnt[0]), int(cnt[1]), hex(int(cnt[4])), hex(int(cnt[5])))
    # the guarded class itself: x right above a power of four must be REJECTED by the two-iteration form whatever the seed
    # (rrt_unit_rk4_lean drives such radii through the fall-back; here: the probe accepts none of them, so nothing to count)


def test_seeded_roots_around_the_powers_of_two_under_a_dense_seed_sweep(g):
    """Where sqrt(x) comes closest to a rounding tie -- the floats within 64 ulps of every power of two 2^e, e in [0, 30) (the
    march's r^2 range) -- with 2^14 seeds per x and form spread over each form's whole acceptance interval: every accepted
    root is sqrtf(x).  Round 4 found the one class that was not (two-iteration roots of x = 4^k (1 + 2^-23)) and guards it: it
    must now be rejected for every seed, and the test sees that the guard costs nothing else (> 99 % accepted overall)."""
    import ctypes as C
    import torch
    from relativisticraytracer_amd import _lib
    cnt = torch.zeros(6, dtype=torch.int64, device="cuda")
    _lib.check(_lib.load_test().rrt_selfcheck_sqrt_boundaries(0, 30, 64, 1 << 14, 1.45e-4, 8.9e-3, C.c_void_p(cnt.data_ptr()), None), "selfcheck_sqrt_boundaries")
    torch.cuda.synchronize()
    assert int(cnt[0]) == 0 and int(cnt[1]) == 0, f"{int(cnt[0])} + {int(cnt[1])} mismatches, e.g. x bits {int(cnt[4]):#x} seed bits {int(cnt[5]):#x}"
    n = 30 * 129 * (1 << 14) * 2
    assert int(cnt[2]) + int(cnt[3]) == n
    assert int(cnt[3]) >= 15 * (1 << 14)                      # the guarded x (one per even exponent) rejected for every seed
    assert int(cnt[2]) > 0.99 * n


def test_divide_known_exception_is_what_the_documents_say(g):
    """"Bit-exact division" has ONE documented hole (rrt_device.h: div_seeded): a denominator whose
    significand lies an odd number of ulps d <= ~11 below 2 has a reciprocal within d^2/4 * 2^-46 of a rounding tie, closer
    than the 2^-42 the once-refined reciprocal carries, so the refined reciprocal can come out one ulp low and the quotient
    with it.  Measured rate on march-shaped operands: 2 in 3.5e13 divides (profiles/r03_div_rounds_probe.txt) = 5.7e-14 per
    divide, 0.004 per 4K frame.  The operand set that probe recorded -- 0x33666662 / 0x4bfffffb -- reproduces with a seed
    three ulps (2^-21.4: march quality) off the reciprocal and not with a seed within one ulp: pinned here so that any
    change of div_seeded that moves this behaviour is noticed."""
    import torch
    a = np.array([0x33666662] * 4, np.uint32).view(np.float32)
    b = np.array([0x4bfffffb] * 4, np.uint32).view(np.float32)
    seed = np.array([0x33000003, 0x33000004, 0x33000002, 0x33000006], np.uint32).view(np.float32)
    out = torch.empty(4, device="cuda")
    g.unit("div_seeded", 4, g.dev(a), g.dev(b), g.dev(seed), out)
    got = g.host(out).view(np.uint32)
    ieee = (a / b).view(np.uint32)
    assert ieee[0] == 0x26e66667
    assert np.array_equal(got[:3], ieee[:3])                          # seeds within one ulp of 1/b: the IEEE quotient
    assert got[3] == 0x26e66666                                       # the documented exception: one ulp low
    # and the rule away from those denominators: march-quality seeds give the IEEE quotient
    rng = np.random.default_rng(5)
    n = 1 << 20
    bb = rng.uniform(1.0, 2.0, n).astype(np.float32) * np.float32(2.0) ** rng.integers(-20, 60, n).astype(np.float32)
    aa = (rng.uniform(1.0, 2.0, n) * 2.0 ** rng.integers(-40, 40, n) * rng.choice([-1.0, 1.0], n)).astype(np.float32)
    sd = ((1.0 / bb.astype(np.float64)) * (1.0 + rng.uniform(-4e-7, 4e-7, n))).astype(np.float32)
    far = (bb.view(np.uint32) & 0x7fffff) < 0x7fffe0                  # significand more than 32 ulps below 2
    out = torch.empty(n, device="cuda")
    g.unit("div_seeded", n, g.dev(aa), g.dev(bb), g.dev(sd), out)
    assert np.array_equal(g.host(out).view(np.uint32)[far], (aa / bb).view(np.uint32)[far])


def test_seeded_sqrt_is_correctly_rounded_whenever_it_accepts(g):
    """sqrt_seeded_yh (the march's square root without v_rsq: Goldschmidt from the previous stage's (1/r, 1/2r)) ==
    IEEE sqrtf for EVERY float of [1, 4) (two binades = every mantissa with either exponent parity) and of
    [2^14, 2^16), with estimates off by 0 ... +-1.2e-2, one and two iterations, wherever it accepts its own result
    (round 3: acceptance on the FIRST residual); it does accept the estimates the march produces (errors <= 1.4e-4
    with one iteration, <= 8.9e-3 with two), it hands on y == 2h exactly, and it rejects the seed 2/sqrt(x), from
    which the iteration converges to MINUS the root."""
    import ctypes as C
    import torch
    from relativisticraytracer_amd import _lib
    for lo in (0x3f800000, 0x3f800000 + (14 << 23)):
        cnt = torch.zeros(4, dtype=torch.int64, device="cuda")
        _lib.check(_lib.load_test().rrt_selfcheck_sqrt_seeded(lo, lo + (2 << 23), C.c_void_p(cnt.data_ptr()), None), "selfcheck_sqrt_seeded")
        torch.cuda.synchronize()
        assert int(cnt[0]) == 0, f"{int(cnt[0])} mismatches, e.g. x bits {int(cnt[1]):#x} seed bits {int(cnt[2]):#x}"
        n = 2 << 23
        # accepted per x: ladder, 1 iteration |delta| <= 1.4e-4 (10 seeds), 2 iterations |delta| <= 5e-3 (16); random
        # seeds within 1.45e-4 (8 + 8) and, of the 8 within 9.5e-3, the ~94 % below the two-iteration tolerance
        assert int(cnt[3]) >= n * (10 + 16 + 8 + 8 + 7) * 0.98, int(cnt[3])


def test_media_sqrt_and_divide_cores_are_correctly_rounded(g):
    """The volumetric code's scaling-free sqrt / divide (rrt_device.h: sqrt_tame, rrt_div_tame) == the IEEE forms:
    sqrt on EVERY float of [2^-40, 1) (the range above 1 is covered by the march's check), divide on 2^32 random
    tame operand pairs."""
    import ctypes as C
    import torch
    from relativisticraytracer_amd import _lib
    cnt = torch.zeros(4, dtype=torch.int64, device="cuda")
    lo, hi = 0x3f800000 - (40 << 23), 0x3f800000
    _lib.check(_lib.load_test().rrt_selfcheck_sqrt(lo, hi, C.c_void_p(cnt.data_ptr()), None), "selfcheck_sqrt")
    torch.cuda.synchronize()
    assert int(cnt[0]) == 0, f"sqrt: {int(cnt[0])} mismatches, e.g. bits {int(cnt[1]):#x}"
    cnt.zero_()
    _lib.check(_lib.load_test().rrt_selfcheck_div_tame(1 << 32, 777, C.c_void_p(cnt.data_ptr()), None), "selfcheck_div_tame")
    torch.cuda.synchronize()
    assert int(cnt[0]) == 0, f"div: {int(cnt[0])} mismatches, e.g. {int(cnt[1]):#x} / {int(cnt[2]):#x}"


def test_division_by_compile_time_constants_is_correctly_rounded(g):
    """rrt_div_const (round 3: three instructions, the folded correctly rounded reciprocal + one Markstein step) == IEEE `/`
    for EVERY dividend of magnitude 2^[-40, 40), both signs, and +0, for each of the seven constants the media code
    divides by (smoothstep edges, rim taper, ISCO_RADIUS, DISK_TEMP_REF)."""
    import ctypes as C
    import torch
    from relativisticraytracer_amd import _lib
    cnt = torch.zeros(4, dtype=torch.int64, device="cuda")
    lo, hi = (127 - 40) << 23, (127 + 40) << 23
    _lib.check(_lib.load_test().rrt_selfcheck_div_const(lo, hi, C.c_void_p(cnt.data_ptr()), None), "selfcheck_div_const")
    torch.cuda.synchronize()
    assert int(cnt[0]) == 0, f"{int(cnt[0])} mismatches, e.g. dividend bits {int(cnt[1]):#x} with constant #{int(cnt[2])}"


def test_unit_kernels_empty_and_bad_args(g):
    import torch
    from relativisticraytracer_amd import _lib
    z = torch.empty(0, device="cuda")
    g.unit("hash31", 0, z, z)                         # n = 0 is a no-op
    assert _lib.load_test().rrt_unit_hash31(4, None, None, None) == 1        # RRT_ERR_INVALID_ARGUMENT
    assert _lib.load_test().rrt_unit_fbm(4, z.data_ptr(), 99, z.data_ptr(), None) == 1
