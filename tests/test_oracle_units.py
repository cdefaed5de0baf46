"""The oracle restatement against the REFERENCE's own functions.

tests/golden/units_ref.npz holds inputs and outputs of the reference headers
(/root/reference/include/{math_utils,geodesics,integrators,densities}.h and
camera_effects/post_processing.h) compiled by g++ -- see oracle/ref_units.cpp
and tests/golden/make_golden.py.  The oracle (libm mode) must reproduce every
one of them bit for bit; that is what "pinned" means in oracle/rrt_oracle.h.
"""
import numpy as np
import pytest

from conftest import same_bits

SPINS = (0.0, 0.9, 0.99)
TIMES = (0.0, 1.0, 12.5)


def test_constants_match_config_h(units_ref):
    c = units_ref["constants"]
    # EVENT_HORIZON ISCO DISK_OUT DISK_H DISK_LUM DISK_OPAC EXPOSURE CLOUD_H CLOUD_OUT CLOUD_OPAC CLOUD_LUM STEP MAX_STEPS T_REF PI W H FPS
    expect = np.float32([2.0, 10.0, 25.0, 0.8, 6.0, 0.4, 0.8, 0.5, 25.0, 0.3, 0.4, 0.3, 2000, 1.5e7,
                         3.1415926535, 1000, 700, 24])
    assert np.array_equal(c, expect)


@pytest.mark.parametrize("spin", SPINS)
def test_geodesic_acc(po, units_ref, spin):
    got = po.units().geodesic_acc(units_ref["geo_p"], units_ref["geo_v"], spin)
    assert same_bits(got, units_ref[f"geodesic_acc_a{spin:g}"])


@pytest.mark.parametrize("spin", SPINS)
def test_rk4_step(po, units_ref, spin):
    p, v = po.units().rk4(units_ref["geo_p"], units_ref["geo_v"], units_ref["rk4_h"], spin)
    assert same_bits(p, units_ref[f"rk4_p_a{spin:g}"])
    assert same_bits(v, units_ref[f"rk4_v_a{spin:g}"])


@pytest.mark.parametrize("spin", SPINS)
def test_rk4_chain_of_fifty_steps(po, rk4_chain_ref, spin):
    """The restatement's integrate_rk4 under the march's zone rule and horizon test (restated here in numpy binary32)
    reproduces the reference's 50-step chains bit for bit at every recorded step."""
    f = np.float32
    c = rk4_chain_ref
    p, v = c["p0"].copy(), c["v0"].copy()
    live = np.ones(len(p), bool)
    steps = np.zeros(len(p), np.int32)
    marks = set(int(k) for k in c["marks"])
    for k in range(1, max(marks) + 1):
        x, y, z = p[:, 0], p[:, 1], p[:, 2]
        r = np.sqrt((x * x + y * y) + z * z)
        live &= ~(r < f(2.0) * f(1.01))
        h = np.where(r < f(18.0), f(0.3) * f(0.1),
                     np.where((np.abs(y) < f(0.8) * f(5.0)) & (r < f(25.0) + f(5.0)), f(0.3) * f(0.3), f(0.3))).astype(np.float32)
        pn, vn = po.units().rk4(p[live], v[live], h[live], spin)
        p[live], v[live] = pn, vn
        steps[live] += 1
        if k in marks:
            assert same_bits(p, c[f"p_a{spin:g}_k{k}"]) and same_bits(v, c[f"v_a{spin:g}_k{k}"]), k
            assert np.array_equal(steps, c[f"steps_a{spin:g}_k{k}"])
    assert (~live).sum() >= 3 and live[:256].all()      # some rays meet the horizon test; the far-out waves never do


def test_hash31_lattice(po, units_ref):
    got = po.units().hash31(units_ref["lattice"])
    assert np.array_equal(got.view(np.uint32), units_ref["hash31"].view(np.uint32))
    assert units_ref["hash31"].min() < 0          # range is (-1, 1): negative lattice coords give negatives


def test_noise_and_fbm(po, units_ref):
    u = po.units()
    assert same_bits(u.noise3d(units_ref["noise_p"]), units_ref["noise3d"])
    assert same_bits(u.fbm(units_ref["noise_p"], 2), units_ref["fbm2"])
    assert same_bits(u.fbm(units_ref["noise_p"], 5), units_ref["fbm5"])


@pytest.mark.parametrize("t", TIMES)
def test_densities(po, units_ref, t):
    u = po.units()
    acc = u.accretion_density(units_ref["disk_p"], t)
    dust = u.dust_density(units_ref["cloud_p"], t)
    assert same_bits(acc, units_ref[f"accretion_t{t:g}"])
    assert same_bits(dust, units_ref[f"dust_t{t:g}"])
    assert (acc > 0.001).sum() > 50 and (dust > 0.001).sum() > 50      # the vectors exercise the live branch


@pytest.mark.parametrize("spin", SPINS)
def test_redshift(po, units_ref, spin):
    got = po.units().redshift(units_ref["disk_p"], units_ref["geo_v"], spin)
    assert same_bits(got, units_ref[f"redshift_disk_a{spin:g}"])


def test_disk_temperature(po, units_ref):
    got = po.units().disk_temperature(units_ref["temp_r"])
    assert same_bits(got, units_ref["disk_temperature"])
    assert (got == 0).any() and (got > 0).any()


def test_smoothstep_and_postfx(po, units_ref):
    u = po.units()
    assert same_bits(u.smoothstep(units_ref["ss_e0"], units_ref["ss_e1"], units_ref["ss_x"]), units_ref["smoothstep"])
    assert same_bits(u.lens(units_ref["uv"], 0.15), units_ref["lens_k0.15"])
    assert same_bits(u.vignette(units_ref["rgb"], units_ref["uv"], 0.4), units_ref["vignette_i0.4"])
    assert same_bits(u.bloom(units_ref["rgb"], 0.8), units_ref["bloom_t0.8"])


def test_live_reference_build_agrees_with_fixtures(po, units_ref):
    """Where oracle/_ref exists (build container), the fixtures must be what it produces now."""
    if not po.ref_available():
        pytest.skip("oracle/_ref not built here (the reference does not travel)")
    ref = po.ref_units()
    assert same_bits(ref.fbm(units_ref["noise_p"], 5), units_ref["fbm5"])
    p, v = ref.rk4(units_ref["geo_p"], units_ref["geo_v"], units_ref["rk4_h"], 0.9)
    assert same_bits(p, units_ref["rk4_p_a0.9"]) and same_bits(v, units_ref["rk4_v_a0.9"])
    assert same_bits(ref.dust_density(units_ref["cloud_p"], 1.0), units_ref["dust_t1"])


def test_empty_inputs(po):
    u = po.units()
    e3 = np.zeros((0, 3), np.float32)
    assert u.hash31(e3).shape == (0,)
    assert u.geodesic_acc(e3, e3, 0.9).shape == (0, 3)
