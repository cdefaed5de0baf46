"""Makes the density functions' tolerance against the reference's vectors exact.

`getAccretionDensity` / `getDustCloudDensity` (densities.h:20-132) are ill-conditioned in their transcendentals: a 1-ulp
change of sinf / cosf at angle ~ time * omega moves the noise coordinates by ~1e-6, the octaves scale that by up to 17.7,
`pow(2.8 * max(0, n - 0.32), 1.6)` amplifies it again where n ~ 0.32, and `base < 0.001f` (densities.h:85) is a hard
gate.  So single samples differ by up to ~1e-3 relative between ANY two math libraries, and a flat "1e-4 relative"
cannot hold point by point (the per-pixel integral damps it: tests/test_gate_accounting.py).  This test replaces the
statistical bar by an accounting:

  A   = the function in PORTABLE math (== the HIP kernels bit for bit: tests/test_gpu_units.py);
  B   = the function in LIBM math (== the reference's own header compiled by g++, bit for bit: asserted below);
  N_s = LIBM math with every transcendental RESULT moved by a pseudo-random <= 2 ulp (atan2f: 3), 32 seeds -- the
        spread a GPU math library of CUDA's documented error class would put on the REFERENCE's own value.

Claim asserted for every one of the 6 x 1024 reference points: |A - B| <= 1e-4 |B| + 1e-6, or |A - B| is no larger than
the largest |N_s - B| -- the deviation is inside the cone the reference's own expression spans under library-level
rounding noise, i.e. it is conditioning (or the hard gate flipping), not an error of the implementation.
"""
import numpy as np
import pytest

TIMES = (0.0, 1.0, 12.5)
SEEDS = range(1, 33)


@pytest.mark.parametrize("t", TIMES)
def test_every_density_deviation_from_the_reference_is_within_its_own_ulp_cone(po, units_ref, t):
    U = po.units()
    for fn, pts, ref in ((U.accretion_density, units_ref["disk_p"], units_ref[f"accretion_t{t:g}"]),
                         (U.dust_density, units_ref["cloud_p"], units_ref[f"dust_t{t:g}"])):
        A = fn(pts, t, po.MATH_PORTABLE)
        B = fn(pts, t, po.MATH_LIBM)
        assert np.array_equal(B.view(np.uint32), ref.view(np.uint32))            # B IS the reference's output
        cone = np.max([np.abs(fn(pts, t, po.MATH_NUDGED_BASE + s) - B) for s in SEEDS], axis=0)
        err = np.abs(A - B)
        tol = 1e-4 * np.abs(B) + 1e-6
        miss = err > tol
        assert np.all(err[miss] <= cone[miss]), (err[miss] / cone[miss]).max()
        # the well-conditioned points (cone inside the tolerance) are therefore all inside 1e-4 ...
        assert not np.any(miss & (cone <= tol))
        # ... and the misses stay the small minority the GPU test's statistical bar describes
        assert miss.mean() <= 0.03 and np.all(err <= 5e-3 * np.abs(B) + 1e-5)


def test_nudged_mode_moves_results_by_a_few_ulps_only(po):
    """The probe itself: nudged libm stays within 2 ulp (atan2f 3) of libm on every unit function that is a single
    transcendental call deep (getDiskTemperature = one powf), and seed 0 offsets differ from seed 1."""
    r = np.linspace(10.0, 60.0, 4096, dtype=np.float32)
    base = po.units().disk_temperature(r, po.MATH_LIBM)
    seen = set()
    for s in (1, 2, 3):
        got = po.units().disk_temperature(r, po.MATH_NUDGED_BASE + s)
        d = got.view(np.int32).astype(np.int64) - base.view(np.int32).astype(np.int64)
        # powf moved by <= 2 ulp, then one multiply by 1.5e7 (the product's ulp can be half as large: <= 4, + rounding)
        assert 1 <= np.abs(d).max() <= 5
        seen.add(tuple(d[:64]))
    assert len(seen) == 3
