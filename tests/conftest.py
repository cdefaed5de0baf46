import os
import sys

import numpy as np
import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")

# the library's test hooks (rrt_debug_fake_device) only work in a process that asked for them before loading it
os.environ.setdefault("RRT_ENABLE_TEST_HOOKS", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def po():
    """The CPU oracle binding (test infrastructure)."""
    from oracle import pyoracle
    pyoracle.build()          # compiles oracle/librrt_oracle.so if missing or stale (gcc only)
    return pyoracle


@pytest.fixture(scope="session")
def units_ref():
    return dict(np.load(os.path.join(GOLDEN, "units_ref.npz")))


@pytest.fixture(scope="session")
def rk4_chain_ref():
    """50-step chains of the reference's own integrate_rk4 under the march's step-size rule (make_golden.py::make_rk4_chains)."""
    return dict(np.load(os.path.join(GOLDEN, "rk4_chain_ref.npz")))


@pytest.fixture(scope="session")
def frames_gold():
    return dict(np.load(os.path.join(GOLDEN, "frames_oracle.npz")))


@pytest.fixture(scope="session")
def frames_ref():
    """Frames from the reference's OWN raymarch_kernel body (tests/golden/make_golden.py::make_frames_ref)."""
    return dict(np.load(os.path.join(GOLDEN, "frames_ref.npz")))


@pytest.fixture(scope="session")
def frames_ref_fma():
    """The reference's kernel body compiled with floating-point CONTRACTION (make_golden.py::make_frames_ref_fma), on the
    frames_ref.npz scenes and four 320x180 views (B1-B4, whose strict reference frames are in the same file)."""
    return dict(np.load(os.path.join(GOLDEN, "frames_ref_fma.npz")))


def contraction_counts(strict8, strict_steps, other8, other_steps):
    """How far a frame is from the strict reference frame, in the only units the reference's kernel outputs: pixels with a
    differing byte, pixels with a byte off by more than one LSB, pixels with another RK4 step count.  rgba8 (h, w, 4) bottom-up,
    steps (h*w,) top-down (the fixtures' layouts).  Returns (counts dict, deviant mask (h, w) bottom-up)."""
    h, w = strict8.shape[:2]
    d = np.abs(strict8.astype(int) - other8.astype(int))[..., :3].max(axis=2)
    sd = (np.asarray(strict_steps).astype(int) != np.asarray(other_steps).astype(int)).reshape(h, w)[::-1]
    return ({"pixels": h * w, "bytes_differ": int((d > 0).sum()), "off_by_more_than_1": int((d > 1).sum()), "steps_differ": int(sd.sum()),
             "deviant": int(((d > 1) | sd).sum()), "max_byte_diff": int(d.max())}, (d > 1) | sd)


@pytest.fixture(scope="session")
def camera_ref():
    return dict(np.load(os.path.join(GOLDEN, "camera_ref.npz")))


@pytest.fixture(scope="session")
def sky():
    from relativisticraytracer_amd.sky import synthetic_sky
    return synthetic_sky()


def same_bits(a, b):
    """Bit equality of float arrays, except that +0 and -0 compare equal."""
    a = np.asarray(a, np.float32); b = np.asarray(b, np.float32)
    return a.shape == b.shape and bool(np.all((a.view(np.uint32) == b.view(np.uint32)) | ((a == 0) & (b == 0))))


def ulp_diff(a, b):
    """|a-b| in units of ulp(b) (float32)."""
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    ulp = np.spacing(np.abs(b).astype(np.float32)).astype(np.float64)
    return np.abs(a - b) / ulp
