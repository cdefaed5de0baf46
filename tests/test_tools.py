"""The profile summariser's choice of kernels (tools/summarize_profile.py): records that bench.py quotes come out of it."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load(name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "tools", name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_headline_kernel_is_the_table_instantiation_not_the_arithmetic_noise_leg():
    """Round 5's b-f summaries averaged the strict table kernel with bench.py's headline_arithmetic_noise launches (same grid, MEDIA 1):
    half the real HBM traffic in the bench line.  The headline is the strict production instantiation with the HIGHEST media value."""
    n = "void (anonymous namespace)::raymarch_pixels<true, %d, %s, %d>((anonymous namespace)::FrameArgs)"
    names = [n % (2, "false", 0), n % (1, "false", 0), n % (2, "false", 1), n % (2, "false", 2), n % (2, "true", 0), n % (2, "true", 2),
             "void (anonymous namespace)::march_defer<true, 0, false>((anonymous namespace)::FrameArgs)"]
    kind = _load("summarize_profile").make_kind(names)
    assert [kind(x) for x in names] == ["kernel", "arithmetic_noise_kernel", "fast_mode_kernel", "fmad_mode_kernel", None, None, None]
    # a banded table (MEDIA 3) in the timed loop: that one is the headline, a dense-table launch beside it is not
    names3 = [n % (3, "false", 0), n % (2, "false", 0), n % (1, "false", 0)]
    kind3 = _load("summarize_profile").make_kind(names3)
    assert [kind3(x) for x in names3] == ["kernel", None, "arithmetic_noise_kernel"]
    # --no-noise-table runs: the arithmetic-noise kernel IS the headline
    kind1 = _load("summarize_profile").make_kind([n % (1, "false", 0)])
    assert kind1(n % (1, "false", 0)) == "kernel"
