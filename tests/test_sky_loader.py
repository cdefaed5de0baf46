"""the sky LOADER.  The reference decodes assets/skyboxes/skybox2.jpg with its vendored stb_image
(stbi_load(filename, ..., 4), src/main.cpp:240).  tests/golden/sky_ref.npz holds what that decoder -- compiled from the
reference's own header where it lies, oracle/ref_stb.c -- produces: sizes, sha256 of the full decodes, crops, and how this
package's PIL fallback differs from it.  The package reads a RAW sky (decoded once, shipped) texel for texel."""
import hashlib
import os

import numpy as np
import pytest

from conftest import GOLDEN

ASSETS = "/root/reference/assets/skyboxes"
in_build_container = pytest.mark.skipif(not os.path.exists(os.path.join(ASSETS, "skybox2.jpg")),
                                        reason="the reference's assets are only present in the build container")


@pytest.fixture(scope="module")
def sky_ref():
    return dict(np.load(os.path.join(GOLDEN, "sky_ref.npz")))


def test_fixture_describes_the_reference_assets(sky_ref):
    assert sky_ref["skybox2_jpg_size"].tolist() == [4096, 2048, 3]            # what loadSkybox() prints (main.cpp:265)
    assert sky_ref["skybox_png_size"].tolist() == [1024, 1024, 3]             # (a JPEG under a .png name)
    assert bool(sky_ref["skybox2_jpg_alpha_all_255"])                          # 3-channel file, 4 requested: alpha = 255
    for k in range(6):
        assert sky_ref[f"skybox2_jpg_crop{k}"].shape == (64, 64, 4)
    # this package's PIL fallback is NOT the reference's decoder: 0.9 % of the colour bytes differ, by at most 3
    vals, counts = sky_ref["skybox2_jpg_pil_minus_stb_values"], sky_ref["skybox2_jpg_pil_minus_stb_counts"]
    assert counts.sum() == 4096 * 2048 * 3 and int(np.abs(vals).max()) == 3
    frac = 1.0 - counts[vals == 0][0] / counts.sum()
    assert 0.005 < frac < 0.02


def test_raw_sky_round_trip_and_refusals(sky_ref, tmp_path):
    from relativisticraytracer_amd import sky
    crop = sky_ref["skybox2_jpg_bigcrop"]
    p = tmp_path / "crop.rrtsky"
    sky.save_sky_raw(str(p), crop)
    assert sky.is_raw_sky(str(p)) and np.array_equal(sky.load_sky_raw(str(p)), crop)
    assert np.array_equal(sky.load_sky(str(p)), crop)                          # load_sky recognises the raw file
    blob = open(p, "rb").read()
    (tmp_path / "short.rrtsky").write_bytes(blob[:-5])
    (tmp_path / "long.rrtsky").write_bytes(blob + b"x")
    (tmp_path / "nohdr.rrtsky").write_bytes(b"RRTSKY1\nabc\n" + blob[20:])
    (tmp_path / "other.bin").write_bytes(b"P6\n1 1\n255\n\0\0\0")
    for bad in ("short.rrtsky", "long.rrtsky", "nohdr.rrtsky", "other.bin"):
        with pytest.raises(ValueError):
            sky.load_sky_raw(str(tmp_path / bad))
    with pytest.raises(ValueError):
        sky.save_sky_raw(str(tmp_path / "x"), np.zeros((4, 4, 3), np.uint8))


def test_oracle_renders_the_fixture_frame_with_the_real_sky_crop(po, sky_ref):
    """The restatement (libm mode) with the 256x128 crop of the reference's decode as its sky == the frame the reference's
    own kernel body rendered with it: bytes and step counts."""
    w, h, spin, vol, t = sky_ref["frame_scene"]
    a = sky_ref["frame_camera"]
    cam = po.camera(a[0], a[1], a[2], a[3])
    r = po.render(cam, po.default_effects(use_ca=1), po.default_params(spin=float(np.float32(spin)), volumetrics=int(vol), math_mode=po.MATH_LIBM),
                  float(np.float32(t)), int(w), int(h), sky_ref["skybox2_jpg_bigcrop"], want=("rgba8", "diag"))
    assert np.array_equal(r["rgba8"], sky_ref["frame_rgba8"])
    assert np.array_equal(r["steps"], sky_ref["frame_steps"].astype(np.int32))
    assert len(np.unique(r["rgba8"][..., :3].reshape(-1, 3), axis=0)) > 500            # the sky really shows


@in_build_container
def test_reference_decoder_reproduces_the_fixture_and_the_converter_ships_it(po, sky_ref, tmp_path):
    """Build container: oracle/_ref/libref_stb.so decodes the reference's asset to the committed digest and crops;
    tools/sky_to_raw.py writes that decode as a raw sky which load_sky returns texel for texel; PIL's decode of the same
    file has the committed difference histogram (same PIL as the one that made the fixture: this image has one)."""
    import subprocess, sys
    from PIL import Image
    from relativisticraytracer_amd import sky
    po.build(ref=True)
    path = os.path.join(ASSETS, "skybox2.jpg")
    assert hashlib.sha256(open(path, "rb").read()).digest() == sky_ref["skybox2_jpg_file_sha256"].tobytes()
    stb, channels = po.ref_stb_load(path)
    assert channels == 3 and stb.shape == (2048, 4096, 4)
    assert hashlib.sha256(stb.tobytes()).digest() == sky_ref["skybox2_jpg_stb_sha256"].tobytes()
    for k in range(6):
        r, c = sky_ref[f"skybox2_jpg_crop{k}_at"]
        assert np.array_equal(stb[r:r + 64, c:c + 64], sky_ref[f"skybox2_jpg_crop{k}"])
    out = tmp_path / "sky.rrtsky"
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    subprocess.run([sys.executable, os.path.join(root, "tools", "sky_to_raw.py"), path, str(out)], check=True, capture_output=True)
    assert np.array_equal(sky.load_sky(str(out)), stb)
    pil = sky.load_sky(path)                                                       # the fallback decoder
    d = pil[..., :3].astype(np.int16) - stb[..., :3].astype(np.int16)
    vals, counts = np.unique(d, return_counts=True)
    assert np.array_equal(vals, sky_ref["skybox2_jpg_pil_minus_stb_values"]) and np.array_equal(counts, sky_ref["skybox2_jpg_pil_minus_stb_counts"])
    with pytest.raises(RuntimeError):
        po.ref_stb_load(str(tmp_path / "missing.jpg"))
