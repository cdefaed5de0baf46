"""Decode a sky image ONCE with the reference's own decoder and write it as a raw sky (relativisticraytracer_amd.sky:
save_sky_raw), which the package's load_sky reads back texel for texel -- SURVEY.md row f1: "JPEG decode via a decoder
other than stb_image may differ by +-1/255: decode once, ship as raw".
The decoder is oracle/_ref/libref_stb.so = /root/reference/include/stb_image.h compiled where it lies (oracle/Makefile),
so this runs in the build container; the raw file travels.  --decoder pil writes what the package's own fallback decodes.
    python tools/sky_to_raw.py /root/reference/assets/skyboxes/skybox2.jpg sky.rrtsky"""
import argparse, hashlib, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from relativisticraytracer_amd import sky


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("image"); ap.add_argument("out")
    ap.add_argument("--decoder", choices=("stb", "pil"), default="stb")
    a = ap.parse_args()
    if a.decoder == "stb":
        from oracle import pyoracle as po
        if not po.ref_stb_available():
            po.build(ref=True)
        rgba, channels = po.ref_stb_load(a.image)
    else:
        from PIL import Image
        import numpy as np
        with Image.open(a.image) as im:
            rgba = np.ascontiguousarray(np.asarray(im.convert("RGBA"), dtype=np.uint8)); channels = len(im.getbands())
    sky.save_sky_raw(a.out, rgba)
    print(f"{a.out}: {rgba.shape[1]}x{rgba.shape[0]} RGBA8 ({channels} channels in the file), decoder {a.decoder}, "
          f"sha256 {hashlib.sha256(rgba.tobytes()).hexdigest()}")


if __name__ == "__main__":
    main()
