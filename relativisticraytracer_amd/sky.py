"""Synthetic equirectangular sky (RGBA8), the bench/test stand-in for the
reference's assets/skyboxes/skybox2.jpg (loaded at src/main.cpp:497).

Integer arithmetic only, so the bytes are identical on every machine.
"""
import numpy as np


def _mix(a):
    a = a.astype(np.uint64)
    a = (a ^ (a >> np.uint64(16))) * np.uint64(0x7FEB352D) & np.uint64(0xFFFFFFFF)
    a = (a ^ (a >> np.uint64(15))) * np.uint64(0x846CA68B) & np.uint64(0xFFFFFFFF)
    a = a ^ (a >> np.uint64(16))
    return a.astype(np.uint32)


def synthetic_sky(width=2048, height=1024, seed=1):
    """Low-frequency gradient + a brighter band along the equator + sparse stars."""
    j, i = np.meshgrid(np.arange(height, dtype=np.int64), np.arange(width, dtype=np.int64), indexing="ij")
    tri_i = np.abs((i * 4 * 256 // width) % 512 - 256)          # 0..256 triangle wave, 2 periods
    tri_j = np.abs((j * 2 * 256 // height) % 512 - 256)
    band = np.clip(96 - np.abs(j - height // 2) * 96 * 6 // height, 0, 96)
    r = 10 + tri_i * 30 // 256 + band * 2 // 3
    g = 12 + tri_j * 26 // 256 + band // 2
    b = 28 + (tri_i + tri_j) * 20 // 256 + band
    h = _mix((i + j * width + np.int64(seed) * 0x9E3779B1).astype(np.uint64) & np.uint64(0xFFFFFFFF))
    star = (h % np.uint32(641)) == 0
    mag = 96 + ((h >> np.uint32(11)) % np.uint32(160)).astype(np.int64)
    tint = ((h >> np.uint32(20)) % np.uint32(48)).astype(np.int64)
    r = np.where(star, np.minimum(255, mag + tint), r)
    g = np.where(star, mag, g)
    b = np.where(star, np.minimum(255, mag + 48 - tint), b)
    out = np.empty((height, width, 4), np.uint8)
    out[..., 0] = np.clip(r, 0, 255)
    out[..., 1] = np.clip(g, 0, 255)
    out[..., 2] = np.clip(b, 0, 255)
    out[..., 3] = 255
    return out


RAW_MAGIC = b"RRTSKY1\n"


def save_sky_raw(path, rgba8):
    """Write a decoded sky as a raw file: b"RRTSKY1\\n", b"<width> <height>\\n", then height x width RGBA8 texels, row 0 =
    top -- the bytes stbi_load(..., 4) returns (src/main.cpp:240) and cudaMemcpy2DToArray uploads (:247), so that a sky
    can be DECODED ONCE with the reference's own decoder and shipped (JPEG decoders differ)."""
    a = np.ascontiguousarray(rgba8, dtype=np.uint8)
    if a.ndim != 3 or a.shape[2] != 4:
        raise ValueError("sky must be an (H, W, 4) uint8 array")
    with open(path, "wb") as fh:
        fh.write(RAW_MAGIC + f"{a.shape[1]} {a.shape[0]}\n".encode())
        fh.write(a.tobytes())


def load_sky_raw(path):
    """Read a raw sky written by save_sky_raw (or tools/sky_to_raw.py): exactly the texels that were decoded, no decoder
    involved.  Raises ValueError on a file that is not one."""
    with open(path, "rb") as fh:
        if fh.read(len(RAW_MAGIC)) != RAW_MAGIC:
            raise ValueError(f"{path}: not a raw sky (no RRTSKY1 header)")
        dims = fh.readline(64).split()
        if len(dims) != 2:
            raise ValueError(f"{path}: bad raw sky header")
        w, h = int(dims[0]), int(dims[1])
        if w <= 0 or h <= 0 or w * h > (1 << 30):
            raise ValueError(f"{path}: bad raw sky size {w}x{h}")
        data = fh.read(w * h * 4 + 1)
    if len(data) != w * h * 4:
        raise ValueError(f"{path}: raw sky holds {len(data)} bytes, expected {w * h * 4}")
    return np.frombuffer(data, np.uint8).reshape(h, w, 4).copy()


def is_raw_sky(path):
    try:
        with open(path, "rb") as fh:
            return fh.read(len(RAW_MAGIC)) == RAW_MAGIC
    except OSError:
        return False


def load_sky(path):
    """An equirectangular sky as (H, W, 4) uint8 RGBA, row 0 = top -- what the reference gets from
    stbi_load(filename, ..., 4) (src/main.cpp:240).
    A raw sky (save_sky_raw / tools/sky_to_raw.py) is read as it is: texel for texel what the reference's decoder gave.
    Anything else is decoded with PIL, which is NOT the reference's decoder: on the reference's own asset
    assets/skyboxes/skybox2.jpg (4096x2048) PIL's libjpeg and stb_image v2.30 disagree on 0.90 % of the colour bytes --
    by 1 on 226 364 of them, by 2 on 1 397, by 3 on one (tests/golden/sky_ref.npz, tests/test_sky_loader.py).  For a
    reference-exact render decode once with the reference's decoder and ship the raw file."""
    if is_raw_sky(path):
        return load_sky_raw(path)
    from PIL import Image
    with Image.open(path) as im:
        return np.ascontiguousarray(np.asarray(im.convert("RGBA"), dtype=np.uint8))
