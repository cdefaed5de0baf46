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


def load_sky(path):
    """Decode an equirectangular image to (H, W, 4) uint8 RGBA, row 0 = top -- what the reference gets
    from stbi_load(..., 4) (src/main.cpp:240).  Decoders differ by +-1/255 on JPEGs; this uses PIL."""
    from PIL import Image
    with Image.open(path) as im:
        return np.ascontiguousarray(np.asarray(im.convert("RGBA"), dtype=np.uint8))
