"""Independent numpy statement of the per-pixel glue around the march (raymarcher.cu:20-34 and :124-173):
pixel -> uv -> lens distortion -> ray direction, sky lookup, bloom, vignette, tone map, u8 truncation,
bottom-up rows.  With max_steps = 0 the march loop does not run, so the oracle's frame is exactly this glue;
the test rebuilds it in float32 numpy without reading the oracle's source and compares.  (The march itself
is pinned through its unit functions, tests/test_oracle_units.py.)"""
import numpy as np

f32 = np.float32


def _numpy_frame(cam, w, h, sky, fx, frac_bits=8):
    pos, fwd, right, up = [np.asarray(v, f32) for v in cam]
    x = np.arange(w, dtype=f32)[None, :].repeat(h, 0)
    y = np.arange(h, dtype=f32)[:, None].repeat(w, 1)
    uvx, uvy = x / f32(w), y / f32(h)
    if fx["lens"]:
        tx, ty = uvx - f32(0.5), uvy - f32(0.5)
        r2 = tx * tx + ty * ty
        f = f32(1.0) + r2 * f32(fx["k"])
        uvx, uvy = tx * f + f32(0.5), ty * f + f32(0.5)
    u = (uvx * f32(2) - f32(1)) * (f32(w) / f32(h))
    v = uvy * f32(2) - f32(1)
    d = fwd[None, None, :] + (right[None, None, :] * u[..., None] + up[None, None, :] * v[..., None])
    mag = np.sqrt(d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1] + d[..., 2] * d[..., 2])
    d = d / mag[..., None]
    d = d / np.sqrt((d * d).sum(-1, dtype=f32))[..., None]          # normalize(vel) again at :129
    pi = f32(3.1415926535)
    sh, sw = sky.shape[:2]

    def sample(off):
        phi = np.arctan2(d[..., 2], d[..., 0]).astype(f32) + f32(off)
        theta = np.arcsin(d[..., 1]).astype(f32)
        tx = f32(0.5) + phi / (f32(2) * pi)
        ty = f32(0.5) - theta / pi
        xb, yb = tx * f32(sw) - f32(0.5), ty * f32(sh) - f32(0.5)
        i, j = np.floor(xb), np.floor(yb)
        a, b = xb - i, yb - j
        q = f32(1 << frac_bits)
        a, b = np.floor(a * q + f32(0.5)) / q, np.floor(b * q + f32(0.5)) / q
        i0, i1 = np.mod(i, sw).astype(int), np.mod(i + 1, sw).astype(int)
        j0, j1 = np.clip(j, 0, sh - 1).astype(int), np.clip(j + 1, 0, sh - 1).astype(int)
        t = sky.astype(f32) / f32(255)
        return ((1 - a) * (1 - b))[..., None] * t[j0, i0] + (a * (1 - b))[..., None] * t[j0, i1] + \
               ((1 - a) * b)[..., None] * t[j1, i0] + (a * b)[..., None] * t[j1, i1]

    off = fx["ca"] if fx["use_ca"] else 0.0
    hdr = np.stack([sample(off)[..., 0], sample(0.0)[..., 1], sample(-off)[..., 2]], -1).astype(f32)
    if fx["bloom"]:
        lum = hdr[..., 0] * f32(0.2126) + hdr[..., 1] * f32(0.7152) + hdr[..., 2] * f32(0.0722)
        hdr = hdr + np.where((lum > f32(fx["bt"]))[..., None], hdr, f32(0)) * f32(fx["bi"])
    if fx["vig"]:
        dd = np.sqrt((uvx - f32(0.5)) ** 2 + (uvy - f32(0.5)) ** 2).astype(f32)
        t = np.clip((dd * f32(fx["vi"]) - f32(0.8)) / (f32(0.2) - f32(0.8)), 0, 1).astype(f32)
        hdr = hdr * (t * t * (f32(3) - f32(2) * t))[..., None]
    ldr = (f32(1) - np.exp(-hdr * f32(0.8))).astype(f32)
    u8 = np.concatenate([(ldr * f32(255)).astype(np.uint8), np.full((h, w, 1), 255, np.uint8)], -1)
    return u8[::-1], ldr[::-1]                                          # row (h-1-y), raymarcher.cu:168


def test_glue_matches_numpy_statement(po, sky):
    cam = ((3.0, 40.0, -20.0), (0.1, -0.8, 0.59), None, None)
    fwd = np.asarray(cam[1], f32); fwd /= np.linalg.norm(fwd)
    right = np.cross([0, 1, 0], fwd).astype(f32); right /= np.linalg.norm(right)
    up = np.cross(fwd, right).astype(f32)
    cam = (cam[0], fwd, right, up)
    w, h = 53, 31
    for fxo in (dict(lens=1, k=0.15, bloom=1, bt=0.8, bi=0.5, vig=1, vi=0.4, use_ca=0, ca=0.005),
                dict(lens=0, k=0.15, bloom=0, bt=0.8, bi=0.5, vig=0, vi=0.4, use_ca=1, ca=0.01),
                dict(lens=1, k=0.4, bloom=1, bt=0.1, bi=1.5, vig=1, vi=1.2, use_ca=1, ca=0.005)):
        fx = po.default_effects(use_lens=fxo["lens"], distortion_amount=fxo["k"], use_bloom=fxo["bloom"],
                                bloom_threshold=fxo["bt"], bloom_intensity=fxo["bi"], use_vignette=fxo["vig"],
                                vignette_intensity=fxo["vi"], use_ca=fxo["use_ca"], ca_amount=fxo["ca"])
        o = po.render(po.camera(*cam), fx, po.default_params(max_steps=0), 0.0, w, h, sky, want=("rgba8", "ldr"))
        u8, ldr = _numpy_frame(cam, w, h, sky, fxo)
        err = np.abs(o["ldr"][..., :3] - ldr)
        assert np.percentile(err, 99) <= 2e-5 and err.max() <= 0.02, (fxo, err.max())   # weight-bucket flips near texel edges
        du = np.abs(o["rgba8"].astype(int) - u8.astype(int))
        assert (du > 1).mean() <= 0.01 and np.all(o["rgba8"][..., 3] == 255)


def test_radiative_transfer_block_matches_numpy_statement(po, units_ref):
    """raymarcher.cu:71-116 written a second time, in float32 numpy, around the PINNED unit functions
    (redshift factor and disk temperature come from the reference's own vectors / oracle units)."""
    rng = np.random.default_rng(21)
    n = 1024
    p = units_ref["disk_p"][:n]; vel = units_ref["geo_v"][:n]
    d_disk = np.where(rng.random(n) < 0.7, rng.exponential(0.4, n), 0).astype(f32)
    d_cloud = np.where(rng.random(n) < 0.5, rng.exponential(0.2, n), 0).astype(f32)
    d_disk[:8] = [0.001, 0.0010001, 0, 0, 5, 0.0005, 30, 0.002]; d_cloud[:8] = [0, 0, 0.001, 0.0011, 0, 0.0005, 12, 0.5]
    h = f32(0.3) * f32([0.3, 0.1])[rng.integers(0, 2, n)]
    rad0 = np.concatenate([rng.exponential(0.3, (n, 3)), rng.uniform(0, 1, (n, 1))], 1).astype(f32)
    for spin in (0.0, 0.9):
        got = po.rt_sample(d_disk, d_cloud, p, vel, h, spin, rad0)
        g = po.units().redshift(p, vel, spin)                              # pinned (geodesics.h:11-25)
        r = np.sqrt((p * p).sum(1, dtype=f32)).astype(f32)
        T = po.units().disk_temperature(r)                                 # pinned (densities.h:12-15)
        tn = (T / f32(1.5e7)).astype(f32)
        on_d, on_c = d_disk > f32(0.001), d_cloud > f32(0.001)
        bol = (g ** f32(4) * tn ** f32(0.5) * d_disk * f32(6.0)).astype(f32)
        ct = (g * tn ** f32(0.4) * f32(2.5)).astype(f32)
        e = np.zeros((n, 3), f32)
        e[:, 0] += np.where(on_d, bol, 0)
        e[:, 1] += np.where(on_d, np.minimum(f32(0.25), f32(0.12) * ct) * bol, 0)
        e[:, 2] += np.where(on_d, np.maximum(f32(0), f32(0.01) * (ct - f32(2))) * bol, 0)
        light = (f32(0.5) + f32(3) * (f32(10) / np.maximum(r, f32(10))) ** f32(1.2)).astype(f32)
        ci = (d_cloud * f32(0.4) * light).astype(f32)
        s = np.clip((g - f32(0.7)) / (f32(1.3) - f32(0.7)), 0, 1).astype(f32); s = s * s * (f32(3) - f32(2) * s)
        for k, (base, a, b) in enumerate(((0.60, 1.2, 0.8), (0.65, 0.8, 1.1), (0.80, 0.6, 1.4))):
            e[:, k] += np.where(on_c, f32(base) * ci * (f32(a) + s * (f32(b) - f32(a))), 0)
        opac = np.where(on_d, d_disk * f32(0.4), 0) + np.where(on_c, d_cloud * f32(0.3), 0)
        st = np.exp(-(opac * h)).astype(f32)
        any_on = on_d | on_c
        fac = (f32(1) - st) * rad0[:, 3]
        want = rad0.copy()
        want[:, :3] += np.where(any_on[:, None], e * fac[:, None], 0)
        want[:, 3] = np.where(any_on, rad0[:, 3] * st, rad0[:, 3])
        assert np.allclose(got, want, rtol=3e-6, atol=1e-7), spin
        assert np.array_equal(got[~any_on], rad0[~any_on])                 # the d > 0.001f gates
        assert (~any_on).sum() > 50 and any_on.sum() > 500
