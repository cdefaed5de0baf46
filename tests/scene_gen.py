"""Seeded random scenes for the live comparisons with the reference's kernel body (tests/test_oracle_frames.py on
the CPU, tests/test_gpu_frames.py on the GPU): camera anywhere from inside the disk to far out, mostly looking
towards the hole, an orthonormal basis as CameraController::getCUDAStateFrom builds it (right = worldUp x forward,
up = forward x right; src/main.cpp:141-167), either sign of spin, media on or off, random time and effects, ragged
frame sizes."""
import numpy as np


def random_scene(rng, case):
    w, h = int(rng.integers(9, 56)), int(rng.integers(5, 34))
    rad = float(np.exp(rng.uniform(np.log(3.0), np.log(120.0))))
    ang = float(rng.uniform(0, 2 * np.pi))
    height = float(rng.normal(0, 0.15) * rad if case % 2 else rng.normal(0, 0.6))
    pos = np.array([rad * np.cos(ang), height, rad * np.sin(ang)], np.float32)
    fwd = -pos / np.linalg.norm(pos) + rng.normal(0, 0.35, 3)
    fwd = (fwd / np.linalg.norm(fwd)).astype(np.float32)
    right = np.cross(np.array([0, 1, 0], np.float32), fwd)
    right = (right / np.linalg.norm(right)).astype(np.float32)
    up = np.cross(fwd, right).astype(np.float32)
    f32 = lambda v: float(np.float32(v))
    return {
        "w": w, "h": h, "cam": np.stack([pos, fwd, right, up]).astype(np.float32),
        "spin": float(rng.choice([0.0, 0.3, 0.9, 0.99, -0.7])), "vol": int(rng.random() < 0.85),
        "t": f32(rng.uniform(0, 30)),
        "fx": dict(use_bloom=int(rng.integers(2)), use_vignette=int(rng.integers(2)), use_ca=int(rng.integers(2)),
                   use_lens=int(rng.integers(2)), bloom_threshold=f32(rng.uniform(0.3, 1.2)),
                   bloom_intensity=f32(rng.uniform(0.1, 1.0)), vignette_intensity=f32(rng.uniform(0.1, 0.8)),
                   ca_amount=f32(rng.uniform(0.0, 0.01)), distortion_amount=f32(rng.uniform(-0.2, 0.2))),
    }
