"""dev tool (GPU): cost of one media evaluation as the render kernels run it (rrt_unit_media_lut: early-outs on, table
switches on), per sample class, from wave-coherent synthetic sample points (64 neighbours a few hundredths of a unit
apart, as the 8x8-pixel wavefronts of a 4K frame produce).  Prints microseconds per million samples and the implied
VALU issue slots per sample (at 1024 SIMDs x 2.25 GHz / 2 cycles per wave-instruction)."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import relativisticraytracer_amd as rrt
import gpu_util as g

nt = rrt.NoiseTable(16.0)
rng = np.random.default_rng(3)
waves = 1 << 16


def pts(y_lo, y_hi, rc_lo=10.5, rc_hi=24.5, spread=0.03):
    rc = rng.uniform(rc_lo, rc_hi, waves); ang = rng.uniform(-np.pi, np.pi, waves)
    yc = rng.uniform(y_lo, y_hi, waves) * rng.choice([-1.0, 1.0], waves)
    c = np.stack([rc * np.cos(ang), yc, rc * np.sin(ang)], 1)
    return (c[:, None, :] + spread * rng.uniform(-0.5, 0.5, (waves, 64, 3))).reshape(-1, 3).astype(np.float32)


classes = {"cloud zone core |y|<0.5 (accretion + dust, 24 noise3D)": pts(0.0, 0.5),
           "disk only 0.9<|y|<1.8 (accretion, 5 noise3D)": pts(0.9, 1.8),
           "disk tail 2.6<|y|<3.2 at rc<14 (accretion envelope small)": pts(2.6, 3.2, 10.5, 14.0),
           "beyond the slab 3.7<|y|<3.95 (early-out)": pts(3.7, 3.95),
           "outside the radial gate rc<9.5": pts(0.0, 3.0, 4.0, 9.5),
           "incoherent cloud core (spread 3: tables off)": pts(0.0, 0.5, spread=3.0)}
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
rate = 1024 * 2.25e9 / 2           # wave-instructions per second, whole chip
for name, p in classes.items():
    n = len(p)
    d = g.dev(p)
    disk = torch.empty(n, device="cuda"); dust = torch.empty(n, device="cuda")
    cnt = torch.zeros(1, dtype=torch.int32, device="cuda")
    for tab in (nt.id,):
        g.unit("media_lut", n, d, 7.5, tab, disk, dust, cnt)
        torch.cuda.synchronize()
        ts = []
        for _ in range(3):
            e0.record(); g.unit("media_lut", n, d, 7.5, tab, disk, dust, cnt); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        ms = min(ts)
        print(f"{name:62s} {ms * 1e3 / (n / 1e6):8.1f} us/Msample  ~{ms * 1e-3 * rate / (n / 64):7.0f} issue slots/sample  "
              f"live disk {float((disk > 0.001).float().mean()):.2f} dust {float((dust > 0.001).float().mean()):.2f}", flush=True)
nt.destroy()
