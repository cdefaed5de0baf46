"""dev tool (GPU): fit the coarse probe's cost model (csrc/rrt_hip.hip: probe_costs) to measured costs.
For each 4K view: the probe's three counts per row tile (RRT_PROBE_WEIGHTS = one unit weight at a time: steps, accretion
samples, dust samples) and the MEASURED cost of the same row tiles (rrt_tile_order's per-wave-tile clocks / 16 of a
single-kernel launch with the noise tables, summed per 16-row tile).  Least squares for (w_step, w_acc, w_dust); prints
the fit, its residuals per view, and how well a map dealt by the FITTED estimate balances the MEASURED costs of 8 shards
against t mod 8.  -> profiles/r04_probe_cost_fit.txt"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import relativisticraytracer_amd as rrt
from relativisticraytracer_amd.sky import synthetic_sky

W, H, R, G = 3840, 2160, 16, 8
VIEWS = {"default": ((0.0, 10.0, -60.0), 0.0, -10.0, 1.0), "key1": ((15.0, 3.0, -30.0), -26.6, -5.1, 6.0),
         "grazing": ((35.0, 0.8, 10.0), -106.0, -1.2, 12.0), "key3": ((5.0, 1.5, 50.0), -174.3, -1.7, 18.0),
         "skimmer": ((4.2, 0.6, 4.2), -90.0, -5.7, 14.0), "orbit": ((40.0, 2.0, 0.0), -90.0, 0.0, 3.0)}
tex = rrt.SkyTexture(synthetic_sky())
fx = rrt.CameraEffects()
nt = rrt.NoiseTable(32.0)
n_tiles = (H + R - 1) // R
feats, meas = {}, {}
for name, (pos, yaw, pitch, t) in VIEWS.items():
    cam = rrt.CameraState.from_angles(pos, yaw, pitch)
    prm = rrt.RenderParams(spin=0.9, noise_table=nt.id)
    f = []
    for wts in ("1,0,0", "0,1,0", "0,0,1"):
        os.environ["RRT_PROBE_WEIGHTS"] = wts
        f.append(rrt.probe_tile_costs(W, H, R, t, cam, fx, prm).astype(np.float64))
    os.environ.pop("RRT_PROBE_WEIGHTS")
    feats[name] = np.stack(f, 1)
    order = rrt.TileOrder(); order.set_seeding(False)
    out = torch.zeros(H * W * 4, dtype=torch.uint8, device="cuda")
    rrt.launch_raymarch(out, W, H, t, cam, tex, fx, rrt.RenderParams(spin=0.9, noise_table=nt.id, tile_order=order.id))
    torch.cuda.synchronize()
    cost = order.info(arrays=True)["cost"].astype(np.float64).reshape(H // 8, W // 8)       # wave tiles: row blocks of 8 LOCAL rows
    # local row lr of the full frame is image row lr (row map: one tile of `height` rows), so row block rb = image rows [8 rb, 8 rb + 8)
    meas[name] = cost.reshape(n_tiles, R // 8, W // 8).sum(axis=(1, 2)) / (W // 8) * (1.0)  # mean per wave-tile column x (R / 8) rows
    order.destroy()
A = np.concatenate([feats[k] for k in VIEWS]); b = np.concatenate([meas[k] for k in VIEWS])
A = A / (W / 16)            # probe_tile_costs sums the cells of a row: per cell (one cell per 2 wave-tile columns)
wts, *_ = np.linalg.lstsq(A, b, rcond=None)
print(f"# probe cost model fit over {len(VIEWS)} 4K views, {len(b)} row tiles: measured wave-tile cost (clocks / 16, single kernel + noise tables)")
print(f"w_step = {wts[0]:.1f}  w_acc = {wts[1]:.1f}  w_dust = {wts[2]:.1f}   (shipped: RRT_PROBE_W_STEP/ACC/DUST in csrc/rrt_hip.hip)")
for name in VIEWS:
    est = feats[name] / (W / 16) @ wts
    m = meas[name]
    rel = (est - m) / m.mean()
    dealt = rrt.balance_tiles(est.astype(np.float32), G)
    load = np.bincount(dealt, weights=m, minlength=G); modulo = np.bincount(np.arange(n_tiles) % G, weights=m, minlength=G)
    print(f"{name:8s} rms residual {np.sqrt((rel ** 2).mean()):.3f} of the mean tile, worst {np.abs(rel).max():.3f};  8 shards, MEASURED cost: "
          f"dealt by the estimate max/mean {load.max() / load.mean():.4f} (min/max {load.min() / load.max():.4f}),  t mod 8 max/mean "
          f"{modulo.max() / modulo.mean():.4f} (min/max {modulo.min() / modulo.max():.4f})")
