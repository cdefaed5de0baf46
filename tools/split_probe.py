"""dev tool (GPU): is it worth running the rays that never come near the hole in a lean (64-VGPR, 8 waves/SIMD)
kernel CONCURRENTLY with the media-carrying kernel (96 VGPRs, 5 waves/SIMD) for the rest?  Crude rehearsal with what
the library already has: the rows of the 4K bench frame whose pixels differ between volumetrics on / off (+ a margin)
go through the full kernel on one stream, the rows above and below through the no-media kernel on another (same bytes
there), against the single full launch."""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import relativisticraytracer_amd as rrt
from relativisticraytracer_amd.sky import synthetic_sky
w, h = 3840, 2160
tex = rrt.SkyTexture(synthetic_sky()); cam = rrt.CameraState.default(); fx = rrt.CameraEffects()
nt = rrt.NoiseTable(32.0)
full = torch.zeros(h * w * 4, dtype=torch.uint8, device="cuda"); nov = torch.zeros_like(full); out = torch.zeros_like(full)
pv = rrt.RenderParams(spin=0.9, noise_table=nt.id); pn = rrt.RenderParams(spin=0.9, volumetrics=0)
rrt.launch_raymarch(full, w, h, 1.0, cam, tex, fx, pv); rrt.launch_raymarch(nov, w, h, 1.0, cam, tex, fx, pn)
torch.cuda.synchronize()
diff = (full.view(h, w, 4) != nov.view(h, w, 4)).any(dim=2).any(dim=1).nonzero().flatten()
lo, hi = int(diff.min()), int(diff.max()) + 1           # stored rows (bottom-up); image rows y = h-1-row
print("rows that differ between media on/off (stored order):", lo, hi, "of", h)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)


def t(fn, n=4):
    fn(); torch.cuda.synchronize(); ts = []
    for _ in range(n):
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return min(ts)


def single():
    rrt.launch_raymarch(out, w, h, 1.0, cam, tex, fx, pv)


for margin in (64, 160, 320):
    a, b = max(0, lo - margin) // 8 * 8, min(h, (hi + margin + 7) // 8 * 8)      # stored rows [a, b) = image rows [h-b, h-a)
    y0, y1 = h - b, h - a

    def split():
        cur = torch.cuda.current_stream()
        s1.wait_stream(cur); s2.wait_stream(cur)
        with torch.cuda.stream(s1):
            rrt.launch_raymarch_rows(out[a * w * 4:], w, h, y0, y1, 1.0, cam, tex, fx, pv)
        with torch.cuda.stream(s2):
            if y0 > 0:
                rrt.launch_raymarch_rows(out[b * w * 4:], w, h, 0, y0, 1.0, cam, tex, fx, pn)
            if y1 < h:
                rrt.launch_raymarch_rows(out, w, h, y1, h, 1.0, cam, tex, fx, pn)
        cur.wait_stream(s1); cur.wait_stream(s2)

    out.zero_(); split(); torch.cuda.synchronize()
    same = bool(torch.equal(out, full))
    ts, tsp = t(single), t(split)

    def inner_only():
        rrt.launch_raymarch_rows(out[a * w * 4:], w, h, y0, y1, 1.0, cam, tex, fx, pv)

    def outer_only():
        if y0 > 0:
            rrt.launch_raymarch_rows(out[b * w * 4:], w, h, 0, y0, 1.0, cam, tex, fx, pn)
        if y1 < h:
            rrt.launch_raymarch_rows(out, w, h, y1, h, 1.0, cam, tex, fx, pn)
    print(f"margin {margin}: media rows {y1 - y0} of {h}; single {ts:.2f} ms, split on two streams {tsp:.2f} ms "
          f"(inner alone {t(inner_only):.2f}, outer alone {t(outer_only):.2f}); bytes equal: {same}", flush=True)
