"""dev tool (GPU, RRT_WAVETIME variant): per-wave duration of the composite pass for shard 0 of 8."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
os.environ["RRT_LIB_OVERRIDE"] = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "relativisticraytracer_amd/lib/variants/wavetime.so")
import numpy as np, torch
import relativisticraytracer_amd as rrt
from relativisticraytracer_amd import _lib
from relativisticraytracer_amd.sky import synthetic_sky
w, h, R, n = 3840, 2160, 16, 8
tex = rrt.SkyTexture(synthetic_sky()); ws = rrt.Workspace(8 << 30)
cam = rrt.CameraState.default(); fx = rrt.CameraEffects(); prm = rrt.RenderParams(spin=0.9, workspace=ws.id, path_policy=2)
buf = torch.zeros(h * w * 4, dtype=torch.uint8, device="cuda")
for rep in range(2):
    rrt.launch_raymarch_tiles(buf, w, h, R, 0, n, 1.0, cam, tex, fx, prm); torch.cuda.synchronize()
rows = rrt.tile_shard_rows(h, R, 0, n)
n_waves = ((w + 7) // 8) * ((rows + 7) // 8)
hdr = np.zeros((n_waves, 4), np.uint32)
_lib.check(_lib.load().rrt_workspace_read(ws.id, 256, hdr.nbytes, hdr.ctypes.data_as(C.c_void_p)), "read")
st = hdr[:, 2]; pad = hdr[:, 3]
walk = (pad & 0xFFFFFF).astype(np.float64) * 16 / 2.29e9 * 1e3; shade = (pad >> 24).astype(np.float64) * 256 / 2.29e9 * 1e3
d = walk[st == 1]; sh = shade[st == 1]; nr = hdr[:, 1][st == 1]
print("waves", n_waves, "deferred", int((st == 1).sum()), "runs max", int(nr.max()))
print("walk ms: mean %.4f p50 %.4f p99 %.4f max %.4f | shade ms: mean %.4f max %.4f" % (d.mean(), np.median(d), np.percentile(d, 99), d.max(), sh.mean(), sh.max()))
i = np.argsort(d)[-5:]; print("top walk:", [(round(float(d[k]), 3), int(nr[k])) for k in i])
