"""Helpers shared by the -m gpu tests: torch is only the device-memory plumbing."""
import ctypes as C

import numpy as np


def dev(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def host(t):
    import torch
    torch.cuda.synchronize()
    return t.cpu().numpy()


class HookSky:
    """A sky in librrt_hip_TEST.so's registry (the test library is a separate library: the package's SkyTexture lives in the
    product library's and is unknown to rrt_unit_sky_sample)."""

    def __init__(self, rgba8):
        from relativisticraytracer_amd import _lib
        arr = np.ascontiguousarray(rgba8, dtype=np.uint8)
        h = C.c_ulonglong(0)
        _lib.check(_lib.load_test().rrt_sky_create(arr.ctypes.data_as(C.c_void_p), arr.shape[1], arr.shape[0], C.byref(h)), "rrt_sky_create")
        self.handle = h.value

    def destroy(self):
        from relativisticraytracer_amd import _lib
        if self.handle:
            _lib.load_test().rrt_sky_destroy(self.handle)
            self.handle = 0


class HookNoiseTable:
    """A noise table in the test library's registry (see HookSky)."""

    def __init__(self, t_max, t0=0.0, coverage=0):
        from relativisticraytracer_amd import _lib
        i = C.c_int(0)
        _lib.check(_lib.load_test().rrt_noise_table_create_window(float(t0), float(t_max), int(coverage), C.byref(i)), "rrt_noise_table_create_window")
        self.id = i.value

    def info(self):
        from relativisticraytracer_amd import _lib
        t, b, boxes = C.c_float(0), C.c_size_t(0), (C.c_int * 12)()
        _lib.check(_lib.load_test().rrt_noise_table_info(self.id, C.byref(t), C.byref(b), C.byref(boxes)), "rrt_noise_table_info")
        return {"t_max": t.value, "bytes": b.value, "accretion_box": list(boxes[:6]), "dust_box": list(boxes[6:])}

    def destroy(self):
        from relativisticraytracer_amd import _lib
        if self.id:
            _lib.load_test().rrt_noise_table_destroy(self.id)
            self.id = 0


def unit(name, *args):
    """Call rrt_unit_<name> (include/rrt_test.h, librrt_hip_test.so) through the C ABI; tensors are passed as device pointers."""
    import torch
    from relativisticraytracer_amd import _lib
    lib = _lib.load_test()
    conv = []
    for a in args:
        if hasattr(a, "data_ptr"):
            conv.append(C.c_void_p(a.data_ptr()))
        elif a is None:
            conv.append(C.c_void_p(0))
        else:
            conv.append(a)
    conv.append(C.c_void_p(torch.cuda.current_stream().cuda_stream))
    _lib.check(getattr(lib, "rrt_unit_" + name)(*conv), "rrt_unit_" + name)


def render_gpu(w, h, spin, vol, cam, time, sky_tex, fx=None, debug=True, max_steps=2000, frac_bits=8, arith_mode=0,
               noise_table=0):
    """Full-frame render through rrt_launch_raymarch(_ex); returns numpy arrays."""
    import torch
    import relativisticraytracer_amd as rrt
    fx = fx or rrt.CameraEffects()
    prm = rrt.RenderParams(spin=spin, volumetrics=vol, max_steps=max_steps, sky_frac_bits=frac_bits,
                           arith_mode=arith_mode, noise_table=noise_table)
    out = torch.zeros(h * w * 4, dtype=torch.uint8, device="cuda")
    res = {}
    if debug:
        n = w * h
        bufs = dict(ldr=torch.zeros(n * 4, device="cuda"), hdr=torch.zeros(n * 4, device="cuda"),
                    steps=torch.zeros(n, dtype=torch.int32, device="cuda"),
                    hit=torch.zeros(n, dtype=torch.int32, device="cuda"),
                    pos=torch.zeros(n * 3, device="cuda"), vel=torch.zeros(n * 3, device="cuda"),
                    rad=torch.zeros(n * 4, device="cuda"),
                    lut_oob=torch.zeros(1, dtype=torch.int32, device="cuda"))
        rrt.launch_raymarch_debug(out, w, h, time, cam, sky_tex, fx, prm, **bufs)
        torch.cuda.synchronize()
        res = {k: v.cpu().numpy() for k, v in bufs.items()}
        res["ldr"] = res["ldr"].reshape(h, w, 4); res["hdr"] = res["hdr"].reshape(h, w, 4)
        res["pos"] = res["pos"].reshape(n, 3); res["vel"] = res["vel"].reshape(n, 3)
        res["rad"] = res["rad"].reshape(n, 4)
    else:
        rrt.launch_raymarch(out, w, h, time, cam, sky_tex, fx, prm)
        torch.cuda.synchronize()
    res["rgba8"] = out.cpu().numpy().reshape(h, w, 4)
    return res
