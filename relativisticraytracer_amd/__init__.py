"""relativisticraytracer_amd -- MI355X (gfx950) implementation of the per-pixel
geodesic ray-march hot path of levi2234/RelativisticRayTracer.

Host-side mirror of the reference's interface for this path
(include/raymarcher.h:11-19, include/camera_effects/camera_settings.h:4-17):

    cam = rrt.CameraState.from_angles((0, 10, -60), yaw=0, pitch=-10)
    sky = rrt.SkyTexture(rgba8_numpy)
    out = torch.empty(h * w * 4, dtype=torch.uint8, device="cuda")
    rrt.launch_raymarch(out, w, h, time, cam, sky, rrt.CameraEffects())

Everything goes through the C ABI of librrt_hip.so (include/rrt.h).  torch is
only used for device memory / streams and is optional: any object with a
`data_ptr()` or a plain integer device address is accepted.  There is no CPU
implementation in this package.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import RRTError, rrt_camera, rrt_debug_outputs, rrt_effects, rrt_params  # noqa: F401

__all__ = ["CameraState", "CameraEffects", "RenderParams", "SkyTexture", "Workspace", "NoiseTable", "launch_raymarch",
           "set_launch_defaults", "get_launch_defaults",
           "launch_raymarch_rows", "launch_raymarch_tiles", "assemble_tiles", "assemble_all_tiles",
           "tile_shard_rows",
           "launch_raymarch_debug", "RRTError", "device_count", "abi_version", "TileOrder", "TileMap",
           "probe_tile_costs", "balance_tiles", "launch_raymarch_tilemap", "assemble_all_tilemap", "clock_probe", "clock_probe_ghz"]


def _ptr(x):
    if x is None:
        return None
    if isinstance(x, int):
        return C.c_void_p(x)
    if hasattr(x, "data_ptr"):
        return C.c_void_p(x.data_ptr())
    raise TypeError(f"expected a device tensor or an integer device address, got {type(x)}")


def _stream(stream):
    if stream is None:
        try:
            import torch
            if torch.cuda.is_available():
                return C.c_void_p(torch.cuda.current_stream().cuda_stream)
        except Exception:
            pass
        return None
    if isinstance(stream, int):
        return C.c_void_p(stream)
    return C.c_void_p(stream.cuda_stream)


class CameraState(rrt_camera):
    """Reference `struct CameraState` (include/raymarcher.h:11-16)."""

    def __init__(self, pos=(0, 0, 0), forward=(0, 0, 1), right=(1, 0, 0), up=(0, 1, 0)):
        super().__init__()
        for name, v in (("pos", pos), ("forward", forward), ("right", right), ("up", up)):
            arr = getattr(self, name)
            for k in range(3):
                arr[k] = float(v[k])

    @classmethod
    def from_angles(cls, pos, yaw, pitch):
        """CameraController::getCUDAStateFrom (src/main.cpp:141-167); degrees."""
        out = cls()
        p = (C.c_float * 3)(*[float(v) for v in pos])
        _lib.check(_lib.load().rrt_camera_from_angles(C.byref(p), float(yaw), float(pitch), C.byref(out)),
                   "rrt_camera_from_angles")
        return out

    @classmethod
    def default(cls):
        """The reference's start-up camera (src/main.cpp:128-130)."""
        return cls.from_angles((0.0, 10.0, -60.0), 0.0, -10.0)

    def as_array(self):
        return np.array([list(self.pos), list(self.forward), list(self.right), list(self.up)], np.float32)


class CameraEffects(rrt_effects):
    """Reference `struct CameraEffects` with its default member initialisers
    (camera_settings.h:5-16); attribute names follow the reference."""

    _ALIASES = {"useBloom": "use_bloom", "bloomThreshold": "bloom_threshold",
                "bloomIntensity": "bloom_intensity", "useVignette": "use_vignette",
                "vignetteIntensity": "vignette_intensity",
                "useChromaticAberration": "use_chromatic_aberration", "caAmount": "ca_amount",
                "useLensDistortion": "use_lens_distortion", "distortionAmount": "distortion_amount"}

    def __init__(self, **kw):
        super().__init__()
        _lib.check(_lib.load().rrt_effects_default(C.byref(self)), "rrt_effects_default")
        for k, v in kw.items():
            setattr(self, k, v)

    def __setattr__(self, k, v):
        k = self._ALIASES.get(k, k)
        if k.startswith("use_"):
            v = 1 if v else 0
        super().__setattr__(k, v)

    def __getattr__(self, k):
        alias = type(self)._ALIASES.get(k)
        if alias is None:
            raise AttributeError(k)
        return getattr(self, alias)


class RenderParams(rrt_params):
    """Scene constants of include/config.h as run-time parameters (defaults == config.h)."""

    def __init__(self, **kw):
        super().__init__()
        _lib.check(_lib.load().rrt_params_init(C.byref(self), C.sizeof(rrt_params)), "rrt_params_init")
        for k, v in kw.items():
            if not hasattr(self, k):
                raise AttributeError(k)
            setattr(self, k, v)


class SkyTexture:
    """Device-resident RGBA8 equirectangular sky; stands in for the reference's
    cudaTextureObject_t (src/main.cpp:237-266)."""

    def __init__(self, rgba8):
        arr = np.ascontiguousarray(rgba8, dtype=np.uint8)
        if arr.ndim != 3 or arr.shape[2] != 4:
            raise ValueError("sky must be an (H, W, 4) uint8 array")
        self.height, self.width = arr.shape[:2]
        h = C.c_ulonglong(0)
        _lib.check(_lib.load().rrt_sky_create(arr.ctypes.data_as(C.c_void_p), self.width, self.height,
                                              C.byref(h)), "rrt_sky_create")
        self.handle = h.value

    def destroy(self):
        if getattr(self, "handle", 0):
            _lib.load().rrt_sky_destroy(self.handle)
            self.handle = 0

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


class Workspace:
    """Caller-owned HBM pool for the three-pass path (RenderParams.workspace = ws.id)."""

    def __init__(self, nbytes):
        i = C.c_int(0)
        _lib.check(_lib.load().rrt_workspace_create(int(nbytes), C.byref(i)), "rrt_workspace_create")
        self.id, self.nbytes = i.value, int(nbytes)

    def stats(self):
        """Of the last launch (synchronous): rows pooled over all its rounds, wavefronts still suspended after the last
        round (finished in line), rounds enqueued / with work, rows of the fullest round, rows the pool holds at most."""
        rows, ovf = C.c_uint(0), C.c_uint(0)
        _lib.check(_lib.load().rrt_workspace_stats(self.id, C.byref(rows), C.byref(ovf)), "rrt_workspace_stats")
        a, b, c, d = C.c_uint(0), C.c_uint(0), C.c_uint(0), C.c_uint(0)
        _lib.check(_lib.load().rrt_workspace_rounds(self.id, C.byref(a), C.byref(b), C.byref(c), C.byref(d)), "rrt_workspace_rounds")
        return {"rows_used": rows.value, "overflow_waves": ovf.value, "rounds_enqueued": a.value, "rounds_with_work": b.value,
                "peak_rows": c.value, "pool_rows": d.value}

    def destroy(self):
        if getattr(self, "id", 0):
            _lib.load().rrt_workspace_destroy(self.id)
            self.id = 0

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


class TileOrder:
    """Cost-ordered dispatch (RenderParams.tile_order = order.id): a launch through the object records what every 8x8-pixel
    wave tile cost, and the next launch with the same geometry dispatches longest-first.  Same pixels; removes the drain
    of views whose long rays are not in the middle of the frame.  Launches through one object are serialised on the
    device: frames that should overlap need one each."""

    def __init__(self):
        i = C.c_int(0)
        _lib.check(_lib.load().rrt_tile_order_create(C.byref(i)), "rrt_tile_order_create")
        self.id = i.value

    def info(self, arrays=False):
        """{"launches", "ordered_launches", "n_tiles"}; with arrays=True also "perm" (the order the next matching launch
        will use) and "cost" (shader clocks / 16 per wave tile of the last launch) as numpy arrays (synchronises)."""
        import numpy as np
        lib = _lib.load()
        a, b, n = C.c_ulonglong(0), C.c_ulonglong(0), C.c_uint(0)
        _lib.check(lib.rrt_tile_order_info(self.id, C.byref(a), C.byref(b), C.byref(n), None, None, 0), "rrt_tile_order_info")
        out = {"launches": a.value, "ordered_launches": b.value, "n_tiles": n.value}
        if arrays and n.value:
            perm = np.empty(n.value, np.uint32); cost = np.empty(n.value, np.uint32)
            _lib.check(lib.rrt_tile_order_info(self.id, None, None, None, perm.ctypes.data, cost.ctypes.data, n.value), "rrt_tile_order_info")
            out["perm"], out["cost"] = perm, cost
        return out

    def set_seeding(self, on):
        """on (default): a launch without history for its geometry takes its order from a coarse probe of the view."""
        _lib.check(_lib.load().rrt_tile_order_set_seeding(self.id, 1 if on else 0), "rrt_tile_order_set_seeding")

    def seeded_launches(self):
        n = C.c_ulonglong(0)
        _lib.check(_lib.load().rrt_tile_order_seeded(self.id, C.byref(n)), "rrt_tile_order_seeded")
        return n.value

    def destroy(self):
        if getattr(self, "id", 0):
            # the object must be destroyed under the device that owns it (include/rrt.h); a refused destroy frees nothing and keeps
            # the handle valid, so the status is NOT ignored here: the wrapper keeps its id and the caller hears of it
            _lib.check(_lib.load().rrt_tile_order_destroy(self.id), "rrt_tile_order_destroy")
            self.id = 0

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


class TileMap:
    """Explicit tile -> shard assignment (cost-weighted instead of t mod n_shards).  shard_of_tile[t] for
    every row tile t of `tile_rows` image rows; device-resident, tied to the current device."""

    def __init__(self, height, tile_rows, n_shards, shard_of_tile):
        m = np.ascontiguousarray(shard_of_tile, dtype=np.int32)
        if m.shape != ((height + tile_rows - 1) // tile_rows,):
            raise ValueError("shard_of_tile must have one entry per row tile")
        i = C.c_int(0)
        _lib.check(_lib.load().rrt_tile_map_create(height, tile_rows, n_shards, m.ctypes.data, C.byref(i)), "rrt_tile_map_create")
        self.id, self.height, self.tile_rows, self.n_shards, self.shard_of_tile = i.value, height, tile_rows, n_shards, m

    def shard_rows(self, shard):
        r = C.c_int(0)
        _lib.check(_lib.load().rrt_tile_map_shard_rows(self.id, shard, C.byref(r), None), "rrt_tile_map_shard_rows")
        return r.value

    def max_shard_rows(self):
        r = C.c_int(0)
        _lib.check(_lib.load().rrt_tile_map_shard_rows(self.id, 0, None, C.byref(r)), "rrt_tile_map_shard_rows")
        return r.value

    def destroy(self):
        if getattr(self, "id", 0):
            _lib.check(_lib.load().rrt_tile_map_destroy(self.id), "rrt_tile_map_destroy")      # as TileOrder.destroy
            self.id = 0

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


def probe_tile_costs(w, h, tile_rows, time, cam, effects, params=None, stream=None):
    """Estimated cost of every row tile from the coarse march-only probe of the view (synchronous; float32 array)."""
    n = (h + tile_rows - 1) // tile_rows
    out = np.zeros(n, np.float32)
    _lib.check(_lib.load().rrt_probe_tile_costs(w, h, tile_rows, float(time), C.byref(cam), C.byref(effects),
                                                C.byref(params) if params is not None else None, out.ctypes.data, n,
                                                _stream(stream)), "rrt_probe_tile_costs")
    return out


def balance_tiles(tile_cost, n_shards, max_tiles_per_shard=0):
    """rrt_tile_map_balance: tiles dealt longest-first to the least loaded shard (deterministic host arithmetic)."""
    c = np.ascontiguousarray(tile_cost, dtype=np.float32)
    out = np.zeros(len(c), np.int32)
    _lib.check(_lib.load().rrt_tile_map_balance(len(c), c.ctypes.data, n_shards, max_tiles_per_shard, out.ctypes.data),
               "rrt_tile_map_balance")
    return out


def clock_probe(d_counters2, duration_us, stream=None):
    """rrt_clock_probe, asynchronous: one wavefront sleeps `duration_us` on `stream` and leaves the shader-clock and
    100 MHz counter deltas in d_counters2 (two int64): d[0] / d[1] * 0.1 = GHz held over that time."""
    _lib.check(_lib.load().rrt_clock_probe(_ptr(d_counters2), int(duration_us), _stream(stream)), "rrt_clock_probe")


def clock_probe_ghz(duration_us=20000, stream=None):
    """The shader clock the chip holds right now (GHz): rrt_clock_probe on `stream` (default: a side stream, so that it
    runs BESIDE whatever the current stream is busy with), synchronous."""
    import torch
    buf = torch.zeros(2, dtype=torch.int64, device="cuda")
    st = stream if stream is not None else torch.cuda.Stream()
    _lib.check(_lib.load().rrt_clock_probe(_ptr(buf), int(duration_us), _stream(st)), "rrt_clock_probe")
    st.synchronize()
    c = buf.cpu().numpy()
    return float(c[0]) / max(float(c[1]), 1.0) * 0.1


TABLE_FULL, TABLE_COARSE, TABLE_COARSEST = 0, 1, 2      # rrt.h: which noise call families a table serves
TABLE_BANDED, TABLE_DENSE = 16, 32                        # ... ORed in: force a layout of the dust families (default: automatic)


class NoiseTable:
    """Caller-owned lattice-hash tables for the volumetric noise (RenderParams.noise_table = nt.id): hash31 of every
    lattice point the table-served noise3D calls can reach for t0 <= time <= t1 (NoiseTable(t_max) = [0, t_max]).
    A launch whose time lies outside the window renders with the arithmetic kernels: same bytes, slower."""

    def __init__(self, t_max=32.0, t0=0.0, coverage=TABLE_FULL):
        i = C.c_int(0)
        _lib.check(_lib.load().rrt_noise_table_create_window(float(t0), float(t_max), int(coverage), C.byref(i)),
                   "rrt_noise_table_create_window")
        self.id, self.t0, self.t1, self.t_max, self.coverage = i.value, float(t0), float(t_max), float(t_max), int(coverage)

    @classmethod
    def window(cls, t0, t1, coverage=TABLE_FULL):
        return cls(t_max=t1, t0=t0, coverage=coverage)

    def covers(self, time):
        return self.t0 <= time <= self.t1

    def info(self):
        t, b, box = C.c_float(0), C.c_size_t(0), (C.c_int * 12)()
        _lib.check(_lib.load().rrt_noise_table_info(self.id, C.byref(t), C.byref(b), C.byref(box)), "rrt_noise_table_info")
        t0, t1, cov, dev = C.c_float(0), C.c_float(0), C.c_int(0), C.c_int(0)
        _lib.check(_lib.load().rrt_noise_table_window(self.id, C.byref(t0), C.byref(t1), C.byref(cov), C.byref(dev)),
                   "rrt_noise_table_window")
        return {"t_max": t.value, "t0": t0.value, "t1": t1.value, "coverage": cov.value, "device": dev.value,
                "bytes": b.value, "accretion_box": list(box[:6]), "dust_box": list(box[6:])}

    @staticmethod
    def plan(t_max, t0=0.0, coverage=TABLE_FULL):
        """Host arithmetic only: size and boxes of such a table; raises RRTError(INVALID_ARGUMENT) for a box that
        create would refuse as well."""
        b, box = C.c_size_t(0), (C.c_int * 12)()
        _lib.check(_lib.load().rrt_noise_table_plan_window(float(t0), float(t_max), int(coverage), C.byref(b), C.byref(box)),
                   "rrt_noise_table_plan_window")
        return {"t_max": float(t_max), "t0": float(t0), "coverage": int(coverage), "bytes": b.value,
                "accretion_box": list(box[:6]), "dust_box": list(box[6:])}

    @staticmethod
    def plan_layout(t_max, t0=0.0, coverage=TABLE_FULL):
        """Host arithmetic only: {"banded": bool} and, for a banded table, "n_bands", "w_min", "w_scale", "band_boxes"
        (3, n_bands, 6) -- ridge octave 1, ridge octave 2, detail octave -- and "acc_octave_boxes" (4, 6)."""
        banded, nb, w0, ws = C.c_int(0), C.c_int(0), C.c_float(0), C.c_float(0)
        boxes = np.zeros((3, 64, 6), np.int32); acc = np.zeros((4, 6), np.int32)
        _lib.check(_lib.load().rrt_noise_table_plan_layout(float(t0), float(t_max), int(coverage), C.byref(banded), C.byref(nb), C.byref(w0),
                                                           C.byref(ws), boxes.ctypes.data, 64, acc.ctypes.data), "rrt_noise_table_plan_layout")
        if not banded.value:
            return {"banded": False}
        return {"banded": True, "n_bands": nb.value, "w_min": w0.value, "w_scale": ws.value,
                "band_boxes": boxes[:, :nb.value].copy(), "acc_octave_boxes": acc}

    @staticmethod
    def fit(t_from, t_until, budget_bytes):
        """The frame drivers' policy (rrt_noise_table_fit_window): longest window from t_from, richest coverage,
        within the byte budget -> (t1, coverage, bytes); bytes == 0: nothing fits."""
        t1, cov, b = C.c_float(0), C.c_int(0), C.c_size_t(0)
        _lib.check(_lib.load().rrt_noise_table_fit_window(float(t_from), float(t_until), int(budget_bytes), C.byref(t1),
                                                          C.byref(cov), C.byref(b)), "rrt_noise_table_fit_window")
        return t1.value, cov.value, b.value

    def destroy(self):
        if getattr(self, "id", 0):
            _lib.load().rrt_noise_table_destroy(self.id)
            self.id = 0

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


class NoiseWindows:
    """What a frame loop does about the noise tables as the clock runs (src/main.cpp:515 lets simTime grow without
    bound): keeps ONE table whose window holds the current frame time, within a byte budget; when the clock leaves
    the window it builds the next one (a few milliseconds) -- after `sync()`, since frames in flight may still read
    the old table.  `table_id(t)` is what goes into RenderParams.noise_table (0: this frame hashes arithmetically;
    counted in `arith_frames`, never silent)."""

    def __init__(self, t_end, budget_bytes, sync=None, enabled=True):
        self.t_end, self.budget, self.sync, self.enabled = float(t_end), int(budget_bytes), sync, enabled
        self.table = None
        self.failed = None          # (t0, t1) of a window that could not be built: frames in it hash arithmetically, no retry
        self.retry_after = 5.0
        self.builds, self.table_frames, self.arith_frames, self.coarsest = 0, 0, 0, TABLE_FULL
        self.peak_bytes = 0

    def table_id(self, t):
        if not self.enabled:
            return 0
        # a window that could not be built (nothing fits the budget; out of memory) is REMEMBERED: no new fit, no device
        # synchronise and no retry until the clock has left it -- retrying every frame serialised the frames in flight for
        # the rest of the run
        if self.failed is not None and self.failed[0] <= t <= self.failed[1]:
            self.arith_frames += 1
            return 0
        if self.table is None or not self.table.covers(t):
            t1, cov, nbytes = NoiseTable.fit(t, max(t, self.t_end), self.budget)
            if self.sync:
                self.sync()
            if self.table is not None:
                self.table.destroy()
                self.table = None
            self.failed = None
            if nbytes:
                try:
                    self.table = NoiseTable.window(t, t1, cov)
                    self.builds += 1
                    self.coarsest = max(self.coarsest, cov)
                    self.peak_bytes = max(self.peak_bytes, nbytes)
                except _lib.RRTError as e:          # e.g. out of memory: warn once, carry on without
                    self.failed = (t, t1)
                    if not getattr(self, "_warned", False):
                        print(f"noise table [{t:g}, {t1:g}] not built ({e}); hashing arithmetically", flush=True)
                        self._warned = True
            else:
                self.failed = (t, t + self.retry_after)      # nothing fits: look again after retry_after seconds of sim time
        if self.table is not None and self.table.covers(t):
            self.table_frames += 1
            return self.table.id
        self.arith_frames += 1
        return 0

    def summary(self):
        return {"builds": self.builds, "table_frames": self.table_frames, "arith_frames": self.arith_frames,
                "coarsest_coverage": self.coarsest, "peak_bytes": self.peak_bytes, "budget_bytes": self.budget}

    def close(self):
        if self.table is not None:
            self.table.destroy()
            self.table = None


def set_launch_defaults(params):
    """Parameters of the reference-signature C++ entry point launch_raymarch() (None: config.h defaults)."""
    _lib.check(_lib.load().rrt_set_launch_defaults(C.byref(params) if params is not None else None),
               "rrt_set_launch_defaults")


def get_launch_defaults():
    out = RenderParams()
    _lib.check(_lib.load().rrt_get_launch_defaults_sized(C.byref(out), C.sizeof(rrt_params)), "rrt_get_launch_defaults")
    return out


def _sky_handle(sky):
    return sky.handle if isinstance(sky, SkyTexture) else int(sky)


def abi_version():
    return _lib.load().rrt_abi_version()


def device_count():
    n = C.c_int(0)
    _lib.load().rrt_device_count(C.byref(n))
    return n.value


def launch_raymarch(d_out, w, h, time, cam, skyboxTex, effects, params=None, stream=None):
    """Drop-in for the reference's launch_raymarch (include/raymarcher.h:19): asynchronous,
    writes w*h RGBA8 pixels (bottom-up rows, alpha 255) to the device buffer `d_out`."""
    _lib.check(_lib.load().rrt_launch_raymarch(_ptr(d_out), w, h, float(time), C.byref(cam),
                                               _sky_handle(skyboxTex), C.byref(effects),
                                               C.byref(params) if params is not None else None,
                                               _stream(stream)), "rrt_launch_raymarch")


def launch_raymarch_rows(d_out_rows, w, h, y0, y1, time, cam, skyboxTex, effects, params=None, stream=None):
    _lib.check(_lib.load().rrt_launch_raymarch_rows(_ptr(d_out_rows), w, h, y0, y1, float(time), C.byref(cam),
                                                    _sky_handle(skyboxTex), C.byref(effects),
                                                    C.byref(params) if params is not None else None,
                                                    _stream(stream)), "rrt_launch_raymarch_rows")


def launch_raymarch_tiles(d_out_tiles, w, h, tile_rows, shard, n_shards, time, cam, skyboxTex, effects,
                          params=None, stream=None):
    _lib.check(_lib.load().rrt_launch_raymarch_tiles(_ptr(d_out_tiles), w, h, tile_rows, shard, n_shards,
                                                     float(time), C.byref(cam), _sky_handle(skyboxTex),
                                                     C.byref(effects),
                                                     C.byref(params) if params is not None else None,
                                                     _stream(stream)), "rrt_launch_raymarch_tiles")


def tile_shard_rows(h, tile_rows, shard, n_shards):
    rows = C.c_int(0)
    _lib.check(_lib.load().rrt_tile_shard_rows(h, tile_rows, shard, n_shards, C.byref(rows)),
               "rrt_tile_shard_rows")
    return rows.value


def assemble_tiles(d_frame, d_tiles, w, h, tile_rows, shard, n_shards, stream=None):
    _lib.check(_lib.load().rrt_assemble_tiles(_ptr(d_frame), _ptr(d_tiles), w, h, tile_rows, shard, n_shards,
                                              _stream(stream)), "rrt_assemble_tiles")


def assemble_all_tiles(d_frame, d_tiles_all, shard_stride_bytes, w, h, tile_rows, n_shards, stream=None):
    _lib.check(_lib.load().rrt_assemble_all_tiles(_ptr(d_frame), _ptr(d_tiles_all), shard_stride_bytes, w, h,
                                                  tile_rows, n_shards, _stream(stream)), "rrt_assemble_all_tiles")


def launch_raymarch_tilemap(d_out_tiles, w, h, tile_map, shard, time, cam, skyboxTex, effects, params=None, stream=None):
    _lib.check(_lib.load().rrt_launch_raymarch_tilemap(_ptr(d_out_tiles), w, h, tile_map.id if isinstance(tile_map, TileMap) else int(tile_map),
                                                       shard, float(time), C.byref(cam), _sky_handle(skyboxTex), C.byref(effects),
                                                       C.byref(params) if params is not None else None, _stream(stream)),
               "rrt_launch_raymarch_tilemap")


def assemble_all_tilemap(d_frame, d_tiles_all, shard_stride_bytes, w, h, tile_map, stream=None):
    _lib.check(_lib.load().rrt_assemble_all_tilemap(_ptr(d_frame), _ptr(d_tiles_all), shard_stride_bytes, w, h,
                                                    tile_map.id if isinstance(tile_map, TileMap) else int(tile_map),
                                                    _stream(stream)), "rrt_assemble_all_tilemap")


def launch_raymarch_debug(d_out, w, h, time, cam, skyboxTex, effects, params=None, stream=None, **outs):
    """Full-frame launch that also fills per-ray outputs: ldr, hdr, steps, hit, pos, vel, rad (+ lut_oob, one
    uint32 counter of out-of-box noise-table reads)."""
    dbg = rrt_debug_outputs()
    for k, v in outs.items():
        setattr(dbg, "d_" + k, _ptr(v).value if v is not None else None)
    _lib.check(_lib.load().rrt_launch_raymarch_ex(_ptr(d_out), w, h, float(time), C.byref(cam),
                                                  _sky_handle(skyboxTex), C.byref(effects),
                                                  C.byref(params) if params is not None else None,
                                                  C.byref(dbg), _stream(stream)), "rrt_launch_raymarch_ex")
