/*
 * rrt.h -- C ABI of librrt_hip.so, the MI355X (gfx950) implementation of the
 * per-pixel geodesic ray-march hot path of levi2234/RelativisticRayTracer.
 *
 * This is the drop-in boundary.  The one entry point of the reference's path is
 *
 *     void launch_raymarch(uchar4* d_out, int w, int h, float time,
 *                          CameraState cam, cudaTextureObject_t skyboxTex,
 *                          CameraEffects effects);
 *                                   -- reference include/raymarcher.h:19,
 *                                      defined src/raymarcher.cu:176-180,
 *                                      called from src/main.cpp:467
 *
 * `include/raymarcher.h` of this repo re-declares it source-compatibly (C++);
 * librrt_hip.so exports it as a real C++ symbol, under the reference's own
 * mangled name too, over rrt_launch_raymarch() below; everything here is
 * plain C: pointers, ints, floats and PODs -- no HIP or torch types.
 *
 * Conventions
 *   - every function returns an rrt_status (0 = RRT_OK); nothing throws;
 *   - `d_*` pointers are DEVICE pointers owned by the caller;
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream);
 *     launches are asynchronous exactly like the reference's (raymarcher.cu:179);
 *   - the library keeps no per-call state and allocates nothing in a launch,
 *     so launches may be captured into a hipGraph (what a launch needs beyond its arguments -- a pool, noise tables, a
 *     tile-order object -- are caller-owned objects created beforehand; the one exception is documented at
 *     rrt_tile_order: its FIRST launch of a larger geometry sizes its buffers, and captured launches ignore it).
 */
#ifndef RRT_H
#define RRT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RRT_ABI_VERSION 5      /* 5: RRT_ARITH_FMAD, rrt_params.nudge_ulps / .nudge_seed, rrt_params_init (an rrt_params of ABI 4's
                                     48 bytes is still accepted: the new fields read as 0);
                                  4: rrt_params.struct_size (leading), .pool_rounds, .pass_chains, rrt_tile_map_*, rrt_probe_tile_costs,
                                     rrt_clock_probe; 3: rrt_params.tile_order, rrt_tile_order_* */

typedef enum {
    RRT_OK = 0,
    RRT_ERR_INVALID_ARGUMENT = 1,
    RRT_ERR_NO_DEVICE = 2,
    RRT_ERR_HIP = 3,          /* a HIP runtime call failed; see rrt_last_hip_error() */
    RRT_ERR_BAD_HANDLE = 4,
    RRT_ERR_OUT_OF_MEMORY = 5,
    RRT_ERR_ABI_MISMATCH = 6  /* an rrt_params whose struct_size is not this library's: built against another include/rrt.h */
} rrt_status;

/* Camera basis handed to the kernel.  Layout == reference `struct CameraState`
 * (include/raymarcher.h:11-16): four packed float3, 48 bytes. */
typedef struct rrt_camera {
    float pos[3];
    float forward[3];
    float right[3];
    float up[3];
} rrt_camera;

/* Per-pixel camera effects.  Layout == reference `struct CameraEffects`
 * (include/camera_effects/camera_settings.h:4-17): 36 bytes, bools padded to
 * 4-byte slots (offsets 0,4,8,12,16,20,24,28,32). */
typedef struct rrt_effects {
    uint8_t use_bloom;       uint8_t _pad0[3];
    float bloom_threshold;
    float bloom_intensity;
    uint8_t use_vignette;    uint8_t _pad1[3];
    float vignette_intensity;
    uint8_t use_chromatic_aberration; uint8_t _pad2[3];
    float ca_amount;
    uint8_t use_lens_distortion;      uint8_t _pad3[3];
    float distortion_amount;
} rrt_effects;

/* Scene / quality parameters.  Defaults == the reference's compile-time
 * constants (include/config.h:18-48); `spin` replaces the SPIN_A macro
 * (config.h:21) so that Kerr a=0.9 / 0.99 are run-time settings. */
typedef struct rrt_params {
    uint32_t struct_size;    /* sizeof(rrt_params) of the header the caller was compiled against; rrt_params_default()
                                fills it in.  Every entry point that takes an rrt_params accepts this header's size and
                                ABI 4's 48 bytes (a prefix: the fields behind it read as their defaults) and refuses any
                                other value with RRT_ERR_ABI_MISMATCH (objects built against the ABI <= 3 header, whose
                                struct began with `spin`, must be recompiled)                                */
    float spin;              /* SPIN_A            config.h:21  default 0.0  */
    int32_t max_steps;       /* MAX_STEPS         config.h:48  default 2000 */
    int32_t volumetrics;     /* 1 = full disk + dust (reference behaviour);
                                0 = "skybox only": both densities read 0,
                                zone-dependent step sizes unchanged        */
    int32_t sky_frac_bits;   /* bilinear weight bits of the sky sampler:
                                8 = CUDA-texture-like (default), 0 = exact  */
    int32_t arith_mode;      /* RRT_ARITH_STRICT (default): every operation rounded as the
                                reference source writes it -- the bit-parity path.
                                RRT_ARITH_FMAD: the geodesic integrator with multiply-adds FUSED and
                                division / square root still correctly rounded -- the arithmetic class
                                of the reference's own build (nvcc defaults: -fmad=true, IEEE div/sqrt);
                                media, sky and post-FX code unchanged.  Within the 1e-4 tolerance of the
                                strict frame on every pixel whose strict value is itself stable under a
                                few-ulp nudge of its primary ray (tests/test_gpu_tolerance.py).
                                RRT_ARITH_FAST: fused multiply-adds and 1-ulp reciprocal
                                square roots, no correctly rounded divide; informational.       */
    int32_t workspace;       /* 0 (default): single kernel, media sampled in line by the marching
                                lane.  An rrt_workspace id: three-pass path -- the march only
                                records in-medium sample points, which the whole chip then evaluates
                                and a last pass composites in march order.  Same bytes out; removes
                                the per-wavefront long pole of disk-grazing rays, which is what
                                strong scaling over GPUs needs.           */
    int32_t path_policy;     /* with a workspace: RRT_PATH_AUTO (default) takes the three-pass path for
                                launches of <= 1.5 M rays (where it is faster) and the single kernel
                                otherwise; RRT_PATH_SINGLE / RRT_PATH_THREE_PASS force one             */
    int32_t noise_table;     /* 0 (default): every noise3D hashes its eight lattice corners arithmetically.
                                An rrt_noise_table id: the low octaves of the disk / dust noise read the corner
                                hashes from a precomputed lattice table whenever the 64 rays of a wavefront
                                share a few lattice cells (4K / 8K frames: most of them) -- same bits, about
                                half the media cost.  A launch whose `time` lies outside
                                the table's window [t0, t1] (rrt_noise_table_window) hashes arithmetically.  */
    int32_t tile_order;      /* 0 (default): wave tiles are dispatched in the static order (row blocks from the middle
                                of the frame outwards).  An rrt_tile_order id: the launch records what every wave tile
                                cost and the next launch of the same geometry through that object dispatches
                                longest-first -- same pixels; removes the drain of views whose long rays are not in the
                                middle (a 4K frame from inside the disk: 53 -> 46 ms), nothing to gain on the
                                reference's default view.  With no history for the launch's geometry the order comes
                                from a coarse march-only probe of the same view (one ray per 16x16 pixels, run on the
                                launch's stream right before it).  Both paths; a launch that is being captured into a
                                hipGraph renders in the static order and leaves the object alone.              */
    int32_t pool_rounds;     /* three-pass path: the workspace pool is reused in ROUNDS -- march until the pool is full,
                                evaluate and composite what was pooled, resume the suspended rays -- so that any pool
                                serves any view.  0 (default): automatic -- as many rounds as the workspace's previous
                                launch needed, plus one, at least 2; n > 0: exactly up to n rounds.  Rays still
                                suspended after the last round finish with the media sampled in line (same bytes).   */
    int32_t pass_chains;     /* three-pass path: 0 (default) = automatic -- a launch of >= 2 048 wavefronts is cut in two
                                along its dispatch order and the halves run their march -> evaluate -> composite chains
                                side by side (the workspace's own second stream), so that one half's evaluation fills the
                                other half's march tail; 1 = one chain; 2 = two whenever possible.  Same bytes.  A caller that
                                keeps SEVERAL launches in flight on streams of its own (frames of an animation) should ask for
                                1: the other frames already fill a launch's tails, and the extra streams only get in each
                                other's way (a rank's share of a 4K frame, three in flight: 6.8 instead of 7.4 ms).  With four
                                or more in flight RRT_PATH_SINGLE is usually faster still -- unless the view has a wavefront
                                that takes longer than the frames in flight together.                   */
    int32_t nudge_ulps;      /* conditioning probe (ABI 5).  0 (default): primary rays exactly as raymarcher.cu:27-34 forms
                                them.  K > 0: every component of every pixel's normalised primary direction is moved by a
                                pseudo-random whole number of ulps in [-K, K] (a hash of pixel and nudge_seed; the oracle has
                                the same function).  Rendering a frame under a few such nudges shows which pixels the
                                reference's own arithmetic does not determine to the tolerance -- near-critical rays, zone
                                and density gates about to flip: how the within-tolerance arithmetic modes are accounted for. */
    uint32_t nudge_seed;
} rrt_params;

#define RRT_PATH_AUTO 0
#define RRT_PATH_SINGLE 1
#define RRT_PATH_THREE_PASS 2

#define RRT_ARITH_STRICT 0
#define RRT_ARITH_FAST 1
#define RRT_ARITH_FMAD 2

/* Opaque sky-texture handle; stands in for cudaTextureObject_t
 * (`unsigned long long`, reference src/main.cpp:231-263). */
typedef unsigned long long rrt_sky_t;

/* Optional per-ray outputs of rrt_launch_raymarch_ex (device pointers, any may
 * be NULL).  Indexing: `ldr`/`hdr` like the RGBA8 frame (bottom-up rows,
 * raymarcher.cu:168); the others top-down, y*width + x. */
typedef struct rrt_debug_outputs {
    float* d_ldr;            /* 4 floats/pixel: tone-mapped r,g,b before the u8 cast, 1 */
    float* d_hdr;            /* 4 floats/pixel: final_hdr after post-FX, 1              */
    int32_t* d_steps;        /* RK4 steps taken                                         */
    int32_t* d_hit;          /* 1 = ray ended on the horizon                            */
    float* d_pos;            /* 3 floats/pixel: final position                          */
    float* d_vel;            /* 3 floats/pixel: final velocity                          */
    float* d_rad;            /* 4 floats/pixel: intensity r,g,b and transmittance       */
    unsigned* d_lut_oob;     /* 1 counter: noise-table reads whose index had to be clamped
                                (must stay 0: the table box covers every reachable cell)  */
} rrt_debug_outputs;

/* ---- library ---- */
int rrt_abi_version(void);
const char* rrt_status_string(int status);
const char* rrt_last_hip_error(void);          /* thread-local text of the last HIP failure */
int rrt_device_count(int* count);
int rrt_path_auto_max_rays(void);             /* RRT_PATH_AUTO takes the three-pass path for launches of at most this many rays (1 500 000) */
/* config.h defaults into the first `size` bytes of an rrt_params and struct_size = size.  Call it through the macro below, so
 * that `size` is the sizeof of the header the CALLER was compiled against: the library then knows which fields the caller
 * has.  Accepted sizes: this header's, and ABI 4's 48 bytes (fields the caller does not have read as their defaults); any
 * other size is RRT_ERR_ABI_MISMATCH here and at every entry point that takes an rrt_params. */
int rrt_params_init(void* prm, uint32_t size);
#define rrt_params_default(p) rrt_params_init((p), (uint32_t)sizeof(rrt_params))
/* (the library also keeps the exports older headers mapped the name to: rrt_params_default_v4 fills ABI 4's 48 bytes -- such a
 * binary keeps working --, rrt_params_default the 36-byte ABI <= 3 layout, whose binaries are refused at their first launch) */
int rrt_effects_default(rrt_effects* fx);      /* camera_settings.h:5-16 defaults          */

/* ---- sky texture: replaces loadSkybox()'s cudaMallocArray + texture object,
 *      reference src/main.cpp:246-263.  RGBA8, row 0 = top of the panorama. ---- */
int rrt_sky_create(const uint8_t* rgba8_host, int width, int height, rrt_sky_t* out);
int rrt_sky_create_from_device(const void* d_rgba8, int width, int height, rrt_sky_t* out); /* borrows */
int rrt_sky_destroy(rrt_sky_t sky);

/* ---- workspace of the three-pass path (rrt_params.workspace): a caller-owned HBM pool, so that a
 *      launch still allocates nothing.  One workspace serves one stream at a time (it owns a second stream of its
 *      own for the second chain, rrt_params.pass_chains; forked from and joined to the caller's by events).  ~1.5 KB per
 *      wave-step that touches the media (handed out in blocks of 8 rows, runs of up to 8 blocks); the 4K bench frame
 *      pools 2.6 GB, an eighth of it 0.3 GB, an eighth of a 4K view from inside the disk 1.9 GB.  The pool is reused in
 *      ROUNDS (rrt_params.pool_rounds): the wavefronts it runs out under are suspended and resumed once the pooled
 *      samples have been evaluated and composited, so any pool serves any view; only rays still suspended after the last
 *      enqueued round are finished by the in-line code (same result either way; one round is the fast case). ---- */
int rrt_workspace_create(size_t bytes, int* out_id);
int rrt_workspace_destroy(int id);
/* after a launch has completed: rows used and wavefronts that fell back (synchronous read) */
int rrt_workspace_stats(int id, unsigned* rows_used, unsigned* overflow_waves);
/* rounds of the last launch (rrt_params.pool_rounds): enqueued, with work for the march, rows of the fullest round, and
 * an upper bound of the pool's rows.  rrt_workspace_stats: rows_used = all rounds together, overflow_waves = wavefronts
 * still suspended after the last round (finished in line). */
int rrt_workspace_rounds(int id, unsigned* rounds_enqueued, unsigned* rounds_with_work, unsigned* peak_rows, unsigned* pool_rows);
/* inspection: copy `bytes` of the pool starting at `offset` to host memory (synchronous) */
int rrt_workspace_read(int id, size_t offset, size_t bytes, void* host_dst);

/* ---- cost-ordered dispatch (rrt_params.tile_order; no counterpart in the reference, whose launch is one fixed grid,
 *      src/raymarcher.cu:176-180).  The object holds, per 8x8-pixel wave tile, the shader clocks the last launch
 *      through it took, and the permutation (sorted on the device right after that launch, on its stream) the next
 *      launch with the same width / height / row map reads.  A launch with another geometry renders in the static order
 *      and starts over.  Launches through one object are serialised on the device, also across streams: give every frame
 *      that should overlap another its own object (the headless drivers: one per slot).  Frames of an animation change
 *      little from one to the next, which is what makes the previous frame's costs a good order for this one.  With NO
 *      history (first launch, new geometry) the order comes from a coarse probe of the view itself: one march-only ray per
 *      16x16 pixels on the launch's stream (~1/250 of the frame's work), costed by a fitted model -- a still image from
 *      inside the disk gets most of the gain too; rrt_tile_order_set_seeding(id, 0) switches that off.  The first
 *      launch of a (larger) geometry allocates the object's buffers -- a synchronising call.  A launch that is being
 *      CAPTURED into a hipGraph ignores the object (static order, nothing recorded): a replayed graph can then never read
 *      a permutation that a later live launch is rewriting.  Works on both paths (single kernel and three-pass). ---- */
int rrt_tile_order_create(int* out_id);
/* Must be called with the device that owns the object CURRENT (the one that was current at create): under another device it
 * returns RRT_ERR_BAD_HANDLE and frees NOTHING -- the handle stays valid and the call can be repeated from the right device
 * (rrt_tile_map_destroy: the same rule).  A caller that ignores the status there leaks the object's device buffers. */
int rrt_tile_order_destroy(int id);
int rrt_tile_order_set_seeding(int id, int on);
int rrt_tile_order_seeded(int id, unsigned long long* seeded_launches);      /* launches ordered by the probe */
/* counters; with perm_host / cost_host (either may be NULL; `capacity` elements each) also, after waiting for the
 * object's last launch, the order the next matching launch will use and the costs the last one recorded */
int rrt_tile_order_info(int id, unsigned long long* launches, unsigned long long* ordered_launches, unsigned* n_tiles,
                        unsigned* perm_host, unsigned* cost_host, unsigned capacity);

/* ---- lattice-hash tables for the volumetric noise (rrt_params.noise_table): hash31 (math_utils.h:91-96) of
 *      every lattice point the low-octave noise3D calls of getAccretionDensity / getDustCloudDensity
 *      (densities.h:54, :95-128) can reach for t0 <= time <= t1, computed once on the device by the same
 *      arithmetic (a few milliseconds).  Caller-owned like the sky and tied to the device it was created on; any
 *      number of launches / streams of that device may read one table concurrently.  A launch whose `time` lies
 *      outside the window renders with the arithmetic kernels: same bytes, slower.
 *      Size: 0.49 GB for [0, 32 s] at full coverage.  A single dense box does not stay bounded as the window slides along
 *      the reference's unbounded simTime (main.cpp:515): the dust coordinates shear with time * (10/rc)^1.5
 *      (densities.h:88-93), so its z extent grows like 0.75 t0 + (t1 - t0); far along the clock the fine dust families
 *      therefore move to the BANDED layout (below), which keeps a 10 s window at t = 500 s at full coverage inside 2 GiB.
 *      Two more knobs: the window and the coverage -- which call families are table-served; the finest ones dominate the volume:
 *          RRT_TABLE_FULL      all table-served families
 *          RRT_TABLE_COARSE    without the 4.41 and 4.0 cells-per-unit dust families (about 1/9 of the dust box)
 *          RRT_TABLE_COARSEST  also without the 2.1 dust family and the finest accretion octave
 *      and rrt_noise_table_fit_window(), the policy the headless drivers use: longest window from t_from, richest
 *      coverage, within a byte budget (*bytes_out == 0: nothing fits, render without a table).
 *      rrt_noise_table_plan*() is host arithmetic only and returns the same RRT_ERR_INVALID_ARGUMENT as create for a
 *      box that cannot be addressed (>= 2^28 lattice points). ---- */
enum { RRT_TABLE_FULL = 0, RRT_TABLE_COARSE = 1, RRT_TABLE_COARSEST = 2,
       /* LAYOUT of the dust families (ABI 5), ORed into `coverage` to force one; neither = automatic.  DENSE: one box for all
        * of them -- what every window near the origin of the clock gets.  BANDED: the three fine families (ridge octaves 1 and
        * 2, the detail octave -- the ones that make the dense box unaddressable minutes into the clock) in one small box per
        * band of the angular rate omega = (10/rc)^1.5, picked per sample from its own radius: the shear of densities.h:88-93
        * only costs every band ITS OWN z range.  [495, 505 s] at full coverage: not addressable dense, ~1.5 GB banded.
        * Same bytes.  Automatic: dense unless it is unaddressable, or over 768 MB and larger than the banded plan.
        * rrt_noise_table_window() reports the layout a table got in this bit of its `coverage`. */
       RRT_TABLE_BANDED = 16, RRT_TABLE_DENSE = 32 };
int rrt_noise_table_create(float t_max, int* out_id);                                  /* = window [0, t_max], full coverage */
int rrt_noise_table_create_window(float t0, float t1, int coverage, int* out_id);
int rrt_noise_table_destroy(int id);
int rrt_noise_table_info(int id, float* t_max, size_t* bytes, int* boxes12);   /* boxes: x0,y0,z0,nx,ny,nz of the accretion and dust boxes */
int rrt_noise_table_window(int id, float* t0, float* t1, int* coverage, int* device);
int rrt_noise_table_plan(float t_max, size_t* bytes, int* boxes12);
int rrt_noise_table_plan_window(float t0, float t1, int coverage, size_t* bytes, int* boxes12);
/* the layout such a table gets (host arithmetic): *banded = 0 / 1; for a banded one the number of omega bands, the rule
 * band = clamp((int)((omega - w_min) * w_scale)), the box (x0, y0, z0, nx, ny, nz) of every (family, band) -- family 0 / 1 / 2
 * = ridge octave 1 / ridge octave 2 / detail octave, band_boxes[(family * cap_bands + band) * 6 ...], cap_bands >= n_bands
 * (64 always is) -- and of the four accretion octaves (acc_octave_boxes[24]).  Either array may be NULL. */
int rrt_noise_table_plan_layout(float t0, float t1, int coverage, int* banded, int* n_bands, float* w_min, float* w_scale,
                                int32_t* band_boxes, int cap_bands, int32_t* acc_octave_boxes);
int rrt_noise_table_fit_window(float t_from, float t_until, size_t budget_bytes, float* t1_out, int* coverage_out, size_t* bytes_out);

/* Handles and devices: a sky, workspace or noise table belongs to the HIP device that was current when it was
 * created (a borrowed sky: the device that owns the pointer), and a launch or copy that names it under another
 * current device returns RRT_ERR_BAD_HANDLE.  (Tests drive those checks without a second GPU through
 * rrt_debug_fake_device, include/rrt_test.h.) */

/* The shader clock the chip HOLDS, measured on the device: one wavefront sleeps for `duration_us` (<= 2 000 000) on
 * `stream` and reads the shader-clock counter (s_memtime) and the constant 100 MHz counter (s_memrealtime) at both ends;
 * d_counters2[0] / d_counters2[1] * 0.1 = GHz.  Launched on a second stream beside the frames of a measurement it says what
 * clock the roofline's peak should be priced at (bench.py: roofline.clock_ghz). */
int rrt_clock_probe(unsigned long long* d_counters2, unsigned duration_us, void* stream);

/* ---- parameters of the reference-signature entry point launch_raymarch() (include/raymarcher.h), which has
 *      no argument for them: spin, max_steps, volumetrics, a workspace, a noise table ...  NULL restores the
 *      config.h defaults.  Nothing is allocated on the caller's behalf: objects named here are the caller's. ---- */
int rrt_set_launch_defaults(const rrt_params* prm);
/* ---- ... or, in ONE call, let the library own them (round 6).  For a host that makes only the drop-in edits and keeps calling
 *      launch_raymarch(): rrt_launch_auto_resources(1, base, table_budget_bytes, pool_bytes) creates ON THE CURRENT DEVICE a pool
 *      (pool_bytes; 0: none), a tile order and -- lazily, at the first launch -- lattice-hash tables over a window of `time` that fits
 *      table_budget_bytes (0: none), and every launch_raymarch() under that device then runs with `base` (NULL: config.h; its
 *      spin, max_steps, arith_mode ... are kept, its object fields replaced) + those objects: same bytes, the full speed of the
 *      path.  THE DOCUMENTED EXCEPTION to "a launch allocates nothing": the launch_raymarch() call whose `time` has left the table's
 *      window (the reference's simTime grows without bound, main.cpp:515) waits for the device, destroys the table and builds the
 *      next window (milliseconds) before it launches.  Calls under another current device use the plain defaults above.
 *      rrt_launch_auto_resources(0, NULL, 0, 0), from the same device, destroys everything.  Explicit rrt_launch_raymarch*()
 *      calls are never affected. ---- */
int rrt_launch_auto_resources(int on, const rrt_params* base, size_t table_budget_bytes, size_t pool_bytes);
int rrt_launch_auto_resources_info(int* on, int* table_builds, float* table_t0, float* table_t1, size_t* table_bytes);   /* any pointer may be NULL */
int rrt_get_launch_defaults_sized(void* out, uint32_t size);        /* through the macro: size = the caller's sizeof(rrt_params) */
#define rrt_get_launch_defaults(out) rrt_get_launch_defaults_sized((out), (uint32_t)sizeof(rrt_params))
/* launch_raymarch() with plain C types (what both C++ symbols of that name forward to): cam12 = pos, forward,
 * right, up; effects36 = the 36 bytes of struct CameraEffects (== rrt_effects); null stream, asynchronous. */
int rrt_launch_raymarch_compat(void* d_out_rgba8, int width, int height, float time, const float* cam12,
                               rrt_sky_t sky, const void* effects36);

/* ---- the hot path.  Replaces launch_raymarch, reference include/raymarcher.h:19 /
 *      src/raymarcher.cu:176-180.  Writes width*height RGBA8 pixels, alpha 255,
 *      bottom-up rows, to d_out_rgba8.  prm == NULL -> config.h defaults.
 *      Size limits (RRT_ERR_INVALID_ARGUMENT beyond them): width*height < 2^31 pixels, height <= 524 280
 *      (65 535 row-blocks of 8 rows, HIP's gridDim.y; the reference's 16x16 launch stops at 1 048 560). ---- */
int rrt_launch_raymarch(void* d_out_rgba8, int width, int height, float time,
                        const rrt_camera* cam, rrt_sky_t sky, const rrt_effects* fx,
                        const rrt_params* prm, void* stream);

/* Row-range variant for sharding the image plane (no counterpart in the
 * reference, which is single-GPU): renders image rows y0 <= y < y1 (y as in
 * raymarcher.cu:17, i.e. before the bottom-up flip) of the full width x height
 * frame.  d_out_rows receives (y1-y0)*width pixels; local row k holds image
 * row y1-1-k, so concatenating the shards in DESCENDING y order reproduces the
 * full bottom-up frame. */
int rrt_launch_raymarch_rows(void* d_out_rows, int width, int height, int y0, int y1, float time,
                             const rrt_camera* cam, rrt_sky_t sky, const rrt_effects* fx,
                             const rrt_params* prm, void* stream);

/* Interleaved row-tile variant: tile t = image rows [t*tile_rows, (t+1)*tile_rows)
 * belongs to shard (t mod n_shards).  Renders all tiles of `shard` into
 * d_out_tiles, tile-major in increasing t, each tile stored bottom-up like
 * rrt_launch_raymarch_rows.  rrt_tile_shard_rows() gives the buffer's rows (x width x 4 bytes). */
int rrt_launch_raymarch_tiles(void* d_out_tiles, int width, int height, int tile_rows,
                              int shard, int n_shards, float time,
                              const rrt_camera* cam, rrt_sky_t sky, const rrt_effects* fx,
                              const rrt_params* prm, void* stream);
int rrt_tile_shard_rows(int height, int tile_rows, int shard, int n_shards, int* rows);
/* Scatter one shard's tile buffer into a full bottom-up frame (device to device). */
int rrt_assemble_tiles(void* d_frame_rgba8, const void* d_tiles, int width, int height,
                       int tile_rows, int shard, int n_shards, void* stream);

/* Same for ALL shards in one launch: shard s's buffer starts at d_tiles_all + s*shard_stride_bytes
 * (the layout a gather into one allocation produces). */
int rrt_assemble_all_tiles(void* d_frame_rgba8, const void* d_tiles_all, size_t shard_stride_bytes,
                           int width, int height, int tile_rows, int n_shards, void* stream);

/* ---- cost-weighted tile -> shard assignment ("cost-model-weighted assignment"; the reference is
 *      single-GPU).  The rows through the hole and the disk cost several times the sky rows; t mod n_shards evens that
 *      out to ~9 % at 8 shards of a 4K frame, a map dealt by COST to ~1 %.  rrt_probe_tile_costs() estimates every row
 *      tile's cost from a coarse march-only probe of the view (deterministic: every rank computes the same numbers from
 *      the same camera, so no exchange is needed), rrt_tile_map_balance() deals the tiles longest-first to the least
 *      loaded shard (host arithmetic), rrt_tile_map_create() makes the assignment a device-resident object, and the
 *      two entry points below are rrt_launch_raymarch_tiles / rrt_assemble_all_tiles for such a map.  Buffer layout:
 *      a shard's tiles in increasing t, tile-major, each tile bottom-up. ---- */
int rrt_tile_map_create(int height, int tile_rows, int n_shards, const int32_t* shard_of_tile, int* out_id);
int rrt_tile_map_destroy(int id);
int rrt_tile_map_shard_rows(int id, int shard, int* rows, int* max_rows);      /* rows of `shard`'s buffer; of the largest */
int rrt_tile_map_balance(int n_tiles, const float* tile_cost, int n_shards, int max_tiles_per_shard, int32_t* shard_of_tile_out);
int rrt_probe_tile_costs(int width, int height, int tile_rows, float time, const rrt_camera* cam, const rrt_effects* fx,
                         const rrt_params* prm, float* tile_cost_host, int n_tiles, void* stream);
int rrt_launch_raymarch_tilemap(void* d_out_tiles, int width, int height, int tile_map, int shard, float time,
                                const rrt_camera* cam, rrt_sky_t sky, const rrt_effects* fx,
                                const rrt_params* prm, void* stream);
int rrt_assemble_all_tilemap(void* d_frame_rgba8, const void* d_tiles_all, size_t shard_stride_bytes,
                             int width, int height, int tile_map, void* stream);

/* Full-frame launch that also fills per-ray debug outputs (parity tests). */
int rrt_launch_raymarch_ex(void* d_out_rgba8, int width, int height, float time,
                           const rrt_camera* cam, rrt_sky_t sky, const rrt_effects* fx,
                           const rrt_params* prm, const rrt_debug_outputs* dbg, void* stream);

/* ---- which path a rank's share takes while several frames of a sequence are in flight (host only; no GPU call) ----
 * New in this repo (the reference renders one frame at a time on one GPU: src/main.cpp:505-529).  A launch of <= 1.5 M rays
 * with a pool can take the three-pass path (RRT_PATH_AUTO) or the single kernel (RRT_PATH_SINGLE); under frames in flight the
 * single kernel is 3-10 % faster unless the share holds a wavefront that outlasts them (then 20 % slower).  The object cuts the
 * sequence into windows, tries the other path for a few frames at a window's start, compares SUSTAINED frame times (the mean
 * interval between the ends of consecutive frames' renders on the rank) and keeps the faster; frames_in_flight single-kernel
 * frames in a row that take > 1.5 x the three-pass mean end the experiment at once (csrc/rrt_path_chooser.cpp).  The bytes do not
 * depend on it.
 *   id = create(frames_in_flight, window_frames (0: 48));  per frame k = 1, 2, ...: policy(id, k, &p) -> rrt_params.path_policy;
 *   later, when frame k's times are known: report(id, k, ms). */
typedef struct rrt_path_chooser_stats {
    int32_t incumbent;                 /* RRT_PATH_AUTO / RRT_PATH_SINGLE: what the current window renders with outside its trial */
    int32_t windows, trials, trials_aborted, switches, outliers;
    int32_t frames[2];                 /* frames handed to [0] the automatic (three-pass) path, [1] the single kernel */
    float last_three_pass_mean_ms;
} rrt_path_chooser_stats;
int rrt_path_chooser_create(int frames_in_flight, int window_frames, int* out_id);
int rrt_path_chooser_destroy(int id);
int rrt_path_chooser_policy(int id, int frame, int* policy_out);
int rrt_path_chooser_report(int id, int frame, float sustained_ms);
int rrt_path_chooser_get_stats(int id, rrt_path_chooser_stats* out);


/* ---- host-side camera helpers (host C++ in the reference too) ---- */
/* CameraController::getCUDAStateFrom, src/main.cpp:141-167 (degrees; note its 3.14159f) */
int rrt_camera_from_angles(const float pos[3], float yaw_deg, float pitch_deg, rrt_camera* out);
/* catmull_rom / lerp_angle, src/camera_paths.cpp:6-29 */
int rrt_catmull_rom(const float p0[3], const float p1[3], const float p2[3], const float p3[3], float t, float out[3]);
int rrt_lerp_angle(float a, float b, float t, float* out);
/* the three built-in keyframe paths, src/camera_paths.cpp:31-73 (0 "Gargantua Fly-By",
 * 1 "Event Horizon Focus", 2 "Horizon Skimmer"); keyframes are 6 floats: time, x, y, z, yaw, pitch */
int rrt_path_count(void);
int rrt_path_info(int path, const char** name, int* n_keys, float* t_end);
int rrt_path_keyframes(int path, float* out6, int cap_keys);
/* PathController::getInterpolatedState, src/main.cpp:176-203 */
int rrt_path_camera_at(int path, float path_time, rrt_camera* out);
/* recording clock of the main loop, src/main.cpp:511-516: times seen by 1-based frame k */
int rrt_recording_clock(int frame_k, int fps, float* sim_time, float* path_time);

#ifdef __cplusplus
}
#endif
#endif /* RRT_H */
