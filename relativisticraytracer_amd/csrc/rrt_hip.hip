/*
 * rrt_hip.hip -- gfx950 kernels and the C ABI (include/rrt.h) of librrt_hip.so.
 *
 * Replaces the reference's only CUDA translation unit, src/raymarcher.cu
 * (raymarch_kernel :15-174 and launch_raymarch :176-180).  Built with
 *   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize -mllvm -enable-post-misched=false -fPIC -shared
 * (relativisticraytracer_amd/build.py).  gfx950 only: no other target, no
 * CUDA dual path.
 *
 * Kernels
 *   raymarch_pixels<SPIN,VOL,DEBUG,FAST>   single-kernel path: one ray per lane, media sampled in line
 *   march_defer / eval_sample_rows / composite_and_shade (/ pool_next_round)
 *                                          three-pass path through a caller-owned workspace: in rounds over the pool, as two
 *                                          chains on two streams (round 4)
 *   probe_costs / probe_to_tiles           coarse march-only probe of a view: first-frame dispatch order, row-tile costs
 *   clock_probe_kernel                     the shader clock the chip holds
 *   assemble_tiles_kernel / assemble_all_kernel   scatter gathered row-tile shards into the frame
 *   k_* / k_selfcheck_*                    array wrappers of the device functions (tests only)
 * Host
 *   C ABI of include/rrt.h, sky and workspace registries, camera basis / path playback
 */
#include <hip/hip_runtime.h>

#include <atomic>
#include <memory>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <unordered_map>
#include <vector>
#include <algorithm>

#include "../../include/rrt.h"
#include "rrt_device.h"
#include "rrt_tile_sort.h"              /* the native radix sort behind rrt_tile_order */

namespace {

using namespace rrt;

thread_local char g_hip_err[256] = "";

int hip_fail(hipError_t e, const char* what) {
    snprintf(g_hip_err, sizeof(g_hip_err), "%s: %s", what, hipGetErrorString(e));
    return RRT_ERR_HIP;
}
#define RRT_HIP(call)                                       \
    do {                                                    \
        hipError_t e_ = (call);                             \
        if (e_ != hipSuccess) return hip_fail(e_, #call);   \
    } while (0)

/* ------------------------------------------------------------------ sky handles
 * A handle is an id into a process-wide registry, never a raw pointer: a stale or made-up handle is
 * reported as RRT_ERR_BAD_HANDLE instead of being dereferenced. */
/* Every handle records the HIP device it was created on, and every entry point that would dereference its
 * allocation from a kernel or a copy compares that with the calling thread's current device: a sky, workspace or
 * noise table used under another hipSetDevice() is RRT_ERR_BAD_HANDLE, not a wild device pointer inside a kernel
 * (the one-process N-GPU driver and the process-global launch defaults make that mistake easy).
 * rrt_debug_fake_device() lets a CPU-only test drive those checks. */
std::atomic<int> g_fake_device{-1};
bool test_hooks_enabled() {       /* decided once, from the environment the process was started with */
    static const bool on = [] { const char* e = getenv("RRT_ENABLE_TEST_HOOKS"); return e && e[0] == '1'; }();
    return on;
}
int current_device() {
    const int fake = g_fake_device.load(std::memory_order_relaxed);
    if (fake >= 0) return fake;
    int d = -1;
    if (hipGetDevice(&d) != hipSuccess) { (void)hipGetLastError(); return -1; }
    return d;
}
/* -1 on either side (no HIP device could be asked) does not fail a launch by itself */
bool on_current_device(int object_device) {
    if (object_device < 0) return true;
    const int d = current_device();
    return d < 0 || d == object_device;
}

struct SkyObject {
    uint8_t* d_texels;
    int w, h;
    bool owned;
    int device;
};
std::mutex g_sky_mu;
std::unordered_map<unsigned long long, SkyObject> g_sky;
unsigned long long g_sky_next = 0x5254000000000001ull;

bool sky_lookup(rrt_sky_t h, SkyObject& out) {
    std::lock_guard<std::mutex> lk(g_sky_mu);
    auto it = g_sky.find(h);
    if (it == g_sky.end()) return false;
    out = it->second;
    return true;
}
rrt_sky_t sky_register(const SkyObject& s) {
    std::lock_guard<std::mutex> lk(g_sky_mu);
    rrt_sky_t h = g_sky_next++;
    g_sky.emplace(h, s);
    return h;
}

/* ------------------------------------------------------------------ deferred-sampling workspace
 * Three-pass path: pass 1 marches the geodesics only and appends the in-medium
 * sample points of a wavefront, one 64-lane "row" per march step that needs one, to a bump-allocated
 * pool in HBM; pass 2 evaluates the densities + emission of every row with the whole chip, wherever
 * the row came from; pass 3 composites each ray's samples in march order and shades the pixel.
 * The pool is handed out in blocks of kBlockRows rows, and blocks in runs of consecutive blocks whose
 * length doubles (1, 2, 4 ... kMaxRun = 8) every time a wave comes back for more: one atomic per run (a
 * single counter saturates near 90 atomics/us) and only ~log2(n) dependent pointer hops when pass 3
 * walks a heavy wave's samples.  Block layout: kBlockRows x five SoA float[64] planes (p.xyz, vel.x, vel.z in;
 * ex, ey, ez, transmittance out in planes 0-3), then a trailer {lane mask of each row; in the first
 * block of a run: start and length of the wave's next run}.  Unused rows keep a zero mask. */
/* Round 4: the pool is reused in ROUNDS.  A round = march until the pool is full (waves the pool ran out under are
 * SUSPENDED: pre-step state and the radiance composited so far are saved per ray) -> evaluate the pooled rows -> composite
 * them; the next round resumes the suspended waves into the emptied pool.  Any pool serves any view; the in-line fall-back
 * only finishes what is still suspended after the last round the host enqueued. */
struct DeferCounters {
    unsigned next_block, overflow_waves;      /* of the current round */
    unsigned last_overflow;                   /* waves suspended when this round started (0: nothing left to do) */
    unsigned rounds_run, rounds_with_work;    /* rounds enqueued so far / rounds whose march had something to do */
    unsigned peak_blocks;                     /* most blocks any round used */
    unsigned suspended_left;                  /* waves still suspended after the last round: finished in line */
    unsigned pad;
    unsigned long long total_blocks;
};
/* state: 0 untouched, 1 marched to its end this round, 2 suspended (the pool ran out under it), 3 shaded.
 * flags bit 0: the rays' radiance so far is saved in `finals` (planes 7-10). */
struct WaveHdr { unsigned first_block, n_runs, state, flags; };
/* Longest run of blocks a wave takes at once.  Runs double (1, 2, 4 ...) up to this, and a wave's last run is on average half
 * empty: with 32 (rounds 1-3) an eighth of the 4K frame from inside the disk allocated 1.46 M rows for 1.2 M samples' worth and
 * needed a second -- sparse, latency-bound -- round in a 2 GiB pool; with 8 it allocates 1.28 M, fits, and takes 7.9 instead of
 * 9.6 ms; the bench view is indifferent (profiles/r04_max_run_ab.txt).  Shorter runs mean more atomics (one per run: ~20 k per
 * such frame, spread over milliseconds) and more link hops in pass 3 (a 2000-row wave: 32 instead of 12). */
#ifndef RRT_MAX_RUN
#define RRT_MAX_RUN 8
#endif
constexpr unsigned kMaxRun = RRT_MAX_RUN;
constexpr unsigned kMaxRunsWalked = 4096;                          /* runs of one wave that pass 3 will walk */
constexpr unsigned kBlockRows = 8;
/* Round 5: a row is FIVE float[64] planes in (p.xyz, vel.x, vel.z), was six.  The only consumer of the sample's velocity is
 * calculateRedshiftFactor's cos_theta = dot(ray_vel, gas_dir) (geodesics.h:18-19), and gas_dir.y is +0 exactly (0 / mag), so
 * vel.y only ever enters as vel.y * 0 = +-0 added to vel.x * gas_dir.x: it can change the sign of a zero cos_theta and nothing
 * else (1 - v * (+-0) = 1) -- unless vel.y is not finite, when the product is NaN; the march poisons vel.x with NaN in that
 * case, which makes cos_theta the same NaN.  17 % less pool traffic on the way in (profiles/r05_pass_counters_*.txt). */
constexpr unsigned kRowPlanes = 5;
constexpr unsigned kRowData = kRowPlanes * 256;
constexpr unsigned kBlockTrailer = kBlockRows * kRowData;         /* masks[kBlockRows] (u64), then next (u32) */
constexpr unsigned kBlockBytes = kBlockTrailer + kBlockRows * 8 + 64;
constexpr unsigned kNoBlock = 0xffffffffu;

constexpr int kMaxChains = 2;
constexpr size_t kCounterStride = 64;      /* bytes between the chains' DeferCounters at the head of the workspace */
static_assert(sizeof(DeferCounters) <= kCounterStride, "one counter block per chain");
struct WorkspaceObject {
    uint8_t* d_base; size_t bytes; int device;
    DeferCounters* h_stats;      /* pinned host copy (one per chain) of the counters its last launch left (asynchronous, behind that launch) */
    hipStream_t side;            /* the second chain's stream (round 4), with the two events that fork it from and join it to the caller's */
    hipEvent_t forked, joined;
};
std::mutex g_ws_mu;
std::unordered_map<int, WorkspaceObject> g_ws;
int g_ws_next = 1;

/* ------------------------------------------------------------------ lattice-hash tables (rrt_noise_table)
 * Two dense boxes of the integer lattice, one for the accretion fbm and one for the dust-cloud noise calls
 * (layout and use: NoiseLut in rrt_device.h).  The boxes are computed on the host from the coordinate ranges
 * those calls can reach for t0 <= time <= t1 (lut_boxes below); a launch with a time outside that window
 * simply runs the arithmetic kernels. */
struct LutBox { int x0, y0, z0, nx, ny, nz; };
struct NoiseTableObject {
    float4* d_cells;          /* accretion box, then dust box */
    size_t bytes;
    float t0, t1;             /* launches with t0 <= time <= t1 read the table */
    int coverage;             /* RRT_TABLE_FULL / _COARSE / _COARSEST */
    unsigned acc_families, dust_families;
    LutBox acc, dust;
    int device;
};
std::mutex g_nt_mu;
std::unordered_map<int, NoiseTableObject> g_nt;
int g_nt_next = 1;

struct Interval {
    double lo, hi;
    Interval scaled(double s) const { return s >= 0 ? Interval{lo * s, hi * s} : Interval{hi * s, lo * s}; }
    Interval shifted(double a, double b) const { return Interval{lo + a, hi + b}; }      /* + [a, b] */
    Interval widened(double w) const { return Interval{lo - w, hi + w}; }
};
struct Reach {                 /* running union of the lattice points a noise3D call family can touch */
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
    void add(const Interval c[3]) {
        for (int k = 0; k < 3; ++k) { lo[k] = std::fmin(lo[k], c[k].lo); hi[k] = std::fmax(hi[k], c[k].hi); }
    }
    /* `octaves` octaves of fbm starting at c: p -> p*2.05 + 10 (math_utils.h:116) */
    void add_fbm(Interval c[3], int octaves) {
        for (int o = 0; o < octaves; ++o) {
            add(c);
            for (int k = 0; k < 3; ++k) c[k] = c[k].scaled(2.05).shifted(10.0, 10.0);
        }
    }
    LutBox box() const {       /* floor(lo) .. floor(hi) + 1 are the corners used; two cells of slack on every side */
        LutBox b;
        int l[3], h[3];
        for (int k = 0; k < 3; ++k) { l[k] = (int)std::floor(lo[k]) - 2; h[k] = (int)std::floor(hi[k]) + 1 + 2; }
        b.x0 = l[0]; b.y0 = l[1]; b.z0 = l[2];
        b.nx = h[0] - l[0] + 1; b.ny = h[1] - l[1] + 1; b.nz = h[2] - l[2] + 1;
        return b;
    }
};

/* Which noise3D call families a table of a given coverage serves (bit layout = the `from_table` words of
 * accretion_density_at / dust_density_at in rrt_device.h).  The finest families dominate the volume of the dust box
 * (it grows with the cube of the scale: 4.41^3 = 86 against 2.1^3 = 9 and 1), so a coarser coverage buys a table an
 * order of magnitude smaller for sequences that run long. */
void coverage_families(int coverage, unsigned& acc, unsigned& dust) {
    acc = (1u << rrt::kLutAccOctaves) - 1u;
    dust = 0xfu | (((1u << rrt::kLutRidgeOctaves) - 1u) << 4) | (rrt::kLutDetail ? 256u : 0u);
    if (coverage >= RRT_TABLE_COARSE) dust &= ~((1u << 6) | 256u);              /* without ridge octave 2 (4.41 cells per unit) and the detail octave (4.0) */
    if (coverage >= RRT_TABLE_COARSEST) { dust &= ~(1u << 5); acc &= 0x7u; }    /* without ridge octave 1 (2.1) and accretion octave 3 (3.9) */
}

/* Coordinate ranges of the table-served noise calls for t0 <= time <= t1, from the constants of
 * densities.h (every bound is taken generously: the functions only run for rc in [10, 25], the
 * accretion one for |y| < 4 and the dust one for |y| < 0.75 -- the zone tests of raymarcher.cu:57-58 --
 * |sin|, |cos| <= 1 + 1e-6, |atan2| <= pi + 1e-6, |noise3D| < 1 + 1e-6, hence |fbm(.,2)| < 0.76).
 * The dust box does NOT stay bounded for a window that slides: its z coordinate is 10 (azimuth - time * omega)
 * with omega = (10/rc)^1.5 in [0.253, 1] (densities.h:88-93) -- differential rotation -- so the reachable z range
 * is 10 [-(pi + max t omega), pi - min t omega]: its width grows like 0.75 t0 + (t1 - t0).  A window bounds it from
 * both sides, a coarser coverage cuts the scale factor. */
void lut_boxes(double t0, double t1, unsigned acc_fam, unsigned dust_fam, LutBox& acc, LutBox& dust) {
    const double pi = 3.14159265358979 + 1e-5;
    const double slack = 1e-6 * (std::fabs(t0) + std::fabs(t1)) + 1e-3;      /* float rounding of time * rate at large times */
    {   /* getAccretionDensity, densities.h:44-54: (rc cos, 4y, rc sin)*0.45 + (0, 0.35 t, 0) */
        Reach r;
        Interval c[3] = {Interval{-25.0, 25.0}.scaled(0.45).widened(1e-3),
                         Interval{-16.0, 16.0}.scaled(0.45).shifted(0.35 * t0, 0.35 * t1).widened(slack),
                         Interval{-25.0, 25.0}.scaled(0.45).widened(1e-3)};
        int octaves = 0;
        while (octaves < rrt::kLutAccOctaves && ((acc_fam >> octaves) & 1u)) ++octaves;
        if (octaves == 0) octaves = 1;                         /* never an empty box */
        r.add_fbm(c, octaves);
        acc = r.box();
    }
    {   /* getDustCloudDensity, densities.h:93: coords = (0.8 rc, 15 y, 10 (phi - t*omega)), omega in [0.25, 1] */
        Reach r;
        const double w_lo = 0.25, w_hi = 1.0;
        const double tw_max = t1 >= 0.0 ? t1 * w_hi : t1 * w_lo;          /* max of t * omega over the window */
        const double tw_min = t0 >= 0.0 ? t0 * w_lo : t0 * w_hi;          /* min */
        const Interval sc[3] = {Interval{8.0, 20.0}.widened(1e-3), Interval{-11.25, 11.25}.widened(1e-3),
                                Interval{-(pi + tw_max) * 10.0, (pi - tw_min) * 10.0}.widened(1e-2 + 10.0 * slack)};
        const double off1[3][3] = {{0, 0, 0}, {1, 2, 3}, {4, 5, 6}}, off2[3][3] = {{0, 0, 0}, {2, 1, 0}, {0, 3, 1}};
        const int w1_oct = (dust_fam & 2u) ? 2 : 1, w2_oct = (dust_fam & 8u) ? 2 : 1;
        for (int k = 0; k < 3; ++k) {                          /* :95-99 */
            Interval c[3];
            for (int ax = 0; ax < 3; ++ax) c[ax] = sc[ax].scaled(0.15).shifted(off1[k][ax], off1[k][ax]);
            r.add_fbm(c, w1_oct);
        }
        if (dust_fam & 4u) for (int k = 0; k < 3; ++k) {       /* :101-106: (coords + 3 w1)*0.4 + offsets */
            Interval c[3];
            for (int ax = 0; ax < 3; ++ax) c[ax] = sc[ax].widened(3.0 * 0.76).scaled(0.4).shifted(off2[k][ax], off2[k][ax]);
            r.add_fbm(c, w2_oct);
        }
        double freq = 1.0;
        for (int k = 0; k < rrt::kLutRidgeOctaves; ++k) {      /* :111-120: (coords + 1.5 w2)*freq */
            if ((dust_fam >> (4 + k)) & 1u) {
                Interval c[3];
                for (int ax = 0; ax < 3; ++ax) c[ax] = sc[ax].widened(1.5 * 0.76).scaled(freq);
                r.add(c);
            }
            freq *= 2.1;
        }
        if (rrt::kLutDetail && (dust_fam & 256u)) {            /* :127: (coords + 1.5 w2)*4 + (0, 0.5 t, 0), first octave only */
            Interval c[3];
            for (int ax = 0; ax < 3; ++ax) c[ax] = sc[ax].widened(1.5 * 0.76).scaled(4.0);
            c[1] = c[1].shifted(0.5 * t0, 0.5 * t1).widened(slack);
            r.add_fbm(c, 1);
        }
        dust = r.box();
    }
}

/* noise3d_lut multiplies with 24-bit operands and addresses records with 32-bit byte offsets */
bool lut_box_addressable(const LutBox& b) {
    const size_t n = (size_t)b.nx * b.ny * b.nz;
    return (size_t)b.nx * b.ny < ((size_t)1 << 23) && n < ((size_t)1 << 28) && b.nz < (1 << 23);
}

/* boxes + byte size of a table over [t0, t1] at `coverage`; RRT_ERR_INVALID_ARGUMENT for what create would refuse */
int plan_table(float t0, float t1, int coverage, NoiseTableObject& nt) {
    if (!(t0 <= t1) || !(t0 >= -1.0e4f) || !(t1 <= 1.0e4f)) return RRT_ERR_INVALID_ARGUMENT;
    if (coverage < RRT_TABLE_FULL || coverage > RRT_TABLE_COARSEST) return RRT_ERR_INVALID_ARGUMENT;
    memset(&nt, 0, sizeof(nt));
    nt.t0 = t0; nt.t1 = t1; nt.coverage = coverage; nt.device = -1;
    coverage_families(coverage, nt.acc_families, nt.dust_families);
    lut_boxes((double)t0, (double)t1, nt.acc_families, nt.dust_families, nt.acc, nt.dust);
    if (!lut_box_addressable(nt.acc) || !lut_box_addressable(nt.dust)) return RRT_ERR_INVALID_ARGUMENT;
    nt.bytes = ((size_t)nt.acc.nx * nt.acc.ny * nt.acc.nz + (size_t)nt.dust.nx * nt.dust.ny * nt.dust.nz) * sizeof(float4);
    return RRT_OK;
}

__global__ __launch_bounds__(256) void build_noise_table(float4* cells, LutBox b) {
    const size_t n = (size_t)b.nx * b.ny * b.nz;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const int x = (int)(i % b.nx), y = (int)((i / b.nx) % b.ny), z = (int)(i / ((size_t)b.nx * b.ny));
        /* "+ 0.0f": the lattice coordinate as noise3d() forms it (ix + 0.0f, ix + 1.0f; rrt_device.h) */
        const float fx = (float)(x + b.x0) + 0.0f, fy = (float)(y + b.y0) + 0.0f, fz = (float)(z + b.z0) + 0.0f;
        const float h00 = hash31(fx, fy, fz), h10 = hash31(fx + 1.0f, fy, fz);
        const float h01 = hash31(fx, fy + 1.0f, fz), h11 = hash31(fx + 1.0f, fy + 1.0f, fz);
        cells[i] = make_float4(h00, h10 - h00, h01, h11 - h01);
    }
}

NoiseLut make_lut(const float4* cells, const LutBox& b, unsigned families) {
    NoiseLut L;
    L.cells = cells;
    L.families = families;
    L.nx = b.nx; L.nxy = b.nx * b.ny;
    L.origin = (b.z0 * b.ny + b.y0) * b.nx + b.x0;
    L.last = (unsigned)((size_t)b.nx * b.ny * b.nz - (size_t)L.nxy - 1);
    return L;
}

/* ------------------------------------------------------------------ kernel arguments */
struct RowMap {        /* local row -> image row, and where its pixels go */
    int n_local_rows;  /* rows rendered by this launch                               */
    int y_base;        /* first image row of tile 0 of shard 0                        */
    int tile_rows;     /* R                                                           */
    int shard;         /* s                                                           */
    int n_shards;      /* G: local tile k is image tile s + k*G                       */
    const int* tile_of_local;   /* rrt_tile_map: local tile k is image tile tile_of_local[k] (increasing); NULL: the rule above */
};

struct FrameArgs {
    uchar4* out;
    int width, height;
    float time;
    rrt_camera cam;
    SkyTex sky;
    /* effects (camera_settings.h) */
    int use_bloom, use_vignette, use_ca, use_lens;
    float bloom_threshold, bloom_intensity, vignette_intensity, ca_amount, distortion_amount;
    /* params */
    float spin, drag_c;
    int max_steps;
    int nudge_ulps; unsigned nudge_seed;      /* rrt_params.nudge_ulps / .nudge_seed: the conditioning probe (primary_ray) */
    RowMap rows;
    rrt_debug_outputs dbg;
    /* deferred-sampling workspace (three-pass path), all NULL for the single-kernel path */
    struct DeferCounters* ctr;
    struct WaveHdr* hdr;
    float* finals;          /* 7 arrays of n_lanes: vx, vy, vz, code (steps | hit << 31 | resume << 30), px, py, pz */
    size_t n_lanes;
    uint8_t* sample_blocks;
    unsigned block_capacity;
    /* lattice-hash tables (rrt_noise_table); only read by the kernels instantiated with MEDIA == 2 */
    NoiseLut lut_acc, lut_dust;
    /* cost-ordered dispatch of the single-kernel path (rrt_tile_order): dispatch slot -> wave tile, and where a wave
     * leaves the clocks it took; both NULL: the static centre-out order */
    const unsigned* tile_perm;
    unsigned* tile_cost;
    int tile_order_id;      /* host side only: rrt_params.tile_order */
    /* a launch that covers only dispatch rows grid_row_base + k * grid_row_stride, k < gridDim.y, of a frame's grid_rows rows
     * of wave tiles (the three-pass path's chains, round 4; stride 2 = every other row, round 5); grid_rows == 0: the kernel's
     * own grid is the whole launch */
    int grid_rows, grid_row_base, grid_row_stride;
};

/* ------------------------------------------------------------------ explicit tile -> shard maps (rrt_tile_map)
 * SURVEY.md 8e's "cost-model-weighted assignment": instead of tile t -> shard t mod G, any assignment.  The object keeps a
 * device image of the lists the kernels index (a shard's tiles in increasing t; every tile's shard and place), tied to
 * the device it was created on. */
struct TileMapObject {
    int height, tile_rows, n_shards, n_tiles, device;
    std::vector<int> shard_of_tile, offset, rows;    /* offset[s]: where shard s's tiles start in tile_of_local */
    int* d_img = nullptr;
};
std::mutex g_tm_mu;
std::unordered_map<int, std::shared_ptr<TileMapObject>> g_tm;
int g_tm_next = 1;
std::shared_ptr<TileMapObject> tile_map_lookup(int id) {
    std::lock_guard<std::mutex> lk(g_tm_mu);
    auto it = g_tm.find(id);
    return it == g_tm.end() ? nullptr : it->second;
}

/* ------------------------------------------------------------------ cost-ordered dispatch (rrt_tile_order)
 * Workgroups are dispatched in blockIdx order and a wave tile's cost is only known once it has been rendered, so the
 * static order (row blocks from the middle outwards) is right for the reference's default view and wrong wherever the
 * longest rays are somewhere else: from inside the disk a 4K frame spends 8 % of its time draining.
 * An rrt_tile_order object remembers, per wave tile, the clocks the previous launch through it took, and dispatches the
 * next launch of the same geometry longest-first (one radix sort of n_tiles keys after the frame, ~20 us: rrt_tile_sort.h).  Any order
 * renders the same pixels.  All launches through one object are serialised on the device (an event chains them across
 * streams): frames that should overlap need an object each. */
struct TileOrderObject {
    int device;
    unsigned* d_cost;        /* clocks >> 4 of each wave tile, written by the render kernel */
    unsigned* d_sorted;      /* the sort's keys between its two passes (before the sort: scratch of the cost probe) */
    unsigned* d_iota;        /* ... and the tiles that travel with them */
    unsigned* d_perm[2];     /* dispatch slot -> wave tile; [cur] is the one the next matching launch reads */
    void* d_temp; size_t temp_bytes;      /* the sort's (digit, block) counters (rrt_tile_sort.h) */
    size_t n_cap;
    int cur;
    bool have;               /* d_perm[cur] holds an order for the geometry below */
    unsigned grid_x, grid_y;
    int width, height; RowMap rows;
    hipEvent_t chained;      /* after the last sort */
    unsigned long long launches, ordered, seeded;
    bool no_seed;            /* rrt_tile_order_set_seeding(id, 0): a geometry without history renders in the static order */
    bool dead;               /* destroyed (a launch that was waiting for `mu` must not touch the buffers) */
    std::mutex mu;           /* launches through one object are serialised on the host as well */
};
bool same_row_map(const RowMap& a, const RowMap& b) {
    return a.n_local_rows == b.n_local_rows && a.y_base == b.y_base && a.tile_rows == b.tile_rows && a.shard == b.shard &&
           a.n_shards == b.n_shards && a.tile_of_local == b.tile_of_local;
}
/* the registry lock only covers the lookup; an object is pinned by its shared_ptr and serialised by its own mutex, so
 * threads driving different objects (different GPUs) never wait for each other */
std::mutex g_to_mu;
std::unordered_map<int, std::shared_ptr<TileOrderObject>> g_to;
int g_to_next = 1;
std::shared_ptr<TileOrderObject> tile_order_lookup(int id) {
    std::lock_guard<std::mutex> lk(g_to_mu);
    auto it = g_to.find(id);
    return it == g_to.end() ? nullptr : it->second;
}

/* image row of local row `lr`, and the local output row it is stored at */
__device__ __forceinline__ bool map_row(const RowMap& m, int height, int lr, int& y, int& out_row) {
    if (lr >= m.n_local_rows) return false;
    int k = lr / m.tile_rows;
    int rr = lr - k * m.tile_rows;
    int t = m.tile_of_local ? m.tile_of_local[k] : m.shard + k * m.n_shards;
    int ty0 = m.y_base + t * m.tile_rows;
    y = ty0 + rr;
    if (y >= height) return false;
    int rows_k = min(m.tile_rows, height - ty0);
    out_row = k * m.tile_rows + (rows_k - 1 - rr);     /* each tile bottom-up, raymarcher.cu:168 */
    return true;
}

/* v moved by k ulps, k uniform in [-K, K] from a 32-bit mix of (x, y, seed, component).  The bit pattern is stepped as a
 * sign-magnitude integer, so a step across zero lands on the small float of the other sign; never used on non-finite v. */
__device__ __forceinline__ uint32_t nudge_mix(uint32_t v) {
    v ^= v >> 16; v *= 0x7feb352du; v ^= v >> 15; v *= 0x846ca68bu; v ^= v >> 16;
    return v;
}
__device__ __forceinline__ float nudge_component(float v, int K, unsigned seed, int x, int y, unsigned comp) {
    const uint32_t h = nudge_mix(nudge_mix((uint32_t)x * 0x9e3779b1u + (uint32_t)y) ^ (seed * 0x85ebca6bu + comp * 0xc2b2ae35u));
    const int k = (int)(h % (uint32_t)(2 * K + 1)) - K;
    const uint32_t b = rrt_f2u(v);
    int m = (int)(b & 0x7fffffffu);                  /* magnitude as an integer; sign apart */
    m = (b >> 31) ? -m : m;
    m += k;
    const uint32_t out = m < 0 ? (0x80000000u | (uint32_t)(-m)) : (uint32_t)m;
    return rrt_u2f(out);
}

/* Primary ray of pixel (x, y): raymarcher.cu:20-34 (+ lens distortion, post_processing.h:19-24). */
__device__ __forceinline__ void primary_ray(const FrameArgs& a, int x, int y, float& uvx, float& uvy, v3& p, v3& vel) {
    uvx = (float)x / (float)a.width;
    uvy = (float)y / (float)a.height;
    if (a.use_lens) lens_distort(uvx, uvy, a.distortion_amount);
    float u_coord = uvx * 2.0f - 1.0f;
    float v_coord = uvy * 2.0f - 1.0f;
    float aspect = (float)a.width / (float)a.height;
    u_coord *= aspect;
    const v3 cfw = mk(a.cam.forward[0], a.cam.forward[1], a.cam.forward[2]);
    const v3 crt = mk(a.cam.right[0], a.cam.right[1], a.cam.right[2]);
    const v3 cup = mk(a.cam.up[0], a.cam.up[1], a.cam.up[2]);
    p = mk(a.cam.pos[0], a.cam.pos[1], a.cam.pos[2]);
    vel = normalize(add(cfw, add(mul(crt, u_coord), mul(cup, v_coord))));
    if (__builtin_expect(a.nudge_ulps != 0, 0)) {
        /* conditioning probe (rrt_params.nudge_ulps; oracle: rrto_nudge_direction, the same function): every component of
         * the unit direction moves by a whole number of ulps in [-K, K] drawn from a hash of (x, y, seed, component) */
        vel.x = nudge_component(vel.x, a.nudge_ulps, a.nudge_seed, x, y, 0u);
        vel.y = nudge_component(vel.y, a.nudge_ulps, a.nudge_seed, x, y, 1u);
        vel.z = nudge_component(vel.z, a.nudge_ulps, a.nudge_seed, x, y, 2u);
    }
}

/* Everything after the march: sky, composition, post-FX, tone map, RGBA8 store -- raymarcher.cu:124-173. */
template <bool DEBUG>
__device__ __forceinline__ void shade_and_store(const FrameArgs& a, int x, int y, int out_row, float uvx, float uvy,
                                                bool hit, v3 p, v3 vel, Radiance acc, int steps) {
    float bg_r = 0.f, bg_g = 0.f, bg_b = 0.f;
    if (!hit) {
        v3 d = normalize(vel);
        float s[4];
        if (a.use_ca) {
            sample_sky(a.sky, d, a.ca_amount, s);  bg_r = s[0];
            sample_sky(a.sky, d, 0.0f, s);         bg_g = s[1];
            sample_sky(a.sky, d, -a.ca_amount, s); bg_b = s[2];
        } else {                                   /* offset 0: the three lookups coincide */
            sample_sky(a.sky, d, 0.0f, s);
            bg_r = s[0]; bg_g = s[1]; bg_b = s[2];
        }
    }
    float hx = acc.r + bg_r * acc.t;
    float hy = acc.g + bg_g * acc.t;
    float hz = acc.b + bg_b * acc.t;

    /* raymarcher.cu:154-161, post_processing.h:13-31 */
    if (a.use_bloom) {
        const v3 bl = bloom_part(mk(hx, hy, hz), a.bloom_threshold);
        hx = hx + bl.x * a.bloom_intensity;
        hy = hy + bl.y * a.bloom_intensity;
        hz = hz + bl.z * a.bloom_intensity;
    }
    if (a.use_vignette) {
        const v3 vg = vignette(mk(hx, hy, hz), uvx, uvy, a.vignette_intensity);
        hx = vg.x; hy = vg.y; hz = vg.z;
    }

    /* raymarcher.cu:164-173 */
    float out_r = 1.0f - rrt_expf(-hx * kExposure);
    float out_g = 1.0f - rrt_expf(-hy * kExposure);
    float out_b = 1.0f - rrt_expf(-hz * kExposure);
    const size_t oi = (size_t)out_row * a.width + x;
    a.out[oi] = make_uchar4((unsigned char)(int)(out_r * 255.0f), (unsigned char)(int)(out_g * 255.0f),
                            (unsigned char)(int)(out_b * 255.0f), 255);
    if (DEBUG) {
        const size_t di = (size_t)y * a.width + x;
        if (a.dbg.d_ldr) { float* q = a.dbg.d_ldr + 4 * oi; q[0] = out_r; q[1] = out_g; q[2] = out_b; q[3] = 1.0f; }
        if (a.dbg.d_hdr) { float* q = a.dbg.d_hdr + 4 * oi; q[0] = hx; q[1] = hy; q[2] = hz; q[3] = 1.0f; }
        if (a.dbg.d_steps) a.dbg.d_steps[di] = steps;
        if (a.dbg.d_hit) a.dbg.d_hit[di] = hit ? 1 : 0;
        if (a.dbg.d_pos) { float* q = a.dbg.d_pos + 3 * di; q[0] = p.x; q[1] = p.y; q[2] = p.z; }
        if (a.dbg.d_vel) { float* q = a.dbg.d_vel + 3 * di; q[0] = vel.x; q[1] = vel.y; q[2] = vel.z; }
        if (a.dbg.d_rad) { float* q = a.dbg.d_rad + 4 * di; q[0] = acc.r; q[1] = acc.g; q[2] = acc.b; q[3] = acc.t; }
    }
}

/* The same with the strict square root seeded by an estimate of 1/r (rrt_device.h: sqrt_seeded): `seed` = 1/|p4| of
 * the previous step, whose end point differs from this position by O(h^2); 0 on a ray's first step (falls back). */
template <bool FAST>
__device__ __forceinline__ void march_radius_seeded(v3 rel_p, float seed, float& r2, float& r, float& y) {
    if (FAST) {
        r2 = dot_fma(rel_p, rel_p);
        y = __builtin_amdgcn_rsqf(r2);
        r = r2 * y;
        if (__builtin_expect(__any(!(r2 >= 1.0f)), 0)) {
            if (!(r2 >= 1.0f)) { r = sqrtf(r2); y = 1.0f; }
        }
    } else {
        r2 = dot(rel_p, rel_p);
        stage_radius<1>(r2, seed, r, y);
    }
}

/* The step size takes three values (the `in_cloud_zone` arm of raymarcher.cu:62 is unreachable: the
 * cloud zone lies inside the disk zone); h*0.5f and h/6.0f (integrators.h:31,57) are folded per value
 * at compile time. */
constexpr float kHVac = kStepSize, kHNear = kStepSize * 0.1f, kHDisk = kStepSize * 0.3f;

/*
 * Per-pixel pipeline, one ray per lane (reference raymarch_kernel, src/raymarcher.cu:15-174).
 * A 256-thread workgroup covers a 16x16 pixel block as four 8x8 wave tiles so that the 64 rays of a
 * wavefront stay spatially coherent (similar step counts, similar zone entry).
 */
/* radius of the pre-step position exactly as the march sees it (strict: correctly rounded root of the unfused r2; FMAD: the
 * correctly rounded root of the fused r2; fast: r2*rsq) */
constexpr int kArithStrict = RRT_ARITH_STRICT, kArithFast = RRT_ARITH_FAST, kArithFmad = RRT_ARITH_FMAD;
template <int ARITH>
__device__ __forceinline__ void march_radius(v3 rel_p, float& r2, float& r, float& y) {
    if (ARITH == kArithFast) {
        r2 = dot_fma(rel_p, rel_p);
        y = __builtin_amdgcn_rsqf(r2);
        r = r2 * y;
    } else {
        r2 = ARITH == kArithFmad ? dot_fma(rel_p, rel_p) : dot(rel_p, rel_p);
        sqrt_rsq(r2, r, y);
    }
    if (__builtin_expect(__any(!(r2 >= 1.0f)), 0)) {
        if (!(r2 >= 1.0f)) { r = sqrtf(r2); y = 1.0f; }
    }
}

/* One RK4 step of the march (integrate_rk4, integrators.h:23-59) from the loop-top radius of the pre-step
 * position.  (A variant without the per-stage `r < 1` guards -- 5 fewer vector instructions per step, repeated
 * with guards in the unreachable case -- was measured and dropped: the longer basic blocks it leaves let the
 * scheduler interleave independent chains, and on gfx950 a VALU instruction issued 2-6 slots after its producer
 * costs 10-15 % more than one issued right behind it; profiles/README.md, round 2.)
 * -DRRT_SEEDED_SQRT=0 builds the v_rsq-based stage radii instead (A/B: profiles/README.md). */
#ifndef RRT_SEEDED_SQRT
#define RRT_SEEDED_SQRT 1
#endif
template <bool SPIN, bool FAST>
__device__ __forceinline__ void march_step(v3& p, v3& vel, float h, float hh, float h6, float drag_c, float r2, float r, float y,
                                           float& y_seed) {
    if (FAST) integrate_rk4_fast<SPIN>(p, vel, h, hh, h6, drag_c, r2, y);
    else if (RRT_SEEDED_SQRT) integrate_rk4_seeded<SPIN>(p, vel, h, hh, h6, drag_c, r2, r, y, y_seed);
    else integrate_rk4_r<SPIN>(p, vel, h, hh, h6, drag_c, r2, r, y);
}

/* Step size of raymarcher.cu:54-62 from the zone flags; h*0.5f is exact, h/6.0f is folded per value. */
__device__ __forceinline__ void zone_step(bool near_bh, bool in_disk, float& h, float& hh, float& h6) {
    h = near_bh ? kHNear : (in_disk ? kHDisk : kHVac);
    hh = 0.5f * h;                                      /* == h * 0.5f of integrators.h:31, one multiply */
    h6 = near_bh ? kHNear / 6.0f : (in_disk ? kHDisk / 6.0f : kHVac / 6.0f);
}

/* Round 3: RRT_MARCH_V2 = the lean RK4 step (rrt_device.h: integrate_rk4_lean) and, with
 * RRT_VACUUM_PATH, a wave-uniform vacuum step.  -DRRT_MARCH_V2=0 builds round 2's loop (A/B: profiles/README.md). */
#ifndef RRT_MARCH_V2
#define RRT_MARCH_V2 1
#endif
#ifndef RRT_VACUUM_PATH
#define RRT_VACUUM_PATH 1
#endif
#ifndef RRT_HORIZON_IN_GENERIC
#define RRT_HORIZON_IN_GENERIC 0
#endif

/* r >= kVacuumR rules out the horizon test (r < 2.02) and every zone of raymarcher.cu:56-58 (near_bh r < 18, disk zone
 * r < 30, cloud zone r < 25): the step is h = STEP_SIZE_M with no media sample.  About nine steps in ten of the bench
 * frame are taken by wavefronts whose 64 rays are all out there. */
constexpr float kVacuumR = kDiskOut + 5.0f;

/* The whole march of one ray with the media sampled in line: raymarcher.cu:41-121.
 * MEDIA: 0 = densities read 0 ("skybox only"), 1 = full media, 2 = full media with the lattice-hash tables.
 * `i`: in = first step (0, or where a resumed ray stopped), out = steps taken.  When every lane starts at the
 * same step the loop counter stays in a scalar register; the per-ray count is written once, at the exit. */
template <bool SPIN, int MEDIA, bool FAST>
__device__ __forceinline__ void march_inline_v1(const FrameArgs& a, v3& p, v3& vel, Radiance& acc, bool& hit, int& i,
                                                unsigned* oob) {
    int steps = i > a.max_steps ? i : a.max_steps;      /* if the loop runs out */
    float y_seed = 0.0f;                                /* 1/r estimate for the next step's radius; 0: none yet */
    for (int k = i; k < a.max_steps; ++k) {
        const v3 rel_p = p;                             /* p - MASS_POS, MASS_POS = 0 */
        float r2, r, y;
        if (RRT_SEEDED_SQRT) march_radius_seeded<FAST>(rel_p, y_seed, r2, r, y);
        else march_radius<FAST>(rel_p, r2, r, y);
        if (r < kEventHorizon * 1.01f) { hit = true; acc.t = 0.0f; steps = k; break; }

        const bool near_bh = r < 18.0f;
        const bool in_disk = fabsf(rel_p.y) < kDiskH * 5.0f && r < kDiskOut + 5.0f;
        const bool in_cloud = fabsf(rel_p.y) < kCloudH * 1.5f && r < kCloudOut;
        float h, hh, h6;
        zone_step(near_bh, in_disk, h, hh, h6);

        march_step<SPIN, FAST>(p, vel, h, hh, h6, a.drag_c, r2, r, y, y_seed);

        if (MEDIA != 0 && (in_disk || in_cloud)) {
            float d_disk, d_cloud;
            media_densities<MEDIA == 2>(rel_p, a.time, in_disk, in_cloud, a.lut_acc, a.lut_dust, oob, d_disk, d_cloud);
            accumulate_sample(acc, d_disk, d_cloud, rel_p, r, vel, h, a.spin);
        }
        if (r > 250.0f && dot(rel_p, vel) > 0.0f) { steps = k + 1; break; }
    }
    i = steps;
}

template <bool SPIN, int MEDIA, int ARITH>
__device__ __forceinline__ void march_inline(const FrameArgs& a, v3& p, v3& vel, Radiance& acc, bool& hit, int& i,
                                             unsigned* oob) {
    constexpr bool FMA = ARITH == kArithFmad;           /* the lean loop with fused multiply-adds (rrt_device.h: integrate_rk4_lean) */
    if constexpr (ARITH == kArithFast || !RRT_MARCH_V2) {
        march_inline_v1<SPIN, MEDIA, ARITH == kArithFast>(a, p, vel, acc, hit, i, oob);
    } else {
        int steps = i > a.max_steps ? i : a.max_steps;  /* if the loop runs out */
        float ys = 0.0f, hs = 0.0f;                     /* (1/r, 1/(2r)) estimate for the next loop-top radius; 0: none yet */
        float hcp = 0.0f;                               /* 1/(2r) at the previous vacuum step's stage 3 (seed extrapolation) */
        for (int k = i; k < a.max_steps; ++k) {
            const v3 rel_p = p;                         /* p - MASS_POS, MASS_POS = 0 */
            const float r2 = FMA ? dot_fma(rel_p, rel_p) : dot(rel_p, rel_p);
            float r, y, hy;
            const bool rejected = sqrt_seeded_yh<1>(r2, ys, hs, r, y, hy);
            /* wave-uniform: every live lane holds an accepted radius >= kVacuumR (two compares, scalar logic) */
            const unsigned long long rej_mask = __builtin_amdgcn_ballot_w64(rejected);
            const bool vacuum = RRT_VACUUM_PATH && (rej_mask | __builtin_amdgcn_ballot_w64(!(r >= kVacuumR))) == 0ull;
#if RRT_HORIZON_IN_GENERIC
            if (vacuum) {
                integrate_rk4_lean<SPIN, true, FMA>(p, vel, 0.f, 0.f, 0.f, a.drag_c, r2, r, y, hy, ys, hs, hcp);
            } else {
                if (rej_mask != 0ull) {
                    bool small;
                    if (rejected) radius_fallback(r2, r, y, hy, small);
                }
                if (r < kEventHorizon * 1.01f) { hit = true; acc.t = 0.0f; steps = k; break; }
#else
            if (!vacuum && rej_mask != 0ull) {
                bool small;                             /* r < 1 ends the ray at the horizon test below */
                if (rejected) radius_fallback(r2, r, y, hy, small);
            }
            if (r < kEventHorizon * 1.01f) { hit = true; acc.t = 0.0f; steps = k; break; }

            if (vacuum) {
                integrate_rk4_lean<SPIN, true, FMA>(p, vel, 0.f, 0.f, 0.f, a.drag_c, r2, r, y, hy, ys, hs, hcp);
            } else {
#endif
                const bool near_bh = r < 18.0f;
                const bool in_disk = fabsf(rel_p.y) < kDiskH * 5.0f && r < kDiskOut + 5.0f;
                const bool in_cloud = fabsf(rel_p.y) < kCloudH * 1.5f && r < kCloudOut;
                float h, hh, h6;
                zone_step(near_bh, in_disk, h, hh, h6);
                integrate_rk4_lean<SPIN, false, FMA>(p, vel, h, hh, h6, a.drag_c, r2, r, y, hy, ys, hs, hcp);
                if (MEDIA != 0 && (in_disk || in_cloud)) {
                    float d_disk, d_cloud;
                    media_densities<MEDIA == 2>(rel_p, a.time, in_disk, in_cloud, a.lut_acc, a.lut_dust, oob, d_disk, d_cloud);
                    accumulate_sample(acc, d_disk, d_cloud, rel_p, r, vel, h, a.spin);
                }
            }
            if (r > 250.0f && (FMA ? dot_fma(rel_p, vel) : dot(rel_p, vel)) > 0.0f) { steps = k + 1; break; }     /* raymarcher.cu:120 */
        }
        i = steps;
    }
}

/* Workgroup geometry of the per-ray kernels.  A wavefront covers a compact kTileW x kTileH pixel tile (8x8
 * unless RRT_TILE_W says otherwise), so its 64 rays stay spatially coherent: similar step counts, similar
 * zone entry, and sample points that share lattice cells of the low noise octaves (which is what makes the
 * noise tables pay).  RRT_WG_WAVES = 1: one wavefront per workgroup (a CU slot is released as soon as that wave
 * is done -- no waiting for three siblings); 4: a 2x2 block of wave tiles per 256-thread workgroup. */
#ifndef RRT_WG_WAVES
#define RRT_WG_WAVES 1
#endif
#ifndef RRT_TILE_W
#define RRT_TILE_W 8
#endif
constexpr int kWGWaves = RRT_WG_WAVES;
constexpr int kWGThreads = 64 * kWGWaves;
constexpr int kTileW = RRT_TILE_W, kTileH = 64 / kTileW;
static_assert(kTileW * kTileH == 64 && (kTileW & (kTileW - 1)) == 0, "a wave tile is 64 pixels, power-of-two wide");
constexpr int kWGPixX = kWGWaves == 4 ? 2 * kTileW : kTileW, kWGPixY = kWGWaves == 4 ? 2 * kTileH : kTileH;
constexpr int kMaxGridY = 65535;               /* HIP's limit for gridDim.y: a launch covers at most kMaxGridY * kWGPixY rows */

/* Workgroups are dispatched in blockIdx order; the rows through the middle of the frame hold the
 * longest rays (shadow edge, disk), so row-blocks are visited from the middle outwards: mid, mid+1,
 * mid-1, ...  Longest-first shortens the tail of a launch; it changes no pixel. */
__device__ __forceinline__ int dispatch_row(const FrameArgs& a) { return a.grid_row_base + (int)blockIdx.y * a.grid_row_stride; }
__device__ __forceinline__ int row_block(const FrameArgs& a) {
    const int nb = a.grid_rows ? a.grid_rows : (int)gridDim.y, j = dispatch_row(a), mid = (nb - 1) >> 1;
    return (j & 1) ? mid + ((j + 1) >> 1) : mid - (j >> 1);
}
/* Workgroups go to the 8 XCDs round-robin in linear-id order (statically: ids = k mod 8 all run on one XCD), and
 * each XCD has its own L2.  RRT_XCD_RUN = L > 0 deals the tile columns of a grid row out so that the workgroups
 * sharing an XCD cover runs of L adjacent columns (a bijection inside every group of 8 L workgroups; it changes which
 * workgroup renders a tile, no pixel).  Measured (profiles/r02_xcd_columns_ab.txt): one contiguous run per XCD -- the
 * usual GEMM recipe -- is 1.5-2x SLOWER here, because an XCD then owns a vertical stripe of the image and the
 * stripes through the hole and the disk cost several times the outer ones: the round-robin interleave is what
 * balances this kernel.  Default 0 = identity. */
#ifndef RRT_XCD_RUN
#define RRT_XCD_RUN 0
#endif
__device__ __forceinline__ int tile_column(const FrameArgs& a) {
    const int bx = blockIdx.x;
    if (RRT_XCD_RUN <= 0) return bx;
    constexpr int L = RRT_XCD_RUN > 0 ? RRT_XCD_RUN : 1, G = 8 * L;
    const int base = (bx / G) * G;
    if (base + G > (int)gridDim.x) return bx;                 /* the ragged last group keeps its place */
    const unsigned id = (unsigned)dispatch_row(a) * gridDim.x + bx;          /* dispatch order: id % 8 labels the XCD */
    return base + (int)(id & 7u) * L + (((bx - base) >> 3) % L);
}
/* A wave tile's cost is its lifetime in shader clocks / 16, clamped to 22 bits (a wave that lives 30 ms); the order only
 * needs bits 6..21 of it (0.5 us steps), which is what the radix sort looks at: two 8-bit passes. */
constexpr unsigned kTileCostMax = (1u << 22) - 1u;
constexpr int kTileCostSortLo = 6, kTileCostSortHi = 22;
static_assert(kTileCostSortHi - kTileCostSortLo == 16 && (kTileCostMax >> kTileCostSortHi) == 0u, "rrt_tile_sort.h sorts 16 key bits in two passes");

/* the wave tile (row_block * gridDim.x + column) this workgroup renders: the static order above, or the launch's
 * cost-ordered permutation */
__device__ __forceinline__ unsigned wave_tile(const FrameArgs& a) {
    if (a.tile_perm) return a.tile_perm[(unsigned)dispatch_row(a) * gridDim.x + blockIdx.x];
    return (unsigned)(row_block(a) * (int)gridDim.x + tile_column(a));
}
__device__ __forceinline__ bool lane_pixel(const FrameArgs& a, int& x, int& y, int& out_row) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int col, rb;
    if (a.tile_perm) { const unsigned t = wave_tile(a); rb = (int)(t / gridDim.x); col = (int)(t - (unsigned)rb * gridDim.x); }
    else { col = tile_column(a); rb = row_block(a); }
    x = col * kWGPixX + (wave & 1) * kTileW + (lane & (kTileW - 1));
    const int lr = rb * kWGPixY + (wave >> 1) * kTileH + lane / kTileW;
    return x < a.width && map_row(a.rows, a.height, lr, y, out_row);
}
/* max over the live lanes of a wave; lanes that are not executing contribute 0 */
__device__ __forceinline__ unsigned wave_max_u32(unsigned v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        unsigned other = (unsigned)__shfl_xor((int)v, o);
        v = other > v ? other : v;
    }
    return v;
}
/* where a wavefront keeps its bookkeeping in the three-pass workspace: by the wave TILE it renders, so that the three passes
 * (and the rounds) find the same slot whatever order they are dispatched in */
__device__ __forceinline__ unsigned wave_slot(const FrameArgs& a) {
    return wave_tile(a) * (unsigned)kWGWaves + (threadIdx.x >> 6);
}
/* add this wave's lifetime (clocks / 16) to its tile's cost (every lane stores the same word) */
__device__ __forceinline__ void add_tile_cost(const FrameArgs& a, unsigned long long t_start) {
#ifdef RRT_WAVE_TIMELINE     /* dev probe (tools/wave_timeline.py): when the wave started and ended, in microseconds mod 65536, instead of its cost */
    if (RRT_WAVE_TIMELINE == 1) {
        a.tile_cost[wave_tile(a)] = (unsigned)((t_start / 100ull) & 0xffffull) | ((unsigned)((__builtin_amdgcn_s_memrealtime() / 100ull) & 0xffffull) << 16);
        return;
    }
#endif
    const unsigned long long dt = (__builtin_readcyclecounter() - t_start) >> 4;
    const unsigned t = wave_tile(a);
    const unsigned long long sum = (unsigned long long)a.tile_cost[t] + dt;
    a.tile_cost[t] = sum > kTileCostMax ? kTileCostMax : (unsigned)sum;
}

/* Single-kernel path: one ray per lane, media sampled in line (reference raymarch_kernel,
 * src/raymarcher.cu:15-174). */
/* Register budgets.  The bare march takes what the ILP-first scheduler wants (73 VGPRs = 7 waves per SIMD; it keeps its
 * rate down to 4: profiles/r03_march_occupancy_probe.txt).  The kernels that carry the media code are held to 96 = 5 waves:
 * the table-served media code loses 3.6 % at 4 waves and gains 0.7 % at 6 (80 VGPRs: no spill under the default scheduler
 * since the rrt_math.h rewrite, eight spills under max-ilp, which is worth more: r03_media_waves_ab.txt, r03_sched_strategy_ab.txt). */
#ifndef RRT_MEDIA_WAVES
#define RRT_MEDIA_WAVES 5
#endif
template <bool SPIN, int MEDIA, bool DEBUG, int ARITH>
__global__ __launch_bounds__(kWGThreads, (MEDIA != 0 && !DEBUG ? RRT_MEDIA_WAVES : 1))      /* 2nd: minimum waves per SIMD */
void raymarch_pixels(const FrameArgs a) {
#if defined(RRT_OCC_PROBE_LDS)      /* dev probe: cap the occupancy of the kernel WITHOUT media code through its LDS footprint (8192 B per one-wave
                                     * workgroup = 20 workgroups per CU = 5 waves per SIMD): what the march loses at the media kernels' occupancy */
    __shared__ volatile int occ_pad[RRT_OCC_PROBE_LDS / 4];
    if (MEDIA == 0) occ_pad[threadIdx.x] = 0;
#endif
    const unsigned long long t_start = a.tile_cost ? __builtin_readcyclecounter() : 0ull;
    int x, y, out_row;
    if (!lane_pixel(a, x, y, out_row)) return;
    float uvx, uvy;
    v3 p, vel;
    primary_ray(a, x, y, uvx, uvy, p, vel);
    Radiance acc = {0.f, 0.f, 0.f, 1.0f};
    bool hit = false;
    int i = 0;
    march_inline<SPIN, MEDIA, ARITH>(a, p, vel, acc, hit, i, DEBUG ? a.dbg.d_lut_oob : nullptr);
    shade_and_store<DEBUG>(a, x, y, out_row, uvx, uvy, hit, p, vel, acc, i);
    if (a.tile_cost) {          /* what this wave cost, for the next launch's order (every lane stores the same word) */
        const unsigned long long dt = (__builtin_readcyclecounter() - t_start) >> 4;
        a.tile_cost[wave_tile(a)] = dt > kTileCostMax ? kTileCostMax : (unsigned)dt;
    }
}

/* ---- three-pass path, pass 1: geodesics only; sample points of in-medium steps go to the pool ---- */
/* amdgpu_num_sgpr(80): gfx950 admits 8 waves per SIMD only up to 80 SGPRs (7 for 82-96); this loop needs
 * the occupancy (measured: 4 waves/SIMD is 15 % slower than 8).
 * RESUME (rounds after the first): only wavefronts the pool ran out under (state 2) do anything -- their suspended rays
 * carry on from the saved pre-step state; rays of the same wave that had already ended stay as they are. */
#ifndef RRT_DEFER_WAVES
#define RRT_DEFER_WAVES 8
#endif
#if RRT_DEFER_WAVES >= 8
#define RRT_DEFER_SGPR_ATTR __attribute__((amdgpu_num_sgpr(80)))
#else
#define RRT_DEFER_SGPR_ATTR
#endif
template <bool SPIN, int ARITH, bool RESUME>
__global__ __launch_bounds__(kWGThreads, RRT_DEFER_WAVES) RRT_DEFER_SGPR_ATTR void march_defer(const FrameArgs a) {
    constexpr bool FAST = ARITH == kArithFast, FMA = ARITH == kArithFmad;
    if (RESUME && a.ctr->last_overflow == 0u) return;            /* nothing was suspended: the whole grid leaves at once */
#ifdef RRT_WAVE_TIMELINE
    const unsigned long long t_start = a.tile_cost ? __builtin_amdgcn_s_memrealtime() : 0ull;
#else
    const unsigned long long t_start = a.tile_cost ? __builtin_readcyclecounter() : 0ull;
#endif
    int x = 0, y = 0, out_row = 0;
    const bool valid = lane_pixel(a, x, y, out_row);
    if (!__any(valid)) return;
    /* lanes without a pixel stay alive (all 64 lanes take part in the wave-level bookkeeping below); they march nothing */
    const int lane = threadIdx.x & 63;
    const unsigned wid = wave_slot(a);
    const size_t li = (size_t)wid * 64 + lane;
    v3 p = mk(1000.f, 0.f, 0.f), vel = mk(0.f, 0.f, 0.f);
    bool hit = false;
    bool active = valid;                                          /* this lane marches in this round */
    int i = 0;
    if (RESUME) {
        if (a.hdr[wid].state != 2u) return;                       /* wave-uniform: marched to its end, or shaded already */
        const unsigned code = reinterpret_cast<const unsigned*>(a.finals)[3 * a.n_lanes + li];
        vel = mk(a.finals[li], a.finals[a.n_lanes + li], a.finals[2 * a.n_lanes + li]);
        p = mk(a.finals[4 * a.n_lanes + li], a.finals[5 * a.n_lanes + li], a.finals[6 * a.n_lanes + li]);
        hit = (code >> 31) != 0;
        i = (int)(code & 0x3fffffffu);
        active = valid && (code & 0x40000000u) != 0;
    } else if (valid) {
        float uvx, uvy;
        primary_ray(a, x, y, uvx, uvy, p, vel);
    }

    /* wave-level bookkeeping; identical in every lane that is still marching */
    unsigned first_block = kNoBlock, n_runs = 0;
    unsigned run_start = kNoBlock, run_len = 0, run_blk = 0;      /* current run; block run_start + run_blk in use */
    unsigned used = kBlockRows;                                   /* rows used in the current block */
    bool overflow = false;                                        /* this lane stopped because the pool is full */

    constexpr bool LEAN = !FAST && RRT_MARCH_V2;               /* round 3's step (march_inline has the notes) */
    float y_seed = 0.0f, h_seed = 0.0f, hc_prev = 0.0f;        /* a resumed ray starts without seeds: its first root takes the
                                                                * v_rsq fall-back, which is the same correctly rounded root */
    for (; active && i < a.max_steps; ++i) {
        const v3 rel_p = p;
        float r2, r, yv, hv = 0.0f;
        bool vacuum = false;
        if constexpr (LEAN) {
            r2 = FMA ? dot_fma(rel_p, rel_p) : dot(rel_p, rel_p);
            const bool rejected = sqrt_seeded_yh<1>(r2, y_seed, h_seed, r, yv, hv);
            const unsigned long long rej_mask = __builtin_amdgcn_ballot_w64(rejected);
            vacuum = RRT_VACUUM_PATH && (rej_mask | __builtin_amdgcn_ballot_w64(!(r >= kVacuumR))) == 0ull;
            if (!vacuum && rej_mask != 0ull) {
                bool small;
                if (rejected) radius_fallback(r2, r, yv, hv, small);
            }
        } else if (RRT_SEEDED_SQRT) march_radius_seeded<FAST>(rel_p, y_seed, r2, r, yv);
        else march_radius<ARITH>(rel_p, r2, r, yv);
        if (r < kEventHorizon * 1.01f) { hit = true; break; }

        bool in_disk = false, in_cloud = false;
        float h = kHVac, hh = 0.5f * kHVac, h6 = kHVac / 6.0f;
        if (!vacuum) {                                             /* wave-uniform */
            const bool near_bh = r < 18.0f;
            in_disk = fabsf(rel_p.y) < kDiskH * 5.0f && r < kDiskOut + 5.0f;
            in_cloud = fabsf(rel_p.y) < kCloudH * 1.5f && r < kCloudOut;
            zone_step(near_bh, in_disk, h, hh, h6);
        }

        /* Both density functions return 0 unless the cylindrical radius rc = sqrtf(x*x + 0*0 + z*z) is in
         * [ISCO, DISK_OUT] (densities.h:21-22, :70-71), and behind disk_point()'s first exact early-out, y^2 rc > 135; only
         * the other steps need a sample.  Round 5: the test here is a cheap SUPERSET of that gate on rc^2 -- no square root:
         * rc = RN(sqrt(rc2)) in [10, 25] implies rc2 in (99.9999, 625.0002), and RN(RN(y y) rc) <= 135 implies
         * (y y)^2 rc2 <= 135^2 (1 + 2e-6) -- the few samples it admits beyond the exact gate evaluate to the identity in pass 2
         * (disk_point() returns false) and are dropped there, so the bytes cannot depend on it.
         * The row is reserved BEFORE the step is taken: if the pool is full the lane stops here with its
         * pre-step state intact, and the next round (or, after the last one, pass 3 in line) resumes it. */
        unsigned long long need_mask = 0ull;
        bool need = false;
        float* row_f = nullptr;
        if (!vacuum && __any(in_disk || in_cloud)) {
            if (in_disk || in_cloud) {
                const float rc2 = rel_p.x * rel_p.x + 0.0f * 0.0f + rel_p.z * rel_p.z;
                const float yy = rel_p.y * rel_p.y;
                need = rc2 >= 99.99f && rc2 <= 625.01f && (yy * yy) * rc2 <= 18227.0f;          /* 135^2 = 18225 */
            }
            need_mask = __ballot(need);
        }
        if (need_mask != 0ull) {
            const int leader = __ffsll((long long)__ballot(1)) - 1;
            if (used == kBlockRows) {                              /* wave-uniform: next block */
                if (run_start != kNoBlock && run_blk + 1 < run_len) {
                    ++run_blk;                                     /* still inside the current run */
                } else {                                           /* take a new run, twice as long */
                    const unsigned len = run_len == 0 ? 1u : (run_len * 2 > kMaxRun ? kMaxRun : run_len * 2);
                    unsigned start = 0;
                    if (lane == leader) start = atomicAdd(&a.ctr->next_block, len);
                    start = __builtin_amdgcn_readfirstlane(start);
                    if (start + len > a.block_capacity) {
                        /* the pool is full: blocks [start, capacity) now belong to this failed run; give them empty
                         * masks so that pass 2 finds nothing in them */
                        if (lane == leader) {
                            for (unsigned b2 = start; b2 < a.block_capacity; ++b2) {
                                ulonglong2* m = reinterpret_cast<ulonglong2*>(a.sample_blocks + (size_t)b2 * kBlockBytes + kBlockTrailer);
#pragma unroll
                                for (unsigned k = 0; k < kBlockRows / 2; ++k) m[k] = make_ulonglong2(0ull, 0ull);
                            }
                        }
                        overflow = true;
                        break;
                    }
                    if (lane == leader) {                          /* lanes 0-7 may have left the loop already */
                        for (unsigned b2 = 0; b2 < len; ++b2) {
                            ulonglong2* m = reinterpret_cast<ulonglong2*>(a.sample_blocks + (size_t)(start + b2) * kBlockBytes +
                                                                          kBlockTrailer);
#pragma unroll
                            for (unsigned k = 0; k < kBlockRows / 2; ++k) m[k] = make_ulonglong2(0ull, 0ull);
                        }
                        unsigned* link = reinterpret_cast<unsigned*>(a.sample_blocks + (size_t)start * kBlockBytes +
                                                                     kBlockTrailer + kBlockRows * 8);
                        link[0] = kNoBlock; link[1] = 0u;
                        if (run_start != kNoBlock) {
                            unsigned* prev = reinterpret_cast<unsigned*>(a.sample_blocks + (size_t)run_start * kBlockBytes +
                                                                         kBlockTrailer + kBlockRows * 8);
                            prev[0] = start; prev[1] = len;
                        }
                    }
                    if (first_block == kNoBlock) first_block = start;
                    run_start = start; run_len = len; run_blk = 0;
                    ++n_runs;
                }
                used = 0;
            }
            uint8_t* base = a.sample_blocks + (size_t)(run_start + run_blk) * kBlockBytes;
            if (lane == leader) reinterpret_cast<unsigned long long*>(base + kBlockTrailer)[used] = need_mask;
            row_f = reinterpret_cast<float*>(base + used * kRowData) + lane;
            ++used;
        }

        if constexpr (LEAN) {
            if (vacuum) integrate_rk4_lean<SPIN, true, FMA>(p, vel, 0.f, 0.f, 0.f, a.drag_c, r2, r, yv, hv, y_seed, h_seed, hc_prev);
            else integrate_rk4_lean<SPIN, false, FMA>(p, vel, h, hh, h6, a.drag_c, r2, r, yv, hv, y_seed, h_seed, hc_prev);
        } else march_step<SPIN, FAST>(p, vel, h, hh, h6, a.drag_c, r2, r, yv, y_seed);

        if (!vacuum && need) {                                                /* pre-step position, post-step velocity */
            row_f[0] = rel_p.x; row_f[64] = rel_p.y; row_f[128] = rel_p.z;
            /* vel.y is not stored (kRowPlanes); a non-finite one travels as a NaN in vel.x (vel.y - vel.y is 0 iff finite) */
            row_f[192] = (vel.y - vel.y == 0.0f) ? vel.x : __builtin_nanf(""); row_f[256] = vel.z;
        }
        if (r > 250.0f && (FMA ? dot_fma(rel_p, vel) : dot(rel_p, vel)) > 0.0f) { ++i; break; }
    }

    /* wave-level epilogue: all 64 lanes are here */
    const bool any_overflow = __any(overflow);
    /* lanes that left the loop early hold stale copies of the bookkeeping: the lane that ran longest has
     * the final run count, and any lane that saw the first allocation has first_block */
    n_runs = wave_max_u32(n_runs);
    first_block = ~wave_max_u32(~first_block);
    /* terminal (or, on overflow, resumable) state of every ray; pass 3 shades all pixels -- keeping the sky
     * and post-FX code with its scalar operands out of this kernel keeps it at 8 waves per SIMD */
    a.finals[li] = vel.x;
    a.finals[a.n_lanes + li] = vel.y;
    a.finals[2 * a.n_lanes + li] = vel.z;
    reinterpret_cast<unsigned*>(a.finals)[3 * a.n_lanes + li] =
        (unsigned)i | (hit ? 0x80000000u : 0u) | (overflow ? 0x40000000u : 0u);
    a.finals[4 * a.n_lanes + li] = p.x;
    a.finals[5 * a.n_lanes + li] = p.y;
    a.finals[6 * a.n_lanes + li] = p.z;
    if (lane == 0) {
        a.hdr[wid].first_block = first_block; a.hdr[wid].n_runs = n_runs; a.hdr[wid].state = any_overflow ? 2u : 1u;
        if (any_overflow) atomicAdd(&a.ctr->overflow_waves, 1u);
    }
    if (a.tile_cost) add_tile_cost(a, t_start);
}

/* zero the workspace's counters and wave headers.  A kernel, not hipMemsetAsync: the memset NODE a captured launch turned
 * into did not reliably clear them on a second replay of the graph (counters came back holding the previous replay's values
 * plus stray words; round 4, tests/test_gpu_frames.py::test_streams_graph_capture_and_borrowed_sky) */
__global__ __launch_bounds__(256) void zero_words(uint4* p, size_t n16) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n16) p[i] = make_uint4(0u, 0u, 0u, 0u);
}

/* between two rounds (one thread): close the round's statistics and empty the pool */
__global__ void pool_next_round(DeferCounters* c, unsigned capacity, int last) {
    const unsigned used = c->next_block < capacity ? c->next_block : capacity;
    if (used > c->peak_blocks) c->peak_blocks = used;
    c->total_blocks += used;
    if (c->rounds_run == 0u || c->last_overflow != 0u) ++c->rounds_with_work;
    ++c->rounds_run;
    c->last_overflow = c->overflow_waves;
    if (last) c->suspended_left = c->overflow_waves;
    c->next_block = 0u; c->overflow_waves = 0u;
}

/* ---- pass 2: densities + emission of every pooled sample row, grid-stride over the pool ---- */
template <int ARITH, bool LUT>
__global__ __launch_bounds__(256) void eval_sample_rows(const FrameArgs a) {
    const int lane = threadIdx.x & 63;
    const unsigned n_blk = min(a.ctr->next_block, a.block_capacity);
    const unsigned total = n_blk * kBlockRows;
    const unsigned n_waves = gridDim.x * 4u;
    /* (Round 5 also tried this loop software-pipelined by hand -- the next row's planes and the mask after it in flight while a
     * row is evaluated -- and measured nothing, 58.2 against 58.6-59.0 ms on the key-1 frame: the head-of-row round trip is not
     * what keeps this kernel at 0.62-0.65 of the VALU issue rate; profiles/r05_eval_prefetch_ab.txt.) */
    for (unsigned row = blockIdx.x * 4u + (threadIdx.x >> 6); row < total; row += n_waves) {
        uint8_t* blk = a.sample_blocks + (size_t)(row / kBlockRows) * kBlockBytes;
        const unsigned k = row % kBlockRows;
        unsigned long long* mask_p = reinterpret_cast<unsigned long long*>(blk + kBlockTrailer) + k;
        const unsigned long long mask = *mask_p;
        if (mask == 0ull) continue;                                   /* wave-uniform */
        const bool mine = (mask >> lane) & 1ull;
        float* f = reinterpret_cast<float*>(blk + k * kRowData) + lane;
        bool has = false;
        float ex = 0.f, ey = 0.f, ez = 0.f, s = 1.0f;
        if (mine) {
            const v3 rel_p = mk(f[0], f[64], f[128]);
            const v3 vel = mk(f[192], 0.0f, f[256]);                  /* vel.y: see kRowPlanes */
            float r2, r, yv;
            march_radius<ARITH>(rel_p, r2, r, yv);
            const bool near_bh = r < 18.0f;
            const bool in_disk = fabsf(rel_p.y) < kDiskH * 5.0f && r < kDiskOut + 5.0f;
            const bool in_cloud = fabsf(rel_p.y) < kCloudH * 1.5f && r < kCloudOut;
            const float h = near_bh ? kHNear : (in_disk ? kHDisk : kHVac);
            float d_disk, d_cloud;
            media_densities<LUT>(rel_p, a.time, in_disk, in_cloud, a.lut_acc, a.lut_dust, nullptr, d_disk, d_cloud);
            has = sample_emission(d_disk, d_cloud, rel_p, r, vel, h, a.spin, ex, ey, ez, s);
        }
        /* Round 5: a sample raymarcher.cu:71 would not take is the identity of the accumulation -- the single kernel skips it,
         * and so does pass 3 now: such lanes leave the row's mask instead of storing (0, 0, 0, 1) (the march pools every sample
         * inside the radial gate and the slab: from inside the disk most of them are real, on the bench view most are not). */
        const unsigned long long live = __ballot(has);
        if (has) { f[0] = ex; f[64] = ey; f[128] = ez; f[192] = s; }
        if (live != mask && lane == (int)(__ffsll((long long)mask) - 1)) *mask_p = live;
    }
}

/* ---- pass 3: composite each ray's samples in march order; shade the rays of wavefronts that have reached their end;
 *      LAST (the last round the host enqueued): rays still suspended are finished with the media sampled in line ---- */
template <bool SPIN, int ARITH, bool LUT, bool LAST>
__global__ __launch_bounds__(kWGThreads) void composite_and_shade(const FrameArgs a) {
    if (a.ctr->rounds_run != 0u && a.ctr->last_overflow == 0u) return;      /* a later round with nothing left: all leave */
    const unsigned long long t_start = a.tile_cost ? __builtin_readcyclecounter() : 0ull;
    int x, y, out_row;
    if (!lane_pixel(a, x, y, out_row)) return;
    const int lane = threadIdx.x & 63;
    const unsigned wid = wave_slot(a);
    const unsigned state = a.hdr[wid].state;
    if (state != 1u && state != 2u) return;                        /* shaded in an earlier round */
    const size_t li = (size_t)wid * 64 + lane;
    Radiance acc = {0.f, 0.f, 0.f, 1.0f};
    if (a.hdr[wid].flags & 1u) {                                   /* the radiance composited in earlier rounds */
        acc.r = a.finals[7 * a.n_lanes + li]; acc.g = a.finals[8 * a.n_lanes + li];
        acc.b = a.finals[9 * a.n_lanes + li]; acc.t = a.finals[10 * a.n_lanes + li];
    }
    unsigned run_start = a.hdr[wid].first_block, run_len = 1;
    const unsigned n_runs = min(a.hdr[wid].n_runs, kMaxRunsWalked);
    for (unsigned rn = 0; rn < n_runs; ++rn) {
        const unsigned* link = reinterpret_cast<const unsigned*>(a.sample_blocks + (size_t)run_start * kBlockBytes +
                                                                 kBlockTrailer + kBlockRows * 8);
        const unsigned next_start = link[0], next_len = link[1];
        if (run_len > kMaxRun || run_start + run_len > a.block_capacity) break;    /* never walk outside the pool */
        /* Blocks of a run are consecutive, so several can be in flight at once: the walk is bound by load
         * latency (a heavy wave has ~250 blocks), not by arithmetic.  kNB blocks' loads are issued together,
         * then their rows are accumulated in order. */
        constexpr unsigned kNB = 4;
        for (unsigned b = 0; b < run_len; b += kNB) {
            float e[kNB][kBlockRows][4];
            unsigned long long m[kNB][kBlockRows];
            /* unconditional, address-independent loads (rows a lane does not own are read and ignored): conditional loads
             * serialise into one memory round trip per row -- also when the condition is wave-uniform: skipping the rows whose
             * mask pass 2 has emptied cost this pass 30-150 % (round 5, profiles/r05_pass_counters.txt) for 25 % fewer bytes */
#pragma unroll
            for (unsigned j = 0; j < kNB; ++j) {
                const uint8_t* blk = a.sample_blocks + (size_t)(run_start + (b + j < run_len ? b + j : b)) * kBlockBytes;
                const unsigned long long* masks = reinterpret_cast<const unsigned long long*>(blk + kBlockTrailer);
#pragma unroll
                for (unsigned k = 0; k < kBlockRows; ++k) {
                    m[j][k] = masks[k];
                    const float* f = reinterpret_cast<const float*>(blk + k * kRowData) + lane;
                    e[j][k][0] = f[0]; e[j][k][1] = f[64]; e[j][k][2] = f[128]; e[j][k][3] = f[192];
                }
            }
#pragma unroll
            for (unsigned j = 0; j < kNB; ++j) {
                const bool live = b + j < run_len;                 /* wave-uniform */
#pragma unroll
                for (unsigned k = 0; k < kBlockRows; ++k)
                    if (live && ((m[j][k] >> lane) & 1ull))
                        accumulate_emission(acc, e[j][k][0], e[j][k][1], e[j][k][2], e[j][k][3]);
            }
        }
        run_start = next_start; run_len = next_len;
    }
    const int leader = __ffsll((long long)__ballot(1)) - 1;
    if (!LAST && state == 2u) {
        /* the wave is suspended: keep what has been composited; the next round's march resumes its rays */
        a.finals[7 * a.n_lanes + li] = acc.r; a.finals[8 * a.n_lanes + li] = acc.g;
        a.finals[9 * a.n_lanes + li] = acc.b; a.finals[10 * a.n_lanes + li] = acc.t;
        if (lane == leader) a.hdr[wid].flags = 1u;
        if (a.tile_cost) add_tile_cost(a, t_start);
        return;
    }
    float uvx, uvy;
    v3 p, vel;
    primary_ray(a, x, y, uvx, uvy, p, vel);
    vel = mk(a.finals[li], a.finals[a.n_lanes + li], a.finals[2 * a.n_lanes + li]);
    const unsigned code = reinterpret_cast<const unsigned*>(a.finals)[3 * a.n_lanes + li];
    bool hit = (code >> 31) != 0;
    int steps = (int)(code & 0x3fffffffu);
    if (LAST && state == 2u && (code & 0x40000000u)) {
        /* the pool ran out under this ray at step `steps` and no round is left: carry on from its saved pre-step state
         * with the media sampled in line -- the samples composited above come first, exactly as in the single kernel */
        p = mk(a.finals[4 * a.n_lanes + li], a.finals[5 * a.n_lanes + li], a.finals[6 * a.n_lanes + li]);
        march_inline<SPIN, LUT ? 2 : 1, ARITH>(a, p, vel, acc, hit, steps, nullptr);
    }
    if (hit) acc.t = 0.0f;                                         /* raymarcher.cu:49 */
    shade_and_store<false>(a, x, y, out_row, uvx, uvy, hit, p, vel, acc, steps);
    if (lane == leader) a.hdr[wid].state = 3u;
#ifndef RRT_WAVE_TIMELINE
    if (a.tile_cost) add_tile_cost(a, t_start);
#endif
}

/* ---- coarse cost probe (round 4): one march-only ray per cell of stride_x x stride_y pixels of the launch's row map.
 * No media evaluation, no shading: the geodesic with the march's own step rule, counting steps and the steps that would
 * need an accretion / a dust sample.  cost = w_step steps + w_acc n_acc + w_dust n_dust, in the unit of the measured tile
 * costs (wave clocks / 16: a 1000-step wave at full occupancy ~ 1.7e5), so that the same radix sort orders both.  Used
 * (i) to dispatch the FIRST frame of a geometry longest-first (rrt_tile_order has no history yet) and (ii) on the host, to
 * weigh row tiles before they are dealt to the GPUs (rrt_probe_tile_costs -> rrt_tile_map_balance).  The weights are a
 * least-squares fit of this model to measured wave-tile costs of six 4K views (tools/probe_fit.py,
 * profiles/r04_probe_cost_fit.txt). */
struct ProbeArgs {
    unsigned* cell_cost;       /* cells_y x cells_x */
    int cells_x, cells_y, stride_x, stride_y;
    float w_step, w_acc, w_dust;
    float ring_steps;          /* max_steps: what a wave on the critical curve marches */
};
#ifndef RRT_PROBE_W_STEP
#define RRT_PROBE_W_STEP 180.0f
#define RRT_PROBE_W_ACC 730.0f
#define RRT_PROBE_W_DUST 630.0f
#endif
constexpr int kProbeStride = 16;

/* The probe is latency-bound, not throughput-bound: a few hundred wavefronts on a chip with 8 192 wave slots, each a
 * serial march -- and a lone wavefront retires a dependent instruction every ~10 clocks (profiles/r04_wave_timeline_default.txt),
 * so a faithful 1000-step march took 1.6 ms however few rays there were.  It therefore marches COARSELY: kProbeStepScale times the
 * reference's step in every zone (0.3 / 0.09 / 0.03 -> 1.2 / 0.36 / 0.12; classic RK4 is still well inside its accuracy range for
 * a cost estimate) in the fast arithmetic (FMA, v_rsq): ~0.15 ms.  Every probe step stands for kProbeStepScale real ones. */
constexpr int kProbeStepScale = 4;
template <bool SPIN>
__global__ __launch_bounds__(64) void probe_costs(const FrameArgs a, const ProbeArgs q) {
    const int lane = threadIdx.x & 63;
    const int cx = blockIdx.x * 8 + (lane & 7), cy = blockIdx.y * 8 + (lane >> 3);
    if (cx >= q.cells_x || cy >= q.cells_y) return;
    /* the cell's representative pixel: its centre, in the LOCAL rows of the launch (a shard probes its own tiles only) */
    int lr = cy * q.stride_y + q.stride_y / 2, y = 0, out_row = 0;
    if (lr >= a.rows.n_local_rows) lr = a.rows.n_local_rows - 1;
    if (!map_row(a.rows, a.height, lr, y, out_row) && !map_row(a.rows, a.height, cy * q.stride_y, y, out_row)) {
        q.cell_cost[cy * q.cells_x + cx] = 0u;
        return;
    }
    const int xc = cx * q.stride_x + q.stride_x / 2;
    const int x = xc < a.width ? xc : a.width - 1;
    float uvx, uvy;
    v3 p, vel;
    primary_ray(a, x, y, uvx, uvy, p, vel);
    const int max_probe = (a.max_steps + kProbeStepScale - 1) / kProbeStepScale;
    int steps = max_probe;
    unsigned n_acc = 0, n_dust = 0;
    bool captured = false;
    for (int k = 0; k < max_probe; ++k) {
        const v3 rel_p = p;
        const float r2 = dot_fma(rel_p, rel_p);
        const float yv = __builtin_amdgcn_rsqf(r2);
        const float r = r2 * yv;
        if (!(r >= kEventHorizon * 1.01f)) { steps = k; captured = true; break; }              /* horizon (or NaN) */
        const bool near_bh = r < 18.0f;
        const bool in_disk = fabsf(rel_p.y) < kDiskH * 5.0f && r < kDiskOut + 5.0f;
        const bool in_cloud = fabsf(rel_p.y) < kCloudH * 1.5f && r < kCloudOut;
        const float h = (float)kProbeStepScale * (near_bh ? kHNear : (in_disk ? kHDisk : kHVac));
        integrate_rk4_fast<SPIN>(p, vel, h, 0.5f * h, h * (1.0f / 6.0f), a.drag_c, r2, yv);
        if (in_disk || in_cloud) {
            const float rc2 = rel_p.x * rel_p.x + rel_p.z * rel_p.z;
            if (rc2 >= kIsco * kIsco && rc2 <= kDiskOut * kDiskOut) {
                /* the slab early-out of disk_point(): y^2 rc > 135 ends both densities twelve instructions in */
                if (rel_p.y * rel_p.y * rel_p.y * rel_p.y * rc2 <= 135.0f * 135.0f) { n_acc += in_disk; n_dust += in_cloud; }
            }
        }
        if (r > 250.0f && dot(rel_p, vel) > 0.0f) { steps = k + 1; break; }
    }
    const float c = (float)kProbeStepScale * (q.w_step * (float)steps + q.w_acc * (float)n_acc + q.w_dust * (float)n_dust);
    /* bit 0: the ray ended on the horizon.  Cells whose neighbours disagree about that hold the CRITICAL CURVE (the edge
     * of the shadow), whose rays orbit until MAX_STEPS: a ring a pixel or two wide that a sample every 16 pixels mostly
     * misses, and the longest waves of the frame (probe_to_tiles prices those cells at max_steps). */
    const unsigned ci = c >= (float)kTileCostMax ? kTileCostMax : (unsigned)c;
    q.cell_cost[cy * q.cells_x + cx] = (ci & ~1u) | (captured ? 1u : 0u);
}
/* every wave tile (tiles_x x tiles_y of kWGPixX x kWGPixY pixels) takes the cost of the probe cell it lies in */
__global__ __launch_bounds__(256) void probe_to_tiles(unsigned* tile_cost, unsigned tiles_x, unsigned tiles_y, ProbeArgs q) {
    const unsigned t = blockIdx.x * 256u + threadIdx.x;
    if (t >= tiles_x * tiles_y) return;
    const unsigned rb = t / tiles_x, col = t - rb * tiles_x;
    int cx = (int)(col * kWGPixX) / q.stride_x, cy = (int)(rb * kWGPixY) / q.stride_y;
    cx = cx < q.cells_x ? cx : q.cells_x - 1; cy = cy < q.cells_y ? cy : q.cells_y - 1;
    const unsigned own = q.cell_cost[cy * q.cells_x + cx];
    unsigned cost = own;
    /* 3 x 3 neighbourhood: the largest estimate (thin structures between samples), and -- where captured and escaping
     * samples meet -- the cost of a wave that marches to max_steps */
    bool mixed = false;
    for (int dy = -1; dy <= 1; ++dy)
        for (int dx = -1; dx <= 1; ++dx) {
            const int nx = cx + dx, ny = cy + dy;
            if (nx < 0 || ny < 0 || nx >= q.cells_x || ny >= q.cells_y) continue;
            const unsigned c = q.cell_cost[ny * q.cells_x + nx];
            mixed = mixed || ((c ^ own) & 1u);
            cost = c > cost ? c : cost;
        }
    if (mixed) {
        const float ring = q.w_step * q.ring_steps;
        const unsigned rc = ring >= (float)kTileCostMax ? kTileCostMax : (unsigned)ring;
        cost = rc > cost ? rc : cost;
    }
    tile_cost[t] = cost;
}

/* one wavefront sleeps for `ticks` of the 100 MHz counter and reports both counters' deltas (rrt_clock_probe) */
__global__ __launch_bounds__(64) void clock_probe_kernel(unsigned long long* out, unsigned long long ticks) {
    const unsigned long long c0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long r1 = r0;
    for (unsigned it = 0; it < (1u << 24) && r1 - r0 < ticks; ++it) {       /* bounded: ~1.3 us per turn */
        __builtin_amdgcn_s_sleep(127);
        r1 = __builtin_amdgcn_s_memrealtime();
    }
    const unsigned long long c1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) { out[0] = c1 - c0; out[1] = r1 - r0; }
}

/* scatter one shard's tile buffer into the full bottom-up frame */
__global__ __launch_bounds__(256) void assemble_tiles_kernel(uchar4* frame, const uchar4* tiles, int width,
                                                            int height, RowMap m) {
    const int x = blockIdx.x * 256 + threadIdx.x;
    if (x >= width) return;
    for (int lr = blockIdx.y; lr < m.n_local_rows; lr += gridDim.y) {        /* gridDim.y is capped at kMaxGridY */
        int y, out_row;
        if (map_row(m, height, lr, y, out_row))
            frame[(size_t)(height - 1 - y) * width + x] = tiles[(size_t)out_row * width + x];
    }
}

/* all shards in one launch: `tiles` holds n_shards buffers, `shard_stride` pixels apart */
__global__ __launch_bounds__(256) void assemble_all_kernel(uchar4* frame, const uchar4* tiles, size_t shard_stride,
                                                          int width, int height, int tile_rows, int n_shards) {
    const int x = blockIdx.x * 256 + threadIdx.x;
    if (x >= width) return;
    for (int y = blockIdx.y; y < height; y += gridDim.y) {                   /* image row; gridDim.y capped */
        const int t = y / tile_rows, rr = y - t * tile_rows;
        const int shard = t % n_shards, k = t / n_shards;
        const int rows_k = min(tile_rows, height - t * tile_rows);
        const size_t src_row = (size_t)k * tile_rows + (rows_k - 1 - rr);
        frame[(size_t)(height - 1 - y) * width + x] = tiles[shard * shard_stride + src_row * width + x];
    }
}

/* all shards of an explicit tile map (rrt_tile_map) in one launch */
__global__ __launch_bounds__(256) void assemble_map_kernel(uchar4* frame, const uchar4* tiles, size_t shard_stride, int width,
                                                          int height, int tile_rows, const int* shard_of_tile, const int* k_of_tile) {
    const int x = blockIdx.x * 256 + threadIdx.x;
    if (x >= width) return;
    for (int y = blockIdx.y; y < height; y += gridDim.y) {
        const int t = y / tile_rows, rr = y - t * tile_rows;
        const int rows_k = min(tile_rows, height - t * tile_rows);
        const size_t src_row = (size_t)k_of_tile[t] * tile_rows + (rows_k - 1 - rr);
        frame[(size_t)(height - 1 - y) * width + x] = tiles[shard_of_tile[t] * shard_stride + src_row * width + x];
    }
}

/* ------------------------------------------------------------------ unit kernels */
__device__ __forceinline__ v3 ld3(const float* a, int i) { return mk(a[3 * i], a[3 * i + 1], a[3 * i + 2]); }
__device__ __forceinline__ void st3(float* a, int i, v3 v) { a[3 * i] = v.x; a[3 * i + 1] = v.y; a[3 * i + 2] = v.z; }

__global__ void k_geodesic_acc(int n, const float* p, const float* v, float spin, float* out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float drag_c = (2.0f * spin) * 2.0f;
    st3(out, i, geodesic_acc<true>(ld3(p, i), ld3(v, i), drag_c));      /* the march's own code path */
}
__global__ void k_rk4(int n, float* p, float* v, const float* h, float spin) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    v3 pp = ld3(p, i), vv = ld3(v, i);
    float drag_c = (2.0f * spin) * 2.0f;
    if (spin != 0.0f) integrate_rk4<true>(pp, vv, h[i], drag_c);
    else integrate_rk4<false>(pp, vv, h[i], drag_c);
    st3(p, i, pp); st3(v, i, vv);
}
/* The PRODUCTION step (round 3's integrate_rk4_lean, what every render kernel runs) as a chain of n_steps steps per
 * element, driven exactly as march_inline drives it: loop-top radius from the seed pair the previous step handed on
 * (seeded Goldschmidt root, v_rsq fall-back where the seed is rejected), horizon test r < 2.02 (the ray stops; the lean
 * step's stage 1 relies on it), then
 *   h == NULL: the march's own zone rule for the step size (raymarcher.cu:56-62) and the wave-uniform VACUUM step when all
 *              64 lanes of the wavefront hold an accepted radius >= 30 -- both template instances, the extrapolated
 *              seeds and the fall-backs are exercised by the inputs of tests/test_gpu_units.py;
 *   h != NULL: the generic step with the caller's step size on every step.
 * seed_scale: the first loop-top root's seed is seed_scale / r (0: none, as a ray's first step; 1.3: a bad seed that must be
 * rejected; 1.00005: an imperfect one that is accepted).  steps[i] = steps taken before the horizon test stopped the ray. */
template <bool SPIN>
__global__ __launch_bounds__(64) void k_rk4_lean(int n, float* p, float* v, const float* h_in, float drag_c, int n_steps,
                                                 float seed_scale, int* steps_out) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    const bool valid = i < n;
    v3 pp = valid ? ld3(p, i) : mk(1000.f, 0.f, 0.f), vv = valid ? ld3(v, i) : mk(0.f, 0.f, 0.f);
    float ys = 0.0f, hs = 0.0f, hcp = 0.0f;
    if (seed_scale != 0.0f) {
        float r0, y0;
        sqrt_rsq(dot(pp, pp), r0, y0);
        ys = seed_scale * y0; hs = 0.5f * ys;
    }
    int k = 0;
    for (; k < (valid ? n_steps : 0); ++k) {
        const v3 rel_p = pp;
        const float r2 = dot(rel_p, rel_p);
        float r, y, hy;
        const bool rejected = sqrt_seeded_yh<1>(r2, ys, hs, r, y, hy);
        const unsigned long long rej_mask = __builtin_amdgcn_ballot_w64(rejected);
        const bool vacuum = h_in == nullptr && RRT_VACUUM_PATH && (rej_mask | __builtin_amdgcn_ballot_w64(!(r >= kVacuumR))) == 0ull;
        if (!vacuum && rej_mask != 0ull) {
            bool small;
            if (rejected) radius_fallback(r2, r, y, hy, small);
        }
        if (r < kEventHorizon * 1.01f) break;
        if (vacuum) {
            integrate_rk4_lean<SPIN, true>(pp, vv, 0.f, 0.f, 0.f, drag_c, r2, r, y, hy, ys, hs, hcp);
        } else {
            float h, hh, h6;
            if (h_in) { h = h_in[i]; hh = 0.5f * h; h6 = h / 6.0f; }
            else {
                const bool near_bh = r < 18.0f;
                const bool in_disk = fabsf(rel_p.y) < kDiskH * 5.0f && r < kDiskOut + 5.0f;
                zone_step(near_bh, in_disk, h, hh, h6);
            }
            integrate_rk4_lean<SPIN, false>(pp, vv, h, hh, h6, drag_c, r2, r, y, hy, ys, hs, hcp);
        }
    }
    if (valid) { st3(p, i, pp); st3(v, i, vv); if (steps_out) steps_out[i] = k; }
}
/* the march's divide on explicit operands: out = div_seeded(a, b, seed) */
__global__ void k_div_seeded(int n, const float* a, const float* b, const float* seed, float* out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = div_seeded(a[i], b[i], seed[i]);
}
__global__ void k_hash31(int n, const float* p, float* out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = hash31(p[3 * i], p[3 * i + 1], p[3 * i + 2]);
}
__global__ void k_noise3d(int n, const float* p, float* out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = noise3d(ld3(p, i));
}
__global__ void k_fbm(int n, const float* p, int oct, float* out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    v3 q = ld3(p, i);
    float v = 0.0f, amp = 0.5f;
    for (int o = 0; o < oct; ++o) {
        v += amp * noise3d(q);
        q = mk(q.x * 2.05f + 10.0f, q.y * 2.05f + 10.0f, q.z * 2.05f + 10.0f);
        amp *= 0.5f;
    }
    out[i] = v;
}
__global__ void k_accretion(int n, const float* p, float time, float* out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = accretion_density<false, false>(ld3(p, i), time, NoiseLut{}, nullptr);
}
__global__ void k_dust(int n, const float* p, float time, float* out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = dust_density<false, false>(ld3(p, i), time, NoiseLut{}, nullptr);
}
__global__ void k_redshift(int n, const float* p, const float* vel, float spin, float* out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = redshift_factor(ld3(p, i), ld3(vel, i), spin);          /* the literal (IEEE-division) form */
}
__global__ void k_math(int fn, int n, const float* a, const float* b, float* out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float r = 0.0f;
    switch (fn) {
        case 0: r = rrt_expf(a[i]); break;
        case 1: r = rrt_powf(a[i], b[i]); break;
        case 2: r = rrt_sinf(a[i]); break;
        case 3: r = rrt_cosf(a[i]); break;
        case 4: r = rrt_atan2f(a[i], b[i]); break;
        case 5: r = rrt_asinf(a[i]); break;
        default: break;
    }
    out[i] = r;
}
__global__ void k_sky(int n, const float* dir, float off, SkyTex sky, float* out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float s[4];
    sample_sky(sky, ld3(dir, i), off, s);
    out[4 * i] = s[0]; out[4 * i + 1] = s[1]; out[4 * i + 2] = s[2]; out[4 * i + 3] = s[3];
}

__global__ void k_disk_temperature(int n, const float* r, float* out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = disk_temperature(r[i]);
}
__global__ void k_smoothstep(int n, const float* e0, const float* e1, const float* x, float* out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = smoothstep(e0[i], e1[i], x[i]);
}
/* what: 0 lens (uv -> uv), 1 vignette (rgb, uv -> rgb), 2 bloom contribution (rgb -> rgb) */
__global__ void k_postfx(int what, int n, const float* rgb, const float* uv, float param, float* out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (what == 0) {
        float ux = uv[2 * i], uy = uv[2 * i + 1];
        lens_distort(ux, uy, param);
        out[2 * i] = ux; out[2 * i + 1] = uy;
    } else if (what == 1) {
        st3(out, i, vignette(ld3(rgb, i), uv[2 * i], uv[2 * i + 1], param));
    } else {
        st3(out, i, bloom_part(ld3(rgb, i), param));
    }
}
/* the radiative-transfer block raymarcher.cu:71-116 on one sample per element; rad = (I_r, I_g, I_b, T) in/out.
 * r = length(p) exactly as the march holds it. */
__global__ void k_rt_sample(int n, const float* d_disk, const float* d_cloud, const float* p, const float* vel,
                            const float* h, float spin, float* rad) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const v3 rp = ld3(p, i);
    float r2, r, y;
    march_radius<kArithStrict>(rp, r2, r, y);
    Radiance acc = {rad[4 * i], rad[4 * i + 1], rad[4 * i + 2], rad[4 * i + 3]};
    accumulate_sample(acc, d_disk[i], d_cloud[i], rp, r, ld3(vel, i), h[i], spin);
    rad[4 * i] = acc.r; rad[4 * i + 1] = acc.g; rad[4 * i + 2] = acc.b; rad[4 * i + 3] = acc.t;
}
/* noise3D through the lattice-hash table (which: 0 accretion box, 1 dust box); counts[0] += reads the clamp had to move */
__global__ void k_noise3d_lut(int n, const float* p, NoiseLut L, float* out, unsigned* counts) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = noise3d_lut(L, ld3(p, i), counts);
}
/* the two density functions exactly as the render kernels call them (early-out, table switches) */
__global__ void k_media_lut(int n, const float* p, float time, NoiseLut la, NoiseLut ld, float* out_disk, float* out_dust,
                            unsigned* counts) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const v3 q = ld3(p, i);
    const float r = length(q);                           /* the zone tests of raymarcher.cu:57-58 gate the calls */
    const bool in_disk = fabsf(q.y) < kDiskH * 5.0f && r < kDiskOut + 5.0f;
    const bool in_cloud = fabsf(q.y) < kCloudH * 1.5f && r < kCloudOut;
    media_densities<true>(q, time, in_disk, in_cloud, la, ld, counts, out_disk[i], out_dust[i]);
}

/*
 * Self-checks of the march loop's sqrt/divide cores against the hardware-IEEE forms (sqrtf, `/`).
 * sqrt: every float whose bit pattern lies in [lo, hi).  div: `n` pseudo-random cases shaped
 * like the loop's operands: r2 log-uniform in [1, 2^28), seeds from sqrt_rsq(r2), numerators
 * log-uniform in 2^[-40, 40) with random sign.  counters[0] += mismatches; counters[1..3] keep
 * one failing case.
 */
__device__ __forceinline__ uint32_t mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
__global__ void k_selfcheck_sqrt(uint32_t lo, uint32_t hi, unsigned long long* counters) {
    uint64_t idx = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    unsigned bad = 0;
    for (uint64_t b = (uint64_t)lo + idx; b < hi; b += stride) {
        float x = rrt_u2f((uint32_t)b);
        float r, y;
        sqrt_rsq(x, r, y);
        float want = sqrtf(x);
        if (rrt_f2u(r) != rrt_f2u(want)) { ++bad; counters[1] = b; }
    }
    if (bad) atomicAdd(counters, (unsigned long long)bad);
}
__global__ void k_selfcheck_div(unsigned long long n, uint32_t seed, unsigned long long* counters) {
    uint64_t idx = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    unsigned bad = 0;
    for (uint64_t k = idx; k < n; k += stride) {
        uint32_t h1 = mix32((uint32_t)k * 2654435761u + seed), h2 = mix32(h1 ^ (uint32_t)(k >> 32) ^ 0x9e3779b9u);
        uint32_t h3 = mix32(h2 + 0x85ebca6bu);
        float r2 = rrt_u2f(0x3f800000u + (h1 % (28u << 23)));                 /* [1, 2^28) */
        float num = rrt_u2f(((87u << 23) + (h2 % (80u << 23))) | (h3 & 0x80000000u));   /* +-2^[-40,40) */
        float c = rrt_u2f(0x3f000000u + (h3 & 0x01ffffffu));                 /* [0.5, 8): drag constants */
        float r, y;
        sqrt_rsq(r2, r, y);
        float y2 = y * y, y3 = y2 * y;
        float d2 = r2 * r, d1 = (r2 * r2) * r;
        float q1 = div_seeded(num, d1, y3 * y2), q2 = div_seeded(c, d2, y3);
        float w1 = num / d1, w2 = c / d2;
        if (rrt_f2u(q1) != rrt_f2u(w1)) { ++bad; counters[1] = rrt_f2u(num); counters[2] = rrt_f2u(d1); }
        if (rrt_f2u(q2) != rrt_f2u(w2)) { ++bad; counters[1] = rrt_f2u(c); counters[2] = rrt_f2u(d2); counters[3] = 2; }
    }
    if (bad) atomicAdd(counters, (unsigned long long)bad);
}

/* The same two divides with the reciprocal-root seed AS THE MARCH PRODUCES IT (round 4): y comes out of
 * sqrt_seeded_yh<1> / <2> started from an estimate that is off by up to the acceptance tolerance of each form (uniform in
 * +-1.45e-4 for the one-iteration root -- which also covers the linearly extrapolated seeds of the vacuum step --, +-8.9e-3
 * for the two-iteration one), not out of the v_rsq-based sqrt_rsq that k_selfcheck_div uses: such a y carries up to 1.5 e^2 =
 * 3.4e-8 of its own error into y^3 and y^5, i.e. the Markstein cores start from a seed ~1.7x worse than k_selfcheck_div's.
 * Rejected roots are skipped (the march takes the sqrt_rsq fall-back there).  The seeded ROOT is checked first (against
 * the v_rsq-based correctly rounded one): counters (8 x uint64) [0] += accepted one-iteration roots that are not correctly
 * rounded, [1] += two-iteration ones, [2] += divide mismatches, [3] += divides checked, [4]/[5] one failing root (x bits,
 * seed bits), [6]/[7] one failing divide (numerator, denominator bits).  tol1 / tol2: half-width of the seed errors tried. */
__global__ void k_selfcheck_div_march(unsigned long long n, uint32_t seed, float tol1, float tol2, unsigned long long* counters) {
    uint64_t idx = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    unsigned bad_root1 = 0, bad_root2 = 0, bad_div = 0;
    unsigned long long checked = 0;
    for (uint64_t k = idx; k < n; k += stride) {
        uint32_t h1 = mix32((uint32_t)k * 2654435761u + seed), h2 = mix32(h1 ^ (uint32_t)(k >> 32) ^ 0x9e3779b9u);
        uint32_t h3 = mix32(h2 + 0x85ebca6bu), h4 = mix32(h3 ^ 0xc2b2ae35u);
        float r2 = rrt_u2f(0x3f800000u + (h1 % (28u << 23)));                 /* [1, 2^28) */
        float num = rrt_u2f(((87u << 23) + (h2 % (80u << 23))) | (h3 & 0x80000000u));   /* +-2^[-40,40) */
        float c = rrt_u2f(0x3f000000u + (h3 & 0x01ffffffu));                 /* [0.5, 8): drag constants */
        float r_ref, y_ref;
        sqrt_rsq(r2, r_ref, y_ref);
        const bool two = (h4 & 1u) != 0;
        const float u = (float)((h4 >> 8) & 0xffffffu) * (2.0f / 16777216.0f) - 1.0f;      /* [-1, 1) */
        const float y0 = y_ref * (1.0f + u * (two ? tol2 : tol1));
        float r, y, hy;
        const bool rejected = two ? sqrt_seeded_yh<2>(r2, y0, 0.5f * y0, r, y, hy) : sqrt_seeded_yh<1>(r2, y0, 0.5f * y0, r, y, hy);
        if (rejected) continue;
        if (rrt_f2u(r) != rrt_f2u(r_ref)) {            /* an ACCEPTED root that is not the correctly rounded one */
            if (two) ++bad_root2; else ++bad_root1;
            counters[4] = rrt_f2u(r2); counters[5] = rrt_f2u(y0);
            continue;
        }
        float y2 = y * y, y3 = y2 * y;
        float d2 = r2 * r, d1 = (r2 * r2) * r;
        float q1 = div_seeded(num, d1, y3 * y2), q2 = div_seeded(c, d2, y3);
        float w1 = num / d1, w2 = c / d2;
        checked += 2;
        if (rrt_f2u(q1) != rrt_f2u(w1)) { ++bad_div; counters[6] = rrt_f2u(num); counters[7] = rrt_f2u(d1); }
        if (rrt_f2u(q2) != rrt_f2u(w2)) { ++bad_div; counters[6] = rrt_f2u(c); counters[7] = rrt_f2u(d2); }
    }
    if (bad_root1) atomicAdd(counters, (unsigned long long)bad_root1);
    if (bad_root2) atomicAdd(counters + 1, (unsigned long long)bad_root2);
    if (bad_div) atomicAdd(counters + 2, (unsigned long long)bad_div);
    atomicAdd(counters + 3, checked);
}

/* sqrt_seeded against sqrtf: every float whose bits lie in [lo, hi), with estimates of 1/sqrt(x) that are off by
 * 0, +-1e-5 ... +-1.2e-2 relative (a fixed ladder plus 16 pseudo-random errors per x), one and two iterations.  Wherever sqrt_seeded ACCEPTS its result (returns true) the
 * root must be sqrtf(x) bit for bit.  counters[0] += mismatches, [1]/[2] one failing case (x bits, seed bits),
 * [3] += accepted cases (so that a test can see the check was not vacuous). */
__global__ void k_selfcheck_sqrt_seeded(uint32_t lo, uint32_t hi, unsigned long long* counters) {
    uint64_t idx = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    const float deltas[10] = {0.0f, 1e-5f, 5e-5f, 1e-4f, 1.4e-4f, 1.6e-4f, 1e-3f, 5e-3f, 1e-2f, 1.2e-2f};
    unsigned bad = 0;
    unsigned long long accepted = 0;
    for (uint64_t b = (uint64_t)lo + idx; b < hi; b += stride) {
        const float x = rrt_u2f((uint32_t)b);
        const float want = sqrtf(x);
        const float y_exact = (float)(1.0 / sqrt((double)x));
        for (int k = 0; k < 18; ++k) {
            for (int sgn = -1; sgn <= 1; sgn += 2) {
                /* the ladder, then 8 pseudo-random errors per x: 4 inside the one-iteration tolerance, 4 inside the two-iteration one */
                float delta;
                if (k < 10) delta = deltas[k];
                else {
                    const uint32_t hsh = mix32((uint32_t)b * 2654435761u + (uint32_t)(k * 2 + (sgn > 0)));
                    delta = (float)(hsh & 0xffffffu) * (1.0f / 16777216.0f) * (k < 14 ? 1.45e-4f : 9.5e-3f);
                }
                const float seed = y_exact * (1.0f + (float)sgn * delta);
                float r1, y1, r2, y2;
#if RRT_MARCH_V2
                /* the form the march uses since round 3: (y, y/2) handed on, acceptance on the FIRST residual */
                float h1, h2;
                if (!sqrt_seeded_yh<1>(x, seed, 0.5f * seed, r1, y1, h1)) { ++accepted; if (rrt_f2u(r1) != rrt_f2u(want) || y1 != h1 + h1) { ++bad; counters[1] = b; counters[2] = rrt_f2u(seed); } }
                if (!sqrt_seeded_yh<2>(x, seed, 0.5f * seed, r2, y2, h2)) { ++accepted; if (rrt_f2u(r2) != rrt_f2u(want) || y2 != h2 + h2) { ++bad; counters[1] = b; counters[2] = rrt_f2u(seed); } }
                /* a seed of twice the reciprocal root (x*y0^2 = 4) converges to MINUS the root: it must be rejected */
                if (k == 0 && sgn > 0) {
                    if (!sqrt_seeded_yh<2>(x, 2.0f * y_exact, y_exact, r2, y2, h2) || !sqrt_seeded_yh<1>(x, 2.0f * y_exact, y_exact, r1, y1, h1)) { ++bad; counters[1] = b; counters[2] = 4; }
                }
#else
                if (sqrt_seeded<1>(x, seed, r1, y1)) { ++accepted; if (rrt_f2u(r1) != rrt_f2u(want)) { ++bad; counters[1] = b; counters[2] = rrt_f2u(seed); } }
                if (sqrt_seeded<2>(x, seed, r2, y2)) { ++accepted; if (rrt_f2u(r2) != rrt_f2u(want)) { ++bad; counters[1] = b; counters[2] = rrt_f2u(seed); } }
#endif
            }
        }
    }
    if (bad) atomicAdd(counters, (unsigned long long)bad);
    atomicAdd(counters + 3, accepted);
}

/* The seeded roots on the floats AROUND every power of two (x = 2^e (1 + j 2^-23), |j| <= span, e in [e_lo, e_hi)) under a DENSE
 * sweep of seeds: n_seeds estimates per x and form, spread evenly over +-tol1 (one iteration) / +-tol2 (two).  That is where
 * sqrt(x) comes closest to a rounding tie (x = 4^k (1 + 2^-23): 2^-26 ulp) and where round 4 found -- and guarded -- the one
 * class of accepted roots that were not correctly rounded.  counters: [0] / [1] mismatching accepted one- / two-iteration
 * roots, [2] accepted roots checked, [3] rejected ones, [4]/[5] one failing case (x bits, seed bits). */
__global__ void k_selfcheck_sqrt_boundaries(int e_lo, int e_hi, int span, unsigned n_seeds, float tol1, float tol2,
                                            unsigned long long* counters) {
    const uint64_t n_x = (uint64_t)(e_hi - e_lo) * (2 * span + 1);
    const uint64_t total = n_x * n_seeds;
    unsigned bad1 = 0, bad2 = 0;
    unsigned long long ok = 0, rej = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t xi = i / n_seeds;
        const unsigned si = (unsigned)(i - xi * n_seeds);
        const int e = e_lo + (int)(xi / (2 * span + 1)), j = (int)(xi % (2 * span + 1)) - span;
        const uint32_t bits = (uint32_t)((e + 127) << 23) + (uint32_t)j;          /* j < 0 reaches into the binade below */
        const float x = rrt_u2f(bits);
        float r_ref, y_ref;
        sqrt_rsq(x, r_ref, y_ref);
        const float want = sqrtf(x);
        const float u = ((float)si + 0.5f) * (2.0f / (float)n_seeds) - 1.0f;     /* (-1, 1) */
        float r, y, hy;
        const float s1 = y_ref * (1.0f + u * tol1), s2 = y_ref * (1.0f + u * tol2);
        if (!sqrt_seeded_yh<1>(x, s1, 0.5f * s1, r, y, hy)) {
            ++ok;
            if (rrt_f2u(r) != rrt_f2u(want)) { ++bad1; counters[4] = bits; counters[5] = rrt_f2u(s1); }
        } else ++rej;
        if (!sqrt_seeded_yh<2>(x, s2, 0.5f * s2, r, y, hy)) {
            ++ok;
            if (rrt_f2u(r) != rrt_f2u(want)) { ++bad2; counters[4] = bits; counters[5] = rrt_f2u(s2); }
        } else ++rej;
    }
    if (bad1) atomicAdd(counters, (unsigned long long)bad1);
    if (bad2) atomicAdd(counters + 1, (unsigned long long)bad2);
    atomicAdd(counters + 2, ok);
    atomicAdd(counters + 3, rej);
}

/* rrt_div_tame against IEEE `/` on `n` pseudo-random tame operand pairs: |b| in 2^[-40, 40), |a| in 2^[-20, 20) times
 * |b| (so |a/b| in 2^[-20, 20)), random signs, plus a == 0 every 64th case. */
__global__ void k_selfcheck_div_tame(unsigned long long n, uint32_t seed, unsigned long long* counters) {
    uint64_t idx = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    unsigned bad = 0;
    for (uint64_t k = idx; k < n; k += stride) {
        uint32_t h1 = mix32((uint32_t)k * 2654435761u + seed), h2 = mix32(h1 ^ (uint32_t)(k >> 32) ^ 0x9e3779b9u);
        uint32_t h3 = mix32(h2 + 0x85ebca6bu);
        float b = rrt_u2f(((87u << 23) + (h1 % (80u << 23))) | (h3 & 0x80000000u));          /* +-2^[-40, 40) */
        float ratio = rrt_u2f(((107u << 23) + (h2 % (40u << 23))) | ((h3 << 1) & 0x80000000u));  /* +-2^[-20, 20) */
        float a = (k & 63) == 0 ? 0.0f : b * ratio;
        float q = rrt_div_tame(a, b), w = a / b;
        if (rrt_f2u(q) != rrt_f2u(w)) { ++bad; counters[1] = rrt_f2u(a); counters[2] = rrt_f2u(b); }
    }
    if (bad) atomicAdd(counters, (unsigned long long)bad);
}

/* rrt_div_const against IEEE `/` for the constants the media code divides by (smoothstep edges of densities.h:74-77,
 * :124 and raymarcher.cu:97, the rim taper :27, ISCO_RADIUS, DISK_TEMP_REF): EVERY dividend whose bit pattern lies in
 * [lo, hi), both signs, plus +0.  counters[0] += mismatches, [1]/[2] one failing case (dividend bits, constant index). */
__global__ void k_selfcheck_div_const(uint32_t lo, uint32_t hi, unsigned long long* counters) {
    uint64_t idx = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    unsigned bad = 0;
    for (uint64_t b = (uint64_t)lo + idx; b < hi; b += stride) {
        for (int sgn = 0; sgn < 2; ++sgn) {
            const float a = rrt_u2f((uint32_t)b | (sgn ? 0x80000000u : 0u));
#define RRT_CHK(K, B) do { const float q = rrt_div_const(a, (B)), w = a / (B); \
                           if (rrt_f2u(q) != rrt_f2u(w)) { ++bad; counters[1] = rrt_f2u(a); counters[2] = (K); } } while (0)
            RRT_CHK(0, kDiskOut * 0.8f - kDiskOut);          /* -5 */
            RRT_CHK(1, (kIsco + 5.0f) - kIsco);               /* 5 */
            RRT_CHK(2, 0.8f - 0.4f);
            RRT_CHK(3, kDiskOut - kDiskOut * 0.85f);          /* 3.75 */
            RRT_CHK(4, kIsco);                                /* 10 */
            RRT_CHK(5, kDiskTempRef);                         /* 1.5e7 */
            RRT_CHK(6, 1.3f - 0.7f);
#undef RRT_CHK
        }
    }
    if (idx == 0) {
        /* +0 dividends (x - e0 with x == e0): exact, sign included.  A -0 dividend would come back as +0 for a positive
         * constant (the last fma adds +0 to it); none of the use sites can produce one -- every dividend is a
         * difference with a non-zero literal, or a radius / temperature >= 1. */
        const float z = 0.0f;
        if (rrt_f2u(rrt_div_const(z, kDiskOut * 0.8f - kDiskOut)) != rrt_f2u(z / (kDiskOut * 0.8f - kDiskOut))) ++bad;
        if (rrt_f2u(rrt_div_const(z, kIsco)) != rrt_f2u(z / kIsco)) ++bad;
        if (rrt_f2u(rrt_div_const(z, 0.8f - 0.4f)) != rrt_f2u(z / (0.8f - 0.4f))) ++bad;
    }
    if (bad) atomicAdd(counters, (unsigned long long)bad);
}

/* ------------------------------------------------------------------ host helpers */
/* an rrt_params built against another header is refused before any field of it is believed -- except ABI 4's 48-byte
 * layout, a strict prefix of this one: its caller simply has no nudge fields (they read as 0) */
constexpr uint32_t kParamsSizeAbi4 = 48;
static_assert(sizeof(rrt_params) == 56 && offsetof(rrt_params, nudge_ulps) == kParamsSizeAbi4, "rrt_params layout (ABI 5)");
int check_params_abi(const rrt_params* prm) {
    if (prm && prm->struct_size != (uint32_t)sizeof(rrt_params) && prm->struct_size != kParamsSizeAbi4) {
        snprintf(g_hip_err, sizeof(g_hip_err), "rrt_params.struct_size %u, this library's is %zu (ABI %d; ABI 4's %u is accepted too): recompile against include/rrt.h",
                 prm->struct_size, sizeof(rrt_params), RRT_ABI_VERSION, kParamsSizeAbi4);
        return RRT_ERR_ABI_MISMATCH;
    }
    return RRT_OK;
}
/* the caller's struct (NULL: config.h defaults) as this library's layout; check_params_abi() has passed */
void load_params(const rrt_params* in, rrt_params& out) {
    rrt_params_init(&out, (uint32_t)sizeof(out));
    if (in) memcpy(&out, in, in->struct_size < sizeof(out) ? in->struct_size : sizeof(out));
    out.struct_size = (uint32_t)sizeof(out);
}
int check_params_values(const rrt_params* prm_in) {
    rrt_params full;
    load_params(prm_in, full);
    const rrt_params* prm = &full;
    if (prm->max_steps < 0 || prm->sky_frac_bits < 0 || prm->sky_frac_bits > 16) return RRT_ERR_INVALID_ARGUMENT;
    if (!(prm->spin == prm->spin)) return RRT_ERR_INVALID_ARGUMENT;
    if (prm->arith_mode != RRT_ARITH_STRICT && prm->arith_mode != RRT_ARITH_FAST && prm->arith_mode != RRT_ARITH_FMAD) return RRT_ERR_INVALID_ARGUMENT;
    if (prm->workspace < 0) return RRT_ERR_INVALID_ARGUMENT;
    if (prm->path_policy < RRT_PATH_AUTO || prm->path_policy > RRT_PATH_THREE_PASS) return RRT_ERR_INVALID_ARGUMENT;
    if (prm->noise_table < 0 || prm->tile_order < 0) return RRT_ERR_INVALID_ARGUMENT;
    if (prm->pool_rounds < 0 || prm->pool_rounds > 64) return RRT_ERR_INVALID_ARGUMENT;
    if (prm->pass_chains < 0 || prm->pass_chains > kMaxChains) return RRT_ERR_INVALID_ARGUMENT;
    if (prm->nudge_ulps < 0 || prm->nudge_ulps > 4096) return RRT_ERR_INVALID_ARGUMENT;
    return RRT_OK;
}
int check_common(const void* out, int width, int height, const rrt_camera* cam, const rrt_effects* fx,
                 const rrt_params* prm) {
    if (!out || !cam || !fx || width <= 0 || height <= 0) return RRT_ERR_INVALID_ARGUMENT;
    if ((long long)width * height > (1ll << 31) - 1) return RRT_ERR_INVALID_ARGUMENT;
    if (height > kMaxGridY * kWGPixY) return RRT_ERR_INVALID_ARGUMENT;             /* 524 280 rows */
    if (prm) {
        const int rc = check_params_abi(prm);
        if (rc != RRT_OK) return rc;
        return check_params_values(prm);
    }
    return RRT_OK;
}

struct LaunchOpts { int media; int arith; int workspace, policy, pool_rounds, pass_chains; };

int fill_args(FrameArgs& a, LaunchOpts& o, void* out, int width, int height, float time, const rrt_camera* cam,
              rrt_sky_t sky, const rrt_effects* fx, const rrt_params* prm_in) {
    rrt_params prm;
    load_params(prm_in, prm);
    SkyObject so;
    if (!sky_lookup(sky, so) || !on_current_device(so.device)) return RRT_ERR_BAD_HANDLE;
    const SkyObject* s = &so;
    a.out = static_cast<uchar4*>(out);
    a.width = width; a.height = height; a.time = time; a.cam = *cam;
    a.sky.texels = s->d_texels; a.sky.w = s->w; a.sky.h = s->h; a.sky.frac_bits = prm.sky_frac_bits;
    a.use_bloom = fx->use_bloom != 0; a.use_vignette = fx->use_vignette != 0;
    a.use_ca = fx->use_chromatic_aberration != 0; a.use_lens = fx->use_lens_distortion != 0;
    a.bloom_threshold = fx->bloom_threshold; a.bloom_intensity = fx->bloom_intensity;
    a.vignette_intensity = fx->vignette_intensity; a.ca_amount = fx->ca_amount;
    a.distortion_amount = fx->distortion_amount;
    a.spin = prm.spin;
    a.drag_c = (2.0f * prm.spin) * 2.0f;            /* 2.0f * SPIN_A * EVENT_HORIZON, geodesics.h:41 */
    a.max_steps = prm.max_steps;
    a.nudge_ulps = prm.nudge_ulps; a.nudge_seed = prm.nudge_seed;
    memset(&a.dbg, 0, sizeof(a.dbg));
    a.tile_perm = nullptr; a.tile_cost = nullptr; a.tile_order_id = prm.tile_order;
    a.grid_rows = 0; a.grid_row_base = 0; a.grid_row_stride = 1;
    if (prm.tile_order != 0) {                      /* whatever path the launch takes: a stale or foreign id is an error */
        const std::shared_ptr<TileOrderObject> to = tile_order_lookup(prm.tile_order);
        if (!to || !on_current_device(to->device)) return RRT_ERR_BAD_HANDLE;
    }
    o.media = prm.volumetrics != 0 ? 1 : 0;
    memset(&a.lut_acc, 0, sizeof(a.lut_acc)); memset(&a.lut_dust, 0, sizeof(a.lut_dust));
    if (prm.noise_table != 0) {
        NoiseTableObject nt;
        {
            std::lock_guard<std::mutex> lk(g_nt_mu);
            auto it = g_nt.find(prm.noise_table);
            if (it == g_nt.end()) return RRT_ERR_BAD_HANDLE;
            nt = it->second;
        }
        if (!on_current_device(nt.device)) return RRT_ERR_BAD_HANDLE;
        /* the boxes were sized for t0 <= time <= t1; any other time runs the arithmetic kernels (same bytes) */
        if (o.media && time >= nt.t0 && time <= nt.t1) {
            o.media = 2;
            a.lut_acc = make_lut(nt.d_cells, nt.acc, nt.acc_families);
            a.lut_dust = make_lut(nt.d_cells + (size_t)nt.acc.nx * nt.acc.ny * nt.acc.nz, nt.dust, nt.dust_families);
        }
    }
    o.arith = prm.arith_mode;
    o.workspace = prm.workspace;
    o.policy = prm.path_policy;
    o.pool_rounds = prm.pool_rounds;
    o.pass_chains = prm.pass_chains;
    a.ctr = nullptr; a.hdr = nullptr; a.finals = nullptr; a.n_lanes = 0; a.sample_blocks = nullptr; a.block_capacity = 0;
    return RRT_OK;
}

/* Largest max_steps the three-pass bookkeeping can represent: the step count shares a word with the hit /
 * resume flags (30 bits), and pass 3 walks at most kMaxRunsWalked runs of a wave, whose lengths double up to
 * kMaxRun blocks of kBlockRows rows -- a wave can pool one row per step, so it must not need more runs than
 * that.  Launches with more steps take the single kernel (same bytes). */
constexpr long long kThreePassMaxSteps = (long long)(kMaxRunsWalked - 8) * kMaxRun * kBlockRows;   /* ~262 k */
static_assert(kThreePassMaxSteps < (1ll << 30), "step count must fit beside the two flag bits");
constexpr int kFinalPlanes = 11;      /* per ray: vel xyz, code, pos xyz, radiance rgbt */
constexpr int kMaxPoolRounds = 64;

/* How many rounds to enqueue (rrt_params.pool_rounds == 0).  The host cannot ask the device without stalling the
 * stream, so it reads the statistics the workspace's PREVIOUS launch left in pinned host memory (an asynchronous copy
 * behind its last kernel; possibly a frame stale, which is all an animation needs): as many rounds as that launch had
 * work for PLUS ONE spare (a view that changes finds room; an idle round costs three near-empty launches, ~20 us: every
 * wave leaves on one scalar load), twice as many plus one if rays were still suspended at its end.  Rays the enqueued
 * rounds do not finish take the in-line route: the bytes never depend on the guess. */
int auto_pool_rounds(const volatile DeferCounters* h, unsigned capacity) {
    if (!h) return 2;
    const unsigned run = h->rounds_run, work = h->rounds_with_work, left = h->suspended_left, peak = h->peak_blocks;
    if (run == 0u) return 2;                                    /* no history */
    (void)peak; (void)capacity;
    int r = (int)(work > 0u ? work : 1u);
    r = left != 0u ? 2 * r + 1 : r + 1;
    return r > kMaxPoolRounds ? kMaxPoolRounds : r;
}

/* one chain = march -> evaluate -> composite (in rounds) over dispatch rows [row0, row1) of the launch, in its own slice
 * of the pool, on its own stream */
/* the kernels of one arithmetic mode, instantiated per (spin, tables) */
template <int ARITH>
int enqueue_chain_arith(const FrameArgs& a, bool lut, dim3 grid, dim3 block, int rounds, hipStream_t st) {
    const bool spin = a.spin != 0.0f;
    for (int r = 0; r < rounds; ++r) {
        const bool last = r == rounds - 1;
#define RRT_MARCH(S) do { if (r == 0) hipLaunchKernelGGL((march_defer<S, ARITH, false>), grid, block, 0, st, a); \
                          else hipLaunchKernelGGL((march_defer<S, ARITH, true>), grid, block, 0, st, a); } while (0)
        if (spin) RRT_MARCH(true); else RRT_MARCH(false);
#undef RRT_MARCH
        RRT_HIP(hipGetLastError());
        if (lut) hipLaunchKernelGGL((eval_sample_rows<ARITH, true>), dim3(2048), dim3(256), 0, st, a);
        else hipLaunchKernelGGL((eval_sample_rows<ARITH, false>), dim3(2048), dim3(256), 0, st, a);
        RRT_HIP(hipGetLastError());
#define RRT_COMP3(S, L) do { if (last) hipLaunchKernelGGL((composite_and_shade<S, ARITH, L, true>), grid, block, 0, st, a); \
                             else hipLaunchKernelGGL((composite_and_shade<S, ARITH, L, false>), grid, block, 0, st, a); } while (0)
#define RRT_COMP(S) do { if (lut) RRT_COMP3(S, true); else RRT_COMP3(S, false); } while (0)
        if (spin) RRT_COMP(true); else RRT_COMP(false);
#undef RRT_COMP
#undef RRT_COMP3
        RRT_HIP(hipGetLastError());
        hipLaunchKernelGGL(pool_next_round, dim3(1), dim3(1), 0, st, a.ctr, a.block_capacity, last ? 1 : 0);
        RRT_HIP(hipGetLastError());
    }
    return RRT_OK;
}

/* one chain = march -> evaluate -> composite (in rounds) over dispatch rows [row0, row1) of the launch, in its own slice
 * of the pool, on its own stream */
int enqueue_chain(FrameArgs a, int arith, bool lut, dim3 full_grid, int row0, int row_stride, int n_rows, int rounds, hipStream_t st) {
    if (n_rows <= 0) return RRT_OK;
    const dim3 block(kWGThreads), grid(full_grid.x, (unsigned)n_rows);
    a.grid_rows = (int)full_grid.y; a.grid_row_base = row0; a.grid_row_stride = row_stride;
    if (arith == kArithFast) return enqueue_chain_arith<kArithFast>(a, lut, grid, block, rounds, st);
    if (arith == kArithFmad) return enqueue_chain_arith<kArithFmad>(a, lut, grid, block, rounds, st);
    return enqueue_chain_arith<kArithStrict>(a, lut, grid, block, rounds, st);
}

/* Three-pass launch through a workspace.  Returns RRT_OK after enqueuing, or -1 if the workspace cannot
 * hold this launch's bookkeeping plus a useful pool (the caller then uses the single-kernel path).
 *
 * TWO CHAINS (round 4).  A rank's share of a frame is a few rounds of wavefronts, and each of the three kernels ends in a
 * tail: a lone wavefront retires a dependent instruction every ~10 clocks, the SIMDs are only busy with 5-6 of them, so the
 * last-started waves of the march take a full 1.5 ms whatever else has finished, and the chip decays from full to empty
 * over that time (profiles/r04_wave_timeline_default.txt: 1.1 of the 5.4 ms of an eighth of the 4K bench frame).  The
 * launch is therefore cut in two along its dispatch order -- the first half of the dispatch rows (the frame's middle, or the
 * costliest tiles under rrt_tile_order) and the second -- and each half runs its own march -> evaluate -> composite chain
 * on its own stream with its own slice of the pool: the evaluation and compositing of one half fill the march tail of the
 * other.  Same kernels, same arithmetic, same bytes; an eighth of the bench frame 5.7 -> 5.2 ms, of the view from inside
 * the disk 12.4 -> 9.5 ms (profiles/r04_split_chain_probe.txt).  The side stream and its events belong to the workspace;
 * a launch that is being captured into a graph, or a small one, runs one chain. */
int launch_deferred(FrameArgs a, int arith, bool lut, const WorkspaceObject& ws, int pool_rounds, int chains_wanted, hipStream_t st) {
    dim3 grid((a.width + kWGPixX - 1) / kWGPixX, (a.rows.n_local_rows + kWGPixY - 1) / kWGPixY);
    const size_t n_waves = (size_t)grid.x * grid.y * kWGWaves;
    const size_t n_lanes = n_waves * 64;
    auto align = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const size_t off_hdr = 256;
    static_assert(kMaxChains * kCounterStride <= 256, "the chains' counters sit in front of the wave headers");
    const size_t off_fin = align(off_hdr + n_waves * sizeof(WaveHdr));
    const size_t off_rows = align(off_fin + n_lanes * 4 * kFinalPlanes);
    if (ws.bytes < off_rows + (size_t)1024 * kBlockBytes) return -1;
    size_t cap = (ws.bytes - off_rows) / kBlockBytes;
    if (cap > 0x0fffffffu) cap = 0x0fffffffu;
    a.hdr = reinterpret_cast<WaveHdr*>(ws.d_base + off_hdr);
    a.finals = reinterpret_cast<float*>(ws.d_base + off_fin);
    a.n_lanes = n_lanes;
    /* one chain or two */
    int chains = 1;
    if (chains_wanted != 1 && ws.side != nullptr && grid.y >= 8 && cap >= 4096 && n_waves >= (chains_wanted == 2 ? 2u : 2048u)) {
        hipStreamCaptureStatus capst = hipStreamCaptureStatusNone;
        if (st != nullptr && hipStreamIsCapturing(st, &capst) != hipSuccess) { (void)hipGetLastError(); capst = hipStreamCaptureStatusNone; }
        if (capst == hipStreamCaptureStatusNone) chains = 2;
    }
    hipLaunchKernelGGL(zero_words, dim3((unsigned)((off_fin / 16 + 255) / 256)), dim3(256), 0, st,
                       reinterpret_cast<uint4*>(ws.d_base), off_fin / 16);        /* counters + wave headers (off_fin is a multiple of 256) */
    RRT_HIP(hipGetLastError());
    /* the pool's split: by what each chain pooled last time (the heavy half holds most of the media), 65 : 35 without history */
    size_t cap_of[kMaxChains] = {cap, 0};
    /* which dispatch rows a chain takes: the first and the second half of the static order (the frame's middle and the rest) --
     * or, when the order is by COST (rrt_tile_order), every other row each: the halves of a cost-sorted order are "all the
     * expensive tiles" and "all the cheap ones", which made the two chains worse than one (5.15 -> 5.45 ms on an
     * eighth of the bench frame), while even and odd rows of it carry the same cost */
    const bool deal_rows = chains == 2 && a.tile_perm != nullptr;
    int row_first[kMaxChains] = {0, deal_rows ? 1 : (int)grid.y / 2};
    int row_step = deal_rows ? 2 : 1;
    int row_count[kMaxChains] = {(int)grid.y, 0};
    if (chains == 2) {
        row_count[0] = deal_rows ? ((int)grid.y + 1) / 2 : (int)grid.y / 2;
        row_count[1] = (int)grid.y - row_count[0];
    }
    if (chains == 2) {
        double share = deal_rows ? 0.5 : 0.65;
        const volatile DeferCounters* h = ws.h_stats;
        if (h && h[0].rounds_run != 0u && h[1].rounds_run != 0u) {
            const double t0 = (double)h[0].total_blocks, t1 = (double)h[1].total_blocks;
            if (t0 + t1 > 0.0) share = t0 / (t0 + t1);
        }
        share = share < 0.3 ? 0.3 : (share > 0.85 ? 0.85 : share);
        cap_of[0] = (size_t)((double)cap * share);
        cap_of[1] = cap - cap_of[0];
    }
    if (chains == 2) {            /* fork: the side stream starts behind the memset, BEFORE anything of chain 0 is in the caller's stream */
        RRT_HIP(hipEventRecord(ws.forked, st));
        RRT_HIP(hipStreamWaitEvent(ws.side, ws.forked, 0));
    }
    size_t block0 = 0;
    for (int c = 0; c < chains; ++c) {
        FrameArgs b = a;
        b.ctr = reinterpret_cast<DeferCounters*>(ws.d_base + (size_t)c * kCounterStride);
        b.sample_blocks = ws.d_base + off_rows + block0 * kBlockBytes;
        b.block_capacity = (unsigned)cap_of[c];
        const int rounds = pool_rounds > 0 ? pool_rounds : auto_pool_rounds(ws.h_stats ? ws.h_stats + c : nullptr, b.block_capacity);
        const hipStream_t cs = c == 1 ? ws.side : st;
        const int rc = enqueue_chain(b, arith, lut, grid, row_first[c], row_step, row_count[c], rounds, cs);
        if (rc != RRT_OK) return rc;
        /* what this chain needed, for the next launch's round count and pool split (and rrt_workspace_stats) */
        if (ws.h_stats) RRT_HIP(hipMemcpyAsync(ws.h_stats + c, b.ctr, sizeof(DeferCounters), hipMemcpyDeviceToHost, cs));
        if (c == 1) {                                          /* join */
            RRT_HIP(hipEventRecord(ws.joined, ws.side));
            RRT_HIP(hipStreamWaitEvent(st, ws.joined, 0));
        }
        block0 += cap_of[c];
    }
    if (chains == 1 && ws.h_stats) memset(const_cast<DeferCounters*>(ws.h_stats) + 1, 0, sizeof(DeferCounters));   /* no second chain this time */
    return RRT_OK;
}

/* When does the three-pass path pay?  It does ~8 % more work than the single kernel, but its longest
 * wavefront is a bare march (<= ~1.2 ms) instead of a march with every media sample in line (up to ~13 ms
 * on the bench view).  A launch only feels that pole once its own duration gets close to it, i.e. for
 * small launches -- one GPU's share of a frame that is spread over many GPUs.  Measured on the 4K bench
 * frame (profiles/README.md): 1/4 of the frame (2.07 M rays) 11.8 ms in line vs 12.6 ms three-pass;
 * 1/8 (1.04 M rays) 11.7 ms vs 6.8 ms. */
constexpr long long kThreePassMaxRays = 1500000;

/* room for n tiles in a tile-order object (grows only; growing forgets the order) */
int tile_order_reserve(TileOrderObject& o, size_t n) {
    if (n <= o.n_cap) return RRT_OK;
    if (n > ((size_t)1 << 30)) return RRT_ERR_INVALID_ARGUMENT;
    RRT_HIP(hipDeviceSynchronize());                 /* launches through the object may still read the old buffers */
    unsigned** bufs[] = {&o.d_cost, &o.d_sorted, &o.d_iota, &o.d_perm[0], &o.d_perm[1]};
    for (unsigned** b : bufs) { if (*b) (void)hipFree(*b); *b = nullptr; }
    if (o.d_temp) (void)hipFree(o.d_temp);
    o.d_temp = nullptr; o.n_cap = 0; o.have = false;
    const size_t cap = n + n / 8;
    for (unsigned** b : bufs) {
        hipError_t e = hipMalloc(reinterpret_cast<void**>(b), cap * sizeof(unsigned));
        if (e != hipSuccess) return hip_fail(e, "hipMalloc(tile order)");
    }
    const size_t tb = (size_t)2 * 256 * rrt_sort::kMaxBlocks * sizeof(unsigned);      /* counters for any n (2 MB) */
    hipError_t e = hipMalloc(&o.d_temp, tb);
    if (e != hipSuccess) return hip_fail(e, "hipMalloc(tile order sort)");
    o.temp_bytes = tb;
    o.n_cap = cap;
    return RRT_OK;
}

/* enqueue the coarse probe of the launch `a` describes (its row map included) into `cells` */
int enqueue_probe(const FrameArgs& a, ProbeArgs& q, unsigned* cells, int stride_x, int stride_y, bool media, hipStream_t st) {
    q.cell_cost = cells;
    q.stride_x = stride_x; q.stride_y = stride_y;
    q.cells_x = (a.width + stride_x - 1) / stride_x;
    q.cells_y = (a.rows.n_local_rows + stride_y - 1) / stride_y;
    q.w_step = RRT_PROBE_W_STEP; q.w_acc = RRT_PROBE_W_ACC; q.w_dust = RRT_PROBE_W_DUST;
    q.ring_steps = (float)a.max_steps;
    if (!media) { q.w_acc = 0.0f; q.w_dust = 0.0f; }          /* volumetrics off: a sample costs nothing, the steps stay */
    if (const char* e = getenv("RRT_PROBE_WEIGHTS")) {          /* dev: tools/probe_fit.py reads the three counts one at a time */
        float w0, w1, w2;
        if (sscanf(e, "%f,%f,%f", &w0, &w1, &w2) == 3) { q.w_step = w0; q.w_acc = w1; q.w_dust = w2; }
    }
    const dim3 grid((q.cells_x + 7) / 8, (q.cells_y + 7) / 8);
    if (a.spin != 0.0f) hipLaunchKernelGGL((probe_costs<true>), grid, dim3(64), 0, st, a, q);
    else hipLaunchKernelGGL((probe_costs<false>), grid, dim3(64), 0, st, a, q);
    RRT_HIP(hipGetLastError());
    return RRT_OK;
}

int launch(const FrameArgs& a, const LaunchOpts& o, bool debug, hipStream_t st) {
    dim3 block(kWGThreads);
    if (a.rows.n_local_rows == 0) return RRT_OK;
    const bool spin = a.spin != 0.0f;
    dim3 grid((a.width + kWGPixX - 1) / kWGPixX, (a.rows.n_local_rows + kWGPixY - 1) / kWGPixY);
    /* which path */
    WorkspaceObject ws{};
    bool deferred = false;
    if (o.workspace != 0) {
        {
            std::lock_guard<std::mutex> lk(g_ws_mu);
            auto it = g_ws.find(o.workspace);
            if (it == g_ws.end()) return RRT_ERR_BAD_HANDLE;
            ws = it->second;
        }
        if (!on_current_device(ws.device)) return RRT_ERR_BAD_HANDLE;
        const long long rays = (long long)a.width * a.rows.n_local_rows;
        const bool want = o.policy == RRT_PATH_THREE_PASS || (o.policy == RRT_PATH_AUTO && rays <= kThreePassMaxRays);
        deferred = o.media != 0 && !debug && want && a.max_steps <= kThreePassMaxSteps;
    }
    FrameArgs b = a;
    /* cost-ordered dispatch: read the order the previous launch through the object left (same geometry), or -- no history --
     * the order a coarse probe of this very view gives; record this launch's costs; sort them into the other buffer for the
     * next one.  A launch that is being captured into a graph renders in the static order and leaves the object alone: a
     * replayed graph must never read a permutation that a later live launch is rewriting. */
    std::shared_ptr<TileOrderObject> order;
    std::unique_lock<std::mutex> order_lock;
    const size_t n_tiles = (size_t)grid.x * grid.y;
    if (a.tile_order_id != 0 && !debug && kWGWaves == 1) {
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        if (st != nullptr && hipStreamIsCapturing(st, &cap) != hipSuccess) { (void)hipGetLastError(); cap = hipStreamCaptureStatusNone; }
        if (cap == hipStreamCaptureStatusNone) {
            order = tile_order_lookup(a.tile_order_id);
            if (!order || !on_current_device(order->device)) return RRT_ERR_BAD_HANDLE;
            order_lock = std::unique_lock<std::mutex>(order->mu);
            if (order->dead) return RRT_ERR_BAD_HANDLE;            /* destroyed while this thread waited for it */
            int rc = tile_order_reserve(*order, n_tiles);
            if (rc != RRT_OK) return rc;
            if (order->launches > 0) RRT_HIP(hipStreamWaitEvent(st, order->chained, 0));
            const bool same = order->have && order->grid_x == grid.x && order->grid_y == grid.y && order->width == a.width &&
                              order->height == a.height && same_row_map(order->rows, a.rows);
            if (!same && !order->no_seed) {
                /* first launch of this geometry: probe -> per-tile estimate -> order */
                ProbeArgs q;
                rc = enqueue_probe(a, q, order->d_sorted, kProbeStride, kProbeStride, o.media != 0, st);   /* d_sorted: scratch until the sort */
                if (rc != RRT_OK) return rc;
                hipLaunchKernelGGL(probe_to_tiles, dim3((unsigned)((n_tiles + 255) / 256)), dim3(256), 0, st, order->d_cost, grid.x, grid.y, q);
                RRT_HIP(hipGetLastError());
                const int next = order->cur ^ 1;
                RRT_HIP(rrt_sort::enqueue(order->d_cost, order->d_sorted, order->d_iota, static_cast<unsigned*>(order->d_temp),
                                          order->d_perm[next], n_tiles, grid.x, grid.y, kTileCostSortLo, st));
                order->cur = next;
                ++order->seeded;
            }
            const bool ordered = same || !order->no_seed;
            b.tile_perm = ordered ? order->d_perm[order->cur] : nullptr;
            b.tile_cost = order->d_cost;
            /* the one hipMemsetAsync on a launch path.  Safe where the workspace header's was not (zero_words, above): this
             * branch is only entered when `st` is NOT being captured -- a captured launch ignores the tile-order object -- so the
             * memset is never turned into a graph node that a replay would have to re-execute */
            RRT_HIP(hipMemsetAsync(order->d_cost, 0, n_tiles * sizeof(unsigned), st));
            if (same) ++order->ordered;
        }
    }
    bool launched = false;
    if (deferred) {
        const int rc = launch_deferred(b, o.arith, o.media == 2, ws, o.pool_rounds, o.pass_chains, st);
        if (rc > 0) return rc;
        launched = rc == RRT_OK;
    }
    if (!launched) {
        const int media = o.media;
        const int arith = o.arith;
#define RRT_LAUNCH4(S, M, D, F) hipLaunchKernelGGL((raymarch_pixels<S, M, D, F>), grid, block, 0, st, b)
#define RRT_LAUNCH3(S, M, D) do { if (arith == kArithFast) RRT_LAUNCH4(S, M, D, kArithFast); else if (arith == kArithFmad) RRT_LAUNCH4(S, M, D, kArithFmad); \
                                  else RRT_LAUNCH4(S, M, D, kArithStrict); } while (0)
#define RRT_LAUNCH2(S, M) do { if (debug) RRT_LAUNCH3(S, M, true); else RRT_LAUNCH3(S, M, false); } while (0)
#define RRT_LAUNCH1(S) do { if (media == 2) RRT_LAUNCH2(S, 2); else if (media == 1) RRT_LAUNCH2(S, 1); else RRT_LAUNCH2(S, 0); } while (0)
        if (spin) RRT_LAUNCH1(true); else RRT_LAUNCH1(false);
#undef RRT_LAUNCH1
#undef RRT_LAUNCH2
#undef RRT_LAUNCH3
#undef RRT_LAUNCH4
        RRT_HIP(hipGetLastError());
    }
    if (order) {
        const int next = order->cur ^ 1;
        RRT_HIP(rrt_sort::enqueue(order->d_cost, order->d_sorted, order->d_iota, static_cast<unsigned*>(order->d_temp),
                                  order->d_perm[next], n_tiles, grid.x, grid.y, kTileCostSortLo, st));
        RRT_HIP(hipEventRecord(order->chained, st));
        order->cur = next; order->have = true; ++order->launches;
        order->grid_x = grid.x; order->grid_y = grid.y; order->width = a.width; order->height = a.height; order->rows = a.rows;
    }
    return RRT_OK;
}

int shard_rows(int height, int tile_rows, int shard, int n_shards) {
    int n_tiles = (height + tile_rows - 1) / tile_rows;
    int rows = 0;
    for (int t = shard; t < n_tiles; t += n_shards) rows += (t * tile_rows + tile_rows <= height) ? tile_rows : (height - t * tile_rows);
    return rows;
}

template <class F>
int unit_launch(int n, void* stream, F f) {
    if (n < 0) return RRT_ERR_INVALID_ARGUMENT;
    if (n == 0) return RRT_OK;
    f(dim3((n + 255) / 256), dim3(256), static_cast<hipStream_t>(stream));
    RRT_HIP(hipGetLastError());
    return RRT_OK;
}

}  // namespace

/* ====================================================================== C ABI */
extern "C" {

int rrt_abi_version(void) { return RRT_ABI_VERSION; }

const char* rrt_status_string(int s) {
    switch (s) {
        case RRT_OK: return "ok";
        case RRT_ERR_INVALID_ARGUMENT: return "invalid argument";
        case RRT_ERR_NO_DEVICE: return "no HIP device";
        case RRT_ERR_HIP: return "HIP runtime error";
        case RRT_ERR_BAD_HANDLE: return "bad handle (sky, workspace, noise table, tile order or tile map)";
        case RRT_ERR_OUT_OF_MEMORY: return "out of memory";
        case RRT_ERR_ABI_MISMATCH: return "rrt_params from another ABI version (recompile against include/rrt.h)";
        default: return "unknown status";
    }
}

const char* rrt_last_hip_error(void) { return g_hip_err; }

int rrt_device_count(int* count) {
    if (!count) return RRT_ERR_INVALID_ARGUMENT;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { *count = 0; (void)hipGetLastError(); return RRT_ERR_NO_DEVICE; }
    *count = n;
    return n > 0 ? RRT_OK : RRT_ERR_NO_DEVICE;
}

int rrt_params_init(void* p, uint32_t size) {
    if (!p) return RRT_ERR_INVALID_ARGUMENT;
    if (size != (uint32_t)sizeof(rrt_params) && size != kParamsSizeAbi4) return RRT_ERR_ABI_MISMATCH;
    rrt_params d;
    memset(&d, 0, sizeof(d));
    d.spin = 0.0f;          /* SPIN_A    config.h:21 */
    d.max_steps = 2000;     /* MAX_STEPS config.h:48 */
    d.volumetrics = 1;
    d.sky_frac_bits = 8;
    d.struct_size = size;
    memcpy(p, &d, size);
    return RRT_OK;
}

/* what binaries built against the ABI 4 header call: that header's struct is the first 48 bytes of today's, and such a
 * struct is still accepted by every entry point (check_params_abi) */
int rrt_params_default_v4(void* abi4_48) { return rrt_params_init(abi4_48, kParamsSizeAbi4); }

/* The symbol binaries built against the ABI <= 3 header call: it fills the 36 bytes THEIR struct has -- spin, max_steps,
 * volumetrics, sky_frac_bits, arith_mode, workspace, path_policy, noise_table, tile_order -- and not a byte more, so such
 * a binary is refused at its first launch (RRT_ERR_ABI_MISMATCH: its first word is `spin`, not a struct size) instead of
 * having its stack overwritten here. */
#pragma push_macro("rrt_params_default")
#undef rrt_params_default
int rrt_params_default(void* legacy36) {
    if (!legacy36) return RRT_ERR_INVALID_ARGUMENT;
    int32_t w[9] = {0, 2000, 1, 8, 0, 0, 0, 0, 0};
    memcpy(legacy36, w, sizeof(w));
    return RRT_OK;
}
#pragma pop_macro("rrt_params_default")

int rrt_effects_default(rrt_effects* e) {    /* camera_settings.h:5-16 */
    if (!e) return RRT_ERR_INVALID_ARGUMENT;
    memset(e, 0, sizeof(*e));
    e->use_bloom = 1; e->bloom_threshold = 0.8f; e->bloom_intensity = 0.5f;
    e->use_vignette = 1; e->vignette_intensity = 0.4f;
    e->use_chromatic_aberration = 0; e->ca_amount = 0.005f;
    e->use_lens_distortion = 1; e->distortion_amount = 0.15f;
    return RRT_OK;
}

int rrt_sky_create(const uint8_t* rgba8_host, int width, int height, rrt_sky_t* out) {
    if (!rgba8_host || !out || width <= 0 || height <= 0) return RRT_ERR_INVALID_ARGUMENT;
    SkyObject s{nullptr, width, height, true, current_device()};
    size_t bytes = (size_t)width * height * 4;
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&s.d_texels), bytes);
    if (e != hipSuccess) return hip_fail(e, "hipMalloc(sky)");
    e = hipMemcpy(s.d_texels, rgba8_host, bytes, hipMemcpyHostToDevice);
    if (e != hipSuccess) { (void)hipFree(s.d_texels); return hip_fail(e, "hipMemcpy(sky)"); }
    *out = sky_register(s);
    return RRT_OK;
}

int rrt_sky_create_from_device(const void* d_rgba8, int width, int height, rrt_sky_t* out) {
    if (!d_rgba8 || !out || width <= 0 || height <= 0) return RRT_ERR_INVALID_ARGUMENT;
    /* the device that owns the caller's allocation, if the runtime can tell; else the current one */
    int device = current_device();
    hipPointerAttribute_t attr;
    if (g_fake_device.load() < 0 && hipPointerGetAttributes(&attr, d_rgba8) == hipSuccess) device = attr.device;
    else (void)hipGetLastError();
    SkyObject s{const_cast<uint8_t*>(static_cast<const uint8_t*>(d_rgba8)), width, height, false, device};
    *out = sky_register(s);
    return RRT_OK;
}

int rrt_sky_destroy(rrt_sky_t sky) {
    SkyObject s;
    {
        std::lock_guard<std::mutex> lk(g_sky_mu);
        auto it = g_sky.find(sky);
        if (it == g_sky.end()) return RRT_ERR_BAD_HANDLE;
        s = it->second;
        g_sky.erase(it);
    }
    if (s.owned) {
        hipError_t e = hipFree(s.d_texels);
        if (e != hipSuccess) return hip_fail(e, "hipFree(sky)");
    }
    return RRT_OK;
}

int rrt_workspace_create(size_t bytes, int* out) {
    if (!out || bytes < (size_t)1 << 20) return RRT_ERR_INVALID_ARGUMENT;
    WorkspaceObject w{nullptr, bytes, current_device(), nullptr, nullptr, nullptr, nullptr};
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&w.d_base), bytes);
    if (e != hipSuccess) return hip_fail(e, "hipMalloc(workspace)");
    e = hipHostMalloc(reinterpret_cast<void**>(&w.h_stats), kMaxChains * sizeof(DeferCounters), hipHostMallocDefault);
    if (e != hipSuccess) { (void)hipFree(w.d_base); return hip_fail(e, "hipHostMalloc(workspace statistics)"); }
    memset(w.h_stats, 0, kMaxChains * sizeof(DeferCounters));
    /* the second chain's stream and the fork / join events; without them launches run one chain.  Default priority on
     * purpose: a higher-priority stream is served STRICTLY first on this hardware -- the other queue did not start until the
     * whole high-priority chain had finished (kernel trace, profiles/README.md round 4): serial again, and slower */
    if (hipStreamCreateWithFlags(&w.side, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&w.forked, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&w.joined, hipEventDisableTiming) != hipSuccess) {
        (void)hipGetLastError();
        if (w.side) (void)hipStreamDestroy(w.side);
        if (w.forked) (void)hipEventDestroy(w.forked);
        w.side = nullptr; w.forked = nullptr; w.joined = nullptr;
    }
    std::lock_guard<std::mutex> lk(g_ws_mu);
    *out = g_ws_next++;
    g_ws.emplace(*out, w);
    return RRT_OK;
}

int rrt_tile_order_create(int* out) {
    if (!out) return RRT_ERR_INVALID_ARGUMENT;
    auto o = std::make_shared<TileOrderObject>();
    o->device = current_device();
    o->d_cost = o->d_sorted = o->d_iota = o->d_perm[0] = o->d_perm[1] = nullptr;
    o->d_temp = nullptr; o->temp_bytes = 0; o->n_cap = 0; o->cur = 0; o->have = false;
    o->grid_x = o->grid_y = 0; o->width = o->height = 0; o->rows = RowMap{0, 0, 1, 0, 1, nullptr};
    o->launches = o->ordered = o->seeded = 0; o->no_seed = false; o->dead = false;
    RRT_HIP(hipEventCreateWithFlags(&o->chained, hipEventDisableTiming));
    std::lock_guard<std::mutex> lk(g_to_mu);
    *out = g_to_next++;
    g_to.emplace(*out, o);
    return RRT_OK;
}

int rrt_tile_order_set_seeding(int id, int on) {
    const std::shared_ptr<TileOrderObject> o = tile_order_lookup(id);
    if (!o) return RRT_ERR_BAD_HANDLE;
    std::lock_guard<std::mutex> lk(o->mu);
    o->no_seed = on == 0;
    return RRT_OK;
}

int rrt_tile_order_destroy(int id) {
    std::shared_ptr<TileOrderObject> o;
    {
        std::lock_guard<std::mutex> lk(g_to_mu);
        auto it = g_to.find(id);
        if (it == g_to.end()) return RRT_ERR_BAD_HANDLE;
        if (!on_current_device(it->second->device)) return RRT_ERR_BAD_HANDLE;
        o = it->second;
        g_to.erase(it);
    }
    std::lock_guard<std::mutex> lk(o->mu);                          /* after any launch that holds the object */
    o->dead = true;
    if (o->launches > 0) (void)hipEventSynchronize(o->chained);     /* its last launch and sort have finished */
    unsigned* bufs[] = {o->d_cost, o->d_sorted, o->d_iota, o->d_perm[0], o->d_perm[1]};
    for (unsigned* b : bufs) if (b) (void)hipFree(b);
    if (o->d_temp) (void)hipFree(o->d_temp);
    (void)hipEventDestroy(o->chained);
    return RRT_OK;
}

/* counters of an object, and (optionally, after waiting for its last launch) the order the NEXT matching launch will
 * use plus the costs the last one recorded: perm_host / cost_host may be NULL, capacity counts elements */
int rrt_tile_order_info(int id, unsigned long long* launches, unsigned long long* ordered_launches, unsigned* n_tiles,
                        unsigned* perm_host, unsigned* cost_host, unsigned capacity) {
    const std::shared_ptr<TileOrderObject> op = tile_order_lookup(id);
    if (!op) return RRT_ERR_BAD_HANDLE;
    std::lock_guard<std::mutex> lk(op->mu);
    const TileOrderObject& o = *op;
    if (!on_current_device(o.device)) return RRT_ERR_BAD_HANDLE;
    const unsigned n = o.have ? o.grid_x * o.grid_y : 0u;
    if (launches) *launches = o.launches;
    if (ordered_launches) *ordered_launches = o.ordered;
    if (n_tiles) *n_tiles = n;
    if ((perm_host || cost_host) && n > 0) {
        if (capacity < n) return RRT_ERR_INVALID_ARGUMENT;
        RRT_HIP(hipEventSynchronize(o.chained));
        if (perm_host) RRT_HIP(hipMemcpy(perm_host, o.d_perm[o.cur], n * sizeof(unsigned), hipMemcpyDeviceToHost));
        if (cost_host) RRT_HIP(hipMemcpy(cost_host, o.d_cost, n * sizeof(unsigned), hipMemcpyDeviceToHost));
    }
    return RRT_OK;
}

/* launches whose order came from the coarse probe (no history for their geometry) */
int rrt_tile_order_seeded(int id, unsigned long long* seeded_launches) {
    const std::shared_ptr<TileOrderObject> o = tile_order_lookup(id);
    if (!o || !seeded_launches) return o ? RRT_ERR_INVALID_ARGUMENT : RRT_ERR_BAD_HANDLE;
    std::lock_guard<std::mutex> lk(o->mu);
    *seeded_launches = o->seeded;
    return RRT_OK;
}

/* Parameters of the reference-signature entry point launch_raymarch() (include/raymarcher.h), which has no
 * parameter for them: config.h defaults until the application says otherwise.  Nothing is allocated here --
 * a workspace or noise table named in the defaults is created (and destroyed) by the caller. */
std::mutex g_defaults_mu;
rrt_params g_defaults;
bool g_defaults_set = false;

int rrt_set_launch_defaults(const rrt_params* prm) {
    std::lock_guard<std::mutex> lk(g_defaults_mu);
    if (!prm) { g_defaults_set = false; return RRT_OK; }
    int rc = check_params_abi(prm);
    if (rc == RRT_OK) rc = check_params_values(prm);
    if (rc != RRT_OK) return rc;
    load_params(prm, g_defaults);
    g_defaults_set = true;
    return RRT_OK;
}

int rrt_get_launch_defaults_sized(void* out, uint32_t size) {
    if (!out) return RRT_ERR_INVALID_ARGUMENT;
    if (size != (uint32_t)sizeof(rrt_params) && size != kParamsSizeAbi4) return RRT_ERR_ABI_MISMATCH;
    std::lock_guard<std::mutex> lk(g_defaults_mu);
    if (!g_defaults_set) return rrt_params_init(out, size);
    memcpy(out, &g_defaults, size);
    static_cast<rrt_params*>(out)->struct_size = size;
    return RRT_OK;
}
/* the export ABI 4 binaries call: their struct has 48 bytes */
#pragma push_macro("rrt_get_launch_defaults")
#undef rrt_get_launch_defaults
int rrt_get_launch_defaults(void* abi4_48) { return rrt_get_launch_defaults_sized(abi4_48, kParamsSizeAbi4); }
#pragma pop_macro("rrt_get_launch_defaults")

/* launch_raymarch() as the reference spells it, minus the C++ types: cam12 = pos, forward, right, up;
 * effects36 = the 36 bytes of struct CameraEffects.  Asynchronous on the null stream.  The reference's
 * launcher reports nothing (src/raymarcher.cu:176-180); this one returns the status and, the first time a
 * launch fails, says why on stderr. */
int rrt_launch_raymarch_compat(void* d_out_rgba8, int width, int height, float time, const float* cam12,
                               rrt_sky_t sky, const void* effects36) {
    if (!cam12 || !effects36) return RRT_ERR_INVALID_ARGUMENT;
    rrt_camera c;
    memcpy(&c, cam12, sizeof(c));
    rrt_effects fx;
    memcpy(&fx, effects36, sizeof(fx));
    rrt_params prm;
    rrt_get_launch_defaults(&prm);
    const int rc = rrt_launch_raymarch(d_out_rgba8, width, height, time, &c, sky, &fx, &prm, nullptr);
    if (rc != RRT_OK) {
        static std::atomic<bool> said{false};
        if (!said.exchange(true)) {
            fprintf(stderr, "launch_raymarch: %s%s%s (reported once)\n", rrt_status_string(rc),
                    rc == RRT_ERR_HIP ? " -- " : "", rc == RRT_ERR_HIP ? rrt_last_hip_error() : "");
        }
    }
    return rc;
}

int rrt_workspace_destroy(int id) {
    WorkspaceObject w;
    {
        std::lock_guard<std::mutex> lk(g_ws_mu);
        auto it = g_ws.find(id);
        if (it == g_ws.end()) return RRT_ERR_BAD_HANDLE;
        w = it->second;
        g_ws.erase(it);
    }
    if (w.side) { (void)hipStreamSynchronize(w.side); (void)hipStreamDestroy(w.side); }
    if (w.forked) (void)hipEventDestroy(w.forked);
    if (w.joined) (void)hipEventDestroy(w.joined);
    hipError_t e = hipFree(w.d_base);
    if (w.h_stats) (void)hipHostFree(w.h_stats);
    if (e != hipSuccess) return hip_fail(e, "hipFree(workspace)");
    return RRT_OK;
}

}  // extern "C"
namespace {
hipError_t read_chain_counters(const WorkspaceObject& w, DeferCounters* out) {
    uint8_t raw[kMaxChains * kCounterStride];
    const hipError_t e = hipMemcpy(raw, w.d_base, sizeof(raw), hipMemcpyDeviceToHost);
    if (e != hipSuccess) return e;
    for (int k = 0; k < kMaxChains; ++k) memcpy(&out[k], raw + (size_t)k * kCounterStride, sizeof(DeferCounters));
    return hipSuccess;
}
}  // namespace
extern "C" {
int rrt_workspace_stats(int id, unsigned* rows_used, unsigned* overflow_waves) {
    WorkspaceObject w;
    {
        std::lock_guard<std::mutex> lk(g_ws_mu);
        auto it = g_ws.find(id);
        if (it == g_ws.end()) return RRT_ERR_BAD_HANDLE;
        w = it->second;
    }
    if (!on_current_device(w.device)) return RRT_ERR_BAD_HANDLE;
    DeferCounters c[kMaxChains];
    RRT_HIP(read_chain_counters(w, c));
    unsigned long long rows = 0; unsigned left = 0;
    for (int k = 0; k < kMaxChains; ++k) { rows += c[k].total_blocks * kBlockRows; left += c[k].suspended_left; }
    if (rows_used) *rows_used = rows > 0xffffffffull ? 0xffffffffu : (unsigned)rows;
    if (overflow_waves) *overflow_waves = left;
    return RRT_OK;
}

int rrt_workspace_rounds(int id, unsigned* rounds_enqueued, unsigned* rounds_with_work, unsigned* peak_rows, unsigned* pool_rows) {
    WorkspaceObject w;
    {
        std::lock_guard<std::mutex> lk(g_ws_mu);
        auto it = g_ws.find(id);
        if (it == g_ws.end()) return RRT_ERR_BAD_HANDLE;
        w = it->second;
    }
    if (!on_current_device(w.device)) return RRT_ERR_BAD_HANDLE;
    DeferCounters c[kMaxChains];
    RRT_HIP(read_chain_counters(w, c));
    unsigned run = 0, work = 0, peak = 0;
    for (int k = 0; k < kMaxChains; ++k) {        /* rounds: of the chain that needed most; rows of the fullest round: both chains' */
        run = c[k].rounds_run > run ? c[k].rounds_run : run;
        work = c[k].rounds_with_work > work ? c[k].rounds_with_work : work;
        peak += c[k].peak_blocks;
    }
    if (rounds_enqueued) *rounds_enqueued = run;
    if (rounds_with_work) *rounds_with_work = work;
    if (peak_rows) *peak_rows = peak * kBlockRows;
    if (pool_rows) *pool_rows = (unsigned)((w.bytes / kBlockBytes) * kBlockRows);      /* upper bound: before the launch's bookkeeping */
    return RRT_OK;
}

int rrt_workspace_read(int id, size_t offset, size_t bytes, void* host_dst) {
    WorkspaceObject w;
    {
        std::lock_guard<std::mutex> lk(g_ws_mu);
        auto it = g_ws.find(id);
        if (it == g_ws.end()) return RRT_ERR_BAD_HANDLE;
        w = it->second;
    }
    if (!host_dst || offset > w.bytes || bytes > w.bytes - offset) return RRT_ERR_INVALID_ARGUMENT;
    if (!on_current_device(w.device)) return RRT_ERR_BAD_HANDLE;
    RRT_HIP(hipMemcpy(host_dst, w.d_base + offset, bytes, hipMemcpyDeviceToHost));
    return RRT_OK;
}

int rrt_noise_table_create_window(float t0, float t1, int coverage, int* out_id) {
    if (!out_id) return RRT_ERR_INVALID_ARGUMENT;
    NoiseTableObject nt;
    const int rc = plan_table(t0, t1, coverage, nt);
    if (rc != RRT_OK) return rc;
    RRT_HIP(hipGetDevice(&nt.device));
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&nt.d_cells), nt.bytes);
    if (e != hipSuccess) { (void)hipGetLastError(); snprintf(g_hip_err, sizeof(g_hip_err), "hipMalloc(noise table, %zu bytes): %s", nt.bytes, hipGetErrorString(e)); return e == hipErrorOutOfMemory ? RRT_ERR_OUT_OF_MEMORY : RRT_ERR_HIP; }
    hipLaunchKernelGGL(build_noise_table, dim3(4096), dim3(256), 0, nullptr, nt.d_cells, nt.acc);
    hipLaunchKernelGGL(build_noise_table, dim3(4096), dim3(256), 0, nullptr,
                       nt.d_cells + (size_t)nt.acc.nx * nt.acc.ny * nt.acc.nz, nt.dust);
    e = hipGetLastError();
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e != hipSuccess) { (void)hipFree(nt.d_cells); return hip_fail(e, "build_noise_table"); }
    std::lock_guard<std::mutex> lk(g_nt_mu);
    *out_id = g_nt_next++;
    g_nt.emplace(*out_id, nt);
    return RRT_OK;
}

int rrt_noise_table_create(float t_max, int* out_id) {
    if (!(t_max >= 0.0f)) return RRT_ERR_INVALID_ARGUMENT;
    return rrt_noise_table_create_window(0.0f, t_max, RRT_TABLE_FULL, out_id);
}

int rrt_noise_table_destroy(int id) {
    NoiseTableObject nt;
    {
        std::lock_guard<std::mutex> lk(g_nt_mu);
        auto it = g_nt.find(id);
        if (it == g_nt.end()) return RRT_ERR_BAD_HANDLE;
        nt = it->second;
        g_nt.erase(it);
    }
    hipError_t e = hipFree(nt.d_cells);
    if (e != hipSuccess) return hip_fail(e, "hipFree(noise table)");
    return RRT_OK;
}

int rrt_noise_table_info(int id, float* t_max, size_t* bytes, int* boxes12) {
    std::lock_guard<std::mutex> lk(g_nt_mu);
    auto it = g_nt.find(id);
    if (it == g_nt.end()) return RRT_ERR_BAD_HANDLE;
    const NoiseTableObject& nt = it->second;
    if (t_max) *t_max = nt.t1;
    if (bytes) *bytes = nt.bytes;
    if (boxes12) { memcpy(boxes12, &nt.acc, sizeof(LutBox)); memcpy(boxes12 + 6, &nt.dust, sizeof(LutBox)); }
    return RRT_OK;
}

int rrt_noise_table_window(int id, float* t0, float* t1, int* coverage, int* device) {
    std::lock_guard<std::mutex> lk(g_nt_mu);
    auto it = g_nt.find(id);
    if (it == g_nt.end()) return RRT_ERR_BAD_HANDLE;
    if (t0) *t0 = it->second.t0;
    if (t1) *t1 = it->second.t1;
    if (coverage) *coverage = it->second.coverage;
    if (device) *device = it->second.device;
    return RRT_OK;
}

/* boxes only (host arithmetic, no device): what rrt_noise_table_create_window would allocate, with the same
 * RRT_ERR_INVALID_ARGUMENT for a box it would refuse */
int rrt_noise_table_plan_window(float t0, float t1, int coverage, size_t* bytes, int* boxes12) {
    NoiseTableObject nt;
    const int rc = plan_table(t0, t1, coverage, nt);
    if (bytes) *bytes = rc == RRT_OK ? nt.bytes : 0;
    if (rc != RRT_OK) return rc;
    if (boxes12) { memcpy(boxes12, &nt.acc, sizeof(LutBox)); memcpy(boxes12 + 6, &nt.dust, sizeof(LutBox)); }
    return RRT_OK;
}

int rrt_noise_table_plan(float t_max, size_t* bytes, int* boxes12) {
    if (!(t_max >= 0.0f)) return RRT_ERR_INVALID_ARGUMENT;
    return rrt_noise_table_plan_window(0.0f, t_max, RRT_TABLE_FULL, bytes, boxes12);
}

/* The window a frame driver should build next: the longest [t_from, t1], t1 <= t_until, at the richest coverage,
 * whose table fits `budget_bytes` -- the window is halved (down to 0.5 s) before the coverage is lowered, because a
 * rebuild costs milliseconds while a coarser table costs every frame.  RRT_OK with *bytes_out == 0 when nothing fits
 * (the driver then renders without a table: same bytes, slower). */
int rrt_noise_table_fit_window(float t_from, float t_until, size_t budget_bytes, float* t1_out, int* coverage_out, size_t* bytes_out) {
    if (!t1_out || !coverage_out || !bytes_out || !(t_from <= t_until)) return RRT_ERR_INVALID_ARGUMENT;
    *t1_out = t_from; *coverage_out = RRT_TABLE_FULL; *bytes_out = 0;
    for (int cov = RRT_TABLE_FULL; cov <= RRT_TABLE_COARSEST; ++cov) {
        float span = t_until - t_from;
        for (;;) {
            NoiseTableObject nt;
            const float t1 = t_from + span;
            if (plan_table(t_from, t1, cov, nt) == RRT_OK && nt.bytes <= budget_bytes) {
                *t1_out = t1; *coverage_out = cov; *bytes_out = nt.bytes;
                return RRT_OK;
            }
            if (span <= 0.5f) break;
            span = span * 0.5f < 0.5f ? 0.5f : span * 0.5f;
        }
    }
    return RRT_OK;
}

/* test hook: make every device check see `device` as the current one (< 0: ask HIP again) */
int rrt_debug_fake_device(int device) {
    if (!test_hooks_enabled()) return RRT_ERR_INVALID_ARGUMENT;
    g_fake_device.store(device < 0 ? -1 : device);
    return RRT_OK;
}

int rrt_launch_raymarch_rows(void* d_out_rows, int width, int height, int y0, int y1, float time,
                             const rrt_camera* cam, rrt_sky_t sky, const rrt_effects* fx,
                             const rrt_params* prm, void* stream) {
    int rc = check_common(d_out_rows, width, height, cam, fx, prm);
    if (rc) return rc;
    if (y0 < 0 || y1 > height || y0 > y1) return RRT_ERR_INVALID_ARGUMENT;
    FrameArgs a;
    LaunchOpts o;
    rc = fill_args(a, o, d_out_rows, width, height, time, cam, sky, fx, prm);
    if (rc) return rc;
    a.rows = RowMap{y1 - y0, y0, y1 - y0 > 0 ? y1 - y0 : 1, 0, 1, nullptr};
    return launch(a, o, false, static_cast<hipStream_t>(stream));
}

int rrt_launch_raymarch(void* d_out_rgba8, int width, int height, float time, const rrt_camera* cam,
                        rrt_sky_t sky, const rrt_effects* fx, const rrt_params* prm, void* stream) {
    return rrt_launch_raymarch_rows(d_out_rgba8, width, height, 0, height, time, cam, sky, fx, prm, stream);
}

int rrt_launch_raymarch_ex(void* d_out_rgba8, int width, int height, float time, const rrt_camera* cam,
                           rrt_sky_t sky, const rrt_effects* fx, const rrt_params* prm,
                           const rrt_debug_outputs* dbg, void* stream) {
    int rc = check_common(d_out_rgba8, width, height, cam, fx, prm);
    if (rc) return rc;
    FrameArgs a;
    LaunchOpts o;
    rc = fill_args(a, o, d_out_rgba8, width, height, time, cam, sky, fx, prm);
    if (rc) return rc;
    a.rows = RowMap{height, 0, height, 0, 1, nullptr};
    if (dbg) a.dbg = *dbg;
    return launch(a, o, dbg != nullptr, static_cast<hipStream_t>(stream));
}

int rrt_tile_shard_rows(int height, int tile_rows, int shard, int n_shards, int* rows) {
    if (!rows || height <= 0 || tile_rows <= 0 || n_shards <= 0 || shard < 0 || shard >= n_shards)
        return RRT_ERR_INVALID_ARGUMENT;
    *rows = shard_rows(height, tile_rows, shard, n_shards);
    return RRT_OK;
}

int rrt_launch_raymarch_tiles(void* d_out_tiles, int width, int height, int tile_rows, int shard, int n_shards,
                              float time, const rrt_camera* cam, rrt_sky_t sky, const rrt_effects* fx,
                              const rrt_params* prm, void* stream) {
    int rc = check_common(d_out_tiles, width, height, cam, fx, prm);
    if (rc) return rc;
    if (tile_rows <= 0 || n_shards <= 0 || shard < 0 || shard >= n_shards) return RRT_ERR_INVALID_ARGUMENT;
    FrameArgs a;
    LaunchOpts o;
    rc = fill_args(a, o, d_out_tiles, width, height, time, cam, sky, fx, prm);
    if (rc) return rc;
    a.rows = RowMap{shard_rows(height, tile_rows, shard, n_shards), 0, tile_rows, shard, n_shards, nullptr};
    return launch(a, o, false, static_cast<hipStream_t>(stream));
}

int rrt_assemble_tiles(void* d_frame, const void* d_tiles, int width, int height, int tile_rows, int shard,
                       int n_shards, void* stream) {
    if (!d_frame || !d_tiles || width <= 0 || height <= 0 || tile_rows <= 0 || n_shards <= 0 || shard < 0 ||
        shard >= n_shards)
        return RRT_ERR_INVALID_ARGUMENT;
    RowMap m{shard_rows(height, tile_rows, shard, n_shards), 0, tile_rows, shard, n_shards, nullptr};
    if (m.n_local_rows == 0) return RRT_OK;
    dim3 grid((width + 255) / 256, m.n_local_rows < kMaxGridY ? m.n_local_rows : kMaxGridY);
    hipLaunchKernelGGL(assemble_tiles_kernel, grid, dim3(256), 0, static_cast<hipStream_t>(stream),
                       static_cast<uchar4*>(d_frame), static_cast<const uchar4*>(d_tiles), width, height, m);
    RRT_HIP(hipGetLastError());
    return RRT_OK;
}

int rrt_assemble_all_tiles(void* d_frame, const void* d_tiles_all, size_t shard_stride_bytes, int width, int height,
                           int tile_rows, int n_shards, void* stream) {
    if (!d_frame || !d_tiles_all || width <= 0 || height <= 0 || tile_rows <= 0 || n_shards <= 0 ||
        (shard_stride_bytes & 3) != 0)
        return RRT_ERR_INVALID_ARGUMENT;
    if (shard_stride_bytes / 4 < (size_t)shard_rows(height, tile_rows, 0, n_shards) * width) return RRT_ERR_INVALID_ARGUMENT;
    dim3 grid((width + 255) / 256, height < kMaxGridY ? height : kMaxGridY);
    hipLaunchKernelGGL(assemble_all_kernel, grid, dim3(256), 0, static_cast<hipStream_t>(stream),
                       static_cast<uchar4*>(d_frame), static_cast<const uchar4*>(d_tiles_all), shard_stride_bytes / 4,
                       width, height, tile_rows, n_shards);
    RRT_HIP(hipGetLastError());
    return RRT_OK;
}

/* ---- explicit tile -> shard assignment (rrt_tile_map) ---- */
int rrt_tile_map_create(int height, int tile_rows, int n_shards, const int32_t* shard_of_tile, int* out_id) {
    if (!shard_of_tile || !out_id || height <= 0 || tile_rows <= 0 || n_shards <= 0) return RRT_ERR_INVALID_ARGUMENT;
    const int n_tiles = (height + tile_rows - 1) / tile_rows;
    auto m = std::make_shared<TileMapObject>();
    m->height = height; m->tile_rows = tile_rows; m->n_shards = n_shards; m->n_tiles = n_tiles;
    m->device = current_device();
    m->shard_of_tile.assign(shard_of_tile, shard_of_tile + n_tiles);
    m->offset.assign(n_shards + 1, 0);
    m->rows.assign(n_shards, 0);
    for (int t = 0; t < n_tiles; ++t) {
        const int sh = shard_of_tile[t];
        if (sh < 0 || sh >= n_shards) return RRT_ERR_INVALID_ARGUMENT;
        ++m->offset[sh + 1];
        m->rows[sh] += (t + 1) * tile_rows <= height ? tile_rows : height - t * tile_rows;
    }
    for (int sh = 0; sh < n_shards; ++sh) m->offset[sh + 1] += m->offset[sh];
    /* device image: [tile_of_local, grouped by shard, increasing t | shard_of_tile | k_of_tile] */
    std::vector<int> img(3 * (size_t)n_tiles), fill(m->offset.begin(), m->offset.end() - 1);
    for (int t = 0; t < n_tiles; ++t) {
        const int sh = shard_of_tile[t], k = fill[sh] - m->offset[sh];
        img[fill[sh]++] = t;
        img[n_tiles + t] = sh;
        img[2 * (size_t)n_tiles + t] = k;
    }
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&m->d_img), img.size() * sizeof(int));
    if (e != hipSuccess) return hip_fail(e, "hipMalloc(tile map)");
    e = hipMemcpy(m->d_img, img.data(), img.size() * sizeof(int), hipMemcpyHostToDevice);
    if (e != hipSuccess) { (void)hipFree(m->d_img); return hip_fail(e, "hipMemcpy(tile map)"); }
    std::lock_guard<std::mutex> lk(g_tm_mu);
    *out_id = g_tm_next++;
    g_tm.emplace(*out_id, m);
    return RRT_OK;
}

int rrt_tile_map_destroy(int id) {
    std::shared_ptr<TileMapObject> m;
    {
        std::lock_guard<std::mutex> lk(g_tm_mu);
        auto it = g_tm.find(id);
        if (it == g_tm.end()) return RRT_ERR_BAD_HANDLE;
        m = it->second;
        g_tm.erase(it);
    }
    hipError_t e = hipFree(m->d_img);
    if (e != hipSuccess) return hip_fail(e, "hipFree(tile map)");
    return RRT_OK;
}

int rrt_tile_map_shard_rows(int id, int shard, int* rows, int* max_rows) {
    const std::shared_ptr<TileMapObject> m = tile_map_lookup(id);
    if (!m) return RRT_ERR_BAD_HANDLE;
    if (shard < 0 || shard >= m->n_shards) return RRT_ERR_INVALID_ARGUMENT;
    if (rows) *rows = m->rows[shard];
    if (max_rows) { int mx = 0; for (int r : m->rows) mx = r > mx ? r : mx; *max_rows = mx; }
    return RRT_OK;
}

/* Deal row tiles to shards by cost: longest-processing-time-first greedy -- tiles in decreasing cost (ties: increasing
 * index), each to the shard with the smallest load so far (ties: fewest rows, then lowest index).  Deterministic host
 * arithmetic in double: every rank computes the same map from the same costs, no exchange needed.  max_tiles_per_shard
 * (0: none) bounds the buffer a shard needs: ceil(n_tiles / n_shards) + slack is typical. */
int rrt_tile_map_balance(int n_tiles, const float* tile_cost, int n_shards, int max_tiles_per_shard, int32_t* shard_of_tile_out) {
    if (n_tiles <= 0 || n_shards <= 0 || !tile_cost || !shard_of_tile_out || max_tiles_per_shard < 0) return RRT_ERR_INVALID_ARGUMENT;
    if (max_tiles_per_shard > 0 && (long long)max_tiles_per_shard * n_shards < n_tiles) return RRT_ERR_INVALID_ARGUMENT;
    std::vector<int> idx(n_tiles);
    for (int t = 0; t < n_tiles; ++t) {
        if (!(tile_cost[t] >= 0.0f) || !(tile_cost[t] < 3.0e38f)) return RRT_ERR_INVALID_ARGUMENT;    /* NaN, negative, infinite */
        idx[t] = t;
    }
    std::stable_sort(idx.begin(), idx.end(), [&](int x, int y) { return tile_cost[x] > tile_cost[y]; });
    std::vector<double> load(n_shards, 0.0);
    std::vector<int> count(n_shards, 0);
    for (int t : idx) {
        int best = -1;
        for (int sh = 0; sh < n_shards; ++sh) {
            if (max_tiles_per_shard > 0 && count[sh] >= max_tiles_per_shard) continue;
            if (best < 0 || load[sh] < load[best] || (load[sh] == load[best] && count[sh] < count[best])) best = sh;
        }
        shard_of_tile_out[t] = best;
        load[best] += (double)tile_cost[t];
        ++count[best];
    }
    return RRT_OK;
}

int rrt_launch_raymarch_tilemap(void* d_out_tiles, int width, int height, int tile_map, int shard, float time,
                                const rrt_camera* cam, rrt_sky_t sky, const rrt_effects* fx, const rrt_params* prm, void* stream) {
    int rc = check_common(d_out_tiles, width, height, cam, fx, prm);
    if (rc) return rc;
    const std::shared_ptr<TileMapObject> m = tile_map_lookup(tile_map);
    if (!m || !on_current_device(m->device)) return RRT_ERR_BAD_HANDLE;
    if (m->height != height || shard < 0 || shard >= m->n_shards) return RRT_ERR_INVALID_ARGUMENT;
    FrameArgs a;
    LaunchOpts o;
    rc = fill_args(a, o, d_out_tiles, width, height, time, cam, sky, fx, prm);
    if (rc) return rc;
    a.rows = RowMap{m->rows[shard], 0, m->tile_rows, shard, m->n_shards, m->d_img + m->offset[shard]};
    return launch(a, o, false, static_cast<hipStream_t>(stream));
}

int rrt_assemble_all_tilemap(void* d_frame, const void* d_tiles_all, size_t shard_stride_bytes, int width, int height,
                             int tile_map, void* stream) {
    if (!d_frame || !d_tiles_all || width <= 0 || height <= 0 || (shard_stride_bytes & 3) != 0) return RRT_ERR_INVALID_ARGUMENT;
    const std::shared_ptr<TileMapObject> m = tile_map_lookup(tile_map);
    if (!m || !on_current_device(m->device)) return RRT_ERR_BAD_HANDLE;
    if (m->height != height) return RRT_ERR_INVALID_ARGUMENT;
    for (int r : m->rows) if (shard_stride_bytes / 4 < (size_t)r * width) return RRT_ERR_INVALID_ARGUMENT;
    dim3 grid((width + 255) / 256, height < kMaxGridY ? height : kMaxGridY);
    hipLaunchKernelGGL(assemble_map_kernel, grid, dim3(256), 0, static_cast<hipStream_t>(stream), static_cast<uchar4*>(d_frame),
                       static_cast<const uchar4*>(d_tiles_all), shard_stride_bytes / 4, width, height, m->tile_rows,
                       m->d_img + m->n_tiles, m->d_img + 2 * (size_t)m->n_tiles);
    RRT_HIP(hipGetLastError());
    return RRT_OK;
}

/* Estimated cost of every row tile of the frame, from the coarse march-only probe (one ray per 16 pixels in x, per
 * min(16, tile_rows) rows in y): what rrt_tile_map_balance wants for a frame nobody has rendered yet.  Synchronous (it
 * returns host numbers); allocates and frees its own scratch.  The unit is the tile-order object's (wave clocks / 16),
 * summed over the probe cells of the tile and scaled to the tile's pixel count. */
int rrt_probe_tile_costs(int width, int height, int tile_rows, float time, const rrt_camera* cam, const rrt_effects* fx,
                         const rrt_params* prm, float* tile_cost_host, int n_tiles, void* stream) {
    if (!tile_cost_host || tile_rows <= 0 || !cam || !fx || width <= 0 || height <= 0) return RRT_ERR_INVALID_ARGUMENT;
    if (n_tiles != (height + tile_rows - 1) / tile_rows) return RRT_ERR_INVALID_ARGUMENT;
    if (prm) { int rc = check_params_abi(prm); if (rc == RRT_OK) rc = check_params_values(prm); if (rc != RRT_OK) return rc; }
    rrt_params p;
    load_params(prm, p);
    FrameArgs a;
    memset(&a, 0, sizeof(a));
    a.width = width; a.height = height; a.time = time; a.cam = *cam;
    a.use_lens = fx->use_lens_distortion != 0; a.distortion_amount = fx->distortion_amount;
    a.spin = p.spin; a.drag_c = (2.0f * p.spin) * 2.0f; a.max_steps = p.max_steps;
    a.rows = RowMap{height, 0, height, 0, 1, nullptr};
    a.grid_row_stride = 1;
    const int sy = tile_rows < kProbeStride ? tile_rows : kProbeStride;
    ProbeArgs q;
    const int cells_x = (width + kProbeStride - 1) / kProbeStride, cells_y = (height + sy - 1) / sy;
    unsigned* d_cells = nullptr;
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&d_cells), (size_t)cells_x * cells_y * sizeof(unsigned));
    if (e != hipSuccess) return hip_fail(e, "hipMalloc(probe)");
    hipStream_t st = static_cast<hipStream_t>(stream);
    int rc = enqueue_probe(a, q, d_cells, kProbeStride, sy, p.volumetrics != 0, st);
    std::vector<unsigned> cells((size_t)cells_x * cells_y);
    if (rc == RRT_OK) {
        e = hipMemcpyAsync(cells.data(), d_cells, cells.size() * sizeof(unsigned), hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (e != hipSuccess) rc = hip_fail(e, "probe read-back");
    }
    (void)hipFree(d_cells);
    if (rc != RRT_OK) return rc;
    for (int t = 0; t < n_tiles; ++t) {
        const int y0 = t * tile_rows, y1 = (t + 1) * tile_rows < height ? (t + 1) * tile_rows : height;
        double sum = 0.0; int n = 0;
        for (int cy = y0 / sy; cy * sy < y1 && cy < cells_y; ++cy) {
            const int yc = cy * sy + sy / 2 < height ? cy * sy + sy / 2 : height - 1;     /* the row the cell was probed at */
            if (yc < y0 || yc >= y1) continue;
            for (int cx = 0; cx < cells_x; ++cx) sum += (double)cells[(size_t)cy * cells_x + cx];
            ++n;
        }
        if (n == 0) {        /* no probe row inside the tile (only when tile_rows does not divide the stride): nearest one */
            const int cy = (y0 + y1) / 2 / sy < cells_y ? (y0 + y1) / 2 / sy : cells_y - 1;
            for (int cx = 0; cx < cells_x; ++cx) sum += (double)cells[(size_t)cy * cells_x + cx];
            n = 1;
        }
        tile_cost_host[t] = (float)(sum / n * (double)(y1 - y0) / (double)kWGPixY);      /* per 8-row wave tile row of the tile */
    }
    return RRT_OK;
}

int rrt_clock_probe(unsigned long long* d_counters2, unsigned duration_us, void* stream) {
    if (!d_counters2 || duration_us == 0 || duration_us > 2000000u) return RRT_ERR_INVALID_ARGUMENT;
    hipLaunchKernelGGL(clock_probe_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream), d_counters2,
                       (unsigned long long)duration_us * 100ull);
    RRT_HIP(hipGetLastError());
    return RRT_OK;
}

/* ---- unit kernels ---- */
int rrt_unit_geodesic_acc(int n, const float* p, const float* v, float spin, float* out, void* st) {
    if (n > 0 && (!p || !v || !out)) return RRT_ERR_INVALID_ARGUMENT;
    return unit_launch(n, st, [&](dim3 g, dim3 b, hipStream_t s) { hipLaunchKernelGGL(k_geodesic_acc, g, b, 0, s, n, p, v, spin, out); });
}
int rrt_unit_rk4(int n, float* p, float* v, const float* h, float spin, void* st) {
    if (n > 0 && (!p || !v || !h)) return RRT_ERR_INVALID_ARGUMENT;
    return unit_launch(n, st, [&](dim3 g, dim3 b, hipStream_t s) { hipLaunchKernelGGL(k_rk4, g, b, 0, s, n, p, v, h, spin); });
}
int rrt_unit_rk4_lean(int n, float* p, float* v, const float* h, float spin, int n_steps, float seed_scale, int32_t* steps, void* st) {
    if (n < 0 || n_steps < 0 || (n > 0 && (!p || !v)) || !(seed_scale == seed_scale)) return RRT_ERR_INVALID_ARGUMENT;
    if (n == 0) return RRT_OK;
    const float drag_c = (2.0f * spin) * 2.0f;
    const dim3 g((n + 63) / 64), b(64);                     /* one wavefront per workgroup, like the render kernels */
    if (spin != 0.0f) hipLaunchKernelGGL((k_rk4_lean<true>), g, b, 0, static_cast<hipStream_t>(st), n, p, v, h, drag_c, n_steps, seed_scale, steps);
    else hipLaunchKernelGGL((k_rk4_lean<false>), g, b, 0, static_cast<hipStream_t>(st), n, p, v, h, drag_c, n_steps, seed_scale, steps);
    RRT_HIP(hipGetLastError());
    return RRT_OK;
}
int rrt_unit_div_seeded(int n, const float* a, const float* b, const float* seed, float* out, void* st) {
    if (n > 0 && (!a || !b || !seed || !out)) return RRT_ERR_INVALID_ARGUMENT;
    return unit_launch(n, st, [&](dim3 g, dim3 bl, hipStream_t s) { hipLaunchKernelGGL(k_div_seeded, g, bl, 0, s, n, a, b, seed, out); });
}
int rrt_unit_hash31(int n, const float* p, float* out, void* st) {
    if (n > 0 && (!p || !out)) return RRT_ERR_INVALID_ARGUMENT;
    return unit_launch(n, st, [&](dim3 g, dim3 b, hipStream_t s) { hipLaunchKernelGGL(k_hash31, g, b, 0, s, n, p, out); });
}
int rrt_unit_noise3d(int n, const float* p, float* out, void* st) {
    if (n > 0 && (!p || !out)) return RRT_ERR_INVALID_ARGUMENT;
    return unit_launch(n, st, [&](dim3 g, dim3 b, hipStream_t s) { hipLaunchKernelGGL(k_noise3d, g, b, 0, s, n, p, out); });
}
int rrt_unit_fbm(int n, const float* p, int oct, float* out, void* st) {
    if (n > 0 && (!p || !out)) return RRT_ERR_INVALID_ARGUMENT;
    if (oct < 0 || oct > 16) return RRT_ERR_INVALID_ARGUMENT;
    return unit_launch(n, st, [&](dim3 g, dim3 b, hipStream_t s) { hipLaunchKernelGGL(k_fbm, g, b, 0, s, n, p, oct, out); });
}
int rrt_unit_accretion_density(int n, const float* p, float time, float* out, void* st) {
    if (n > 0 && (!p || !out)) return RRT_ERR_INVALID_ARGUMENT;
    return unit_launch(n, st, [&](dim3 g, dim3 b, hipStream_t s) { hipLaunchKernelGGL(k_accretion, g, b, 0, s, n, p, time, out); });
}
int rrt_unit_dust_density(int n, const float* p, float time, float* out, void* st) {
    if (n > 0 && (!p || !out)) return RRT_ERR_INVALID_ARGUMENT;
    return unit_launch(n, st, [&](dim3 g, dim3 b, hipStream_t s) { hipLaunchKernelGGL(k_dust, g, b, 0, s, n, p, time, out); });
}
int rrt_unit_redshift(int n, const float* p, const float* vel, float spin, float* out, void* st) {
    if (n > 0 && (!p || !vel || !out)) return RRT_ERR_INVALID_ARGUMENT;
    return unit_launch(n, st, [&](dim3 g, dim3 b, hipStream_t s) { hipLaunchKernelGGL(k_redshift, g, b, 0, s, n, p, vel, spin, out); });
}
int rrt_unit_math(int fn, int n, const float* a, const float* b, float* out, void* st) {
    if (n > 0 && (!a || !b || !out)) return RRT_ERR_INVALID_ARGUMENT;
    if (fn < 0 || fn > 5) return RRT_ERR_INVALID_ARGUMENT;
    return unit_launch(n, st, [&](dim3 g, dim3 bl, hipStream_t s) { hipLaunchKernelGGL(k_math, g, bl, 0, s, fn, n, a, b, out); });
}
int rrt_unit_sky_sample(int n, const float* dir, float off, rrt_sky_t sky, int frac_bits, float* out, void* st) {
    if (n > 0 && (!dir || !out)) return RRT_ERR_INVALID_ARGUMENT;
    if (frac_bits < 0 || frac_bits > 16) return RRT_ERR_INVALID_ARGUMENT;
    SkyObject so;
    if (!sky_lookup(sky, so)) return RRT_ERR_BAD_HANDLE;
    SkyTex t{so.d_texels, so.w, so.h, frac_bits};
    return unit_launch(n, st, [&](dim3 g, dim3 b, hipStream_t s) { hipLaunchKernelGGL(k_sky, g, b, 0, s, n, dir, off, t, out); });
}

int rrt_unit_disk_temperature(int n, const float* r, float* out, void* st) {
    if (n > 0 && (!r || !out)) return RRT_ERR_INVALID_ARGUMENT;
    return unit_launch(n, st, [&](dim3 g, dim3 b, hipStream_t s) { hipLaunchKernelGGL(k_disk_temperature, g, b, 0, s, n, r, out); });
}
int rrt_unit_smoothstep(int n, const float* e0, const float* e1, const float* x, float* out, void* st) {
    if (n > 0 && (!e0 || !e1 || !x || !out)) return RRT_ERR_INVALID_ARGUMENT;
    return unit_launch(n, st, [&](dim3 g, dim3 b, hipStream_t s) { hipLaunchKernelGGL(k_smoothstep, g, b, 0, s, n, e0, e1, x, out); });
}
int rrt_unit_postfx(int what, int n, const float* rgb, const float* uv, float param, float* out, void* st) {
    if (what < 0 || what > 2) return RRT_ERR_INVALID_ARGUMENT;
    if (n > 0 && (!out || (what != 2 && !uv) || (what != 0 && !rgb))) return RRT_ERR_INVALID_ARGUMENT;
    return unit_launch(n, st, [&](dim3 g, dim3 b, hipStream_t s) { hipLaunchKernelGGL(k_postfx, g, b, 0, s, what, n, rgb, uv, param, out); });
}
int rrt_unit_rt_sample(int n, const float* d_disk, const float* d_cloud, const float* p, const float* vel, const float* h,
                       float spin, float* rad, void* st) {
    if (n > 0 && (!d_disk || !d_cloud || !p || !vel || !h || !rad)) return RRT_ERR_INVALID_ARGUMENT;
    return unit_launch(n, st, [&](dim3 g, dim3 b, hipStream_t s) { hipLaunchKernelGGL(k_rt_sample, g, b, 0, s, n, d_disk, d_cloud, p, vel, h, spin, rad); });
}
int rrt_unit_noise3d_lut(int n, const float* p, int table, int which, float* out, unsigned* d_counts, void* st) {
    if (n > 0 && (!p || !out)) return RRT_ERR_INVALID_ARGUMENT;
    if (which < 0 || which > 1) return RRT_ERR_INVALID_ARGUMENT;
    NoiseTableObject nt;
    {
        std::lock_guard<std::mutex> lk(g_nt_mu);
        auto it = g_nt.find(table);
        if (it == g_nt.end()) return RRT_ERR_BAD_HANDLE;
        nt = it->second;
    }
    if (!on_current_device(nt.device)) return RRT_ERR_BAD_HANDLE;
    const NoiseLut L = which == 0 ? make_lut(nt.d_cells, nt.acc, nt.acc_families)
                                  : make_lut(nt.d_cells + (size_t)nt.acc.nx * nt.acc.ny * nt.acc.nz, nt.dust, nt.dust_families);
    return unit_launch(n, st, [&](dim3 g, dim3 b, hipStream_t s) { hipLaunchKernelGGL(k_noise3d_lut, g, b, 0, s, n, p, L, out, d_counts); });
}
int rrt_unit_media_lut(int n, const float* p, float time, int table, float* out_disk, float* out_dust, unsigned* d_counts, void* st) {
    if (n > 0 && (!p || !out_disk || !out_dust)) return RRT_ERR_INVALID_ARGUMENT;
    NoiseTableObject nt;
    {
        std::lock_guard<std::mutex> lk(g_nt_mu);
        auto it = g_nt.find(table);
        if (it == g_nt.end()) return RRT_ERR_BAD_HANDLE;
        nt = it->second;
    }
    if (!on_current_device(nt.device)) return RRT_ERR_BAD_HANDLE;
    if (!(time >= nt.t0 && time <= nt.t1)) return RRT_ERR_INVALID_ARGUMENT;
    const NoiseLut la = make_lut(nt.d_cells, nt.acc, nt.acc_families);
    const NoiseLut ld = make_lut(nt.d_cells + (size_t)nt.acc.nx * nt.acc.ny * nt.acc.nz, nt.dust, nt.dust_families);
    return unit_launch(n, st, [&](dim3 g, dim3 b, hipStream_t s) { hipLaunchKernelGGL(k_media_lut, g, b, 0, s, n, p, time, la, ld, out_disk, out_dust, d_counts); });
}

int rrt_selfcheck_div_const(uint32_t lo_bits, uint32_t hi_bits, unsigned long long* d_counters, void* st) {
    if (!d_counters || lo_bits > hi_bits || hi_bits > 0x7f800000u) return RRT_ERR_INVALID_ARGUMENT;
    hipLaunchKernelGGL(k_selfcheck_div_const, dim3(4096), dim3(256), 0, static_cast<hipStream_t>(st), lo_bits, hi_bits, d_counters);
    RRT_HIP(hipGetLastError());
    return RRT_OK;
}

int rrt_selfcheck_sqrt(uint32_t lo_bits, uint32_t hi_bits, unsigned long long* d_counters, void* st) {
    if (!d_counters || lo_bits > hi_bits) return RRT_ERR_INVALID_ARGUMENT;
    hipLaunchKernelGGL(k_selfcheck_sqrt, dim3(2048), dim3(256), 0, static_cast<hipStream_t>(st), lo_bits, hi_bits, d_counters);
    RRT_HIP(hipGetLastError());
    return RRT_OK;
}
int rrt_selfcheck_sqrt_seeded(uint32_t lo_bits, uint32_t hi_bits, unsigned long long* d_counters, void* st) {
    if (!d_counters || lo_bits > hi_bits) return RRT_ERR_INVALID_ARGUMENT;
    hipLaunchKernelGGL(k_selfcheck_sqrt_seeded, dim3(2048), dim3(256), 0, static_cast<hipStream_t>(st), lo_bits, hi_bits, d_counters);
    RRT_HIP(hipGetLastError());
    return RRT_OK;
}
int rrt_selfcheck_sqrt_boundaries(int e_lo, int e_hi, int span, unsigned n_seeds, float tol1, float tol2, unsigned long long* d_counters, void* st) {
    if (!d_counters || e_lo >= e_hi || e_lo < -60 || e_hi > 100 || span < 0 || span > 4096 || n_seeds == 0 || !(tol1 >= 0.0f) || !(tol2 >= 0.0f))
        return RRT_ERR_INVALID_ARGUMENT;
    hipLaunchKernelGGL(k_selfcheck_sqrt_boundaries, dim3(2048), dim3(256), 0, static_cast<hipStream_t>(st), e_lo, e_hi, span, n_seeds, tol1, tol2, d_counters);
    RRT_HIP(hipGetLastError());
    return RRT_OK;
}
int rrt_selfcheck_div_tame(unsigned long long n, uint32_t seed, unsigned long long* d_counters, void* st) {
    if (!d_counters) return RRT_ERR_INVALID_ARGUMENT;
    hipLaunchKernelGGL(k_selfcheck_div_tame, dim3(2048), dim3(256), 0, static_cast<hipStream_t>(st), n, seed, d_counters);
    RRT_HIP(hipGetLastError());
    return RRT_OK;
}
int rrt_selfcheck_div_march(unsigned long long n, uint32_t seed, float tol1, float tol2, unsigned long long* d_counters, void* st) {
    if (!d_counters || !(tol1 >= 0.0f) || !(tol2 >= 0.0f)) return RRT_ERR_INVALID_ARGUMENT;
    hipLaunchKernelGGL(k_selfcheck_div_march, dim3(2048), dim3(256), 0, static_cast<hipStream_t>(st), n, seed, tol1, tol2, d_counters);
    RRT_HIP(hipGetLastError());
    return RRT_OK;
}
int rrt_selfcheck_div(unsigned long long n, uint32_t seed, unsigned long long* d_counters, void* st) {
    if (!d_counters) return RRT_ERR_INVALID_ARGUMENT;
    hipLaunchKernelGGL(k_selfcheck_div, dim3(2048), dim3(256), 0, static_cast<hipStream_t>(st), n, seed, d_counters);
    RRT_HIP(hipGetLastError());
    return RRT_OK;
}

/* ---------------------------------------------------------------- camera paths (host)
 * Reference: catmull_rom / lerp_angle / initDefaultPaths, src/camera_paths.cpp:6-73;
 * PathController::getInterpolatedState, src/main.cpp:176-203; fixed recording clock,
 * src/main.cpp:511-516.  The keyframe tables are data and keep the reference's values. */
}  // extern "C"  (re-opened below)

namespace {
struct Key { float time, x, y, z, yaw, pitch; };
struct Path { const char* name; int n; Key keys[6]; };
const Path kPaths[3] = {
    {"Gargantua Fly-By", 5, {{0.0f, 0.0f, 15.0f, -80.0f, 0.0f, -10.6f},
                             {6.0f, 15.0f, 3.0f, -30.0f, -26.6f, -5.1f},
                             {12.0f, 35.0f, 0.8f, 10.0f, -106.0f, -1.2f},
                             {18.0f, 5.0f, 1.5f, 50.0f, -174.3f, -1.7f},
                             {25.0f, -20.0f, 12.0f, 70.0f, -196.0f, -9.3f}}},
    {"Event Horizon Focus", 5, {{0.0f, 40.0f, 2.0f, 0.0f, -90.0f, 0.0f},
                                {8.0f, 0.0f, 5.0f, 40.0f, -180.0f, -5.0f},
                                {16.0f, -40.0f, 2.0f, 0.0f, -270.0f, 0.0f},
                                {24.0f, 0.0f, -5.0f, -40.0f, -360.0f, 5.0f},
                                {32.0f, 40.0f, 2.0f, 0.0f, -450.0f, 0.0f}}},
    {"Horizon Skimmer", 6, {{0.0f, 0.0f, 10.0f, -60.0f, 0.0f, -9.5f},
                            {8.0f, 15.0f, 2.0f, -15.0f, -45.0f, -4.7f},
                            {14.0f, 4.2f, 0.6f, 4.2f, -90.0f, -5.7f},
                            {20.0f, -20.0f, 8.0f, -20.0f, -225.0f, -20.0f},
                            {26.0f, -20.0f, 8.0f, -20.0f, 20.0f, -10.0f},
                            {29.0f, -30.0f, 2.0f, -30.0f, 45.0f, -2.7f}}},
};

float catmull_1d(float a, float b, float c, float d, float t, float t2, float t3) {   /* camera_paths.cpp:10-15 */
    return 0.5f * ((2.0f * b) + (-a + c) * t + (2.0f * a - 5.0f * b + 4.0f * c - d) * t2 +
                   (-a + 3.0f * b - 3.0f * c + d) * t3);
}
float lerp_angle_f(float a, float b, float t) {                                       /* camera_paths.cpp:25-29 */
    float diff = fmodf(b - a + 180.0f, 360.0f) - 180.0f;
    if (diff < -180.0f) diff += 360.0f;
    return a + diff * t;
}
}  // namespace

extern "C" {

int rrt_catmull_rom(const float p0[3], const float p1[3], const float p2[3], const float p3[3], float t,
                    float out[3]) {
    if (!p0 || !p1 || !p2 || !p3 || !out) return RRT_ERR_INVALID_ARGUMENT;
    float t2 = t * t, t3 = t2 * t;
    for (int k = 0; k < 3; ++k) out[k] = catmull_1d(p0[k], p1[k], p2[k], p3[k], t, t2, t3);
    return RRT_OK;
}

int rrt_lerp_angle(float a, float b, float t, float* out) {
    if (!out) return RRT_ERR_INVALID_ARGUMENT;
    *out = lerp_angle_f(a, b, t);
    return RRT_OK;
}

int rrt_path_count(void) { return 3; }

int rrt_path_info(int idx, const char** name, int* n_keys, float* t_end) {
    if (idx < 0 || idx >= 3) return RRT_ERR_INVALID_ARGUMENT;
    if (name) *name = kPaths[idx].name;
    if (n_keys) *n_keys = kPaths[idx].n;
    if (t_end) *t_end = kPaths[idx].keys[kPaths[idx].n - 1].time;
    return RRT_OK;
}

int rrt_path_keyframes(int idx, float* out6, int cap_keys) {
    if (idx < 0 || idx >= 3 || !out6 || cap_keys < kPaths[idx].n) return RRT_ERR_INVALID_ARGUMENT;
    for (int i = 0; i < kPaths[idx].n; ++i) memcpy(out6 + 6 * i, &kPaths[idx].keys[i], 6 * sizeof(float));
    return RRT_OK;
}

/* Camera of a built-in path at path time t -- what PathController::getInterpolatedState (src/main.cpp:176-203)
 * returns: the end poses hold outside the keyed interval; inside, the position is the Catmull-Rom spline through
 * the segment's two keys and their neighbours (the end keys stand in for the missing neighbour), the angles go the
 * short way round (lerp_angle), all with the segment parameter (t - t_from) / (t_to - t_from) in binary32. */
int rrt_path_camera_at(int idx, float t, rrt_camera* out) {
    if (idx < 0 || idx >= 3 || !out) return RRT_ERR_INVALID_ARGUMENT;
    const Key* key = kPaths[idx].keys;
    const int last = kPaths[idx].n - 1;
    auto pose_of = [&](const Key& q) { return rrt_camera_from_angles(&q.x, q.yaw, q.pitch, out); };
    if (t <= key[0].time) return pose_of(key[0]);
    if (t >= key[last].time) return pose_of(key[last]);
    int seg = -1;                                           /* first segment whose closed interval holds t */
    for (int i = 0; i < last && seg < 0; ++i)
        if (t >= key[i].time && t <= key[i + 1].time) seg = i;
    if (seg < 0) return pose_of(key[last]);                 /* t is NaN */
    const Key &from = key[seg], &to = key[seg + 1];
    const Key &before = key[seg > 0 ? seg - 1 : 0], &after = key[seg + 2 < last ? seg + 2 : last];
    const float s = (t - from.time) / (to.time - from.time);
    float pos[3];
    rrt_catmull_rom(&before.x, &from.x, &to.x, &after.x, s, pos);   /* Key holds x, y, z contiguously */
    return rrt_camera_from_angles(pos, lerp_angle_f(from.yaw, to.yaw, s), lerp_angle_f(from.pitch, to.pitch, s), out);
}

/* The recording clock of the reference's main loop (src/main.cpp:511-516): dt = 1.0f / fps,
 * simTime and pathTime are float accumulators advanced BEFORE each render, so frame k
 * (1-based) is rendered at sum_{1..k} dt evaluated in binary32. */
int rrt_recording_clock(int frame_k, int fps, float* sim_time, float* path_time) {
    if (frame_k < 0 || fps <= 0) return RRT_ERR_INVALID_ARGUMENT;
    const float dt = 1.0f / (float)fps;
    float s = 0.0f, p = 0.0f;
    for (int i = 0; i < frame_k; ++i) { s += dt; p += dt; }
    if (sim_time) *sim_time = s;
    if (path_time) *path_time = p;
    return RRT_OK;
}

/* CameraController::getCUDAStateFrom, reference src/main.cpp:141-167 (host C++ there too).
 * Note the reference's 3.14159f, not PI. */
int rrt_camera_from_angles(const float pos[3], float yaw, float pitch, rrt_camera* out) {
    if (!pos || !out) return RRT_ERR_INVALID_ARGUMENT;
    /* float arithmetic in the reference's order (its results are pinned by tests/test_camera_paths.py) */
    const float deg = 3.14159f;                                    /* the reference's literal, not PI */
    const float a_yaw = yaw * deg / 180.0f, a_pitch = pitch * deg / 180.0f;
    const float cp = std::cos(a_pitch);
    float f[3] = {std::sin(a_yaw) * cp, std::sin(a_pitch), std::cos(a_yaw) * cp};
    const float f_len = std::sqrt(f[0] * f[0] + f[1] * f[1] + f[2] * f[2]);
    for (float& c : f) c /= f_len;
    const float world_up[3] = {0.0f, 1.0f, 0.0f};
    auto cross3 = [](const float a[3], const float b[3], float o[3]) {
        o[0] = a[1] * b[2] - a[2] * b[1];
        o[1] = a[2] * b[0] - a[0] * b[2];
        o[2] = a[0] * b[1] - a[1] * b[0];
    };
    float side[3], top[3];
    cross3(world_up, f, side);                                     /* right = worldUp x forward */
    const float side_len = std::sqrt(side[0] * side[0] + side[1] * side[1] + side[2] * side[2]);
    for (float& c : side) c /= side_len;
    cross3(f, side, top);                                          /* up = forward x right */
    for (int k = 0; k < 3; ++k) { out->pos[k] = pos[k]; out->forward[k] = f[k]; out->right[k] = side[k]; out->up[k] = top[k]; }
    return RRT_OK;
}

}  // extern "C"

/* The reference's entry point as a linkable C++ symbol (include/raymarcher.h:19, src/raymarcher.cu:176-180),
 * for translation units built against THIS repository's include/raymarcher.h (HIP vector types).  The twin
 * under the reference's own mangled name (CUDA's `struct uchar4`) is csrc/rrt_compat.cpp. */
#include "../../include/raymarcher.h"
void launch_raymarch(uchar4* d_out, int w, int h, float time, CameraState cam, cudaTextureObject_t skyboxTex,
                     CameraEffects effects) {
    (void)rrt_launch_raymarch_compat(d_out, w, h, time, reinterpret_cast<const float*>(&cam), skyboxTex, &effects);
}
