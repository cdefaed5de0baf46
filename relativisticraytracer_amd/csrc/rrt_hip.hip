/*
 * rrt_hip.hip -- the HIP translation unit of librrt_hip.so: host side of the render path and its C ABI (include/rrt.h).
 *
 * Replaces the reference's only CUDA translation unit, src/raymarcher.cu (raymarch_kernel :15-174 and launch_raymarch
 * :176-180).  Built with
 *   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize -mllvm -enable-post-misched=false
 *         -mllvm -amdgpu-sched-strategy=max-ilp -fPIC
 * (relativisticraytracer_amd/build.py).  gfx950 only: no other target, no CUDA dual path.
 *
 * Sections (round 5: one 3 100-line file became four)
 *   rrt_kernels.h      the kernels and the structures they share with the host (included below: a kernel and its launch
 *                      site share a translation unit)
 *   this file          handle registries (sky, workspace, noise table, tile map, tile order), noise-table planning, path choice
 *                      and launch logic, the C ABI
 *   rrt_camera.cpp     camera basis / path playback / recording clock (plain C++)
 *   rrt_test_hooks.h   rrt_unit_*, rrt_selfcheck_*, rrt_debug_fake_device -- compiled only with -DRRT_TEST_HOOKS, i.e. into
 *                      librrt_hip_test.so; the product library exports none of them
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
#ifdef RRT_TEST_HOOKS
#include "../../include/rrt_test.h"
#endif
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
 * rrt_debug_fake_device() (librrt_hip_test.so) lets a CPU-only test drive those checks. */
#ifdef RRT_TEST_HOOKS            /* librrt_hip_test.so only */
std::atomic<int> g_fake_device{-1};
bool test_hooks_enabled() {       /* decided once, from the environment the process was started with */
    static const bool on = [] { const char* e = getenv("RRT_ENABLE_TEST_HOOKS"); return e && e[0] == '1'; }();
    return on;
}
int faked_device() { return g_fake_device.load(std::memory_order_relaxed); }
#else
constexpr int faked_device() { return -1; }
#endif
int current_device() {
    const int fake = faked_device();
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

/* ------------------------------------------------------------------ device code (kernels, kernel arguments, pool layout) */
#include "rrt_kernels.h"

struct WorkspaceObject {
    uint8_t* d_base; size_t bytes; int device;
    DeferCounters* h_stats;      /* pinned host copy (one per chain) of the counters its last launch left (asynchronous, behind that launch) */
    hipStream_t side;            /* the second chain's stream (round 4), with the two events that fork it from and join it to the caller's */
    hipEvent_t forked, joined;
};
std::mutex g_ws_mu;
std::unordered_map<int, WorkspaceObject> g_ws;
int g_ws_next = 1;

/* ------------------------------------------------------------------ lattice-hash tables: planning + registry */
#include "rrt_noise_plan.h"

/* ------------------------------------------------------------------ rrt_tile_map / rrt_tile_order objects + registries */
#include "rrt_tile_objects.h"


#ifdef RRT_TEST_HOOKS
#define RRT_TEST_HOOKS_PART 1
#include "rrt_test_hooks.h"
#undef RRT_TEST_HOOKS_PART
#endif

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
    memset(&a.lut_acc, 0, sizeof(a.lut_acc)); memset(&a.lut_dust, 0, sizeof(a.lut_dust)); memset(&a.dust_bands, 0, sizeof(a.dust_bands));
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
            o.media = nt.banded ? 3 : 2;
            a.lut_acc = make_lut(nt.d_cells, nt.acc, nt.acc_families);
            a.lut_dust = make_lut(nt.d_cells + dust_cell0(nt), nt.dust, nt.dust_families);
            a.dust_bands = make_bands(nt);
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
 * work for -- PLUS ONE spare unless its fullest round left a fifth of the pool free (a view that changes finds room; an idle
 * round costs four near-empty launches: every wave leaves on one scalar load) --, twice as many plus one if rays were still
 * suspended at its end.  Rays the enqueued
 * rounds do not finish take the in-line route: the bytes never depend on the guess. */
int auto_pool_rounds(const volatile DeferCounters* h, unsigned capacity) {
    if (!h) return 2;
    const unsigned run = h->rounds_run, work = h->rounds_with_work, left = h->suspended_left, peak = h->peak_blocks;
    if (run == 0u) return 2;                                    /* no history */
    int r = (int)(work > 0u ? work : 1u);
    /* round 5: no spare round while the pool has room to spare -- the previous launch finished everything and its fullest
     * round used at most 80 % of this launch's pool: a view that changes from one frame to the next does not outgrow that, and
     * the idle round's four near-empty launches are ~1 % of a rank's share of a frame.  (If it ever does overflow, those rays
     * finish in line -- same bytes -- and the next launch sees suspended_left != 0.) */
    if (left == 0u && (unsigned long long)peak * 5ull <= (unsigned long long)capacity * 4ull) return r > kMaxPoolRounds ? kMaxPoolRounds : r;
    r = left != 0u ? 2 * r + 1 : r + 1;
    return r > kMaxPoolRounds ? kMaxPoolRounds : r;
}

/* the kernels of one arithmetic mode, instantiated per (spin, tables) */
template <int ARITH>
int enqueue_chain_arith(const FrameArgs& a, int media, dim3 grid, dim3 block, int rounds, hipStream_t st) {
    const bool spin = a.spin != 0.0f;
    for (int r = 0; r < rounds; ++r) {
        const bool last = r == rounds - 1;
#define RRT_MARCH(S) do { if (r == 0) hipLaunchKernelGGL((march_defer<S, ARITH, false>), grid, block, 0, st, a); \
                          else hipLaunchKernelGGL((march_defer<S, ARITH, true>), grid, block, 0, st, a); } while (0)
        if (spin) RRT_MARCH(true); else RRT_MARCH(false);
#undef RRT_MARCH
        RRT_HIP(hipGetLastError());
        if (media == 3) hipLaunchKernelGGL((eval_sample_rows<ARITH, 3>), dim3(2048), dim3(256), 0, st, a);
        else if (media == 2) hipLaunchKernelGGL((eval_sample_rows<ARITH, 2>), dim3(2048), dim3(256), 0, st, a);
        else hipLaunchKernelGGL((eval_sample_rows<ARITH, 1>), dim3(2048), dim3(256), 0, st, a);
        RRT_HIP(hipGetLastError());
#define RRT_COMP3(S, L) do { if (last) hipLaunchKernelGGL((composite_and_shade<S, ARITH, L, true>), grid, block, 0, st, a); \
                             else hipLaunchKernelGGL((composite_and_shade<S, ARITH, L, false>), grid, block, 0, st, a); } while (0)
#define RRT_COMP(S) do { if (media == 3) RRT_COMP3(S, 3); else if (media == 2) RRT_COMP3(S, 2); else RRT_COMP3(S, 1); } while (0)
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
int enqueue_chain(FrameArgs a, int arith, int media, dim3 full_grid, int row0, int row_stride, int n_rows, int rounds, hipStream_t st) {
    if (n_rows <= 0) return RRT_OK;
    const dim3 block(kWGThreads), grid(full_grid.x, (unsigned)n_rows);
    a.grid_rows = (int)full_grid.y; a.grid_row_base = row0; a.grid_row_stride = row_stride;
    if (arith == kArithFast) return enqueue_chain_arith<kArithFast>(a, media, grid, block, rounds, st);
    if (arith == kArithFmad) return enqueue_chain_arith<kArithFmad>(a, media, grid, block, rounds, st);
    return enqueue_chain_arith<kArithStrict>(a, media, grid, block, rounds, st);
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
int launch_deferred(FrameArgs a, int arith, int media, const WorkspaceObject& ws, int pool_rounds, int chains_wanted, hipStream_t st) {
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
#ifdef RRT_WS_MEMSET      /* dev builds only: rounds 1-3's hipMemsetAsync, to re-examine round 4's capture failure (LABNOTES.md, round 5) */
    RRT_HIP(hipMemsetAsync(ws.d_base, 0, off_fin, st));
#else
    hipLaunchKernelGGL(zero_words, dim3((unsigned)((off_fin / 16 + 255) / 256)), dim3(256), 0, st,
                       reinterpret_cast<uint4*>(ws.d_base), off_fin / 16);        /* counters + wave headers (off_fin is a multiple of 256) */
    RRT_HIP(hipGetLastError());
#endif
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
        const int rc = enqueue_chain(b, arith, media, grid, row_first[c], row_step, row_count[c], rounds, cs);
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
        const int rc = launch_deferred(b, o.arith, o.media, ws, o.pool_rounds, o.pass_chains, st);
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
#define RRT_LAUNCH1(S) do { if (media == 3) RRT_LAUNCH2(S, 3); else if (media == 2) RRT_LAUNCH2(S, 2); else if (media == 1) RRT_LAUNCH2(S, 1); else RRT_LAUNCH2(S, 0); } while (0)
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

}  // namespace

/* ====================================================================== C ABI */
extern "C" {

int rrt_abi_version(void) { return RRT_ABI_VERSION; }

/* RRT_PATH_AUTO's threshold, for hosts that size their own policy on it (the drivers leave rrt_tile_order off for launches
 * that will take the three-pass path): one definition instead of a copy in every driver */
int rrt_path_auto_max_rays(void) { return (int)kThreePassMaxRays; }

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
    if (faked_device() < 0 && hipPointerGetAttributes(&attr, d_rgba8) == hipSuccess) device = attr.device;
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
}  // extern "C"
namespace {
/* rrt_launch_auto_resources (round 6): the objects the library owns on behalf of a host that only ever calls the
 * reference-signature launch_raymarch() */
struct AutoResources {
    bool on = false;
    int device = -1;
    int pool = 0, order = 0, table = 0;
    size_t table_budget = 0;
    float t0 = 0.0f, t1 = -1.0f;           /* the window the table (or the remembered "nothing fits") covers; empty at first */
    int table_builds = 0;
    rrt_params base;
};
std::mutex g_auto_mu;
AutoResources g_auto;

void auto_release_locked() {
    if (g_auto.table) rrt_noise_table_destroy(g_auto.table);
    if (g_auto.order) rrt_tile_order_destroy(g_auto.order);
    if (g_auto.pool) rrt_workspace_destroy(g_auto.pool);
    g_auto = AutoResources();
}

/* the parameters of one launch_raymarch() call under auto resources; slides the table's window when `time` has left it */
bool auto_params(float time, rrt_params& prm) {
    std::lock_guard<std::mutex> lk(g_auto_mu);
    if (!g_auto.on) return false;
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev != g_auto.device) return false;       /* another device is current: plain defaults */
    if (g_auto.table_budget > 0 && g_auto.base.volumetrics && !(time >= g_auto.t0 && time <= g_auto.t1)) {
        /* The one place a launch_raymarch() call waits for the device and allocates -- opted into by the caller: the frames that
         * may still read the old table drain, the next window is fitted to the budget and built (milliseconds). */
        (void)hipDeviceSynchronize();
        if (g_auto.table) { rrt_noise_table_destroy(g_auto.table); g_auto.table = 0; }
        float t1 = time; int cov = RRT_TABLE_FULL; size_t bytes = 0;
        rrt_noise_table_fit_window(time, time + 120.0f, g_auto.table_budget, &t1, &cov, &bytes);
        if (bytes != 0 && rrt_noise_table_create_window(time, t1, cov, &g_auto.table) == RRT_OK) ++g_auto.table_builds;
        else { g_auto.table = 0; if (bytes == 0) t1 = time + 5.0f; }                  /* nothing fits / no memory: look again after 5 s */
        g_auto.t0 = time; g_auto.t1 = t1;
    }
    prm = g_auto.base;
    prm.workspace = g_auto.pool; prm.tile_order = g_auto.order; prm.noise_table = g_auto.table;
    return true;
}
}  // namespace
extern "C" {

int rrt_launch_auto_resources(int on, const rrt_params* base, size_t table_budget_bytes, size_t pool_bytes) {
    std::lock_guard<std::mutex> lk(g_auto_mu);
    if (g_auto.on) {                                   /* a second call replaces the first's objects; both from their device */
        int dev = -1;
        if (hipGetDevice(&dev) != hipSuccess || dev != g_auto.device) return RRT_ERR_BAD_HANDLE;
        (void)hipDeviceSynchronize();
        auto_release_locked();
    }
    if (!on) return RRT_OK;
    rrt_params b;
    rrt_params_init(&b, (uint32_t)sizeof(rrt_params));
    if (base) {
        int rc = check_params_abi(base);
        if (rc == RRT_OK) rc = check_params_values(base);
        if (rc != RRT_OK) return rc;
        load_params(base, b);
    }
    AutoResources a;
    a.base = b;
    if (hipGetDevice(&a.device) != hipSuccess) return RRT_ERR_NO_DEVICE;
    a.table_budget = table_budget_bytes;
    int rc = RRT_OK;
    if (pool_bytes > 0) rc = rrt_workspace_create(pool_bytes, &a.pool);
    if (rc == RRT_OK) rc = rrt_tile_order_create(&a.order);
    if (rc != RRT_OK) {
        if (a.pool) rrt_workspace_destroy(a.pool);
        return rc;
    }
    a.on = true;
    g_auto = a;
    return RRT_OK;
}

int rrt_launch_auto_resources_info(int* on, int* table_builds, float* table_t0, float* table_t1, size_t* table_bytes) {
    std::lock_guard<std::mutex> lk(g_auto_mu);
    if (on) *on = g_auto.on ? 1 : 0;
    if (table_builds) *table_builds = g_auto.table_builds;
    if (table_t0) *table_t0 = g_auto.t0;
    if (table_t1) *table_t1 = g_auto.t1;
    if (table_bytes) {
        *table_bytes = 0;
        if (g_auto.table) rrt_noise_table_info(g_auto.table, nullptr, table_bytes, nullptr);
    }
    return RRT_OK;
}

int rrt_launch_raymarch_compat(void* d_out_rgba8, int width, int height, float time, const float* cam12,
                               rrt_sky_t sky, const void* effects36) {
    if (!cam12 || !effects36) return RRT_ERR_INVALID_ARGUMENT;
    rrt_camera c;
    memcpy(&c, cam12, sizeof(c));
    rrt_effects fx;
    memcpy(&fx, effects36, sizeof(fx));
    rrt_params prm;
    if (!auto_params(time, prm)) rrt_get_launch_defaults(&prm);
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
    auto fill_box = [&](const LutBox& bx, size_t cell0) {
        const size_t n = (size_t)bx.nx * bx.ny * bx.nz;
        hipLaunchKernelGGL(build_noise_table, dim3((unsigned)std::min<size_t>(4096, (n + 255) / 256)), dim3(256), 0, nullptr, nt.d_cells + cell0, bx);
    };
    auto record = [&](const LutBox& bx, unsigned cell0) {
        const NoiseLut L = make_lut(nullptr, bx, 0u);
        rrt::BandLut r;
        memset(&r, 0, sizeof(r));
        r.cell0 = cell0; r.origin = L.origin; r.nx = L.nx; r.nxy = L.nxy; r.last = L.last;
        return r;
    };
    if (!nt.banded) fill_box(nt.acc, 0);
    fill_box(nt.dust, dust_cell0(nt));
    e = hipGetLastError();
    if (e == hipSuccess && nt.banded) {
        std::vector<rrt::BandLut> recs((size_t)rrt::kBandFamilies * nt.bands.n_bands + rrt::kLutAccOctaves);
        for (int o = 0; o < rrt::kLutAccOctaves; ++o) {
            fill_box(nt.bands.acc_box[o], nt.bands.acc_cell0[o]);
            recs[(size_t)rrt::kBandFamilies * nt.bands.n_bands + o] = record(nt.bands.acc_box[o], nt.bands.acc_cell0[o]);
        }
        for (int f = 0; f < rrt::kBandFamilies; ++f)
            for (int b = 0; b < nt.bands.n_bands; ++b) {
                fill_box(nt.bands.box[f][b], nt.bands.cell0[f][b]);
                recs[(size_t)f * nt.bands.n_bands + b] = record(nt.bands.box[f][b], nt.bands.cell0[f][b]);
            }
        e = hipGetLastError();
        if (e == hipSuccess) e = hipMemcpy(reinterpret_cast<char*>(nt.d_cells) + nt.bands.entries_offset, recs.data(),
                                           recs.size() * sizeof(rrt::BandLut), hipMemcpyHostToDevice);
    }
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
    if (coverage) *coverage = it->second.coverage | (it->second.banded ? RRT_TABLE_BANDED : 0);
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

int rrt_noise_table_plan_layout(float t0, float t1, int coverage, int* banded, int* n_bands, float* w_min, float* w_scale,
                                int32_t* band_boxes, int cap_bands, int32_t* acc_octave_boxes) {
    if (!banded || !n_bands) return RRT_ERR_INVALID_ARGUMENT;
    NoiseTableObject nt;
    const int rc = plan_table(t0, t1, coverage, nt);
    if (rc != RRT_OK) return rc;
    *banded = nt.banded ? 1 : 0;
    *n_bands = nt.banded ? nt.bands.n_bands : 0;
    if (w_min) *w_min = nt.bands.w_min;
    if (w_scale) *w_scale = nt.bands.w_scale;
    if (nt.banded && band_boxes) {
        if (cap_bands < nt.bands.n_bands) return RRT_ERR_INVALID_ARGUMENT;
        for (int f = 0; f < rrt::kBandFamilies; ++f)
            for (int b = 0; b < nt.bands.n_bands; ++b) {
                const LutBox& bx = nt.bands.box[f][b];
                const int v[6] = {bx.x0, bx.y0, bx.z0, bx.nx, bx.ny, bx.nz};
                memcpy(band_boxes + ((size_t)f * cap_bands + b) * 6, v, sizeof(v));
            }
    }
    if (nt.banded && acc_octave_boxes)
        for (int o = 0; o < rrt::kLutAccOctaves; ++o) {
            const LutBox& bx = nt.bands.acc_box[o];
            const int v[6] = {bx.x0, bx.y0, bx.z0, bx.nx, bx.ny, bx.nz};
            memcpy(acc_octave_boxes + (size_t)o * 6, v, sizeof(v));
        }
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
        if (!on_current_device(m->device)) return RRT_ERR_BAD_HANDLE;      /* like every other handle: freed under the device that owns it */
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

#ifdef RRT_TEST_HOOKS
#define RRT_TEST_HOOKS_PART 2
#include "rrt_test_hooks.h"
#undef RRT_TEST_HOOKS_PART
#endif

}  // extern "C"

/* The reference's entry point as a linkable C++ symbol (include/raymarcher.h:19, src/raymarcher.cu:176-180),
 * for translation units built against THIS repository's include/raymarcher.h (HIP vector types).  The twin
 * under the reference's own mangled name (CUDA's `struct uchar4`) is csrc/rrt_compat.cpp. */
#include "../../include/raymarcher.h"
void launch_raymarch(uchar4* d_out, int w, int h, float time, CameraState cam, cudaTextureObject_t skyboxTex,
                     CameraEffects effects) {
    (void)rrt_launch_raymarch_compat(d_out, w, h, time, reinterpret_cast<const float*>(&cam), skyboxTex, &effects);
}

