// rrt_headless.cpp -- the reference's main loop (src/main.cpp:482-539) without a window, in C++, on top of
// the C ABI (include/rrt.h).  Per frame k = 1..N exactly what main() does while recording: advance the
// fixed 1/24 s float clock (main.cpp:511-516), take the camera from the active path (getInterpolatedState,
// :176-203) or the start-up camera (:128-130), launch_raymarch (:467), hand the pixels to the recorder
// (captureFrame, :85-97 -- here a raw RGBA stream, the bytes the reference pipes into ffmpeg).
//
//   rrt_headless --width 1000 --height 700 --frames 24 --path 0 --spin 0.9 --out frames.rgba [--sky-seed 1 | --sky file.rrtsky]
//                [--gpus N] [--tile-rows 16] [--workspace-gib G] [--noise-table-gib B | --no-noise-table]
//                [--arith strict|fmad|fast] [--path-window F | --path-window -1 | --path-policy auto|single|three-pass]
//                [--init-timeout 300] [--frame-timeout 120]      (watchdog, seconds; exit status 3 when it fires)
//
// Noise tables: the reference's simTime runs without bound (main.cpp:515) and a table's size grows with the times
// it covers, so each device keeps ONE table over a window of the clock that fits --noise-table-gib (default 2;
// rrt_noise_table_fit_window picks window and coverage) and rebuilds it -- milliseconds -- when the clock leaves the
// window.  Frames that had to render without a table are counted in the summary line, never silent.
//
// Multi-GPU (the reference is single-GPU): ONE process drives N devices.  ncclCommInitAll
// gives one RCCL communicator per device; image tile t (rows [16t, 16t+16)) belongs to device t mod N; every
// frame each device renders its tiles into a compact buffer (rrt_launch_raymarch_tiles), ONE grouped
// ncclSend / ncclRecv exchange -- a gather over xGMI -- lands all shards in one allocation on device 0, and
// device 0 scatters them into the bottom-up frame with one launch (rrt_assemble_all_tiles).  Two frames are in
// flight: frame k lives on stream k mod 2 of every device (own tile / gather / frame buffers and own pool), so
// the next frame's wavefronts fill the drain of this one's kernels and the exchange sits under marching.
// The same scheme as relativisticraytracer_amd/sharding.py (one process per GPU over torch.distributed), which
// bench.py uses; the bytes are identical (tests/test_headless.py).
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <dlfcn.h>
#include <fcntl.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/rrt.h"

namespace {

// the bench/test sky (relativisticraytracer_amd/sky.py: synthetic_sky), integer arithmetic only
uint32_t mix(uint64_t a) {
    a &= 0xFFFFFFFFull;
    a = ((a ^ (a >> 16)) * 0x7FEB352Dull) & 0xFFFFFFFFull;
    a = ((a ^ (a >> 15)) * 0x846CA68Bull) & 0xFFFFFFFFull;
    a = a ^ (a >> 16);
    return (uint32_t)a;
}
std::vector<uint8_t> synthetic_sky(int w, int h, int seed) {
    std::vector<uint8_t> out((size_t)w * h * 4);
    for (int64_t j = 0; j < h; ++j)
        for (int64_t i = 0; i < w; ++i) {
            auto tri = [](int64_t v) { int64_t m = v % 512; if (m < 0) m += 512; m -= 256; return m < 0 ? -m : m; };
            int64_t ti = tri(i * 4 * 256 / w), tj = tri(j * 2 * 256 / h);
            int64_t band = 96 - (j - h / 2 < 0 ? h / 2 - j : j - h / 2) * 96 * 6 / h;
            if (band < 0) band = 0;
            if (band > 96) band = 96;
            int64_t r = 10 + ti * 30 / 256 + band * 2 / 3, g = 12 + tj * 26 / 256 + band / 2, b = 28 + (ti + tj) * 20 / 256 + band;
            uint32_t hsh = mix((uint64_t)(i + j * w + (int64_t)seed * 0x9E3779B1ll));
            if (hsh % 641u == 0) {
                int64_t mag = 96 + (hsh >> 11) % 160u, tint = (hsh >> 20) % 48u;
                r = mag + tint > 255 ? 255 : mag + tint; g = mag; b = mag + 48 - tint > 255 ? 255 : mag + 48 - tint;
            }
            auto cl = [](int64_t v) { return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); };
            uint8_t* t = &out[4 * ((size_t)j * w + i)];
            t[0] = cl(r); t[1] = cl(g); t[2] = cl(b); t[3] = 255;
        }
    return out;
}

// A raw sky (relativisticraytracer_amd/sky.py: save_sky_raw, tools/sky_to_raw.py): "RRTSKY1\n", "<width> <height>\n", then height x width
// RGBA8 texels, row 0 = top -- the bytes the reference's stbi_load(..., 4) returns (src/main.cpp:240), decoded ONCE with the reference's
// own decoder and shipped (JPEG decoders differ: SURVEY.md row f1).  Returns false with a message on anything else.
bool load_sky_raw(const std::string& path, std::vector<uint8_t>& texels, int& w, int& h) {
    FILE* fh = fopen(path.c_str(), "rb");
    if (!fh) { fprintf(stderr, "rrt_headless: cannot open sky %s\n", path.c_str()); return false; }
    char magic[8] = {};
    bool ok = fread(magic, 1, 8, fh) == 8 && memcmp(magic, "RRTSKY1\n", 8) == 0;
    char line[64] = {};
    long long lw = 0, lh = 0;
    ok = ok && fgets(line, sizeof(line), fh) != nullptr && sscanf(line, "%lld %lld", &lw, &lh) == 2 && lw > 0 && lh > 0 && lw * lh <= (1ll << 30);
    if (ok) {
        texels.resize((size_t)(lw * lh * 4));
        ok = fread(texels.data(), 1, texels.size(), fh) == texels.size() && fgetc(fh) == EOF;
    }
    fclose(fh);
    if (!ok) { fprintf(stderr, "rrt_headless: %s is not a raw sky (RRTSKY1 header, width height, RGBA8 texels)\n", path.c_str()); return false; }
    w = (int)lw; h = (int)lh;
    return true;
}

// Progress trace + watchdog.  Every phase of the driver passes a trace point; the last kTraceKeep of them are kept in
// memory (RRT_HEADLESS_TRACE=1 also prints them as they happen).  A watchdog thread looks at the time since the last
// trace point: it says on stderr every 30 s where the driver is waiting, and once the wait exceeds the limit of the
// phase (--init-timeout for the bring-up -- communicator included --, --frame-timeout once frames are being rendered) it
// prints the kept trace and every communicator's ncclCommGetAsyncError and _exit()s with status 3: a sick node or a stuck
// collective ends the run with a diagnosis instead of hanging it (no retry, nothing is re-executed).
// Why the bring-up gets its own, longer limit: /opt/rocm/lib/librccl.so is a 573 MB fat binary; the first collective of a
// process page-faults the gfx950 code object out of it, which took minutes on a box with a cold page cache (the one
// abnormal end of round 3) -- warm_library_pages() below turns that into one sequential read.
constexpr int kTraceKeep = 96;
struct TraceState {
    std::mutex mu;
    std::string lines[kTraceKeep];
    unsigned long long n = 0;
    std::atomic<long long> last_ns{0};
    std::atomic<const char*> phase{"start-up"};
    std::atomic<bool> in_frames{false}, stop{false};
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    std::vector<ncclComm_t> comms;              // filled once, before in_frames (read by the watchdog on expiry only)
};
TraceState g_tr;

long long now_ns() {
    return std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - g_tr.t0).count();
}
void trace(const char* what, int k = -1) {
    static const bool on = getenv("RRT_HEADLESS_TRACE") != nullptr;
    const long long t = now_ns();
    char buf[160];
    if (k >= 0) snprintf(buf, sizeof(buf), "[rrt_headless %8.3f s] %s %d", t * 1e-9, what, k);
    else snprintf(buf, sizeof(buf), "[rrt_headless %8.3f s] %s", t * 1e-9, what);
    {
        std::lock_guard<std::mutex> lk(g_tr.mu);
        g_tr.lines[g_tr.n++ % kTraceKeep] = buf;
    }
    g_tr.phase.store(what);
    g_tr.last_ns.store(t);
    if (on) { fprintf(stderr, "%s\n", buf); fflush(stderr); }
}
void dump_trace() {
    std::lock_guard<std::mutex> lk(g_tr.mu);
    const unsigned long long first = g_tr.n > kTraceKeep ? g_tr.n - kTraceKeep : 0;
    for (unsigned long long i = first; i < g_tr.n; ++i) fprintf(stderr, "  %s\n", g_tr.lines[i % kTraceKeep].c_str());
}
void watchdog(double init_limit_s, double frame_limit_s) {
    long long said = 0;
    while (!g_tr.stop.load()) {
        std::this_thread::sleep_for(std::chrono::milliseconds(200));
        const long long idle = now_ns() - g_tr.last_ns.load();
        const double limit = g_tr.in_frames.load() ? frame_limit_s : init_limit_s;
        if (idle > (said + 1) * 30000000000ll) {
            ++said;
            fprintf(stderr, "rrt_headless: %.0f s in \"%s\" (limit %.0f s)\n", idle * 1e-9, g_tr.phase.load(), limit);
            fflush(stderr);
        }
        if (idle < 30000000000ll) said = 0;
        if (limit > 0.0 && idle * 1e-9 > limit) {
            fprintf(stderr, "rrt_headless: no progress for %.0f s in \"%s\" -- giving up.  Last trace points:\n", idle * 1e-9,
                    g_tr.phase.load());
            dump_trace();
            std::vector<ncclComm_t> comms;
            { std::lock_guard<std::mutex> lk(g_tr.mu); comms = g_tr.comms; }
            for (size_t d = 0; d < comms.size(); ++d) {
                ncclResult_t async = ncclSuccess;
                const ncclResult_t q = ncclCommGetAsyncError(comms[d], &async);
                fprintf(stderr, "  communicator %zu: ncclCommGetAsyncError -> %s, async error: %s\n", d, ncclGetErrorString(q),
                        ncclGetErrorString(async));
            }
            fflush(stderr);
            _exit(3);
        }
    }
}

// Read a shared library's file once, front to back, so that its pages are in the page cache before the loader / the HIP
// runtime fault them in one by one (on a cold overlay file system the page-by-page path is the slow one).
void warm_library_pages(const void* symbol_in_library) {
    Dl_info info;
    if (!dladdr(symbol_in_library, &info) || !info.dli_fname) return;
    const int fd = open(info.dli_fname, O_RDONLY);
    if (fd < 0) return;
    (void)posix_fadvise(fd, 0, 0, POSIX_FADV_SEQUENTIAL);
    std::vector<char> buf(8u << 20);
    while (read(fd, buf.data(), buf.size()) > 0) {}
    close(fd);
}

int fail(const char* what, int rc) {
    fprintf(stderr, "rrt_headless: %s: %s (%s)\n", what, rrt_status_string(rc), rrt_last_hip_error());
    return 1;
}
#define HIPCHK(call)                                                                                          \
    do {                                                                                                      \
        hipError_t e_ = (call);                                                                               \
        if (e_ != hipSuccess) { fprintf(stderr, "rrt_headless: %s: %s\n", #call, hipGetErrorString(e_)); return 1; } \
    } while (0)
#define NCCLCHK(call)                                                                                          \
    do {                                                                                                       \
        ncclResult_t r_ = (call);                                                                              \
        if (r_ != ncclSuccess) { fprintf(stderr, "rrt_headless: %s: %s\n", #call, ncclGetErrorString(r_)); return 1; } \
    } while (0)

constexpr int kMaxSlots = 4;   // capacity of the per-slot arrays; --frames-in-flight picks 1..kMaxSlots of them

struct Device {                // everything one GPU owns
    int id = 0;
    rrt_sky_t sky = 0;
    int noise_table = 0;                          // this device's copy of the current window's table (0: none)
    int pool[kMaxSlots] = {};
    int order[kMaxSlots] = {};                    // rrt_tile_order per slot (0: static dispatch order)
    hipStream_t stream[kMaxSlots] = {};
    void* tiles[kMaxSlots] = {};                  // this device's shard of a frame
    int shard_rows = 0;
    ncclComm_t comm = nullptr;
    void* probe = nullptr;                        // 256 B to send + 256 B per peer to receive: the bring-up exchange
    hipEvent_t comm_free = nullptr;               // after this device's part of the last exchange (orders the next one behind it)
    // per-window path choice (include/rrt.h: rrt_path_chooser_*): the end of every frame's render on this device, kept for
    // 2 * kMaxSlots frames so that frame k - 1's event is still frame k - 1's when frame k is delivered
    int chooser = 0;
    hipEvent_t render_end[2 * kMaxSlots] = {};
};

}  // namespace

int main(int argc, char** argv) {
    int w = 1000, h = 700, frames = 24, fps = 24, path = -1, sky_seed = 1, all_fx = 0;   // config.h:7-9
    int arith = RRT_ARITH_STRICT;  // --arith strict | fmad | fast (--fast = --arith fast)
    int path_policy = -1;          // --path-policy auto|single|three-pass pins rrt_params.path_policy for every launch (no per-window choice)
    int path_window = 0;           // frames per window of the per-rank path choice (0: the library's default, 48); -1: no choice, the
                                   // three-pass path for every small share as in rounds 1-5
    int gpus = 1, tile_rows = 16, workspace_gib = 2, use_table = 1, force_collective = 0;
    int tile_order = -1;           // cost-ordered dispatch: -1 auto (on when frames are rendered one at a time), 0 off, 1 on
    int kSlots = 3;                // frames in flight: frame k renders on stream k mod kSlots while its predecessors are
                                   // gathered / assembled / copied out (a rank's share of a frame is only a few rounds of
                                   // wavefronts; 3 measured best at 8 shards of a 4K frame: profiles/r02_frames_in_flight.txt)
    float spin = 0.0f;
    double table_gib = 2.0;
    double init_timeout = 300.0, frame_timeout = 120.0;       // watchdog limits in seconds (0: none)
    std::string out_path, sky_path;
    for (int i = 1; i < argc; ++i) {
        std::string a = argv[i];
        auto val = [&](int& dst) { if (i + 1 < argc) dst = atoi(argv[++i]); };
        if (a == "--width") val(w); else if (a == "--height") val(h); else if (a == "--frames") val(frames);
        else if (a == "--fps") val(fps); else if (a == "--path") val(path); else if (a == "--sky-seed") val(sky_seed);
        else if (a == "--gpus") val(gpus); else if (a == "--tile-rows") val(tile_rows); else if (a == "--workspace-gib") val(workspace_gib);
        else if (a == "--frames-in-flight") val(kSlots);
        else if (a == "--spin" && i + 1 < argc) spin = (float)atof(argv[++i]);
        else if (a == "--noise-table-gib" && i + 1 < argc) table_gib = atof(argv[++i]);
        else if (a == "--init-timeout" && i + 1 < argc) init_timeout = atof(argv[++i]);
        else if (a == "--frame-timeout" && i + 1 < argc) frame_timeout = atof(argv[++i]);
        else if (a == "--no-noise-table") use_table = 0;
        else if (a == "--tile-order") tile_order = 1; else if (a == "--no-tile-order") tile_order = 0;
        else if (a == "--force-collective") force_collective = 1;     // run the RCCL exchange even with one GPU (self-check)
        else if (a == "--out" && i + 1 < argc) out_path = argv[++i];
        else if (a == "--sky" && i + 1 < argc) sky_path = argv[++i];
        else if (a == "--all-effects") all_fx = 1; else if (a == "--fast") arith = RRT_ARITH_FAST;
        else if (a == "--path-window") val(path_window);
        else if (a == "--path-policy" && i + 1 < argc) {
            const std::string m = argv[++i];
            if (m == "auto") path_policy = RRT_PATH_AUTO; else if (m == "single") path_policy = RRT_PATH_SINGLE; else if (m == "three-pass") path_policy = RRT_PATH_THREE_PASS;
            else { fprintf(stderr, "--path-policy auto | single | three-pass\n"); return 2; }
            path_window = -1;
        }
        else if (a == "--arith" && i + 1 < argc) {
            const std::string m = argv[++i];
            if (m == "strict") arith = RRT_ARITH_STRICT; else if (m == "fmad") arith = RRT_ARITH_FMAD; else if (m == "fast") arith = RRT_ARITH_FAST;
            else { fprintf(stderr, "--arith strict | fmad | fast\n"); return 2; }
        }
        else { fprintf(stderr, "unknown argument %s\n", a.c_str()); return 2; }
    }
    if (w <= 0 || h <= 0 || frames < 0 || fps <= 0 || gpus < 1 || tile_rows < 1 || kSlots < 1 || kSlots > kMaxSlots) {
        fprintf(stderr, "bad arguments\n"); return 2;
    }
    int n_dev = 0, rc;
    if ((rc = rrt_device_count(&n_dev)) != RRT_OK) return fail("no GPU", rc);
    if (gpus > n_dev) { fprintf(stderr, "rrt_headless: --gpus %d but %d device(s) visible\n", gpus, n_dev); return 2; }
    const bool collective = gpus > 1 || force_collective;
    trace("start");
    std::thread dog(watchdog, init_timeout, frame_timeout);
    struct DogStop { std::thread& t; ~DogStop() { g_tr.stop.store(true); t.join(); } } dog_stop{dog};      // every return path
    // the RCCL library's pages, read sequentially in the background while the per-device resources are set up
    std::thread warm;
    if (collective && !getenv("RRT_NO_LIBRARY_WARMUP")) warm = std::thread(warm_library_pages, (const void*)&ncclCommInitAll);
    struct WarmJoin { std::thread& t; ~WarmJoin() { if (t.joinable()) t.join(); } } warm_join{warm};
    // a failing communicator says why (RCCL prints nothing below WARN); the caller's setting wins
    if (collective) setenv("NCCL_DEBUG", "WARN", 0);
    // one process, one node: RCCL's bootstrap and RAS sockets need no interface but the loopback (by default it picks
    // the first non-loopback one -- in a container a veth whose state is not this program's business)
    if (collective) setenv("NCCL_SOCKET_IFNAME", "lo", 0);

    // the recording clock (main.cpp:511-516) says which times the sequence reaches; the tables slide along it
    float seq_end = 0.0f;
    { float unused = 0.0f; rrt_recording_clock(frames, fps, &seq_end, &unused); seq_end += 1.0f; }
    const size_t table_budget = table_gib > 0.0 ? (size_t)(table_gib * (double)(1ull << 30)) : 0;
    if (table_budget == 0) use_table = 0;
    int table_builds = 0, table_frames = 0, arith_frames = 0, coarsest = RRT_TABLE_FULL;
    size_t table_peak = 0;
    bool table_warned = false;
    float win_t0 = 0.0f, win_t1 = -1.0f;         // the window the devices' tables (or the remembered failure) cover; empty at first
    bool win_has_table = false;

    std::vector<uint8_t> sky;
    int sky_w = 2048, sky_h = 1024;
    if (sky_path.empty()) sky = synthetic_sky(sky_w, sky_h, sky_seed);
    else if (!load_sky_raw(sky_path, sky, sky_w, sky_h)) return 2;
    rrt_effects fx; rrt_effects_default(&fx);
    fx.use_chromatic_aberration = (uint8_t)all_fx;

    // ---- per-device resources
    std::vector<Device> dev(gpus);
    size_t shard_stride = 0;                       // bytes between shards in the gathered buffer
    for (int d = 0; d < gpus; ++d) {
        int rows = 0;
        rrt_tile_shard_rows(h, tile_rows, d, gpus, &rows);
        dev[d].shard_rows = rows;
        if ((size_t)rows * w * 4 > shard_stride) shard_stride = (size_t)rows * w * 4;
    }
    shard_stride = (shard_stride + 255) & ~(size_t)255;
    for (int d = 0; d < gpus; ++d) {
        Device& D = dev[d];
        D.id = d;
        HIPCHK(hipSetDevice(d));
        const int rows = collective ? D.shard_rows : h;            // image rows of one launch on this device
        if ((rc = rrt_sky_create(sky.data(), sky_w, sky_h, &D.sky)) != RRT_OK) return fail("sky", rc);
        HIPCHK(hipMalloc(&D.probe, 256 * (size_t)(gpus + 1)));
        HIPCHK(hipEventCreateWithFlags(&D.comm_free, hipEventDisableTiming));
        // the path is a choice only where a launch could take either: a pool, a share under RRT_PATH_AUTO's threshold, frames in
        // flight (one frame at a time the three-pass path's two chains win: DESIGN.md section 5)
        if (path_window >= 0 && kSlots >= 2 && workspace_gib > 0 && (long long)w * rows <= (long long)rrt_path_auto_max_rays()) {
            if ((rc = rrt_path_chooser_create(kSlots, path_window, &D.chooser)) != RRT_OK) return fail("path chooser", rc);
            for (int e = 0; e < 2 * kSlots; ++e) HIPCHK(hipEventCreate(&D.render_end[e]));
        }
        for (int s = 0; s < kSlots; ++s) {
            HIPCHK(hipStreamCreateWithFlags(&D.stream[s], hipStreamNonBlocking));
            HIPCHK(hipMalloc(&D.tiles[s], shard_stride));
            if (workspace_gib > 0 && (rc = rrt_workspace_create(((size_t)workspace_gib << 30) / kSlots, &D.pool[s])) != RRT_OK)
                return fail("workspace", rc);
            // every frame's wave tiles dispatched longest-first by what the slot's previous frame measured (the first frame: by
            // the library's probe of the view): worth it when the frames do not overlap (their drains are exposed) AND the
            // launch takes the single kernel; a wash when frames overlap, and measured harmful on the three-pass path's two
            // chains (it piles every expensive tile into the first chain: an eighth of a 4K frame 5.15 -> 5.45 ms, of a
            // disk-heavy one 7.7 -> 10.6; profiles/r04_shard_kernel_times_*.txt), which is what small launches with a pool take
            const bool three_pass_likely = workspace_gib > 0 && (long long)w * rows <= (long long)rrt_path_auto_max_rays();   // RRT_PATH_AUTO's own threshold
            if ((tile_order == 1 || (tile_order < 0 && kSlots == 1 && !three_pass_likely)) && (rc = rrt_tile_order_create(&D.order[s])) != RRT_OK)
                return fail("tile order", rc);
        }
    }
    trace("per-device resources ready");
    if (collective) {
        std::vector<int> ids(gpus);
        std::vector<ncclComm_t> comms(gpus);
        for (int d = 0; d < gpus; ++d) ids[d] = d;
        if (warm.joinable()) { trace("waiting for the RCCL library pages"); warm.join(); }
        trace("ncclCommInitAll ...");
        NCCLCHK(ncclCommInitAll(comms.data(), gpus, ids.data()));
        trace("ncclCommInitAll done");
        for (int d = 0; d < gpus; ++d) dev[d].comm = comms[d];
        { std::lock_guard<std::mutex> lk(g_tr.mu); g_tr.comms = comms; }
        // One untimed exchange of a few bytes brings up the peer-to-peer channels and loads RCCL's code object now, under
        // the bring-up limit, instead of under frame 1's.
        trace("first exchange (channel set-up) ...");
        NCCLCHK(ncclGroupStart());
        for (int d = 0; d < gpus; ++d) {
            NCCLCHK(ncclSend(dev[d].probe, 256, ncclUint8, 0, dev[d].comm, dev[d].stream[0]));
            NCCLCHK(ncclRecv(static_cast<uint8_t*>(dev[0].probe) + 256 * (size_t)(d + 1), 256, ncclUint8, d, dev[0].comm, dev[0].stream[0]));
        }
        NCCLCHK(ncclGroupEnd());
        for (int d = 0; d < gpus; ++d) { HIPCHK(hipSetDevice(d)); HIPCHK(hipStreamSynchronize(dev[d].stream[0])); }
        trace("first exchange done");
    }
    // device 0: gathered shards + assembled frame, per slot; pinned host frames for the sink
    HIPCHK(hipSetDevice(0));
    void* gathered[kMaxSlots] = {};
    void* frame[kMaxSlots] = {};
    void* host[kMaxSlots] = {};
    hipEvent_t done[kMaxSlots];
    const size_t frame_bytes = (size_t)w * h * 4;
    for (int s = 0; s < kSlots; ++s) {
        HIPCHK(hipMalloc(&gathered[s], shard_stride * gpus));
        HIPCHK(hipMalloc(&frame[s], frame_bytes));
        HIPCHK(hipEventCreateWithFlags(&done[s], hipEventDisableTiming));
    }
    FILE* f = out_path.empty() ? nullptr : fopen(out_path.c_str(), "wb");
    if (!out_path.empty() && !f) { perror("fopen"); return 1; }
    if (f) for (int s = 0; s < kSlots; ++s) HIPCHK(hipHostMalloc(&host[s], frame_bytes, hipHostMallocDefault));

    rrt_camera cam;
    const float start_pos[3] = {0.0f, 10.0f, -60.0f};
    rrt_camera_from_angles(start_pos, 0.0f, -10.0f, &cam);
    const char* path_name = "";
    if (path >= 0 && (rc = rrt_path_info(path, &path_name, nullptr, nullptr)) != RRT_OK) return fail("path", rc);

    // write frame `k`'s pixels once its copy has landed (called one frame late, so that the copy overlaps the next render)
    // (without a sink the wait still happens: the host never runs more than kSlots frames ahead of the device, a device
    // fault is reported at the frame it belongs to, and the watchdog sees frames complete)
    int delivered = 0;
    auto deliver = [&](int slot) -> int {
        HIPCHK(hipEventSynchronize(done[slot]));
        trace("frame complete", ++delivered);
        // frame `delivered` has been rendered on every device (the gather waited for all shards): its sustained time on a device =
        // the interval between the ends of frame delivered - 1's and its own render there
        if (delivered >= 2) for (int d = 0; d < gpus; ++d) if (dev[d].chooser) {
            float ms = 0.0f;
            HIPCHK(hipSetDevice(d));
            if (hipEventElapsedTime(&ms, dev[d].render_end[(delivered - 1) % (2 * kSlots)], dev[d].render_end[delivered % (2 * kSlots)]) == hipSuccess)
                rrt_path_chooser_report(dev[d].chooser, delivered, ms > 0.0f ? ms : 0.0f);   // (a frame that ended before its predecessor: 0)
        }
        if (f && fwrite(host[slot], 1, frame_bytes, f) != frame_bytes) fprintf(stderr, "Warning: Frame write incomplete\n");
        return 0;
    };

    for (int d = 0; d < gpus; ++d) { HIPCHK(hipSetDevice(d)); HIPCHK(hipDeviceSynchronize()); }
    trace("bring-up complete; rendering");
    g_tr.in_frames.store(true);
    auto t0 = std::chrono::steady_clock::now();
    for (int k = 1; k <= frames; ++k) {
        const int slot = k % kSlots;
        float sim_t = 0.0f, path_t = 0.0f;
        rrt_recording_clock(k, fps, &sim_t, &path_t);
        if (path >= 0 && (rc = rrt_path_camera_at(path, path_t, &cam)) != RRT_OK) return fail("camera", rc);
        trace("frame", k);
        // slot reuse: frame k-kSlots used the same buffers; its host copy must have been written out
        if (k > kSlots && deliver(slot)) return 1;
        // 0. noise tables: when the clock has left the window, every device builds the next one (after its frames in
        //    flight, which may still read the old table, have drained).  The window is ONE decision for all devices: if it
        //    cannot be built on any of them (nothing fits the budget, out of memory) it is dropped on all of them and
        //    REMEMBERED -- no new fit, no device synchronise, no retry until the clock has left it (retrying
        //    every frame serialised the frames in flight and rebuilt the other devices' tables for nothing).
        const bool in_window = sim_t >= win_t0 && sim_t <= win_t1;
        if (use_table && !in_window) {
            float t1 = sim_t; int cov = RRT_TABLE_FULL; size_t bytes = 0;
            // trace points around the rebuild: it synchronises every device and creates a table on each, which on a
            // starved box takes time that must neither count against --frame-timeout under the name of the last frame nor
            // leave the watchdog's notices pointing at the wrong phase
            trace("noise table window: fitting", k);
            rrt_noise_table_fit_window(sim_t, seq_end > sim_t ? seq_end : sim_t, table_budget, &t1, &cov, &bytes);
            bool ok = bytes != 0;
            for (int d = 0; d < gpus; ++d) {
                Device& D = dev[d];
                HIPCHK(hipSetDevice(d));
                trace("noise table window: draining and rebuilding on device", d);
                HIPCHK(hipDeviceSynchronize());
                if (D.noise_table) { rrt_noise_table_destroy(D.noise_table); D.noise_table = 0; }
                if (ok && (rc = rrt_noise_table_create_window(sim_t, t1, cov, &D.noise_table)) != RRT_OK) {
                    D.noise_table = 0;
                    ok = false;
                    if (!table_warned) {
                        fprintf(stderr, "rrt_headless: noise table [%g, %g] not built on device %d: %s (%s); hashing arithmetically\n",
                                sim_t, t1, d, rrt_status_string(rc), rrt_last_hip_error());
                        table_warned = true;
                    }
                }
            }
            if (!ok) {      // all or nothing: the devices must agree on which kernels a frame runs for the timing to mean anything
                for (int d = 0; d < gpus; ++d) if (dev[d].noise_table) { HIPCHK(hipSetDevice(d)); rrt_noise_table_destroy(dev[d].noise_table); dev[d].noise_table = 0; }
                if (bytes == 0) t1 = sim_t + 5.0f;          // nothing fits: look again after 5 s of sim time
            }
            win_t0 = sim_t; win_t1 = t1; win_has_table = ok;
            trace("noise table window: done", k);
            if (ok) { ++table_builds; if (cov > coarsest) coarsest = cov; if (bytes > table_peak) table_peak = bytes; }
        }
        if (use_table && win_has_table) ++table_frames; else ++arith_frames;
        // 1. every device renders its tiles
        for (int d = 0; d < gpus; ++d) {
            Device& D = dev[d];
            HIPCHK(hipSetDevice(d));
            rrt_params prm; rrt_params_default(&prm);
            prm.spin = spin; prm.arith_mode = arith;
            prm.workspace = D.pool[slot]; prm.noise_table = D.noise_table; prm.tile_order = D.order[slot];
            // frames in flight fill each other's drains: ONE chain per launch (the second chain's streams only compete with the other
            // frames: 2-7 % per frame, profiles/r05_sustained_chains.txt).  The plain single kernel would be faster still on most views,
            // but its longest wavefront (up to 19 ms on a disk-grazing view) bounds a slot's frame rate; a moving camera keeps the
            // path that is never slow
            prm.pass_chains = kSlots >= 2 ? 1 : 0;
            // ... per window, by measurement: rrt_path_chooser (csrc/rrt_path_chooser.cpp) tries the single kernel for a few frames
            // of a window, keeps it where it sustains the faster frames, and drops it at once on a frame that takes > 1.5 x the
            // three-pass median (same bytes either way)
            if (D.chooser) { int pol = RRT_PATH_AUTO; rrt_path_chooser_policy(D.chooser, k, &pol); prm.path_policy = pol; }
            else if (path_policy >= 0) prm.path_policy = path_policy;
            void* dst = collective ? D.tiles[slot] : frame[slot];
            if (collective) rc = rrt_launch_raymarch_tiles(dst, w, h, tile_rows, d, gpus, sim_t, &cam, D.sky, &fx, &prm, D.stream[slot]);
            else rc = rrt_launch_raymarch(dst, w, h, sim_t, &cam, D.sky, &fx, &prm, D.stream[slot]);
            if (rc != RRT_OK) return fail("launch", rc);
            if (D.chooser) HIPCHK(hipEventRecord(D.render_end[k % (2 * kSlots)], D.stream[slot]));
        }
        // 2. one gather: every device sends its shard, device 0 receives all of them (its own included)
        if (collective) {
            // Successive frames use the SAME communicators from DIFFERENT streams (slot k mod kSlots).  RCCL orders the
            // operations of one communicator by itself; the event chain states that order in the stream graph as well:
            // a device's part of exchange k starts after its part of exchange k-1 has finished, whatever streams they ran on.
            if (k > 1) for (int d = 0; d < gpus; ++d) { HIPCHK(hipSetDevice(d)); HIPCHK(hipStreamWaitEvent(dev[d].stream[slot], dev[d].comm_free, 0)); }
            NCCLCHK(ncclGroupStart());
            for (int d = 0; d < gpus; ++d) {
                const size_t bytes = (size_t)dev[d].shard_rows * w * 4;
                if (bytes == 0) continue;
                NCCLCHK(ncclSend(dev[d].tiles[slot], bytes, ncclUint8, 0, dev[d].comm, dev[d].stream[slot]));
                NCCLCHK(ncclRecv(static_cast<uint8_t*>(gathered[slot]) + (size_t)d * shard_stride, bytes, ncclUint8, d, dev[0].comm,
                                 dev[0].stream[slot]));
            }
            NCCLCHK(ncclGroupEnd());
            for (int d = 0; d < gpus; ++d) { HIPCHK(hipSetDevice(d)); HIPCHK(hipEventRecord(dev[d].comm_free, dev[d].stream[slot])); }
            // 3. device 0 scatters the shards into the bottom-up frame
            HIPCHK(hipSetDevice(0));
            if ((rc = rrt_assemble_all_tiles(frame[slot], gathered[slot], shard_stride, w, h, tile_rows, gpus, dev[0].stream[slot])) != RRT_OK)
                return fail("assemble", rc);
        }
        // 4. hand the frame to the sink: asynchronous copy now, the write happens one frame later
        HIPCHK(hipSetDevice(0));
        if (f) HIPCHK(hipMemcpyAsync(host[slot], frame[slot], frame_bytes, hipMemcpyDeviceToHost, dev[0].stream[slot]));
        HIPCHK(hipEventRecord(done[slot], dev[0].stream[slot]));
    }
    trace("all frames enqueued; draining");
    // drain: the last min(frames, kSlots) frames, oldest first
    for (int k = frames - kSlots + 1; k <= frames; ++k)
        if (k >= 1 && deliver(k % kSlots)) return 1;
    for (int d = 0; d < gpus; ++d) { HIPCHK(hipSetDevice(d)); HIPCHK(hipDeviceSynchronize()); }
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (f) fclose(f);
    std::string choice = "null";
    if (dev[0].chooser) {
        choice = "[";
        for (int d = 0; d < gpus; ++d) {
            rrt_path_chooser_stats st;
            rrt_path_chooser_get_stats(dev[d].chooser, &st);
            char buf[256];
            snprintf(buf, sizeof(buf), "%s{\"device\": %d, \"frames_three_pass\": %d, \"frames_single_kernel\": %d, \"windows\": %d, \"trials\": %d, "
                     "\"trials_aborted\": %d, \"switches\": %d, \"outliers\": %d}", d ? ", " : "", d, st.frames[0], st.frames[1], st.windows, st.trials,
                     st.trials_aborted, st.switches, st.outliers);
            choice += buf;
        }
        choice += "]";
    }
    printf("{\"frames\": %d, \"width\": %d, \"height\": %d, \"n_gpus\": %d, \"seconds\": %.4f, \"fps\": %.3f, \"Mrays_per_s\": %.3f, "
           "\"path\": \"%s\", \"spin\": %g, \"arith_mode\": \"%s\", \"noise_tables\": {\"builds\": %d, \"table_frames\": %d, "
           "\"arith_frames\": %d, \"coarsest_coverage\": %d, \"peak_bytes\": %zu, \"budget_bytes\": %zu}, \"tile_order\": %s, \"collective\": \"%s\", "
           "\"path_choice\": %s}\n",
           frames, w, h, gpus, dt, frames / dt, (double)frames * w * h / dt / 1e6, path_name, spin,
           arith == RRT_ARITH_FAST ? "fast" : (arith == RRT_ARITH_FMAD ? "fmad" : "strict"),
           table_builds, table_frames, arith_frames, coarsest, table_peak, table_budget, dev[0].order[0] ? "true" : "false",
           collective ? "rccl grouped send/recv gather" : "none", choice.c_str());

    for (int d = 0; d < gpus; ++d) {
        Device& D = dev[d];
        HIPCHK(hipSetDevice(d));
        if (D.comm) ncclCommDestroy(D.comm);
        (void)hipFree(D.probe);
        (void)hipEventDestroy(D.comm_free);
        if (D.chooser) { rrt_path_chooser_destroy(D.chooser); for (int e = 0; e < 2 * kSlots; ++e) (void)hipEventDestroy(D.render_end[e]); }
        for (int s = 0; s < kSlots; ++s) {
            if (D.pool[s]) rrt_workspace_destroy(D.pool[s]);
            if (D.order[s]) rrt_tile_order_destroy(D.order[s]);
            (void)hipFree(D.tiles[s]);
            (void)hipStreamDestroy(D.stream[s]);
        }
        if (D.noise_table) rrt_noise_table_destroy(D.noise_table);
        rrt_sky_destroy(D.sky);
    }
    HIPCHK(hipSetDevice(0));
    for (int s = 0; s < kSlots; ++s) { (void)hipFree(gathered[s]); (void)hipFree(frame[s]); if (host[s]) (void)hipHostFree(host[s]); }
    return 0;
}
