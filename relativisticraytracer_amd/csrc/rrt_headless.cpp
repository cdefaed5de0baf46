// rrt_headless.cpp -- the reference's main loop (src/main.cpp:482-539) without a window, in C++, on top of
// the C ABI (include/rrt.h).  Per frame k = 1..N exactly what main() does while recording: advance the
// fixed 1/24 s float clock (main.cpp:511-516), take the camera from the active path (getInterpolatedState,
// :176-203) or the start-up camera (:128-130), launch_raymarch (:467), hand the pixels to the recorder
// (captureFrame, :85-97 -- here a raw RGBA stream, the bytes the reference pipes into ffmpeg).
//
//   rrt_headless --width 1000 --height 700 --frames 24 --path 0 --spin 0.9 --out frames.rgba [--sky-seed 1]
//
// Single GPU; the multi-GPU driver is relativisticraytracer_amd/headless.py (torch.distributed + RCCL).
#include <hip/hip_runtime_api.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
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
            if (band < 0) band = 0; if (band > 96) band = 96;
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

int fail(const char* what, int rc) {
    fprintf(stderr, "rrt_headless: %s: %s (%s)\n", what, rrt_status_string(rc), rrt_last_hip_error());
    return 1;
}

}  // namespace

int main(int argc, char** argv) {
    int w = 1000, h = 700, frames = 24, fps = 24, path = -1, sky_seed = 1, all_fx = 0, fast = 0;   // config.h:7-9
    float spin = 0.0f;
    std::string out_path;
    for (int i = 1; i < argc; ++i) {
        std::string a = argv[i];
        auto val = [&](int& dst) { if (i + 1 < argc) dst = atoi(argv[++i]); };
        if (a == "--width") val(w); else if (a == "--height") val(h); else if (a == "--frames") val(frames);
        else if (a == "--fps") val(fps); else if (a == "--path") val(path); else if (a == "--sky-seed") val(sky_seed);
        else if (a == "--spin" && i + 1 < argc) spin = (float)atof(argv[++i]);
        else if (a == "--out" && i + 1 < argc) out_path = argv[++i];
        else if (a == "--all-effects") all_fx = 1; else if (a == "--fast") fast = 1;
        else { fprintf(stderr, "unknown argument %s\n", a.c_str()); return 2; }
    }
    int n_dev = 0, rc;
    if ((rc = rrt_device_count(&n_dev)) != RRT_OK) return fail("no GPU", rc);

    std::vector<uint8_t> sky = synthetic_sky(2048, 1024, sky_seed);
    rrt_sky_t tex = 0;
    if ((rc = rrt_sky_create(sky.data(), 2048, 1024, &tex)) != RRT_OK) return fail("sky", rc);
    rrt_effects fx; rrt_effects_default(&fx);
    fx.use_chromatic_aberration = (uint8_t)all_fx;
    rrt_params prm; rrt_params_default(&prm);
    prm.spin = spin; prm.arith_mode = fast ? RRT_ARITH_FAST : RRT_ARITH_STRICT;
    int ws = 0;
    if (rrt_workspace_create((size_t)2 << 30, &ws) == RRT_OK) prm.workspace = ws;

    void* d_out = nullptr;
    const size_t bytes = (size_t)w * h * 4;
    if (hipMalloc(&d_out, bytes) != hipSuccess) { fprintf(stderr, "hipMalloc failed\n"); return 1; }
    std::vector<uint8_t> host(bytes);
    FILE* f = out_path.empty() ? nullptr : fopen(out_path.c_str(), "wb");
    if (!out_path.empty() && !f) { perror("fopen"); return 1; }

    rrt_camera cam;
    const float start_pos[3] = {0.0f, 10.0f, -60.0f};
    rrt_camera_from_angles(start_pos, 0.0f, -10.0f, &cam);
    const char* path_name = "";
    if (path >= 0 && (rc = rrt_path_info(path, &path_name, nullptr, nullptr)) != RRT_OK) return fail("path", rc);

    auto t0 = std::chrono::steady_clock::now();
    for (int k = 1; k <= frames; ++k) {
        float sim_t = 0.0f, path_t = 0.0f;
        rrt_recording_clock(k, fps, &sim_t, &path_t);
        if (path >= 0 && (rc = rrt_path_camera_at(path, path_t, &cam)) != RRT_OK) return fail("camera", rc);
        if ((rc = rrt_launch_raymarch(d_out, w, h, sim_t, &cam, tex, &fx, &prm, nullptr)) != RRT_OK) return fail("launch", rc);
        if (f) {
            if (hipMemcpy(host.data(), d_out, bytes, hipMemcpyDeviceToHost) != hipSuccess) { fprintf(stderr, "copy failed\n"); return 1; }
            if (fwrite(host.data(), 1, bytes, f) != bytes) { fprintf(stderr, "Warning: Frame write incomplete\n"); }
        }
    }
    if (hipDeviceSynchronize() != hipSuccess) { fprintf(stderr, "sync failed\n"); return 1; }
    double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (f) fclose(f);
    printf("{\"frames\": %d, \"width\": %d, \"height\": %d, \"seconds\": %.4f, \"fps\": %.3f, \"Mrays_per_s\": %.3f, "
           "\"path\": \"%s\", \"spin\": %g, \"arith_mode\": \"%s\"}\n",
           frames, w, h, dt, frames / dt, (double)frames * w * h / dt / 1e6, path_name, spin, fast ? "fast" : "strict");
    hipFree(d_out);
    rrt_sky_destroy(tex);
    return 0;
}
