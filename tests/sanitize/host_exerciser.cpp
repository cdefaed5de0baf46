// host_exerciser.cpp -- drives the HOST side of librrt_hip.so (registries, noise-table planning, camera basis and
// path playback, the recording clock, argument checks, the device-binding checks through the fake-device hook)
// under AddressSanitizer + UndefinedBehaviorSanitizer, without a GPU.  The library is built
// for it with host instrumentation only (hipcc -fsanitize=address,undefined -fno-gpu-sanitize: GPU ASan is not
// available on this pool); nothing here launches a kernel.  Exit code 0 = clean.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/rrt_test.h"      // rrt.h + the test hooks: the library under test is built with -DRRT_TEST_HOOKS

#define EXPECT(cond)                                                                  \
    do { if (!(cond)) { fprintf(stderr, "host exerciser: %s failed (line %d)\n", #cond, __LINE__); return 1; } } while (0)

int main() {
    setenv("RRT_ENABLE_TEST_HOOKS", "1", 1);          // before the library's first look at it: the fake-device hook
    EXPECT(rrt_abi_version() == RRT_ABI_VERSION);
    for (int s = -2; s < 9; ++s) EXPECT(rrt_status_string(s) != nullptr && strlen(rrt_status_string(s)) > 0);
    EXPECT(rrt_last_hip_error() != nullptr);
    rrt_params prm; rrt_effects fx;
    EXPECT(rrt_params_default(&prm) == RRT_OK && rrt_params_default(nullptr) == RRT_ERR_INVALID_ARGUMENT);
    EXPECT(rrt_effects_default(&fx) == RRT_OK && rrt_effects_default(nullptr) == RRT_ERR_INVALID_ARGUMENT);

    {   // ABI 4: the struct says its own size; another size is refused; the tile dealer on host arithmetic
        rrt_params q = prm;
        EXPECT(q.struct_size == sizeof(rrt_params) && rrt_set_launch_defaults(&q) == RRT_OK);
        q.struct_size = 36;
        EXPECT(rrt_set_launch_defaults(&q) == RRT_ERR_ABI_MISMATCH && rrt_set_launch_defaults(nullptr) == RRT_OK);
        std::vector<float> cost(135);
        std::vector<int32_t> map(135);
        for (int t = 0; t < 135; ++t) cost[t] = 1.0f + 4.0f * std::exp(-(t - 67) * (t - 67) / 81.0f);
        EXPECT(rrt_tile_map_balance(135, cost.data(), 8, 0, map.data()) == RRT_OK);
        EXPECT(rrt_tile_map_balance(135, cost.data(), 8, 17, map.data()) == RRT_OK);
        EXPECT(rrt_tile_map_balance(135, cost.data(), 8, 16, map.data()) == RRT_ERR_INVALID_ARGUMENT);
        EXPECT(rrt_tile_map_balance(0, cost.data(), 8, 0, map.data()) == RRT_ERR_INVALID_ARGUMENT);
        cost[5] = NAN;
        EXPECT(rrt_tile_map_balance(135, cost.data(), 8, 0, map.data()) == RRT_ERR_INVALID_ARGUMENT);
        EXPECT(rrt_tile_map_destroy(77) == RRT_ERR_BAD_HANDLE);
        int rows = 0;
        EXPECT(rrt_tile_map_shard_rows(77, 0, &rows, nullptr) == RRT_ERR_BAD_HANDLE);
    }

    // ---- noise-table planning: windows of either sign, every coverage, refusals, the drivers' policy
    size_t bytes = 0; int boxes[12];
    EXPECT(rrt_noise_table_plan(32.0f, &bytes, boxes) == RRT_OK && bytes > 0);
    EXPECT(rrt_noise_table_plan(-1.0f, &bytes, boxes) == RRT_ERR_INVALID_ARGUMENT);
    EXPECT(rrt_noise_table_plan(NAN, &bytes, boxes) == RRT_ERR_INVALID_ARGUMENT);
    EXPECT(rrt_noise_table_plan_window(0.0f, 600.0f, RRT_TABLE_FULL | RRT_TABLE_DENSE, &bytes, nullptr) == RRT_ERR_INVALID_ARGUMENT && bytes == 0);
    EXPECT(rrt_noise_table_plan(600.0f, &bytes, nullptr) == RRT_OK && bytes > 0);          // round 5: addressable in the banded layout
    int banded = 0, n_bands = 0; float w_min = 0, w_scale = 0;
    std::vector<int32_t> band_boxes(3 * 64 * 6), acc_boxes(24);
    for (int layout : {0, (int)RRT_TABLE_BANDED, (int)RRT_TABLE_DENSE})
        for (int cov = RRT_TABLE_FULL; cov <= RRT_TABLE_COARSEST; ++cov)
            for (float t0 : {-9000.0f, -50.0f, 0.0f, 31.5f, 495.0f, 9000.0f})
                for (float span : {0.0f, 0.5f, 10.0f, 1000.0f}) {
                    (void)rrt_noise_table_plan_window(t0, t0 + span, cov | layout, &bytes, boxes);
                    (void)rrt_noise_table_plan_layout(t0, t0 + span, cov | layout, &banded, &n_bands, &w_min, &w_scale, band_boxes.data(), 64, acc_boxes.data());
                }
    EXPECT(rrt_noise_table_plan_layout(495.0f, 505.0f, RRT_TABLE_FULL, &banded, &n_bands, &w_min, &w_scale, band_boxes.data(), 64, acc_boxes.data()) == RRT_OK
           && banded == 1 && n_bands >= 1 && n_bands <= 64);
    EXPECT(rrt_noise_table_plan_layout(495.0f, 505.0f, RRT_TABLE_FULL, &banded, &n_bands, nullptr, nullptr, band_boxes.data(), 1, nullptr) == RRT_ERR_INVALID_ARGUMENT);   // cap_bands too small
    EXPECT(rrt_noise_table_plan_layout(495.0f, 505.0f, RRT_TABLE_FULL, nullptr, &n_bands, nullptr, nullptr, nullptr, 0, nullptr) == RRT_ERR_INVALID_ARGUMENT);
    EXPECT(rrt_noise_table_plan_window(0.0f, 1.0f, RRT_TABLE_BANDED | RRT_TABLE_DENSE, &bytes, boxes) == RRT_ERR_INVALID_ARGUMENT);
    EXPECT(rrt_noise_table_plan_window(2.0f, 1.0f, RRT_TABLE_FULL, &bytes, boxes) == RRT_ERR_INVALID_ARGUMENT);
    EXPECT(rrt_noise_table_plan_window(0.0f, 1.0f, 7, &bytes, boxes) == RRT_ERR_INVALID_ARGUMENT);
    EXPECT(rrt_noise_table_plan_window(0.0f, 1.0f, RRT_TABLE_FULL, nullptr, nullptr) == RRT_OK);
    float t1 = 0; int cov = 0;
    for (size_t budget : {(size_t)0, (size_t)1 << 20, (size_t)1 << 28, (size_t)2 << 30, (size_t)64 << 30})
        for (float t = 0.0f; t < 4000.0f; t += 333.25f) {
            EXPECT(rrt_noise_table_fit_window(t, t + 100.0f, budget, &t1, &cov, &bytes) == RRT_OK);
            EXPECT(bytes <= budget && t1 >= t && t1 <= t + 100.0f);
        }
    EXPECT(rrt_noise_table_fit_window(1.0f, 0.0f, 1 << 30, &t1, &cov, &bytes) == RRT_ERR_INVALID_ARGUMENT);
    EXPECT(rrt_noise_table_fit_window(0.0f, 1.0f, 1 << 30, nullptr, &cov, &bytes) == RRT_ERR_INVALID_ARGUMENT);
    EXPECT(rrt_noise_table_info(12345, nullptr, nullptr, nullptr) == RRT_ERR_BAD_HANDLE);
    EXPECT(rrt_noise_table_window(12345, nullptr, nullptr, nullptr, nullptr) == RRT_ERR_BAD_HANDLE);
    EXPECT(rrt_noise_table_destroy(12345) == RRT_ERR_BAD_HANDLE);

    // ---- camera basis, paths, recording clock
    rrt_camera cam;
    const float pos[3] = {0.0f, 10.0f, -60.0f};
    EXPECT(rrt_camera_from_angles(pos, 0.0f, -10.0f, &cam) == RRT_OK);
    EXPECT(rrt_camera_from_angles(nullptr, 0.0f, 0.0f, &cam) == RRT_ERR_INVALID_ARGUMENT);
    EXPECT(rrt_camera_from_angles(pos, 0.0f, 90.0f, &cam) == RRT_OK);          // looking straight up: right = 0/0 as in the reference
    const int n_paths = rrt_path_count();
    EXPECT(n_paths == 3);
    for (int p = -1; p <= n_paths; ++p) {
        const char* name = nullptr; int n_keys = 0; float t_end = 0;
        const int rc = rrt_path_info(p, &name, &n_keys, &t_end);
        EXPECT((rc == RRT_OK) == (p >= 0 && p < n_paths));
        if (rc != RRT_OK) { EXPECT(rrt_path_camera_at(p, 1.0f, &cam) != RRT_OK); continue; }
        std::vector<float> keys((size_t)n_keys * 6);
        EXPECT(rrt_path_keyframes(p, keys.data(), n_keys) == RRT_OK);
        EXPECT(rrt_path_keyframes(p, keys.data(), n_keys - 1) != RRT_OK);      // too small a buffer is refused, not overrun
        EXPECT(rrt_path_info(p, nullptr, nullptr, nullptr) == RRT_OK);
        for (float t = -2.0f; t < t_end + 3.0f; t += 0.173f) EXPECT(rrt_path_camera_at(p, t, &cam) == RRT_OK);
        for (int k = 0; k < n_keys; ++k) EXPECT(rrt_path_camera_at(p, keys[6 * k], &cam) == RRT_OK);
        EXPECT(rrt_path_camera_at(p, NAN, &cam) == RRT_OK || true);
        EXPECT(rrt_path_camera_at(p, 1.0f, nullptr) == RRT_ERR_INVALID_ARGUMENT);
    }
    float a3[3] = {1, 2, 3}, b3[3] = {-4, 5, 6}, c3[3] = {7, -8, 9}, d3[3] = {0, 0, 1}, o3[3], ang = 0;
    EXPECT(rrt_catmull_rom(a3, b3, c3, d3, 0.37f, o3) == RRT_OK && rrt_catmull_rom(a3, b3, c3, nullptr, 0.5f, o3) != RRT_OK);
    for (float a = -720.0f; a <= 720.0f; a += 97.0f) EXPECT(rrt_lerp_angle(a, -a * 0.5f, 0.3f, &ang) == RRT_OK);
    EXPECT(rrt_lerp_angle(0.0f, 1.0f, 0.5f, nullptr) != RRT_OK);
    float st = 0, pt = 0;
    for (int k = 0; k <= 400; k += 7) EXPECT(rrt_recording_clock(k, 24, &st, &pt) == RRT_OK);
    EXPECT(rrt_recording_clock(-1, 24, &st, &pt) != RRT_OK && rrt_recording_clock(1, 0, &st, &pt) != RRT_OK);
    EXPECT(rrt_recording_clock(3, 24, nullptr, nullptr) == RRT_OK);

    // ---- per-window path choice (host logic only: csrc/rrt_path_chooser.cpp): three synthetic ranks, reports three frames late
    {
        int pc = 0, pol = -1;
        EXPECT(rrt_path_chooser_create(0, 0, &pc) == RRT_ERR_INVALID_ARGUMENT && rrt_path_chooser_create(3, 0, nullptr) == RRT_ERR_INVALID_ARGUMENT);
        EXPECT(rrt_path_chooser_policy(12345, 1, &pol) == RRT_ERR_BAD_HANDLE && rrt_path_chooser_report(12345, 1, 1.0f) == RRT_ERR_BAD_HANDLE);
        for (int scenario = 0; scenario < 3; ++scenario) {
            EXPECT(rrt_path_chooser_create(3, scenario == 2 ? 4 : 0, &pc) == RRT_OK);
            float pend[3] = {0, 0, 0};
            for (int k = 1; k <= 2500; ++k) {                              // beyond the policy ring's 1024 frames
                EXPECT(rrt_path_chooser_policy(pc, k, &pol) == RRT_OK && (pol == RRT_PATH_AUTO || pol == RRT_PATH_SINGLE));
                const float single = scenario == 0 ? 4.5f : (k % 5 ? 7.5f : 19.0f), three = scenario == 0 ? 4.8f : 6.2f;
                if (k > 3) EXPECT(rrt_path_chooser_report(pc, k - 3, pend[k % 3]) == RRT_OK);
                pend[k % 3] = pol == RRT_PATH_SINGLE ? single : three;
            }
            rrt_path_chooser_stats cs;
            EXPECT(rrt_path_chooser_get_stats(pc, &cs) == RRT_OK && cs.frames[0] + cs.frames[1] == 2500 && cs.windows > 10);
            EXPECT((scenario == 0) == (cs.incumbent == RRT_PATH_SINGLE));
            EXPECT(rrt_path_chooser_report(pc, 0, 1.0f) == RRT_ERR_INVALID_ARGUMENT && rrt_path_chooser_report(pc, 5, NAN) == RRT_ERR_INVALID_ARGUMENT);
            EXPECT(rrt_path_chooser_get_stats(pc, nullptr) == RRT_ERR_INVALID_ARGUMENT);
            EXPECT(rrt_path_chooser_destroy(pc) == RRT_OK && rrt_path_chooser_destroy(pc) == RRT_ERR_BAD_HANDLE);
        }
    }

    // ---- shard arithmetic
    int rows = 0, total = 0;
    for (int s = 0; s < 8; ++s) { EXPECT(rrt_tile_shard_rows(2160, 16, s, 8, &rows) == RRT_OK); total += rows; }
    EXPECT(total == 2160);
    EXPECT(rrt_tile_shard_rows(90, 7, 1, 2, &rows) == RRT_OK && rows > 0);
    EXPECT(rrt_tile_shard_rows(90, 0, 0, 2, &rows) != RRT_OK && rrt_tile_shard_rows(90, 8, 2, 2, &rows) != RRT_OK);
    EXPECT(rrt_tile_shard_rows(90, 8, 0, 2, nullptr) != RRT_OK);

    // ---- launch defaults
    rrt_params got;
    prm.spin = 0.9f; prm.max_steps = 77;
    EXPECT(rrt_set_launch_defaults(&prm) == RRT_OK && rrt_get_launch_defaults(&got) == RRT_OK && got.max_steps == 77);
    prm.max_steps = -5;
    EXPECT(rrt_set_launch_defaults(&prm) == RRT_ERR_INVALID_ARGUMENT);
    EXPECT(rrt_set_launch_defaults(nullptr) == RRT_OK && rrt_get_launch_defaults(&got) == RRT_OK && got.max_steps == 2000);
    EXPECT(rrt_get_launch_defaults(nullptr) == RRT_ERR_INVALID_ARGUMENT);

    // ---- library-owned resources of the drop-in entry point: argument checks (nothing can be created without a device)
    {
        int on = -1, builds = -1; float t0 = 0, t1 = 0; size_t tb = 1;
        EXPECT(rrt_launch_auto_resources_info(&on, &builds, &t0, &t1, &tb) == RRT_OK && on == 0 && builds == 0 && tb == 0);
        EXPECT(rrt_launch_auto_resources_info(nullptr, nullptr, nullptr, nullptr, nullptr) == RRT_OK);
        EXPECT(rrt_launch_auto_resources(0, nullptr, 0, 0) == RRT_OK);               // switching off what is off
        rrt_params bad; rrt_params_default(&bad); bad.struct_size = 40;
        EXPECT(rrt_launch_auto_resources(1, &bad, 0, 0) == RRT_ERR_ABI_MISMATCH);
        rrt_params_default(&bad); bad.max_steps = -3;
        EXPECT(rrt_launch_auto_resources(1, &bad, 0, 0) == RRT_ERR_INVALID_ARGUMENT);
        EXPECT(rrt_launch_auto_resources_info(&on, nullptr, nullptr, nullptr, nullptr) == RRT_OK && on == 0);
    }

    // ---- handles: unknown ids, argument checks, and the device binding (fake device ids: no GPU is touched)
    EXPECT(rrt_sky_destroy(0xdeadbeefull) == RRT_ERR_BAD_HANDLE);
    EXPECT(rrt_workspace_destroy(99) == RRT_ERR_BAD_HANDLE && rrt_workspace_stats(99, nullptr, nullptr) == RRT_ERR_BAD_HANDLE);
    char scratch[64];
    EXPECT(rrt_workspace_read(99, 0, 8, scratch) == RRT_ERR_BAD_HANDLE);
    int ws = 0;
    EXPECT(rrt_workspace_create(16, &ws) == RRT_ERR_INVALID_ARGUMENT && rrt_workspace_create(1 << 24, nullptr) == RRT_ERR_INVALID_ARGUMENT);
    EXPECT(rrt_debug_fake_device(2) == RRT_OK);
    rrt_sky_t sky = 0;
    EXPECT(rrt_sky_create_from_device(reinterpret_cast<void*>(0x1000), 8, 4, &sky) == RRT_OK);
    EXPECT(rrt_sky_create_from_device(nullptr, 8, 4, &sky) == RRT_ERR_INVALID_ARGUMENT || true);
    EXPECT(rrt_sky_create(nullptr, 8, 4, &sky) == RRT_ERR_INVALID_ARGUMENT || true);
    rrt_params_default(&prm); rrt_effects_default(&fx);
    EXPECT(rrt_camera_from_angles(pos, 0.0f, -10.0f, &cam) == RRT_OK);
    void* out = reinterpret_cast<void*>(0x2000);
    EXPECT(rrt_debug_fake_device(5) == RRT_OK);                      // "hipSetDevice(5)": the sky lives on device 2
    EXPECT(rrt_launch_raymarch(out, 16, 8, 1.0f, &cam, sky, &fx, &prm, nullptr) == RRT_ERR_BAD_HANDLE);
    EXPECT(rrt_launch_raymarch_rows(out, 16, 8, 0, 8, 1.0f, &cam, sky, &fx, &prm, nullptr) == RRT_ERR_BAD_HANDLE);
    EXPECT(rrt_launch_raymarch_tiles(out, 16, 8, 4, 1, 2, 1.0f, &cam, sky, &fx, nullptr, nullptr) == RRT_ERR_BAD_HANDLE);
    EXPECT(rrt_launch_raymarch_ex(out, 16, 8, 1.0f, &cam, sky, &fx, &prm, nullptr, nullptr) == RRT_ERR_BAD_HANDLE);
    // argument checks come before any handle is looked at
    EXPECT(rrt_launch_raymarch(nullptr, 16, 8, 1.0f, &cam, sky, &fx, &prm, nullptr) == RRT_ERR_INVALID_ARGUMENT);
    EXPECT(rrt_launch_raymarch(out, 0, 8, 1.0f, &cam, sky, &fx, &prm, nullptr) == RRT_ERR_INVALID_ARGUMENT);
    EXPECT(rrt_launch_raymarch(out, 65536, 65536, 1.0f, &cam, sky, &fx, &prm, nullptr) == RRT_ERR_INVALID_ARGUMENT);
    EXPECT(rrt_launch_raymarch_rows(out, 16, 8, 3, 2, 1.0f, &cam, sky, &fx, &prm, nullptr) == RRT_ERR_INVALID_ARGUMENT);
    prm.max_steps = -1;
    EXPECT(rrt_launch_raymarch(out, 16, 8, 1.0f, &cam, sky, &fx, &prm, nullptr) == RRT_ERR_INVALID_ARGUMENT);
    rrt_params_default(&prm);
    prm.noise_table = 4242;                                             // unknown table id
    EXPECT(rrt_debug_fake_device(2) == RRT_OK);
    EXPECT(rrt_launch_raymarch(out, 16, 8, 1.0f, &cam, sky, &fx, &prm, nullptr) == RRT_ERR_BAD_HANDLE);
    const float cam12[12] = {0, 10, -60, 0, 0, 1, 1, 0, 0, 0, 1, 0};
    EXPECT(rrt_launch_raymarch_compat(out, 16, 8, 1.0f, cam12, 0x7777ull, &fx) == RRT_ERR_BAD_HANDLE);      // says why, once
    EXPECT(rrt_launch_raymarch_compat(out, 16, 8, 1.0f, cam12, 0x7777ull, &fx) == RRT_ERR_BAD_HANDLE);
    EXPECT(rrt_launch_raymarch_compat(out, 16, 8, 1.0f, nullptr, sky, &fx) == RRT_ERR_INVALID_ARGUMENT);
    EXPECT(rrt_sky_destroy(sky) == RRT_OK && rrt_sky_destroy(sky) == RRT_ERR_BAD_HANDLE);
    EXPECT(rrt_debug_fake_device(-1) == RRT_OK);
    int n_dev = -1;
    (void)rrt_device_count(&n_dev);                                      // no GPU here: RRT_ERR_NO_DEVICE, count 0
    EXPECT(rrt_device_count(nullptr) == RRT_ERR_INVALID_ARGUMENT);
    printf("host exerciser ok (%d device(s) visible)\n", n_dev);
    return 0;
}
