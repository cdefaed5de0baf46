/*
 * rrt_camera.cpp -- host-side camera code of librrt_hip.so (plain C++, no HIP): basis from angles, Catmull-Rom path playback,
 * the recording clock.  C ABI: include/rrt.h ("host-side camera helpers").
 */
#include <cmath>
#include <cstring>

#include "../../include/rrt.h"

/* ---------------------------------------------------------------- camera paths (host)
 * Reference: catmull_rom / lerp_angle / initDefaultPaths, src/camera_paths.cpp:6-73;
 * PathController::getInterpolatedState, src/main.cpp:176-203; fixed recording clock,
 * src/main.cpp:511-516.  The keyframe tables are data and keep the reference's values. */
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
