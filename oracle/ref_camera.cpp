/*
 * ref_camera.cpp -- C wrappers around the REFERENCE's camera-path code.
 *
 * TEST INFRASTRUCTURE ONLY.  Linked with /root/reference/src/camera_paths.cpp
 * compiled where it lies (oracle/Makefile) into oracle/_ref/libref_camera.so.
 * Pins catmull_rom, lerp_angle and the three built-in keyframe tables
 * (camera_paths.cpp:6-73).  CameraController::getCUDAStateFrom and
 * PathController::getInterpolatedState live in src/main.cpp; main.cpp as a
 * whole needs GLFW/GLAD, but the line range that holds those two structs does
 * not: it is piped into g++ and linked into the same library (round 3:
 * ref_main_camera_pre.h, ref_main_camera_post.inc, oracle/Makefile).
 */
#include <cuda_runtime.h>
#include "camera_paths.h"

extern "C" {

void ref_catmull_rom(const float* p0, const float* p1, const float* p2, const float* p3, float t, float* out) {
    float3 r = catmull_rom(make_float3(p0[0], p0[1], p0[2]), make_float3(p1[0], p1[1], p1[2]),
                           make_float3(p2[0], p2[1], p2[2]), make_float3(p3[0], p3[1], p3[2]), t);
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
float ref_lerp_angle(float a, float b, float t) { return lerp_angle(a, b, t); }

static void ensure_paths() {
    if (PathManager::instance().getPaths().empty()) initDefaultPaths();
}
int ref_path_count() { ensure_paths(); return (int)PathManager::instance().getPaths().size(); }
int ref_path_len(int idx) {
    ensure_paths();
    const CameraPath* p = PathManager::instance().getPath(idx);
    return p ? (int)p->keyframes.size() : -1;
}
/* out: 6 floats per keyframe: time, pos.xyz, yaw, pitch */
int ref_path_keys(int idx, float* out) {
    ensure_paths();
    const CameraPath* p = PathManager::instance().getPath(idx);
    if (!p) return -1;
    for (size_t i = 0; i < p->keyframes.size(); ++i) {
        const Keyframe& k = p->keyframes[i];
        out[6 * i] = k.time; out[6 * i + 1] = k.pos.x; out[6 * i + 2] = k.pos.y; out[6 * i + 3] = k.pos.z;
        out[6 * i + 4] = k.yaw; out[6 * i + 5] = k.pitch;
    }
    return (int)p->keyframes.size();
}
int ref_path_name(int idx, char* buf, int cap) {
    ensure_paths();
    const CameraPath* p = PathManager::instance().getPath(idx);
    if (!p) return -1;
    int n = 0;
    for (; n < cap - 1 && n < (int)p->name.size(); ++n) buf[n] = p->name[n];
    buf[n] = 0;
    return n;
}

}  // extern "C"
