/*
 * oracle_exerciser.c -- drives every entry point of the CPU restatement (oracle/rrt_oracle.c) under
 * AddressSanitizer + UndefinedBehaviorSanitizer (`make -C oracle asan`).  CPU only, test
 * infrastructure: small frames in both math modes, strided / rect renders, the gate recorder and its replay, the unit
 * functions on awkward inputs (zeros, negative lattice points, huge coordinates, NaN).  Exit code 0 = the
 * sanitizers found nothing; they abort the process otherwise.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../oracle/rrt_oracle.h"

/* CameraController::getCUDAStateFrom (main.cpp:141-167) in plain C, for the exerciser's cameras only */
static void camera_from_angles(const float* pos, float yaw, float pitch, rrto_camera* c) {
    const float ry = yaw * 3.14159f / 180.0f, rp = pitch * 3.14159f / 180.0f;
    float f[3] = {sinf(ry) * cosf(rp), sinf(rp), cosf(ry) * cosf(rp)};
    float m = sqrtf(f[0] * f[0] + f[1] * f[1] + f[2] * f[2]);
    for (int k = 0; k < 3; ++k) f[k] /= m;
    float r[3] = {1.0f * f[2] - 0.0f * f[1], 0.0f * f[0] - 0.0f * f[2], 0.0f * f[1] - 1.0f * f[0]};
    m = sqrtf(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
    for (int k = 0; k < 3; ++k) r[k] /= m;
    const float u[3] = {f[1] * r[2] - f[2] * r[1], f[2] * r[0] - f[0] * r[2], f[0] * r[1] - f[1] * r[0]};
    for (int k = 0; k < 3; ++k) { c->pos[k] = pos[k]; c->forward[k] = f[k]; c->right[k] = r[k]; c->up[k] = u[k]; }
}

static void fill_sky(uint8_t* sky, int w, int h) {
    for (int i = 0; i < w * h; ++i) {
        uint32_t v = (uint32_t)i * 2654435761u;
        sky[4 * i] = (uint8_t)(v >> 24); sky[4 * i + 1] = (uint8_t)(v >> 16); sky[4 * i + 2] = (uint8_t)(v >> 8); sky[4 * i + 3] = 255;
    }
}

int main(void) {
    enum { SW = 64, SH = 32 };
    uint8_t* sky = (uint8_t*)malloc(SW * SH * 4);
    fill_sky(sky, SW, SH);
    rrto_params prm; rrto_default_params(&prm);
    rrto_effects fx; rrto_default_effects(&fx);
    /* cameras: the start-up camera, one inside both media zones, one inside the horizon radius, one far out */
    const float cams[4][5] = {{0.f, 10.f, -60.f, 0.f, -10.f}, {14.f, 0.05f, 3.f, 200.f, 2.f}, {0.f, 0.5f, 0.f, 0.f, 0.f},
                              {3000.f, 40.f, 0.f, -90.f, 0.f}};
    long checksum = 0;
    for (int c = 0; c < 4; ++c) {
        rrto_camera cam;
        camera_from_angles(cams[c], cams[c][3], cams[c][4], &cam);
        for (int mode = 0; mode < 2; ++mode) {
            const int w = 37 + 5 * c, h = 19 + 3 * c;          /* ragged sizes */
            prm.math_mode = mode ? RRTO_MATH_PORTABLE : RRTO_MATH_LIBM;
            prm.spin = c & 1 ? 0.9f : 0.0f;
            prm.volumetrics = 1;
            fx.use_ca = c & 1;
            uint8_t* rgba = (uint8_t*)calloc((size_t)w * h, 4);
            float* ldr = (float*)calloc((size_t)w * h * 4, sizeof(float));
            float* hdr = (float*)calloc((size_t)w * h * 4, sizeof(float));
            rrto_diag d; memset(&d, 0, sizeof(d));
            int32_t* steps = (int32_t*)calloc((size_t)w * h, 4); int32_t* hit = (int32_t*)calloc((size_t)w * h, 4);
            int32_t* nn = (int32_t*)calloc((size_t)w * h, 4); int32_t* ns = (int32_t*)calloc((size_t)w * h, 4);
            int32_t* nd = (int32_t*)calloc((size_t)w * h, 4);
            float* pos = (float*)calloc((size_t)w * h * 3, 4); float* vel = (float*)calloc((size_t)w * h * 3, 4);
            float* rad = (float*)calloc((size_t)w * h * 4, 4);
            d.steps = steps; d.hit = hit; d.n_noise = nn; d.n_samples = ns; d.n_dens = nd; d.pos = pos; d.vel = vel; d.rad = rad;
            if (rrto_render(&cam, &fx, &prm, 1.0f + c, w, h, 0, 0, w, h, 1, 1, sky, SW, SH, rgba, ldr, hdr, &d, 0) != 0) return 2;
            /* a rect and a strided sample of the same frame, fewer outputs */
            if (rrto_render(&cam, &fx, &prm, 1.0f + c, w, h, 3, 2, w - 1, h - 3, 1, 1, sky, SW, SH, rgba, NULL, NULL, NULL, 1) != 0) return 2;
            if (rrto_render(&cam, &fx, &prm, 1.0f + c, w, h, 0, 0, w, h, 3, 2, sky, SW, SH, NULL, ldr, NULL, &d, 2) != 0) return 2;
            /* gate recorder, then replay of the recorded decisions */
            const int cap = 4096, nxs = (w + 2) / 3, nys = (h + 1) / 2;
            uint8_t* glog = (uint8_t*)calloc((size_t)nxs * nys, cap);
            int32_t* gcnt = (int32_t*)calloc((size_t)nxs * nys, 4);
            if (rrto_render_gates(&cam, &fx, &prm, 1.0f + c, w, h, 0, 0, w, h, 3, 2, sky, SW, SH, NULL, ldr, NULL, &d, 0, 1, glog, cap, gcnt) != 0) return 3;
            if (rrto_render_gates(&cam, &fx, &prm, 1.0f + c, w, h, 0, 0, w, h, 3, 2, sky, SW, SH, NULL, ldr, NULL, &d, 0, 2, glog, cap, gcnt) != 0) return 3;
            for (int i = 0; i < w * h; ++i) checksum += rgba[4 * i] + steps[i];
            free(glog); free(gcnt); free(rgba); free(ldr); free(hdr); free(steps); free(hit); free(nn); free(ns); free(nd);
            free(pos); free(vel); free(rad);
        }
    }
    /* bad arguments are refused, not dereferenced */
    {
        rrto_camera cam; camera_from_angles(cams[0], 0.f, -10.f, &cam);
        uint8_t px[4];
        if (rrto_render(&cam, &fx, &prm, 1.0f, 4, 4, -1, 0, 4, 4, 1, 1, sky, SW, SH, px, NULL, NULL, NULL, 1) == 0) return 4;
        if (rrto_render(&cam, &fx, &prm, 1.0f, 4, 4, 0, 0, 5, 4, 1, 1, sky, SW, SH, px, NULL, NULL, NULL, 1) == 0) return 4;
        if (rrto_render(&cam, &fx, &prm, 1.0f, 4, 4, 0, 0, 4, 4, 0, 1, sky, SW, SH, px, NULL, NULL, NULL, 1) == 0) return 4;
    }
    /* unit functions on awkward inputs */
    {
        enum { N = 12 };
        const float pts[N][3] = {{0, 0, 0}, {-0.f, -0.f, -0.f}, {1e-30f, 0, 0}, {-3.5f, -2.25f, -7.75f}, {1e6f, -1e6f, 1e6f},
                                 {10000.f, 10000.f, 10000.f}, {12.f, 0.1f, 5.f}, {24.9f, 0.7f, 0.f}, {0.f, 3.9f, 10.1f},
                                 {NAN, 1.f, 1.f}, {INFINITY, 0.f, 0.f}, {17.f, -0.3f, -17.f}};
        float out[N], out3[N * 3], v[N][3];
        for (int i = 0; i < N; ++i) { v[i][0] = 0.3f; v[i][1] = -0.9f; v[i][2] = 0.2f; }
        for (int mode = 0; mode < 2; ++mode) {
            const int mm = mode ? RRTO_MATH_PORTABLE : RRTO_MATH_LIBM;
            rrto_hash31(N, &pts[0][0], out); rrto_noise3d(N, &pts[0][0], out);
            rrto_fbm(N, &pts[0][0], 5, out); rrto_fbm(N, &pts[0][0], 2, out); rrto_fbm(N, &pts[0][0], 0, out);
            rrto_geodesic_acc(N, &pts[0][0], &v[0][0], 0.9f, out3);
            rrto_accretion_density(N, &pts[0][0], 12.5f, mm, out);
            rrto_dust_density(N, &pts[0][0], 12.5f, mm, out);
            rrto_redshift(N, &pts[0][0], &v[0][0], 0.99f, mm, out);
        }
        rrto_geodesic_acc(0, NULL, NULL, 0.f, NULL);               /* n = 0 touches nothing */
    }
    free(sky);
    printf("oracle exerciser ok (checksum %ld, %d threads)\n", checksum, rrto_max_threads());
    return 0;
}
