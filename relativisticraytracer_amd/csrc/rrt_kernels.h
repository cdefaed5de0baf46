/*
 * rrt_kernels.h -- the gfx950 kernels of the render path and the structures they share with the host side.
 *
 * A SECTION of rrt_hip.hip (included once, inside its anonymous namespace, after `using namespace rrt`): without relocatable
 * device code a kernel and its launch site have to share a translation unit, so the split is textual -- device code here, host
 * registries / launch logic / C ABI in rrt_hip.hip, camera host code in rrt_camera.cpp, test hooks in rrt_test_hooks.h.
 *
 * Replaces the reference's only kernel, raymarch_kernel (src/raymarcher.cu:15-174):
 *   raymarch_pixels<SPIN,MEDIA,DEBUG,ARITH>   single-kernel path: one ray per lane, media sampled in line
 *   march_defer / eval_sample_rows / composite_and_shade (/ pool_next_round, zero_words)
 *                                             three-pass path through a caller-owned workspace, in rounds, as two chains
 *   probe_costs / probe_to_tiles              coarse march-only probe of a view: first-frame dispatch order, row-tile costs
 *   clock_probe_kernel                        the shader clock the chip holds
 *   assemble_tiles_kernel / assemble_all_kernel / assemble_map_kernel   scatter gathered row-tile shards into the frame
 *   build_noise_table                         the lattice-hash tables (rrt_noise_table)
 */
#ifndef RRT_KERNELS_H
#define RRT_KERNELS_H

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
/* ------------------------------------------------------------------ lattice-hash tables: box of lattice points, fill kernel */
struct LutBox { int x0, y0, z0, nx, ny, nz; };

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
    /* lattice-hash tables (rrt_noise_table); only read by the kernels instantiated with MEDIA >= 2; dust_bands only with
     * MEDIA == 3 (the fine dust families in per-omega-band boxes: rrt_device.h, DustBands) */
    NoiseLut lut_acc, lut_dust;
    DustBands dust_bands;
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
 * MEDIA: 0 = densities read 0 ("skybox only"), 1 = full media, 2 = full media with the lattice-hash tables, 3 = with the
 * tables in their banded layout (rrt_device.h: DustBands).
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
            media_densities<MEDIA>(rel_p, a.time, in_disk, in_cloud, a.lut_acc, a.lut_dust, a.dust_bands, oob, d_disk, d_cloud);
            accumulate_sample(acc, d_disk, d_cloud, rel_p, r, vel, h, a.spin);
        }
        if (r > 250.0f && dot(rel_p, vel) > 0.0f) { steps = k + 1; break; }
    }
    i = steps;
}

/* Round 6: the vacuum steps of a wavefront in a loop of their own, every exit of which is wave-uniform.  In the flat loop
 * of rounds 3-5 the vacuum and the generic path met before the back edge, and the step that is taken nine times in ten paid
 * the register copies of that meeting (FMAD: 14 v_mov on 221 arithmetic instructions; strict: 5 on 278).  Here the
 * loop-carried state has one producer; with the body written out twice a step can write its results into the registers of
 * the state before last, which the escape test (pre-step position, post-step velocity: raymarcher.cu:120) has released by
 * then: 220 VALU per FMAD vacuum step, 281 strict, no copy left (tools/isa_histogram.py; 4K bench frame 32.2 -> 30.3 ms
 * FMAD, 37.2 -> 35.9 strict, same bytes: profiles/r06_vac_inner_ab.txt).  A lane that escapes does not leave by itself -- a
 * divergent exit would make the step counter a per-lane value: the WAVE leaves (1), the escaped lanes end their march at the
 * caller, the others come back in.
 * In: the loop-top radius (r2, r, y, hy) of p, accepted and >= kVacuumR in every live lane.  Returns why the wave left:
 * 1 = a lane escaped (`escaped`; k counts its last step), 2 = out of steps, 3 = some lane needs the generic step: (r2, r, y,
 * hy, rejected, rej_mask) are then the loop-top values of the new p. */
#ifndef RRT_VAC_INNER
#define RRT_VAC_INNER 2
#endif
/* the escape test's dot product behind a wave-uniform `some lane is beyond r = 250` (1), or evaluated on every step (0:
 * 0.4 ms slower on the 4K frame) */
#ifndef RRT_VAC_ESC_BRANCH
#define RRT_VAC_ESC_BRANCH 1
#endif
template <bool SPIN, bool FMA>
__device__ __forceinline__ int vacuum_run(v3& p, v3& vel, float drag_c, int& k, int max_steps, float& r2, float& r, float& y, float& hy,
                                          float& ys, float& hs, float& hcp, bool& rejected, unsigned long long& rej_mask, bool& escaped) {
#define RRT_VAC_STEP                                                                                                  \
    {                                                                                                                 \
        const v3 q = p;                                                                                               \
        integrate_rk4_lean<SPIN, true, FMA>(p, vel, 0.f, 0.f, 0.f, drag_c, r2, r, y, hy, ys, hs, hcp);                \
        ++k;                                                                                                          \
        escaped = false;                                                                                              \
        if (!RRT_VAC_ESC_BRANCH || __builtin_amdgcn_ballot_w64(r > 250.0f) != 0ull) { /* raymarcher.cu:120 */         \
            escaped = r > 250.0f && (FMA ? dot_fma(q, vel) : dot(q, vel)) > 0.0f;                                     \
            if (__builtin_amdgcn_ballot_w64(escaped) != 0ull) return 1;                                               \
        }                                                                                                             \
        if (k >= max_steps) return 2;                                                                                 \
        r2 = FMA ? dot_fma(p, p) : dot(p, p);                                                                         \
        rejected = sqrt_seeded_yh<1>(r2, ys, hs, r, y, hy);                                                           \
        rej_mask = __builtin_amdgcn_ballot_w64(rejected);                                                             \
        if ((rej_mask | __builtin_amdgcn_ballot_w64(!(r >= kVacuumR))) != 0ull) return 3;                             \
    }
    for (;;) {
        RRT_VAC_STEP
#if RRT_VAC_INNER >= 2
        RRT_VAC_STEP
#endif
    }
#undef RRT_VAC_STEP
}

/* UK: every lane of the wave enters at the same step `i` (the single kernel: 0), so the step counter is one number per wave;
 * the loop says so once per outer iteration (v_readfirstlane), which keeps the counter and the vacuum loop's exit code in
 * scalar registers whatever the compiler makes of the merged exits of the outer loop. */
template <bool SPIN, int MEDIA, int ARITH, bool UK = false>
__device__ __forceinline__ void march_inline(const FrameArgs& a, v3& p, v3& vel, Radiance& acc, bool& hit, int& i,
                                             unsigned* oob) {
    constexpr bool FMA = ARITH == kArithFmad;           /* the lean loop with fused multiply-adds (rrt_device.h: integrate_rk4_lean) */
    if constexpr (ARITH == kArithFast || !RRT_MARCH_V2) {
        march_inline_v1<SPIN, MEDIA, ARITH == kArithFast>(a, p, vel, acc, hit, i, oob);
    } else {
        int steps = i > a.max_steps ? i : a.max_steps;  /* if the loop runs out */
        float ys = 0.0f, hs = 0.0f;                     /* (1/r, 1/(2r)) estimate for the next loop-top radius; 0: none yet */
        float hcp = 0.0f;                               /* 1/(2r) at the previous vacuum step's stage 3 (seed extrapolation) */
#if RRT_VAC_INNER
        /* vacuum steps in a loop of their own (vacuum_run); #else: the flat loop of rounds 3-5 */
        int k = i;
        while (k < a.max_steps) {
            if (UK) k = __builtin_amdgcn_readfirstlane(k);
            v3 rel_p = p;                               /* p - MASS_POS, MASS_POS = 0 */
            float r2 = FMA ? dot_fma(rel_p, rel_p) : dot(rel_p, rel_p);
            float r, y, hy;
            bool rejected = sqrt_seeded_yh<1>(r2, ys, hs, r, y, hy);
            /* wave-uniform: every live lane holds an accepted radius >= kVacuumR (two compares, scalar logic) */
            unsigned long long rej_mask = __builtin_amdgcn_ballot_w64(rejected);
            if (RRT_VACUUM_PATH && (rej_mask | __builtin_amdgcn_ballot_w64(!(r >= kVacuumR))) == 0ull) {
                bool escaped;
                const int why = vacuum_run<SPIN, FMA>(p, vel, a.drag_c, k, a.max_steps, r2, r, y, hy, ys, hs, hcp, rejected, rej_mask, escaped);
                if (escaped) { steps = k; break; }      /* k counts the step just taken */
                if (why != 3) continue;
                rel_p = p;
            }
            if (rej_mask != 0ull) {
                bool small;                             /* r < 1 ends the ray at the horizon test below */
                if (rejected) radius_fallback(r2, r, y, hy, small);
            }
            if (r < kEventHorizon * 1.01f) { hit = true; acc.t = 0.0f; steps = k; break; }
            const bool near_bh = r < 18.0f;
            const bool in_disk = fabsf(rel_p.y) < kDiskH * 5.0f && r < kDiskOut + 5.0f;
            const bool in_cloud = fabsf(rel_p.y) < kCloudH * 1.5f && r < kCloudOut;
            float h, hh, h6;
            zone_step(near_bh, in_disk, h, hh, h6);
            integrate_rk4_lean<SPIN, false, FMA>(p, vel, h, hh, h6, a.drag_c, r2, r, y, hy, ys, hs, hcp);
            if (MEDIA != 0 && (in_disk || in_cloud)) {
                float d_disk, d_cloud;
                media_densities<MEDIA>(rel_p, a.time, in_disk, in_cloud, a.lut_acc, a.lut_dust, a.dust_bands, oob, d_disk, d_cloud);
                accumulate_sample(acc, d_disk, d_cloud, rel_p, r, vel, h, a.spin);
            }
            ++k;
            if (r > 250.0f && (FMA ? dot_fma(rel_p, vel) : dot(rel_p, vel)) > 0.0f) { steps = k; break; }     /* raymarcher.cu:120 */
        }
#else
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
                    media_densities<MEDIA>(rel_p, a.time, in_disk, in_cloud, a.lut_acc, a.lut_dust, a.dust_bands, oob, d_disk, d_cloud);
                    accumulate_sample(acc, d_disk, d_cloud, rel_p, r, vel, h, a.spin);
                }
            }
            if (r > 250.0f && (FMA ? dot_fma(rel_p, vel) : dot(rel_p, vel)) > 0.0f) { steps = k + 1; break; }     /* raymarcher.cu:120 */
        }
#endif
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
    march_inline<SPIN, MEDIA, ARITH, true>(a, p, vel, acc, hit, i, DEBUG ? a.dbg.d_lut_oob : nullptr);
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
    /* `i` is a per-lane variable only across rounds: the lanes that march in a round all start at the same step (0, or the
     * step the pool ran out at, which is a wave-uniform event) -- but a first round can PROVE it to the compiler-independent
     * v_readfirstlane below, a resumed one only by that argument, so only the first round keeps the counter scalar */
    constexpr bool UK = !RESUME && LEAN && RRT_VAC_INNER;
    for (; active && i < a.max_steps; ++i) {
        if (UK) i = __builtin_amdgcn_readfirstlane(i);
        v3 rel_p = p;
        float r2, r, yv, hv = 0.0f;
        bool vacuum = false;
        if constexpr (LEAN) {
            r2 = FMA ? dot_fma(rel_p, rel_p) : dot(rel_p, rel_p);
            bool rejected = sqrt_seeded_yh<1>(r2, y_seed, h_seed, r, yv, hv);
            unsigned long long rej_mask = __builtin_amdgcn_ballot_w64(rejected);
            vacuum = RRT_VACUUM_PATH && (rej_mask | __builtin_amdgcn_ballot_w64(!(r >= kVacuumR))) == 0ull;
#if RRT_VAC_INNER
            if (vacuum) {                                          /* the vacuum steps in their own loop (vacuum_run) */
                bool escaped;
                const int why = vacuum_run<SPIN, FMA>(p, vel, a.drag_c, i, a.max_steps, r2, r, yv, hv, y_seed, h_seed, hc_prev, rejected,
                                                      rej_mask, escaped);
                if (escaped) break;                                /* i counts the step just taken */
                if (why != 3) { --i; continue; }                   /* (the loop's own ++i) */
                rel_p = p;
                vacuum = false;
            }
#endif
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
/* Round 5: pass 2 is the one latency-bound kernel with spare registers -- 40 % of its wave-cycles sit in s_waitcnt at 3.8
 * resident waves per SIMD (profiles/r05_pmc_diag_before_nt.txt) -- and its code fits 64 VGPRs with two spilled registers: 8 waves
 * per SIMD (the grid is exactly 8 192 waves) instead of 5 is worth 3 % of a full disk-heavy three-pass frame (key 1 56.9 -> 55.3 ms,
 * from inside the disk 57.5 -> 55.7; 6 and 7 waves: < 1 %; a rank's share, whose pass 2 already runs under the other chain's
 * march, does not move: profiles/r05_eval_waves_ab.txt).  The in-line kernels, which share the code, gain nothing from more
 * waves (round 3) and keep 5. */
#ifndef RRT_EVAL_WAVES
#define RRT_EVAL_WAVES 8
#endif
template <int ARITH, int MEDIA>
__global__ __launch_bounds__(256, RRT_EVAL_WAVES) void eval_sample_rows(const FrameArgs a) {
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
            media_densities<MEDIA>(rel_p, a.time, in_disk, in_cloud, a.lut_acc, a.lut_dust, a.dust_bands, nullptr, d_disk, d_cloud);
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
template <bool SPIN, int ARITH, int MEDIA, bool LAST>
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
        march_inline<SPIN, MEDIA, ARITH>(a, p, vel, acc, hit, steps, nullptr);
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

#endif  /* RRT_KERNELS_H */
