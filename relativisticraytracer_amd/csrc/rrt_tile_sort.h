/*
 * rrt_tile_sort.h -- the sort behind cost-ordered dispatch (rrt_tile_order): wave tiles by cost, longest first.
 *
 * Hand-written for gfx950 (round 5; rounds 3-4 called hipcub::DeviceRadixSort, AMD's CUB-compatibility layer).  No
 * counterpart in the reference, whose launch is one fixed grid (src/raymarcher.cu:176-180).
 *
 * What is sorted: n = grid_x * grid_y wave tiles (129 600 for a 4K frame, 518 400 at 8K) on the 16 significant bits of
 * their cost (bits kTileCostSortLo .. kTileCostSortHi-1 of clocks / 16: 0.5 us steps), DESCENDING and STABLE with respect
 * to the static dispatch order (row blocks from the middle of the frame outwards): tiles of equal cost keep the order the
 * static launch would have given them, which is the right order where costs tie (sky tiles) -- hipcub's ties came out in
 * tile-index order, top row first.
 *
 * How: LSD radix sort, two passes of 8 bits, FOUR short launches: histogram, scatter, histogram, scatter.  The problem is small (0.5 MB of
 * keys at 4K) and entirely latency-bound, so the design minimises dependent memory round trips and kernel boundaries rather
 * than maximising parallelism:
 *   - few fat blocks: 2 048 keys per block (64 blocks at 4K), a thread's keys loaded in one batch and kept in registers for both
 *     the counting and the scattering phase of the kernel; 1 024 threads x 2 keys -- a lone wavefront retires an instruction every
 *     ~10 clocks, so a thread's serial instruction count is the time, and sixteen wavefronts on one CU overlap each other's
 *     latencies (256 threads x 16 keys took 36 us per scatter, x 8 keys 20 us, 1 024-key blocks with a scan kernel in between 46 us;
 *     profiles/r05_tile_sort_timing.txt);
 *   - no scan kernel: counters live as hist[block][digit]; a scatter block's thread d walks its digit's column once
 *     (n_blocks coalesced loads shared by four threads per digit, issued together with the key loads), which gives it the digit's
 *     total and the count of the blocks before its own; an LDS scan of the 256 totals finishes the offsets;
 *   - pass 0 reads the costs through the static slot -> tile map; no global atomics anywhere (pass 0's scatter once built pass 1's
 *     counters with them: the high byte of a cost is shared by most tiles of a frame, and same-address atomics made it 20 us);
 *   - 64-wide wavefronts rank their lanes with ballots (eight ballots give every lane the mask of the lanes that hold the
 *     same digit; the popcount below the lane is its rank); the sixteen wavefronts of a block take consecutive sixteenths of its
 *     chunk, LDS holds their per-digit write cursors.
 * No spin-waits, no inter-block communication other than kernel boundaries.
 */
#ifndef RRT_TILE_SORT_H
#define RRT_TILE_SORT_H

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rrt_sort {
namespace {                                   /* internal linkage: the library exports nothing of this */

#ifndef RRT_SORT_THREADS
#define RRT_SORT_THREADS 1024
#endif
constexpr unsigned kThreads = RRT_SORT_THREADS;  /* sixteen wavefronts per block (one CU's worth at 4 per SIMD) */
constexpr unsigned kWaves = kThreads / 64u;
constexpr unsigned kParts = kThreads / 256u;  /* threads per digit in the column walk */
constexpr unsigned kBatch = 2048;             /* keys per block and batch */
constexpr unsigned kItems = kBatch / kThreads;   /* keys per thread and batch: 2 */
static_assert(kThreads % 256u == 0 && kThreads <= 1024u && kItems * kThreads == kBatch, "a block is 256 ... 1 024 threads over 2 048 keys");
constexpr unsigned kMaxBlocks = 1024;         /* a scatter thread walks a column of n_blocks counters: bound it */

/* batches per block: 1 unless n needs more than kMaxBlocks blocks (> 2 M wave tiles) */
inline unsigned reps_for(size_t n) {
    size_t r = 1;
    while ((n + kBatch * r - 1) / (kBatch * r) > kMaxBlocks) r *= 2;
    return (unsigned)r;
}
inline unsigned blocks_for(size_t n) { const size_t c = (size_t)kBatch * reps_for(n); return (unsigned)((n + c - 1) / c); }
/* unsigned words of scratch any sort needs: two hist[block][digit] matrices */
constexpr size_t kScratchWords = (size_t)2 * 256 * kMaxBlocks;

/* dispatch slot -> wave tile of the static order (rrt_kernels.h: row_block): row blocks mid, mid+1, mid-1 ...
 * slot / grid_x by multiplication: with inv_gx = ceil(2^48 / grid_x) the quotient is exact for slot < 2^30, grid_x < 2^18 (the error
 * term slot / 2^48 < 2^-18 < 1 / grid_x) and the product fits 64 bits for quotients < 2^16 (gridDim.y <= 65 535) */
__device__ __forceinline__ unsigned static_tile(unsigned slot, unsigned grid_x, unsigned grid_y, unsigned long long inv_gx) {
    const unsigned j = (unsigned)(((unsigned long long)slot * inv_gx) >> 48), col = slot - j * grid_x;
    const int mid = ((int)grid_y - 1) >> 1;
    const int rb = (j & 1u) ? mid + (int)((j + 1u) >> 1) : mid - (int)(j >> 1);
    return (unsigned)rb * grid_x + col;
}

/* descending order = ascending order of the inverted digit */
__device__ __forceinline__ unsigned digit_of(unsigned key, int shift) { return 255u - ((key >> shift) & 255u); }

struct Pass {
    const unsigned* cost;          /* pass 0: cost per wave tile */
    const unsigned* keys_in;       /* pass 1: keys / tiles as pass 0 left them */
    const unsigned* vals_in;
    unsigned* keys_out;            /* pass 0 only */
    unsigned* vals_out;
    unsigned* hist;                /* [n_blocks][256] counts of THIS pass's input chunks */
    unsigned n, reps, n_blocks, grid_x, grid_y;
    unsigned long long inv_gx;     /* ceil(2^48 / grid_x) */
    int shift;
};

/* element i of the pass's input; an index past the end reads as (key 0, tile 0) and is never written anywhere */
template <int PASS>
__device__ __forceinline__ void load_item(const Pass& p, unsigned i, unsigned& key, unsigned& val) {
    key = 0u; val = 0u;
    if (i < p.n) {
        if (PASS == 0) { val = static_tile(i, p.grid_x, p.grid_y, p.inv_gx); key = p.cost[val]; }
        else { key = p.keys_in[i]; val = p.vals_in[i]; }
    }
}

/* hist[block][d] = digit counts of the block's chunk of the pass's input.  (The first version let pass 0's scatter build pass 1's
 * counters with global atomics, to save this launch: the next digit is the HIGH byte of a cost, a frame's tiles share a few of those,
 * and the same-address atomics made that scatter 20 us where its twin takes 6; a 5 us kernel of its own is the cheaper way.) */
template <int PASS>
__global__ __launch_bounds__(kThreads) void histogram(Pass p) {
    __shared__ unsigned h[256];
    if (threadIdx.x < 256u) h[threadIdx.x] = 0u;
    __syncthreads();
    const unsigned base = blockIdx.x * kBatch * p.reps;
    for (unsigned r = 0; r < p.reps; ++r) {
        unsigned key[kItems], val[kItems];
#pragma unroll
        for (unsigned k = 0; k < kItems; ++k) load_item<PASS>(p, base + r * kBatch + k * kThreads + threadIdx.x, key[k], val[k]);
#pragma unroll
        for (unsigned k = 0; k < kItems; ++k)
            if (base + r * kBatch + k * kThreads + threadIdx.x < p.n) atomicAdd(&h[digit_of(key[k], p.shift)], 1u);
    }
    __syncthreads();
    if (threadIdx.x < 256u) p.hist[blockIdx.x * 256u + threadIdx.x] = h[threadIdx.x];
}

/* Stable scatter of the block's chunk.  Wavefront w owns the w-th sixteenth of the chunk (consecutive keys), 64 keys per
 * step in order; a key's place = (keys with a smaller digit anywhere) + (same digit in earlier blocks) + (same digit in
 * earlier wavefronts of this block) + (same digit earlier in this wavefront's share).
 * 1 024 threads x 2 keys (round 5, second version; the first had 256 x 8 and took 20 us): the serial part of a thread -- the
 * ranking loop and its share of the column walk -- is a quarter as long, and the sixteen wavefronts of the block overlap each
 * other's latencies on the CU's four SIMDs. */
template <int PASS>
__global__ __launch_bounds__(kThreads) void scatter(Pass p) {
    __shared__ unsigned cur[kWaves][256];
    __shared__ unsigned part_tot[kParts][256], part_bef[kParts][256];
    __shared__ unsigned wave_tot[4];
    const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const unsigned share = kItems * 64u * p.reps;                       /* keys per wavefront */
    const unsigned w_base = blockIdx.x * kBatch * p.reps + wave * share;
    /* the keys of the first batch (the only one unless the frame has > 2 M wave tiles): loads in flight ... */
    unsigned key[kItems], val[kItems];
#pragma unroll
    for (unsigned k = 0; k < kItems; ++k) load_item<PASS>(p, w_base + k * 64u + lane, key[k], val[k]);
    /* ... together with digit d's column of the block histograms (the digit's total and what the earlier blocks hold), kParts
     * threads per digit, each every kParts-th block: the 64 blocks of a 4K frame are ONE round trip of 16 loads per thread */
    {
        const unsigned d = threadIdx.x & 255u, part = threadIdx.x >> 8;
        unsigned total = 0u, before = 0u;
        constexpr unsigned kWalk = 16;
        for (unsigned b0 = part; b0 < p.n_blocks; b0 += kWalk * kParts) {
            unsigned v[kWalk];
#pragma unroll
            for (unsigned j = 0; j < kWalk; ++j) v[j] = b0 + j * kParts < p.n_blocks ? p.hist[(b0 + j * kParts) * 256u + d] : 0u;
#pragma unroll
            for (unsigned j = 0; j < kWalk; ++j) { total += v[j]; before += b0 + j * kParts < blockIdx.x ? v[j] : 0u; }
        }
        part_tot[part][d] = total; part_bef[part][d] = before;
    }
    for (unsigned k = threadIdx.x; k < kWaves * 256u; k += kThreads) (&cur[0][0])[k] = 0u;
    __syncthreads();
    /* per-wave digit counts of this block's chunk */
    for (unsigned r = 0; r < p.reps; ++r) {
        if (r > 0) {
#pragma unroll
            for (unsigned k = 0; k < kItems; ++k) load_item<PASS>(p, w_base + r * kItems * 64u + k * 64u + lane, key[k], val[k]);
        }
#pragma unroll
        for (unsigned k = 0; k < kItems; ++k)
            if (w_base + r * kItems * 64u + k * 64u + lane < p.n) atomicAdd(&cur[wave][digit_of(key[k], p.shift)], 1u);
    }
    /* threads 0 .. 255 (wavefronts 0 .. 3): thread d holds digit d's total and the earlier blocks' count; exclusive scan of the
     * 256 totals */
    unsigned total = 0u, before = 0u, incl = 0u;
    if (threadIdx.x < 256u) {
#pragma unroll
        for (unsigned q = 0; q < kParts; ++q) { total += part_tot[q][threadIdx.x]; before += part_bef[q][threadIdx.x]; }
        incl = total;
#pragma unroll
        for (unsigned o = 1; o < 64u; o <<= 1) {
            const unsigned up = (unsigned)__shfl_up((int)incl, o);
            if (lane >= o) incl += up;
        }
        if (lane == 63u) wave_tot[wave] = incl;
    }
    __syncthreads();
    if (threadIdx.x < 256u) {
        unsigned smaller = incl - total;
        for (unsigned w = 0; w < wave; ++w) smaller += wave_tot[w];
        const unsigned d = threadIdx.x;
        unsigned at = smaller + before;                                  /* where this block's keys of digit d start */
#pragma unroll
        for (unsigned w = 0; w < kWaves; ++w) { const unsigned c = cur[w][d]; cur[w][d] = at; at += c; }
    }
    __syncthreads();
    for (unsigned r = 0; r < p.reps; ++r) {
        if (p.reps > 1) {                                                /* more than one batch: reload (the first one too) */
#pragma unroll
            for (unsigned k = 0; k < kItems; ++k) load_item<PASS>(p, w_base + r * kItems * 64u + k * 64u + lane, key[k], val[k]);
        }
#pragma unroll
        for (unsigned k = 0; k < kItems; ++k) {
            const bool valid = w_base + r * kItems * 64u + k * 64u + lane < p.n;
            const unsigned d = digit_of(key[k], p.shift);
            unsigned long long peers = __ballot(valid);
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                const unsigned long long has = __ballot((d >> b) & 1u);
                peers &= ((d >> b) & 1u) ? has : ~has;
            }
            const unsigned long long below = peers & ((1ull << lane) - 1ull);
            if (valid) {
                const unsigned pos = cur[wave][d] + (unsigned)__popcll(below);
                if (PASS == 0) p.keys_out[pos] = key[k];
                p.vals_out[pos] = val[k];
            }
            __builtin_amdgcn_wave_barrier();                    /* every lane has read its cursor before the leaders move them */
            if (valid && below == 0ull) cur[wave][d] += (unsigned)__popcll(peers);
            __builtin_amdgcn_wave_barrier();
        }
    }
}

/* Enqueue the sort on `st`: perm_out[slot] = wave tile, longest first.  keys_tmp / vals_tmp: n words each; scratch:
 * kScratchWords words.  lo_bit: the lowest of the 16 key bits.  Returns the first HIP launch error. */
inline hipError_t enqueue(const unsigned* d_cost, unsigned* d_keys_tmp, unsigned* d_vals_tmp, unsigned* d_scratch, unsigned* d_perm_out,
                          size_t n, unsigned grid_x, unsigned grid_y, int lo_bit, hipStream_t st) {
    if (n == 0) return hipSuccess;
    Pass p{};
    p.n = (unsigned)n; p.reps = reps_for(n); p.n_blocks = blocks_for(n); p.grid_x = grid_x; p.grid_y = grid_y;
    p.inv_gx = ((1ull << 48) + grid_x - 1) / grid_x;
    p.cost = d_cost; p.keys_out = d_keys_tmp; p.vals_out = d_vals_tmp; p.hist = d_scratch; p.shift = lo_bit;
    hipLaunchKernelGGL(histogram<0>, dim3(p.n_blocks), dim3(kThreads), 0, st, p);
    hipLaunchKernelGGL(scatter<0>, dim3(p.n_blocks), dim3(kThreads), 0, st, p);
    Pass q = p;
    q.cost = nullptr; q.keys_in = d_keys_tmp; q.vals_in = d_vals_tmp; q.keys_out = nullptr; q.vals_out = d_perm_out;
    q.hist = d_scratch + (size_t)256 * kMaxBlocks; q.shift = lo_bit + 8;
    hipLaunchKernelGGL(histogram<1>, dim3(p.n_blocks), dim3(kThreads), 0, st, q);
    hipLaunchKernelGGL(scatter<1>, dim3(p.n_blocks), dim3(kThreads), 0, st, q);
    return hipGetLastError();
}

}  // namespace
}  // namespace rrt_sort
#endif
