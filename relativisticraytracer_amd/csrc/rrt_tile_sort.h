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
 * How: LSD radix sort, two passes of 8 bits, each pass = per-block digit histogram -> exclusive scan over (digit, block)
 * -> stable scatter.  64-wide wavefronts rank their lanes with ballots (eight ballots give every lane the mask of the lanes
 * that hold the same digit; popcount below the lane = its rank), four wavefronts of a block split its chunk in order, and
 * LDS holds the block's per-wave digit counters.  Pass 0 reads the costs through the static slot -> tile map and fuses the
 * NEXT pass's block histogram into its scatter (the destination of an element says which block will read it), so a sort is
 * five launches: histogram, scan, scatter(+histogram), scan, scatter.  Everything a pass touches (<= 4 MB at 8K) stays in L2.
 * No spin-waits, no inter-block communication other than kernel boundaries.
 */
#ifndef RRT_TILE_SORT_H
#define RRT_TILE_SORT_H

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rrt_sort {

constexpr unsigned kThreads = 256;            /* four wavefronts per block */
constexpr unsigned kMaxBlocks = 1024;         /* the scan kernel is one block: bound what it has to scan (256 K counters) */
constexpr unsigned kScanThreads = 1024;

/* keys per block: 1024 (four per thread) unless that would need more than kMaxBlocks blocks */
inline unsigned chunk_for(size_t n) {
    size_t c = 1024;
    while ((n + c - 1) / c > kMaxBlocks) c *= 2;
    return (unsigned)c;
}
inline unsigned blocks_for(size_t n) { const unsigned c = chunk_for(n); return (unsigned)((n + c - 1) / c); }
/* unsigned words of scratch a sort of n keys needs: two (digit, block) counter matrices */
inline size_t scratch_words(size_t n) { return (size_t)2 * 256 * blocks_for(n); }

/* dispatch slot -> wave tile of the static order (rrt_hip.hip: row_block): row blocks mid, mid+1, mid-1 ... */
__device__ __forceinline__ unsigned static_tile(unsigned slot, unsigned grid_x, unsigned grid_y) {
    const unsigned j = slot / grid_x, col = slot - j * grid_x;
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
    unsigned* hist;                /* [256][n_blocks] of THIS pass: counts in, exclusive offsets after the scan */
    unsigned* hist_next;           /* pass 0: the next pass's counters (zeroed by the histogram kernel, filled by the scatter) */
    unsigned n, chunk, n_blocks, grid_x, grid_y;
    int shift, shift_next;
};

template <int PASS>
__device__ __forceinline__ void load_item(const Pass& p, unsigned i, unsigned& key, unsigned& val) {
    if (PASS == 0) { val = static_tile(i, p.grid_x, p.grid_y); key = p.cost[val]; }
    else { key = p.keys_in[i]; val = p.vals_in[i]; }
}

/* pass 0 only: per-block digit counts of the block's chunk -> hist[d][block]; zero the next pass's column */
__global__ __launch_bounds__(kThreads) void histogram0(Pass p) {
    __shared__ unsigned h[256];
    h[threadIdx.x] = 0u;
    __syncthreads();
    const unsigned base = blockIdx.x * p.chunk;
    for (unsigned k = threadIdx.x; k < p.chunk; k += kThreads) {
        const unsigned i = base + k;
        if (i < p.n) {
            unsigned key, val;
            load_item<0>(p, i, key, val);
            atomicAdd(&h[digit_of(key, p.shift)], 1u);
        }
    }
    __syncthreads();
    p.hist[threadIdx.x * p.n_blocks + blockIdx.x] = h[threadIdx.x];
    p.hist_next[threadIdx.x * p.n_blocks + blockIdx.x] = 0u;
}

/* exclusive scan of m counters in place, one block */
__global__ __launch_bounds__(kScanThreads) void scan_counters(unsigned* c, unsigned m) {
    __shared__ unsigned wave_sum[kScanThreads / 64];
    const unsigned per = (m + kScanThreads - 1) / kScanThreads;
    const unsigned lo = threadIdx.x * per, hi = lo + per < m ? lo + per : m;
    unsigned s = 0u;
    for (unsigned i = lo; i < hi; ++i) s += c[i];
    /* inclusive scan inside the wavefront */
    unsigned incl = s;
    const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
#pragma unroll
    for (unsigned o = 1; o < 64; o <<= 1) {
        const unsigned up = (unsigned)__shfl_up((int)incl, o);
        if (lane >= o) incl += up;
    }
    if (lane == 63u) wave_sum[wave] = incl;
    __syncthreads();
    unsigned before = 0u;
    for (unsigned w = 0; w < wave; ++w) before += wave_sum[w];
    unsigned run = before + incl - s;
    for (unsigned i = lo; i < hi; ++i) { const unsigned v = c[i]; c[i] = run; run += v; }
}

/* stable scatter of the block's chunk to the offsets of the scanned histogram.  The four wavefronts take consecutive
 * quarters of the chunk; each counts its digits (LDS), the counters become the waves' write cursors, then every wavefront
 * walks its quarter 64 keys at a time: a lane's place = its wave's cursor of that digit + the number of lower lanes that
 * hold the same digit. */
template <int PASS>
__global__ __launch_bounds__(kThreads) void scatter(Pass p) {
    __shared__ unsigned cur[4][256];
    const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    for (unsigned k = threadIdx.x; k < 4u * 256u; k += kThreads) (&cur[0][0])[k] = 0u;
    __syncthreads();
    const unsigned quarter = p.chunk / 4u;
    const unsigned w_base = blockIdx.x * p.chunk + wave * quarter;
    for (unsigned k = lane; k < quarter; k += 64u) {
        const unsigned i = w_base + k;
        if (i < p.n) {
            unsigned key, val;
            load_item<PASS>(p, i, key, val);
            atomicAdd(&cur[wave][digit_of(key, p.shift)], 1u);
        }
    }
    __syncthreads();
    {   /* thread d: the four waves' cursors of digit d */
        const unsigned d = threadIdx.x;
        unsigned at = p.hist[d * p.n_blocks + blockIdx.x];
#pragma unroll
        for (unsigned w = 0; w < 4u; ++w) { const unsigned c = cur[w][d]; cur[w][d] = at; at += c; }
    }
    __syncthreads();
    for (unsigned k0 = 0; k0 < quarter; k0 += 64u) {           /* wave-uniform trip count */
        const unsigned i = w_base + k0 + lane;
        const bool valid = i < p.n;
        unsigned key = 0u, val = 0u;
        if (valid) load_item<PASS>(p, i, key, val);
        const unsigned d = digit_of(key, p.shift);
        unsigned long long peers = __ballot(valid);
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const unsigned long long has = __ballot((d >> b) & 1u);
            peers &= ((d >> b) & 1u) ? has : ~has;
        }
        const unsigned long long below = peers & ((1ull << lane) - 1ull);
        if (valid) {
            const unsigned pos = cur[wave][d] + (unsigned)__popcll(below);
            if (PASS == 0) {
                p.keys_out[pos] = key;
                p.vals_out[pos] = val;
                atomicAdd(&p.hist_next[digit_of(key, p.shift_next) * p.n_blocks + pos / p.chunk], 1u);
            } else {
                p.vals_out[pos] = val;
            }
        }
        __builtin_amdgcn_wave_barrier();                        /* every lane has read its cursor before the leaders move them */
        if (valid && below == 0ull) cur[wave][d] += (unsigned)__popcll(peers);
        __builtin_amdgcn_wave_barrier();
    }
}

/* Enqueue the sort on `st`: perm_out[slot] = wave tile, longest first.  keys_tmp / vals_tmp: n words each; scratch:
 * scratch_words(n) words.  lo_bit: the lowest of the 16 key bits.  Returns the first HIP launch error. */
inline hipError_t enqueue(const unsigned* d_cost, unsigned* d_keys_tmp, unsigned* d_vals_tmp, unsigned* d_scratch, unsigned* d_perm_out,
                          size_t n, unsigned grid_x, unsigned grid_y, int lo_bit, hipStream_t st) {
    if (n == 0) return hipSuccess;
    Pass p{};
    p.n = (unsigned)n; p.chunk = chunk_for(n); p.n_blocks = blocks_for(n); p.grid_x = grid_x; p.grid_y = grid_y;
    unsigned* hist0 = d_scratch;
    unsigned* hist1 = d_scratch + (size_t)256 * p.n_blocks;
    p.cost = d_cost; p.keys_out = d_keys_tmp; p.vals_out = d_vals_tmp;
    p.hist = hist0; p.hist_next = hist1; p.shift = lo_bit; p.shift_next = lo_bit + 8;
    hipLaunchKernelGGL(histogram0, dim3(p.n_blocks), dim3(kThreads), 0, st, p);
    hipLaunchKernelGGL(scan_counters, dim3(1), dim3(kScanThreads), 0, st, hist0, 256u * p.n_blocks);
    hipLaunchKernelGGL(scatter<0>, dim3(p.n_blocks), dim3(kThreads), 0, st, p);
    Pass q = p;
    q.cost = nullptr; q.keys_in = d_keys_tmp; q.vals_in = d_vals_tmp; q.keys_out = nullptr; q.vals_out = d_perm_out;
    q.hist = hist1; q.hist_next = nullptr; q.shift = lo_bit + 8; q.shift_next = 0;
    hipLaunchKernelGGL(scan_counters, dim3(1), dim3(kScanThreads), 0, st, hist1, 256u * p.n_blocks);
    hipLaunchKernelGGL(scatter<1>, dim3(p.n_blocks), dim3(kThreads), 0, st, q);
    return hipGetLastError();
}

}  // namespace rrt_sort
#endif
