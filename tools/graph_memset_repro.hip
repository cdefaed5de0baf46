// dev tool (GPU): does a hipMemsetAsync captured into a hipGraph clear its buffer on EVERY replay?
//
// Round 4 replaced the hipMemsetAsync of the three-pass workspace header by a kernel (zero_words) after a captured launch had
// replayed with stale counters; ADVICE r04 asked for the memset node to be isolated from everything else that round's capture
// path contained.  This is that case and nothing else: capture { memset(buf, 0, bytes); kernel: out[i] = buf[i]; buf[i] += 7 },
// replay it `reps` times, and count the words of `out` that are not 0 after each replay -- for the sizes and alignments the
// workspace header takes (256 B of counters + 16 B per wavefront, rounded to 256).  A correct memset node gives 0 everywhere.
//   hipcc --offload-arch=gfx950 -O2 tools/graph_memset_repro.hip -o tools/graph_memset_repro && tools/graph_memset_repro
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
__global__ void read_then_dirty(unsigned* buf, unsigned* out, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { out[i] = buf[i]; buf[i] += 7u; }
}
// variant 2's kernels: word 1 of the header is a counter every thread bumps (atomics), word 2 counts rounds, the rest is scribbled on
__global__ void bump_header(unsigned* hdr, unsigned* far_rows, size_t n_words) {
    atomicAdd(&hdr[1], 1u);
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (4 + i < n_words) hdr[4 + i] = 0xDEAD0000u + (unsigned)i;
    far_rows[i] = hdr[2];
}
__global__ void read_header(const unsigned* hdr, unsigned* seen, int round) { if (threadIdx.x == 0) seen[round] = hdr[1]; }
__global__ void next_round(unsigned* hdr) { hdr[1] = 0u; hdr[2] += 1u; }

int main() {
    const size_t sizes[] = {256, 4096, 256 + 16 * 1000, 256 + 16 * 16200, 256 + 16 * 129600, (size_t)8 << 20};
    const size_t offsets[] = {0, 256, 4096 + 256};
    int bad_total = 0;
    hipStream_t st;
    CK(hipStreamCreate(&st));
    unsigned* h_pinned;
    CK(hipHostMalloc(&h_pinned, 4096));
    // variant 0: memset + kernel, thread-local capture.  variant 1 (closer to the library's captured three-pass launch): global
    // capture mode, a device-to-pinned-host copy node of the first 64 bytes behind the kernel, a second kernel behind that.
    for (int variant = 0; variant < 2; ++variant)
    for (size_t bytes0 : sizes) for (size_t off : offsets) {
        const size_t bytes = (bytes0 + 255) & ~(size_t)255, n = bytes / 4;
        unsigned *base, *out;
        CK(hipMalloc(&base, bytes + 8192)); CK(hipMalloc(&out, bytes));
        unsigned* buf = base + off / 4;
        CK(hipMemset(base, 0xAB, bytes + 8192));
        hipGraph_t g; hipGraphExec_t ge;
        // the buffer is used LIVE first, as the library's workspace is (it holds a finished launch's counters and headers)
        hipLaunchKernelGGL(read_then_dirty, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, buf, out, n);
        CK(hipStreamSynchronize(st));
        CK(hipStreamBeginCapture(st, variant ? hipStreamCaptureModeGlobal : hipStreamCaptureModeThreadLocal));
        CK(hipMemsetAsync(buf, 0, bytes, st));
        hipLaunchKernelGGL(read_then_dirty, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, buf, out, n);
        if (variant) {
            CK(hipMemcpyAsync(h_pinned, buf, 64, hipMemcpyDeviceToHost, st));
            hipLaunchKernelGGL(read_then_dirty, dim3(1), dim3(64), 0, st, buf, out + 0, (size_t)0);     // (touches nothing: a node behind the copy)
        }
        CK(hipStreamEndCapture(st, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        std::vector<unsigned> h(n);
        for (int rep = 0; rep < 4; ++rep) {
            CK(hipGraphLaunch(ge, st));
            CK(hipStreamSynchronize(st));
            CK(hipMemcpy(h.data(), out, bytes, hipMemcpyDeviceToHost));
            size_t bad = 0, first = n;
            for (size_t i = 0; i < n; ++i) if (h[i] != 0u) { if (first == n) first = i; ++bad; }
            printf("variant %d bytes %9zu offset %5zu replay %d: %zu of %zu words not cleared%s", variant, bytes, off, rep, bad, n, bad ? "" : "\n");
            if (bad) { printf("  (first at word %zu = 0x%08x)\n", first, h[first]); ++bad_total; }
        }
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g)); CK(hipFree(base)); CK(hipFree(out));
    }
    // variant 2: the shape of the library's captured three-pass launch -- a 64 MB allocation whose first `hdr` bytes are the memset
    // region, a non-blocking capture stream in global mode, per round: a kernel with atomics on the header + writes far into
    // the allocation, a reader of the header, a one-thread kernel that rewrites counters; then a device-to-pinned-host copy of the
    // counters; between replays a live memset of ANOTHER buffer on the null stream (torch's d3.zero_()).  `first_words` must read
    // (rounds, 0, ...) after every replay if the memset node cleared the header.
    {
        hipStream_t cs;
        CK(hipStreamCreateWithFlags(&cs, hipStreamNonBlocking));
        unsigned char* big; unsigned* other; unsigned* seen;
        const size_t total = (size_t)64 << 20;
        CK(hipMalloc(&big, total)); CK(hipMalloc(&other, 1 << 20)); CK(hipMalloc(&seen, 4096));
        for (size_t hdr : {(size_t)1792, (size_t)4352, (size_t)(256 + 16 * 16200 + 255) & ~(size_t)255}) {
            CK(hipMemset(big, 0xCD, total));
            auto enqueue = [&](hipStream_t st) -> int {
                CK(hipMemsetAsync(big, 0, hdr, st));
                for (int round = 0; round < 3; ++round) {
                    hipLaunchKernelGGL(bump_header, dim3(64), dim3(64), 0, st, reinterpret_cast<unsigned*>(big), reinterpret_cast<unsigned*>(big + ((size_t)32 << 20)), hdr / 4);
                    hipLaunchKernelGGL(read_header, dim3(1), dim3(64), 0, st, reinterpret_cast<unsigned*>(big), seen, round);
                    hipLaunchKernelGGL(next_round, dim3(1), dim3(1), 0, st, reinterpret_cast<unsigned*>(big));
                }
                CK(hipMemcpyAsync(h_pinned, big, 64, hipMemcpyDeviceToHost, st));
                return 0;
            };
            if (enqueue(cs)) return 2;                                  // live once, as the library's workspace is used before the capture
            CK(hipStreamSynchronize(cs));
            hipGraph_t g; hipGraphExec_t ge;
            CK(hipStreamBeginCapture(cs, hipStreamCaptureModeGlobal));
            if (enqueue(cs)) return 2;
            CK(hipStreamEndCapture(cs, &g));
            CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
            for (int rep = 0; rep < 4; ++rep) {
                CK(hipMemsetAsync(other, 0, 1 << 20, nullptr));         // live work on the null stream in between
                CK(hipStreamSynchronize(nullptr));
                CK(hipGraphLaunch(ge, cs));
                CK(hipStreamSynchronize(cs));
                unsigned h[16];
                CK(hipMemcpy(h, seen, sizeof(h), hipMemcpyDeviceToHost));
                // seen[round] = header word 1 (a counter bump_header adds 64*64 to per round, next_round resets) as the reader saw it
                const bool ok = h[0] == 64u * 64u && h[1] == 64u * 64u && h[2] == 64u * 64u && h_pinned[2] == 3u;
                printf("variant 2 header %6zu bytes replay %d: reader saw %u %u %u, rounds counter %u%s\n", hdr, rep, h[0], h[1], h[2], h_pinned[2], ok ? "" : "   <-- STALE HEADER");
                if (!ok) ++bad_total;
            }
            CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
        }
    }
    // variant 3: variant 0's graph instantiated the way torch.cuda.graph instantiates (hipGraphInstantiateWithFlags,
    // hipGraphInstantiateFlagAutoFreeOnLaunch) -- under torch's capture the defect reproduces (tools/graph_memset_repro_torch.py)
    for (size_t bytes : {(size_t)1792, (size_t)4352, (size_t)259584}) {
        const size_t n = bytes / 4;
        unsigned *buf, *out;
        CK(hipMalloc(&buf, bytes)); CK(hipMalloc(&out, bytes));
        CK(hipMemset(buf, 0xAB, bytes));
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
        CK(hipMemsetAsync(buf, 0, bytes, st));
        hipLaunchKernelGGL(read_then_dirty, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, buf, out, n);
        CK(hipStreamEndCapture(st, &g));
        CK(hipGraphInstantiateWithFlags(&ge, g, hipGraphInstantiateFlagAutoFreeOnLaunch));
        std::vector<unsigned> h(n);
        for (int rep = 0; rep < 4; ++rep) {
            CK(hipGraphLaunch(ge, st));
            CK(hipStreamSynchronize(st));
            CK(hipMemcpy(h.data(), out, bytes, hipMemcpyDeviceToHost));
            size_t bad = 0, first = n, last = 0;
            for (size_t i = 0; i < n; ++i) if (h[i] != 0u) { if (first == n) first = i; last = i; ++bad; }
            printf("variant 3 (AutoFreeOnLaunch) bytes %7zu replay %d: %zu of %zu words not cleared", bytes, rep, bad, n);
            if (bad) { printf("   <-- STALE: words %zu..%zu, first = 0x%08x", first, last, h[first]); ++bad_total; }
            printf("\n");
        }
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g)); CK(hipFree(buf)); CK(hipFree(out));
    }
    // variant 4: the template graph is DESTROYED right after instantiation, as torch.cuda.graph does (it keeps only the executable
    // graph), and the host heap is churned before the launches: does the executable graph still own its memset parameters?
    for (size_t bytes : {(size_t)1792, (size_t)4352, (size_t)259584}) {
        const size_t n = bytes / 4;
        unsigned *buf, *out;
        CK(hipMalloc(&buf, bytes)); CK(hipMalloc(&out, bytes));
        CK(hipMemset(buf, 0xAB, bytes));
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
        CK(hipMemsetAsync(buf, 0, bytes, st));
        hipLaunchKernelGGL(read_then_dirty, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, buf, out, n);
        CK(hipStreamEndCapture(st, &g));
        CK(hipGraphInstantiateWithFlags(&ge, g, hipGraphInstantiateFlagAutoFreeOnLaunch));
        CK(hipGraphDestroy(g));
        std::vector<std::vector<char>> churn;
        for (int k = 0; k < 2000; ++k) churn.emplace_back(64 + 37 * (k % 50), (char)k);       // reuse whatever the template graph freed
        std::vector<unsigned> h(n);
        for (int rep = 0; rep < 4; ++rep) {
            CK(hipGraphLaunch(ge, st));
            CK(hipStreamSynchronize(st));
            CK(hipMemcpy(h.data(), out, bytes, hipMemcpyDeviceToHost));
            size_t bad = 0, first = n, last = 0;
            for (size_t i = 0; i < n; ++i) if (h[i] != 0u) { if (first == n) first = i; last = i; ++bad; }
            printf("variant 4 (template graph destroyed) bytes %7zu replay %d: %zu of %zu words not cleared", bytes, rep, bad, n);
            if (bad) { printf("   <-- STALE: words %zu..%zu, first = 0x%08x", first, last, h[first]); ++bad_total; }
            printf("\n");
        }
        CK(hipGraphExecDestroy(ge)); CK(hipFree(buf)); CK(hipFree(out));
    }
    printf("%s\n", bad_total ? "MEMSET NODE DEFECT REPRODUCED" : "memset nodes cleared their buffers on every replay: not reproduced in isolation");
    return 0;
}
