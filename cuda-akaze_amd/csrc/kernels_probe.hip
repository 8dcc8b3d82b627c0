// kernels_probe.hip -- bandwidth ceilings measured on the box the bench runs on ("confirm with a copy-kernel
// ceiling on the box; report achieved/peak and achieved/copy-ceiling").  Not on the hot path.
//
//   k_copy_probe    float4 grid-stride copy with the streaming kernels' own access shape: 16 B per lane, 1 KiB per
//                   wave-instruction, non-temporal stores.  Bytes moved = 2 x size (read + write).
//   k_gather_probe  one dword per lane from pseudo-random 128-byte lines of a large buffer: the access shape of the
//                   descriptor's point samples; used to calibrate the FETCH_SIZE counter for 4-byte gathers
//                   (MI355X_MICROARCH.md: only 16 B/lane streams are calibrated).
#include "fed_common.h"

template <int NI>
__global__ __launch_bounds__(256) void k_copy_probe(const float4* __restrict__ src, float4* __restrict__ dst, long n4)
{
    const long stride = (long)gridDim.x * 256 * NI;
    long i = (long)blockIdx.x * 256 * NI + threadIdx.x;
    // NI independent 16-byte loads in flight per lane and iteration
    for (; i + 256 * (NI - 1) < n4; i += stride) {
        float4 v[NI];
#pragma unroll
        for (int j = 0; j < NI; j++) v[j] = src[i + 256 * j];
#pragma unroll
        for (int j = 0; j < NI; j++) hak_store_nt(dst + i + 256 * j, v[j]);
    }
    for (; i < n4; i += 256) hak_store_nt(dst + i, src[i]);
}

__global__ __launch_bounds__(256) void k_gather_probe(const unsigned* __restrict__ src, unsigned* __restrict__ sink, long nlines, int per_lane)
{
    // every lane walks its own multiplicative sequence of line indices: no two consecutive touches share a line
    unsigned long long s = ((unsigned long long)blockIdx.x * 256 + threadIdx.x) * 0x9E3779B97F4A7C15ull + 12345ull;
    unsigned acc = 0;
    for (int k = 0; k < per_lane; k += 4) {
        unsigned v[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            s = s * 6364136223846793005ull + 1442695040888963407ull;
            const long line = (long)((s >> 20) % (unsigned long long)nlines);
            v[j] = src[line * 32 + ((s >> 12) & 31)];
        }
        acc += v[0] ^ v[1] ^ v[2] ^ v[3];
    }
    if (acc == 0xDEADBEEFu) sink[0] = acc;                  // keeps the loads alive; practically never true
}

static int probe_time(hipEvent_t a, hipEvent_t b, int iters, double* ms)
{
    float t = 0;
    if (hipEventSynchronize(b) != hipSuccess || hipEventElapsedTime(&t, a, b) != hipSuccess) return 1;
    *ms = (double)t / iters;
    return 0;
}

// copies `bytes` (a multiple of 16) `iters` times; *ms_per_copy = average duration of one copy kernel
int hak_launch_copy_probe(long bytes, int iters, double* ms_per_copy)
{
    float4 *s = nullptr, *d = nullptr;
    if (hipMalloc((void**)&s, (size_t)bytes) != hipSuccess) return 1;
    if (hipMalloc((void**)&d, (size_t)bytes) != hipSuccess) { (void)hipFree(s); return 1; }
    (void)hipMemset(s, 1, (size_t)bytes);
    const long n4 = bytes / 16;
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    // the ceiling is the best of a few launch shapes (loads in flight per lane x blocks per CU)
    double best = 0;
    int rc = 0;
    for (int shape = 0; shape < 6 && !rc; shape++) {
        const unsigned grid = 256u * (shape % 3 == 0 ? 8 : shape % 3 == 1 ? 16 : 32);
        auto launch = [&]() {
            if (shape < 3) k_copy_probe<4><<<grid, 256>>>(s, d, n4);
            else k_copy_probe<8><<<grid, 256>>>(s, d, n4);
        };
        launch();                                           // warm-up (page tables, clocks)
        (void)hipEventRecord(a, nullptr);
        for (int i = 0; i < iters; i++) launch();
        (void)hipEventRecord(b, nullptr);
        double ms = 0;
        rc = probe_time(a, b, iters, &ms) || hipGetLastError() != hipSuccess;
        if (!rc && (best == 0 || ms < best)) best = ms;
    }
    *ms_per_copy = best;
    (void)hipEventDestroy(a); (void)hipEventDestroy(b);
    (void)hipFree(s); (void)hipFree(d);
    return rc;
}

// `per_lane` dword gathers per lane from a `bytes`-sized buffer by `blocks` x 256 lanes, `iters` times
int hak_launch_gather_probe(long bytes, int blocks, int per_lane, int iters, double* ms_per_launch)
{
    unsigned* s = nullptr;
    unsigned* sink = nullptr;
    if (hipMalloc((void**)&s, (size_t)bytes) != hipSuccess) return 1;
    if (hipMalloc((void**)&sink, 64) != hipSuccess) { (void)hipFree(s); return 1; }
    (void)hipMemset(s, 0, (size_t)bytes);
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    k_gather_probe<<<blocks, 256>>>(s, sink, bytes / 128, per_lane);
    (void)hipEventRecord(a, nullptr);
    for (int i = 0; i < iters; i++) k_gather_probe<<<blocks, 256>>>(s, sink, bytes / 128, per_lane);
    (void)hipEventRecord(b, nullptr);
    const int rc = probe_time(a, b, iters, ms_per_launch) || hipGetLastError() != hipSuccess;
    (void)hipEventDestroy(a); (void)hipEventDestroy(b);
    (void)hipFree(s); (void)hipFree(sink);
    return rc;
}
