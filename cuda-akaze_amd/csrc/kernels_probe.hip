// kernels_probe.hip -- bandwidth ceilings measured on the box the bench runs on ("confirm with a copy-kernel
// ceiling on the box; report achieved/peak and achieved/copy-ceiling").  Not on the hot path.
//
//   k_copy_probe    float4 grid-stride copy with the streaming kernels' own access shape: 16 B per lane, 1 KiB per
//                   wave-instruction, non-temporal stores.  Bytes moved = 2 x size (read + write).
//   k_gather_probe  one dword per lane from pseudo-random 128-byte lines of a large buffer: the access shape of the
//                   descriptor's point samples; used to calibrate the FETCH_SIZE counter for 4-byte gathers
//                   (MI355X_MICROARCH.md: only 16 B/lane streams are calibrated).
#include "fed_common.h"

// NI independent 16-byte loads in flight per lane and iteration; NT: non-temporal stores (the streaming kernels' own) or plain ones
template <int NI, bool NT>
__global__ __launch_bounds__(256) void k_copy_probe(const float4* __restrict__ src, float4* __restrict__ dst, long n4)
{
    const long stride = (long)gridDim.x * 256 * NI;
    long i = (long)blockIdx.x * 256 * NI + threadIdx.x;
    for (; i + 256 * (NI - 1) < n4; i += stride) {
        float4 v[NI];
#pragma unroll
        for (int j = 0; j < NI; j++) v[j] = src[i + 256 * j];
#pragma unroll
        for (int j = 0; j < NI; j++) {
            if (NT) hak_store_nt(dst + i + 256 * j, v[j]);
            else dst[i + 256 * j] = v[j];
        }
    }
    for (; i < n4; i += 256) dst[i] = src[i];
}

// read-only and write-only streams of the same shape (what each direction reaches alone)
__global__ __launch_bounds__(256) void k_read_probe(const float4* __restrict__ src, float4* __restrict__ sink, long n4)
{
    const long stride = (long)gridDim.x * 256 * 4;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (long i = (long)blockIdx.x * 256 * 4 + threadIdx.x; i + 768 < n4; i += stride) {
        const float4 a = src[i], b = src[i + 256], c = src[i + 512], d = src[i + 768];
        acc.x += a.x + b.x + c.x + d.x; acc.y += a.y + b.y + c.y + d.y; acc.z += a.z + b.z + c.z + d.z; acc.w += a.w + b.w + c.w + d.w;
    }
    if (acc.x == 123.456f) sink[0] = acc;                   // keeps the loads alive; never true for the constant fill
}
__global__ __launch_bounds__(256) void k_write_probe(float4* __restrict__ dst, long n4)
{
    const long stride = (long)gridDim.x * 256 * 4;
    const float4 v = make_float4(1.f, 2.f, 3.f, 4.f);
    for (long i = (long)blockIdx.x * 256 * 4 + threadIdx.x; i + 768 < n4; i += stride) {
        hak_store_nt(dst + i, v); hak_store_nt(dst + i + 256, v); hak_store_nt(dst + i + 512, v); hak_store_nt(dst + i + 768, v);
    }
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

// ---- the FED family's access shape with the arithmetic taken out: a wave streams down its 256-lane strip (240 stored columns,
// 8-column halo on both sides) through the rows of its segment plus `warm` rows above it, one 16-byte load per lane and row with
// three rows in flight, and stores NW planes with the kernels' own unconditional nt buffer stores.  What k_fed_sf<NS> would take
// if its ~330 VALU instructions per row cost nothing: the floor of the launch geometry and the read : write mix.
template <int NW>
__global__ __launch_bounds__(256) void k_stream_probe(const float* __restrict__ src, float* __restrict__ dst, long stride, long plane,
                                                      int w, int h, int p, int ry, int warm, int nbx, int nby, int nimg)
{
    int bx, by, img;
    if (!hak_xcd_decode(nbx, nby, nimg, bx, by, img)) return;
    const float* L = src + (long)img * stride;
    float* D = dst + (long)img * stride * NW;
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int x0 = bx * 240 - 8 + 4 * lane;
    const int ybeg = (by * 4 + wv) * ry;
    if (ybeg >= h) return;
    const int yend = min(ybeg + ry, h);
    const bool owns = 4 * lane >= 8 && 4 * lane < 248 && x0 < w && x0 >= 0;
    const int xl = min(max(x0, 0), w - 4);
    const __amdgpu_buffer_rsrc_t r = hak_buf_rsrc(D);
    const unsigned col = owns ? (unsigned)x0 * 4u : HAK_BUF_OOB;
    float4 q[3];
    const int t0 = max(ybeg - warm, 0);
#pragma unroll
    for (int i = 0; i < 3; i++) q[i] = hak_load_stream(reinterpret_cast<const float4*>(L + (long)min(t0 + i, h - 1) * p + xl));
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int t = t0; t < yend; t += 3) {
#pragma unroll
        for (int u = 0; u < 3; u++) {
            const int row = t + u;
            const float4 c = q[u];
            q[u] = hak_load_stream(reinterpret_cast<const float4*>(L + (long)min(row + 3, h - 1) * p + xl));
            acc.x += c.x; acc.y += c.y; acc.z += c.z; acc.w += c.w;                 // (keeps the warm-up loads alive)
            const unsigned off = (row >= ybeg && row < yend) ? col + (unsigned)row * (unsigned)p * 4u : HAK_BUF_OOB;
#pragma unroll
            for (int k = 0; k < NW; k++) hak_buf_store_nt(r, off + (unsigned)k * (unsigned)(plane * 4), k ? acc : c);
        }
    }
}

// ---- the streaming Hessian's access shape with the arithmetic taken out (k_hessian_stream<float, S, false>, kernels_hessian_stream.hip):
// strips of 256 - 2 M columns (M = 4 / 8 / 8 / 12 for S = 1..4), row segments cut by hak_stream_rows, 2 S + 1 warm-up rows above and
// 2 S rows below every segment, one 16-byte load per lane and row with PD rows in flight, and per row the interleaved {Lx, Ly} store
// exactly as the kernel issues it: the lane's four pixels of both derivatives turned through per-wave LDS (one staging row + a ring
// row), read back as pixel pairs, two dense 16-byte nt buffer stores per lane.  The block carries the kernel's LDS footprint
// (rings of R = 2 S + 1 + PD rows per wave + staging + candidate buffer) and its __launch_bounds__ occupancy target, so it runs at
// the waves per SIMD the real kernel reaches (3 for S <= 3, 2 for S = 4).  12 B/px compulsory: what the class could do if its
// ~250-300 vector instructions per row cost nothing.
template <int S>
__global__ __launch_bounds__(256, (S <= 3 ? 3 : 2)) void k_hess_probe(const float* __restrict__ src, float* __restrict__ dxy, long stride, int w, int h,
                                                                     int p, int ry, int nbx, int nby, int nimg)
{
    constexpr int PD = S == 3 ? 1 : 2, R = 2 * S + 1 + PD, M = S == 1 ? 4 : S == 4 ? 12 : 8, XV = 256 - 2 * M;
    __shared__ float4 yring[4 * R * 64];
    __shared__ float4 xstage[4 * 64];
    __shared__ unsigned long long cbuf[4 * 256];
    int bx, by, img;
    if (!hak_xcd_decode(nbx, nby, nimg, bx, by, img)) return;
    const float* L = src + (long)img * stride;
    float* D = dxy + (long)img * stride * 2;
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (threadIdx.x == 0) cbuf[0] = 0;                      // (keeps the candidate buffer allocated)
    const int x0 = bx * XV - M + 4 * lane;
    const int ybeg = (by * 4 + wv) * ry;
    if (ybeg >= h) return;
    const int yend = min(ybeg + ry, h);
    const int xl = min(max(x0, 0), p - 4);
    const int sx = x0 - 4 * lane;
    auto pair_off = [&](int q) -> unsigned {
        const int x = sx + q;
        return q >= M && q < M + XV && x >= 0 && x < w ? (unsigned)x * 8u : HAK_BUF_OOB;
    };
    const unsigned o0 = pair_off(2 * lane), o2 = pair_off(128 + 2 * lane);
    const __amdgpu_buffer_rsrc_t r = hak_buf_rsrc(D);
    float4* Y = yring + wv * R * 64;
    float4* XS = xstage + wv * 64;
    const int t0 = max(0, ybeg - 1 - 2 * S), tend = yend + 2 * S;
    float4 q[PD + 1];
#pragma unroll
    for (int i = 0; i < PD; i++) q[i] = hak_load_stream(reinterpret_cast<const float4*>(L + (unsigned)(min(t0 + i, h - 1) * p + xl)));
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    int slot = 0;
    for (int t = t0; t <= tend; t += PD + 1) {
#pragma unroll
        for (int u = 0; u <= PD; u++) {
            const int row = t + u;
            q[(u + PD) % (PD + 1)] = hak_load_stream(reinterpret_cast<const float4*>(L + (unsigned)(min(row + PD, h - 1) * p + xl)));
            const float4 c = q[u];
            acc.x += c.x; acc.y += c.y; acc.z += c.z; acc.w += c.w;
            const int b = row - S;                              // the derivative row this iteration stores
            __builtin_amdgcn_wave_barrier();
            XS[lane] = c;
            Y[slot * 64 + lane] = acc;
            __builtin_amdgcn_wave_barrier();
            const float2* xs2 = reinterpret_cast<const float2*>(XS);
            const float2* ys2 = reinterpret_cast<const float2*>(Y + slot * 64);
            const float2 xa = xs2[lane], ya = ys2[lane], xb = xs2[64 + lane], yb = ys2[64 + lane];
            const unsigned roff = b >= ybeg && b < yend ? (unsigned)(b * p) * 8u : HAK_BUF_OOB;
            hak_buf_store_nt(r, o0 + roff, make_float4(xa.x, ya.x, xa.y, ya.y));
            hak_buf_store_nt(r, o2 + roff, make_float4(xb.x, yb.x, xb.y, yb.y));
            slot = slot + 1 == R ? 0 : slot + 1;
        }
    }
}

static int probe_time(hipEvent_t a, hipEvent_t b, int iters, double* ms)
{
    float t = 0;
    if (hipEventSynchronize(b) != hipSuccess || hipEventElapsedTime(&t, a, b) != hipSuccess) return 1;
    *ms = (double)t / iters;
    return 0;
}

// Copies `bytes` (a multiple of 16) `iters` times per launch shape; *ms_per_copy = average duration of one copy kernel of the
// BEST shape.  shapes_ms (nullable, HAK_COPY_SHAPES entries): every shape's figure, then read-only and write-only.
// Shapes: loads in flight per lane {2, 4, 8} x stores {nt, plain} x grid {8, 16, 32 blocks per CU, one pass (no loop)}.
int hak_launch_copy_probe(long bytes, int iters, double* ms_per_copy, double* shapes_ms)
{
    float4 *s = nullptr, *d = nullptr;
    if (hipMalloc((void**)&s, (size_t)bytes) != hipSuccess) return 1;
    if (hipMalloc((void**)&d, (size_t)bytes) != hipSuccess) { (void)hipFree(s); return 1; }
    (void)hipMemset(s, 1, (size_t)bytes);
    const long n4 = bytes / 16;
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    double best = 0;
    int rc = 0;
    auto timed = [&](auto launch, double* out) {
        launch();                                           // warm-up (page tables, clocks)
        (void)hipEventRecord(a, nullptr);
        for (int i = 0; i < iters; i++) launch();
        (void)hipEventRecord(b, nullptr);
        double ms = 0;
        rc = probe_time(a, b, iters, &ms) || hipGetLastError() != hipSuccess;
        if (!rc && out) *out = ms;
        return ms;
    };
    for (int shape = 0; shape < HAK_COPY_SHAPES - 2 && !rc; shape++) {
        const int ni = 2 << (shape % 3);                    // 2, 4, 8
        const bool nt = (shape / 3) % 2 == 0;
        const int gsel = shape / 6;                         // 0..3
        const long one_pass = (n4 + 256L * ni - 1) / (256L * ni);
        const unsigned grid = gsel == 3 ? (unsigned)(one_pass < 0x7FFFFFFFL ? one_pass : 0x7FFFFFFFL) : 256u * (8u << gsel);
        auto launch = [&]() {
            if (ni == 2) { if (nt) k_copy_probe<2, true><<<grid, 256>>>(s, d, n4); else k_copy_probe<2, false><<<grid, 256>>>(s, d, n4); }
            else if (ni == 4) { if (nt) k_copy_probe<4, true><<<grid, 256>>>(s, d, n4); else k_copy_probe<4, false><<<grid, 256>>>(s, d, n4); }
            else { if (nt) k_copy_probe<8, true><<<grid, 256>>>(s, d, n4); else k_copy_probe<8, false><<<grid, 256>>>(s, d, n4); }
        };
        const double ms = timed(launch, shapes_ms ? shapes_ms + shape : nullptr);
        if (!rc && (best == 0 || ms < best)) best = ms;
    }
    if (!rc && shapes_ms) {
        timed([&]() { k_read_probe<<<256u * 16u, 256>>>(s, d, n4); }, shapes_ms + HAK_COPY_SHAPES - 2);
        timed([&]() { k_write_probe<<<256u * 16u, 256>>>(d, n4); }, shapes_ms + HAK_COPY_SHAPES - 1);
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

// `nimg` images of w x h (pitch = w rounded up to 64): one read stream and `nwrite` (1..3) write streams with the streaming kernels'
// launch geometry (hak_stream_rows) and `warm` warm-up rows per segment; *ms = average kernel time, bytes = compulsory (no halo)
int hak_launch_stream_probe(int w, int h, int nimg, int nwrite, int warm, int iters, double* ms, double* bytes)
{
    if ((w & 3) || w < 16 || h < 8 || nimg < 1 || nwrite < 1 || nwrite > 3 || iters < 1) return 1;
    const int p = (w + 63) / 64 * 64;
    const long plane = (long)h * p;
    if ((long)nwrite * plane * 4 >= (long)HAK_BUF_OOB) return 1;
    float *s = nullptr, *d = nullptr;
    if (hipMalloc((void**)&s, sizeof(float) * (size_t)plane * nimg) != hipSuccess) return 1;
    if (hipMalloc((void**)&d, sizeof(float) * (size_t)plane * nimg * nwrite) != hipSuccess) { (void)hipFree(s); return 1; }
    (void)hipMemset(s, 0, sizeof(float) * (size_t)plane * nimg);
    const int gx = (w + 239) / 240;
    const int ry = hak_stream_rows(h, (long)gx * nimg, 8);
    const int gy = (h + 4 * ry - 1) / (4 * ry);
    const unsigned grid = hak_xcd_grid(gx, gy, nimg);
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    auto launch = [&]() {
        if (nwrite == 1) k_stream_probe<1><<<grid, 256>>>(s, d, plane, plane, w, h, p, ry, warm, gx, gy, nimg);
        else if (nwrite == 2) k_stream_probe<2><<<grid, 256>>>(s, d, plane, plane, w, h, p, ry, warm, gx, gy, nimg);
        else k_stream_probe<3><<<grid, 256>>>(s, d, plane, plane, w, h, p, ry, warm, gx, gy, nimg);
    };
    launch();
    (void)hipEventRecord(a, nullptr);
    for (int i = 0; i < iters; i++) launch();
    (void)hipEventRecord(b, nullptr);
    const int rc = probe_time(a, b, iters, ms) || hipGetLastError() != hipSuccess;
    *bytes = (1.0 + nwrite) * 4.0 * (double)w * h * nimg;
    (void)hipEventDestroy(a); (void)hipEventDestroy(b);
    (void)hipFree(s); (void)hipFree(d);
    return rc;
}

// the Hessian class's launch of one level: `nimg` planes of w x h, dilation S = step (1..4); *bytes = 12 B/px compulsory
int hak_launch_hess_probe(int w, int h, int nimg, int step, int iters, double* ms, double* bytes)
{
    if ((w & 3) || w < 16 || h < 8 || nimg < 1 || step < 1 || step > 4 || iters < 1) return 1;
    const int p = (w + 63) / 64 * 64;
    const long plane = (long)h * p;
    if (2 * plane * 4 >= (long)HAK_BUF_OOB) return 1;
    float *s = nullptr, *d = nullptr;
    if (hipMalloc((void**)&s, sizeof(float) * (size_t)plane * nimg) != hipSuccess) return 1;
    if (hipMalloc((void**)&d, sizeof(float) * (size_t)plane * nimg * 2) != hipSuccess) { (void)hipFree(s); return 1; }
    (void)hipMemset(s, 0, sizeof(float) * (size_t)plane * nimg);
    const int M = step == 1 ? 4 : step == 4 ? 12 : 8, XV = 256 - 2 * M;
    const int gx = (w + XV - 1) / XV;
    const int ry = hak_stream_rows(h, (long)gx * nimg, 16);
    const int gy = (h + 4 * ry - 1) / (4 * ry);
    const unsigned grid = hak_xcd_grid(gx, gy, nimg);
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    auto launch = [&]() {
        switch (step) {
        case 1: k_hess_probe<1><<<grid, 256>>>(s, d, plane, w, h, p, ry, gx, gy, nimg); break;
        case 2: k_hess_probe<2><<<grid, 256>>>(s, d, plane, w, h, p, ry, gx, gy, nimg); break;
        case 3: k_hess_probe<3><<<grid, 256>>>(s, d, plane, w, h, p, ry, gx, gy, nimg); break;
        default: k_hess_probe<4><<<grid, 256>>>(s, d, plane, w, h, p, ry, gx, gy, nimg); break;
        }
    };
    launch();
    (void)hipEventRecord(a, nullptr);
    for (int i = 0; i < iters; i++) launch();
    (void)hipEventRecord(b, nullptr);
    const int rc = probe_time(a, b, iters, ms) || hipGetLastError() != hipSuccess;
    *bytes = 12.0 * (double)w * h * nimg;
    (void)hipEventDestroy(a); (void)hipEventDestroy(b);
    (void)hipFree(s); (void)hipFree(d);
    return rc;
}
