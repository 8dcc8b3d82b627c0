// strip_map.hip -- does the wave -> (strip, row segment) mapping of the streaming kernels matter for HBM throughput?
// A wave streams down `rows` rows of a 256-px strip (240 owned columns, 16-byte load + NW 16-byte nt stores per lane and row,
// like k_fed_sf).  Mapping A (the kernels' today): the 4 waves of a block take 4 consecutive row segments of ONE strip.
// Mapping B: the 4 waves of a block take 4 ADJACENT strips of the same row segment (one row = 3.84 KB contiguous per block).
// Build: hipcc --offload-arch=gfx950 -O3 -w -o build/strip_map tools/probes/strip_map.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));

template <int NW, int MODE, int XV>
__global__ __launch_bounds__(512) void k_strip(const float* __restrict__ src, float* __restrict__ dst, long plane, int w, int h, int p,
                                               int nstrips, int nseg, int rows, int nimg)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    // blocks: x = (strip group | strip), y = (segment | segment group), z = image, flattened with the image slowest
    int strip, seg, img;
    long b = blockIdx.x;
    if (MODE == 2) {
        seg = b % nseg; img = b / nseg;
        strip = wv;
    } else if (MODE == 1) {
        const int gs = (nstrips + 3) / 4;
        const int bx = b % gs; b /= gs;
        seg = b % nseg; img = b / nseg;
        strip = bx * 4 + wv;
    } else {
        const int gseg = (nseg + 3) / 4;
        strip = b % nstrips; b /= nstrips;
        const int by = b % gseg; img = b / gseg;
        seg = by * 4 + wv;
    }
    if (strip >= nstrips || seg >= nseg || img >= nimg) return;
    constexpr int M = (256 - XV) / 2;
    const int x0 = strip * XV - M + 4 * lane;
    const bool owns = 4 * lane >= M && 4 * lane < M + XV && x0 >= 0 && x0 < w;
    const int xl = min(max(x0, 0), p - 4);
    const float* s = src + (long)img * plane;
    float* d = dst + (long)img * plane * NW;
    // MODE 3: strip-major destination -- the strip's rows are contiguous (XV floats per row)
    const long dbase = MODE == 3 ? (long)strip * h * XV - (long)(strip * XV) : 0;
    const int dp = MODE == 3 ? XV : p;
    const int y0 = seg * rows, y1 = min(y0 + rows, h);
    f4 q[3];
    for (int i = 0; i < 3; i++) q[i] = __builtin_nontemporal_load(reinterpret_cast<const f4*>(s + (long)min(y0 + i, h - 1) * p + xl));
    for (int y = y0; y < y1; y++) {
        const f4 v = q[(y - y0) % 3];
        q[(y - y0) % 3] = *reinterpret_cast<const f4*>(s + (long)min(y + 3, h - 1) * p + xl);
        if (owns) {
#pragma unroll
            for (int k = 0; k < NW; k++) __builtin_nontemporal_store(v + (float)k, reinterpret_cast<f4*>(d + (long)k * plane + dbase + (long)y * dp + x0));
        }
    }
}
template <int NW, int MODE, int XV> double run(const float* s, float* d, int w, int h, int p, int nimg, int rows)
{
    const int nstrips = (w + XV - 1) / XV, nseg = (h + rows - 1) / rows;
    const long plane = (long)h * p;
    const long blocks = MODE == 2 ? (long)nseg * nimg : MODE == 1 ? (long)((nstrips + 3) / 4) * nseg * nimg : (long)nstrips * ((nseg + 3) / 4) * nimg;
    const int nt = MODE == 2 ? 64 * nstrips : 256;
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    k_strip<NW, MODE, XV><<<(unsigned)blocks, nt>>>(s, d, plane, w, h, p, nstrips, nseg, rows, nimg);
    (void)hipEventRecord(a);
    for (int i = 0; i < 5; i++) k_strip<NW, MODE, XV><<<(unsigned)blocks, nt>>>(s, d, plane, w, h, p, nstrips, nseg, rows, nimg);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms = 0; (void)hipEventElapsedTime(&ms, a, b);
    return (double)w * h * 4.0 * (1 + NW) * nimg * 5 / (ms * 1e-3) / 1e9;
}
int main()
{
    const int nimg = 256;
    float *s = nullptr, *d = nullptr;
    const long plane = 1080L * 1920;
    if (hipMalloc((void**)&s, plane * 4 * nimg) != hipSuccess || hipMalloc((void**)&d, plane * 4 * nimg * 2) != hipSuccess) { printf("alloc failed\n"); return 1; }
    (void)hipMemset(s, 0, plane * 4 * nimg);
    struct { int w, h, p; } geo[] = {{1920, 1080, 1920}, {960, 540, 1024}, {480, 270, 512}};
    for (auto g : geo) {
        const int n = g.w == 1920 ? nimg : nimg;
        for (int rows : {270, 68}) {
            if (rows > g.h || g.w > 1920) continue;
            printf("%4d x %4d  rows/wave %3d  XV 240  1R+2W: A %5.0f  B %5.0f  C (row of strips per block) %5.0f  D (strip-major dst) %5.0f    1R+1W: A %5.0f  C %5.0f  D %5.0f   GB/s\n", g.w, g.h, rows,
                   run<2, 0, 240>(s, d, g.w, g.h, g.p, n, rows), run<2, 1, 240>(s, d, g.w, g.h, g.p, n, rows),
                   run<2, 2, 240>(s, d, g.w, g.h, g.p, n, rows), run<2, 3, 240>(s, d, g.w, g.h, g.p, n, rows),
                   run<1, 0, 240>(s, d, g.w, g.h, g.p, n, rows), run<1, 2, 240>(s, d, g.w, g.h, g.p, n, rows), run<1, 3, 240>(s, d, g.w, g.h, g.p, n, rows));
        }
    }
    (void)hipFree(s); (void)hipFree(d);
    return 0;
}
