// kernels_base_stream.hip -- octave-0 prologue, pass A, as a register-streaming kernel (both element types).
//
//   hLowPass(img -> Lt(0,0), var soffset^2)       akaze.cpp:325-327 / 589-596, akazed.cu:2336, 204 (gConv2d<R>)
//   hLowPass(img -> smooth, var 1, ksz 5)         akazed.cu:2336 (gConv2d<2>)
//   hScharrContrast: max |Scharr(smooth)|         akazed.cu:2410, 644 (float) / 3208 (fastakaze)
//
// Same outputs as the LDS tile kernels k_base_a<R> (kernels_base.hip) and kf_base<R> (kernels_fast.hip): Lt(0,0), the
// gradient magnitude of the sigma=1 image as a plane for the histogram pass, and its maximum.  Here a wave owns a 248-px
// strip (float4 / 4 bytes per lane) and streams down its row segment like k_fed_sf: when image row t arrives,
//     row passes of both Gaussians on row t (x neighbours -4..+4 are exactly the two neighbour lanes: 8 DPP shifts),
//     Lt row t-R     = column pass over the base ring (2R+1 rows),
//     smooth row t-2 = column pass over the sigma=1 ring (5 rows), kept in a 3-row ring with its two x-neighbour columns,
//     gradient row t-3 = Scharr magnitude of smooth rows t-4 .. t-2.
// No LDS, no barriers; 4 B/px read, 8 B/px written.
// Round 5 (HIST): the reference's contrast maximum is a maximum over the 16-px lattice (hak_on_lattice), so it can be had BEFORE
// this pass from 1 / 256 of the pixels (k_lattice_hmax below: the sigma=1 image and its Scharr magnitude at the lattice points
// only, same expressions); the pass then bins the gradient magnitude on the fly (LDS histograms, flushed per block) and neither
// writes the gradient plane nor needs the second pass that re-read it: 4 B/px read, 4 B/px written.
// MEASURED, OFF BY DEFAULT (HAK_BASE_HIST=1 switches it on; bit-identical, in the GPU tests' alternative list): half the bytes and
// no gain -- prologue class per 512 x 1080p images 3.69-3.73 ms two-pass vs 3.85-3.88 ms (3.97-4.00 with run-aggregated atomics).
// The pass is bound by vector issue, not by HBM: ~300 vector instructions per row and wave (75 per pixel: two separable Gaussians
// and the Scharr magnitude in the reference's unfused mul / add order) = 2.57 ms of issue at two waves per SIMD against 2.96 ms
// measured and a data-movement floor of 2.4 ms; the binning adds ~10 instructions per pixel to exactly that budget, more than the
// dedicated histogram pass (0.70 ms, 6 TB/s read) costs beside it.
// Reflect-101 is applied to the INPUT, as the tile kernels do
// (rows: the row index is reflected; columns: the lane just outside the image loads the mirrored pixels), and every
// stage then indexes plainly.  w % 4 == 0, R <= 4 (a deeper halo than one lane needs the tile kernel).
#include "fed_common.h"


namespace {

constexpr int BS_HX = 4;                                   // one lane of halo on either side
constexpr int BS_XV = 256 - 2 * BS_HX;                      // 248 output columns per wave
constexpr int BS_PD = 3;                                    // rows in flight ahead of the one being consumed
constexpr int BS_RING = 9;                                  // 2R+1 for R = 4; the loop is unrolled by it

template <typename V> struct BsT;
template <> struct BsT<float> { using In = float; using Q = float4; using V4 = float4; };
template <> struct BsT<int> { using In = unsigned char; using Q = unsigned; using V4 = int4; };

template <typename V> struct BsmTaps { SfTaps<V> a; V b[5]; };

// what one lane loads for one row: its aligned 4 pixels, and -- only in the lane just outside the image of an edge
// strip -- the four mirrored pixels.  Kept raw until the row is consumed so that the prefetch never waits.
template <typename V> struct BsRaw { typename BsT<V>::Q q; V e0, e1, e2, e3; };

template <typename V, bool XE>
__device__ __forceinline__ BsRaw<V> bsm_load(const typename BsT<V>::In* __restrict__ s, int row, int sp, int xl, bool edge, int c0, int c1,
                                             int c2, int c3)
{
    using In = typename BsT<V>::In;
    BsRaw<V> r;
    const In* q = s + (long)row * sp;
    r.q = *reinterpret_cast<const typename BsT<V>::Q*>(q + xl);
    r.e0 = r.e1 = r.e2 = r.e3 = 0;
    if (XE && edge) { r.e0 = (V)q[c0]; r.e1 = (V)q[c1]; r.e2 = (V)q[c2]; r.e3 = (V)q[c3]; }
    return r;
}
template <bool XE>
__device__ __forceinline__ float4 bsm_unpack(const BsRaw<float>& r, bool edge)
{
    float4 v = r.q;
    if (XE) { v.x = edge ? r.e0 : v.x; v.y = edge ? r.e1 : v.y; v.z = edge ? r.e2 : v.z; v.w = edge ? r.e3 : v.w; }
    return v;
}
template <bool XE>
__device__ __forceinline__ int4 bsm_unpack(const BsRaw<int>& r, bool edge)
{
    int4 v = make_int4((int)(r.q & 255u), (int)((r.q >> 8) & 255u), (int)((r.q >> 16) & 255u), (int)(r.q >> 24));
    if (XE) { v.x = edge ? r.e0 : v.x; v.y = edge ? r.e1 : v.y; v.z = edge ? r.e2 : v.z; v.w = edge ? r.e3 : v.w; }
    return v;
}

// base Gaussian: c*b0, then += b[k] * (value at -k + value at +k), k = 1..R (akazed.cu:227-239 / 283-288); the integer
// version ends in >> 16 (akazed.cu:2922-2985)
template <int R>
__device__ __forceinline__ float bsm_conv(const float c, const float (&m)[4], const float (&q)[4], const float (&b)[5])
{
    float ws = c * b[0];
#pragma unroll
    for (int k = 1; k <= R; k++) ws += b[k] * (m[k - 1] + q[k - 1]);
    return ws;
}
template <int R>
__device__ __forceinline__ int bsm_conv(const int c, const int (&m)[4], const int (&q)[4], const int (&b)[5])
{
    unsigned ws = (unsigned)b[0] * (unsigned)c;
#pragma unroll
    for (int k = 1; k <= R; k++) ws += (unsigned)b[k] * (unsigned)(m[k - 1] + q[k - 1]);
    return (int)ws >> 16;
}

// Scharr magnitude of the sigma=1 image (float: akazed.cu:664-666, un-normalised; int: akazed.cu:3208-3232)
__device__ __forceinline__ float bsm_mag(float ul, float uc, float ur, float cl, float cr, float ll, float lc, float lr)
{
    const float dx = 10 * (cr - cl) + 3 * (ur + lr - ul - ll);
    const float dy = 10 * (lc - uc) + 3 * (ll + lr - ul - ur);
    return sqrtf(dx * dx + dy * dy);
}
__device__ __forceinline__ int bsm_mag(int ul, int uc, int ur, int cl, int cr, int ll, int lc, int lr)
{
    const int dx = 10 * (cr - cl) + 3 * (ur + lr - ul - ll);
    const int dy = 10 * (lc - uc) + 3 * (ll + lr - ul - ur);
    return (int)(sqrtf((float)(int)((unsigned)dx * (unsigned)dx + (unsigned)dy * (unsigned)dy)) + 0.5f);
}
__device__ __forceinline__ float bsm_max(float a, float b) { return fmaxf(a, b); }
__device__ __forceinline__ int bsm_max(int a, int b) { return max(a, b); }
// histogram bin of one gradient magnitude (akazed.cu:924-928 / 3319-3326); F = the per-image factor (float path: NBINS / hmax as a
// double -- the exact double product truncated = __fmul_rz + float -> int; FAST path: the 16.16 factor, wrapping product)
template <typename V> struct BsHistF;
template <> struct BsHistF<float> { double f; };
template <> struct BsHistF<int> { int f; };
__device__ __forceinline__ int bsm_bin(float g, const BsHistF<float>& F)
{
    const int hi = (int)((double)g * F.f);
    return hi >= HAK_NBINS ? HAK_NBINS - 1 : hi;
}
__device__ __forceinline__ int bsm_bin(int g, const BsHistF<int>& F)
{
    const int hi = (int)((unsigned)g * (unsigned)F.f) >> 16;
    return hi >= HAK_NBINS ? HAK_NBINS - 1 : (hi < 0 ? 0 : hi);
}
#define BS_HIST_COPIES 8       // LDS histograms per block, selected by lane (LDS atomics on one address serialise)

// Lt and the gradient plane as one raw buffer (lower pointer) + byte offsets: unconditional, countable stores (fed_common.h)
struct BsmOut { __amdgpu_buffer_rsrc_t r; unsigned lt, gr; };    // per-lane byte offsets (column + plane) or HAK_BUF_OOB

template <typename V>
struct BsmState {
    using V4 = typename BsT<V>::V4;
    V4 Hb[BS_RING];                     // base row pass, rows t-8 .. t          slot = iteration mod 9
    V4 H1[BS_RING];                     // sigma=1 row pass (5 rows live)
    V4 Sm[3];                           // smooth rows t-4 .. t-2                slot = iteration mod 3
    V SmL[3], SmR[3];                   // smooth at columns x0-1 and x0+4 of those rows
    BsRaw<V> Lq[BS_PD];                 // prefetch ring
    V tmax;
};

template <typename V, int R, int U, bool XE, bool HIST>
__device__ __forceinline__ void bsm_iter(BsmState<V>& S, const int t, const typename BsT<V>::In* __restrict__ s, V* __restrict__ LT,
                                         V* __restrict__ GR, const int sp, const int p, const int xl, const int x0, const int h,
                                         const int ybeg, const int yend, const bool owns, const bool lat, const int hcov, const bool edge, const int c0, const int c1,
                                         const int c2, const int c3, const BsmTaps<V>& tp, const BsmOut& O, int* mine, const BsHistF<V>& HF)
{
    using V4 = typename BsT<V>::V4;
    // ---- image row t arrives (virtual rows outside the image are their mirror rows); request row t + PD
    const BsRaw<V> raw = S.Lq[pmod(U, BS_PD)];
    S.Lq[pmod(U, BS_PD)] = bsm_load<V, XE>(s, hak_refl(min(t + BS_PD, h + 3), h), sp, xl, edge, c0, c1, c2, c3);
    const V4 c = bsm_unpack<XE>(raw, edge);
    // ---- both row passes: the window is columns x0-4 .. x0+7
    {
        const V win[12] = {wave_shr1(c.x), wave_shr1(c.y), wave_shr1(c.z), wave_shr1(c.w), c.x, c.y, c.z, c.w,
                           wave_shl1(c.x), wave_shl1(c.y), wave_shl1(c.z), wave_shl1(c.w)};
        V hb[4], h1[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const V m[4] = {win[3 + j], win[2 + j], win[1 + j], win[j]};
            const V q[4] = {win[5 + j], win[6 + j], win[7 + j], win[8 + j]};
            hb[j] = bsm_conv<R>(win[4 + j], m, q, tp.b);
            h1[j] = sf_conv(win[4 + j], win[3 + j], win[5 + j], win[2 + j], win[6 + j], tp.a);
        }
        S.Hb[pmod(U, BS_RING)] = mk4(hb[0], hb[1], hb[2], hb[3]);
        S.H1[pmod(U, BS_RING)] = mk4(h1[0], h1[1], h1[2], h1[3]);
    }
    // ---- base column pass -> Lt row t - R
    {
        const int y = t - R;
        const V4 cc = S.Hb[pmod(U - R, BS_RING)];
        V4 um[4], dn[4];
#pragma unroll
        for (int k = 1; k <= 4; k++) {
            um[k - 1] = S.Hb[pmod(U - R - (k <= R ? k : 0), BS_RING)];
            dn[k - 1] = S.Hb[pmod(U - R + (k <= R ? k : 0), BS_RING)];
        }
        V4 o;
        {
            const V m[4] = {um[0].x, um[1].x, um[2].x, um[3].x}, q[4] = {dn[0].x, dn[1].x, dn[2].x, dn[3].x};
            o.x = bsm_conv<R>(cc.x, m, q, tp.b);
        }
        {
            const V m[4] = {um[0].y, um[1].y, um[2].y, um[3].y}, q[4] = {dn[0].y, dn[1].y, dn[2].y, dn[3].y};
            o.y = bsm_conv<R>(cc.y, m, q, tp.b);
        }
        {
            const V m[4] = {um[0].z, um[1].z, um[2].z, um[3].z}, q[4] = {dn[0].z, dn[1].z, dn[2].z, dn[3].z};
            o.z = bsm_conv<R>(cc.z, m, q, tp.b);
        }
        {
            const V m[4] = {um[0].w, um[1].w, um[2].w, um[3].w}, q[4] = {dn[0].w, dn[1].w, dn[2].w, dn[3].w};
            o.w = bsm_conv<R>(cc.w, m, q, tp.b);
        }
        hak_buf_store_nt(O.r, O.lt + (y >= ybeg && y < yend ? (unsigned)(y * p) * (unsigned)sizeof(V) : HAK_BUF_OOB), o);
    }
    // ---- sigma=1 column pass -> smooth row t - 2 (never written)
    {
        const V4 cc = S.H1[pmod(U - 2, BS_RING)], u1 = S.H1[pmod(U - 3, BS_RING)], d1 = S.H1[pmod(U - 1, BS_RING)];
        const V4 u2 = S.H1[pmod(U - 4, BS_RING)], d2 = S.H1[pmod(U, BS_RING)];
        V4 sm;
        sm.x = sf_conv(cc.x, u1.x, d1.x, u2.x, d2.x, tp.a);
        sm.y = sf_conv(cc.y, u1.y, d1.y, u2.y, d2.y, tp.a);
        sm.z = sf_conv(cc.z, u1.z, d1.z, u2.z, d2.z, tp.a);
        sm.w = sf_conv(cc.w, u1.w, d1.w, u2.w, d2.w, tp.a);
        S.Sm[pmod(U, 3)] = sm;
        S.SmL[pmod(U, 3)] = wave_shr1(sm.w);
        S.SmR[pmod(U, 3)] = wave_shl1(sm.x);
    }
    // ---- gradient magnitude row t - 3
    {
        const int b = t - 3;
        const V4 su = S.Sm[pmod(U - 2, 3)], sc = S.Sm[pmod(U - 1, 3)], sd = S.Sm[pmod(U, 3)];
        const V uL = S.SmL[pmod(U - 2, 3)], uR = S.SmR[pmod(U - 2, 3)];
        const V cL = S.SmL[pmod(U - 1, 3)], cR = S.SmR[pmod(U - 1, 3)];
        const V dL = S.SmL[pmod(U, 3)], dR = S.SmR[pmod(U, 3)];
        V4 g;
        g.x = bsm_mag(uL, su.x, su.y, cL, sc.y, dL, sd.x, sd.y);
        g.y = bsm_mag(su.x, su.y, su.z, sc.x, sc.z, sd.x, sd.y, sd.z);
        g.z = bsm_mag(su.y, su.z, su.w, sc.y, sc.w, sd.y, sd.z, sd.w);
        g.w = bsm_mag(su.z, su.w, uR, sc.z, cR, sd.z, sd.w, dR);
        const bool inr = b >= ybeg && b < yend;
        if constexpr (HIST) {
            // the maximum is known (k_lattice_hmax ran first): bin the row's magnitudes instead of storing them
            if (inr && owns) {
                // (counting runs of equal bins in registers -- a lane's four pixels mostly share one -- and issuing one atomic per run
                // was measured slower still: the pass is short of vector issue slots, not of LDS atomic throughput)
                atomicAdd(&mine[bsm_bin(g.x, HF)], 1); atomicAdd(&mine[bsm_bin(g.y, HF)], 1);
                atomicAdd(&mine[bsm_bin(g.z, HF)], 1); atomicAdd(&mine[bsm_bin(g.w, HF)], 1);
            }
        } else {
            hak_buf_store_nt(O.r, O.gr + (inr ? (unsigned)(b * p) * (unsigned)sizeof(V) : HAK_BUF_OOB), g);
            // the reference's maximum runs over the 16-px lattice only (hak_on_lattice): x0 % 4 == 0, so only g.x can be on it
            if (inr && lat && (b & 15) == 0 && b < hcov) S.tmax = bsm_max(S.tmax, g.x);
        }
    }
}

template <typename V, int R, bool XE, bool HIST>
__device__ __forceinline__ V bsm_strip(const typename BsT<V>::In* __restrict__ s, V* __restrict__ LT, V* __restrict__ GR, int sp, int p,
                                       int w, int h, int x0, int ybeg, int yend, bool owns, const BsmTaps<V>& tp, int* mine,
                                       const BsHistF<V>& HF)
{
    using V4 = typename BsT<V>::V4;
    constexpr int RR = R < 3 ? 3 : R;                       // rows of context above and below the segment
    // aligned 4-pixel load of every lane, kept inside the row; the lane just outside the image loads mirrored pixels instead
    const int xl = min(max(x0, 0), w - 4);
    const bool edge = XE && (x0 == -4 || x0 == w);
    const int c0 = hak_refl(x0, w), c1 = hak_refl(x0 + 1, w), c2 = hak_refl(x0 + 2, w), c3 = hak_refl(x0 + 3, w);
    BsmOut O;
    {
        V* lo = HIST || LT < GR ? LT : GR;
        O.r = hak_buf_rsrc(lo);
        const unsigned xb = (unsigned)x0 * (unsigned)sizeof(V);
        O.lt = owns ? xb + (unsigned)((LT - lo) * (long)sizeof(V)) : HAK_BUF_OOB;
        O.gr = owns && !HIST ? xb + (unsigned)((GR - lo) * (long)sizeof(V)) : HAK_BUF_OOB;
    }
    const int t0 = ybeg - RR;
    const int tend = yend - 1 + RR;
    // this lane's first pixel is a lattice column the reference's maximum kernel covers (rows: tested per row, b < hcov below)
    const bool lat = owns && (x0 & 15) == 0 && x0 < hak_lattice_cov(w);
    const int hcov = hak_lattice_cov(h);
    BsmState<V> S;
    const V z = 0;
    const V4 z4 = mk4(z, z, z, z);
#pragma unroll
    for (int i = 0; i < BS_RING; i++) { S.Hb[i] = z4; S.H1[i] = z4; }
#pragma unroll
    for (int i = 0; i < 3; i++) { S.Sm[i] = z4; S.SmL[i] = z; S.SmR[i] = z; }
    S.tmax = z;
#pragma unroll
    for (int i = 0; i < BS_PD; i++) S.Lq[i] = bsm_load<V, XE>(s, hak_refl(min(t0 + i, h + 3), h), sp, xl, edge, c0, c1, c2, c3);
    for (int tb = t0; tb <= tend; tb += BS_RING) {
        bsm_iter<V, R, 0, XE, HIST>(S, tb + 0, s, LT, GR, sp, p, xl, x0, h, ybeg, yend, owns, lat, hcov, edge, c0, c1, c2, c3, tp, O, mine, HF);
        bsm_iter<V, R, 1, XE, HIST>(S, tb + 1, s, LT, GR, sp, p, xl, x0, h, ybeg, yend, owns, lat, hcov, edge, c0, c1, c2, c3, tp, O, mine, HF);
        bsm_iter<V, R, 2, XE, HIST>(S, tb + 2, s, LT, GR, sp, p, xl, x0, h, ybeg, yend, owns, lat, hcov, edge, c0, c1, c2, c3, tp, O, mine, HF);
        bsm_iter<V, R, 3, XE, HIST>(S, tb + 3, s, LT, GR, sp, p, xl, x0, h, ybeg, yend, owns, lat, hcov, edge, c0, c1, c2, c3, tp, O, mine, HF);
        bsm_iter<V, R, 4, XE, HIST>(S, tb + 4, s, LT, GR, sp, p, xl, x0, h, ybeg, yend, owns, lat, hcov, edge, c0, c1, c2, c3, tp, O, mine, HF);
        bsm_iter<V, R, 5, XE, HIST>(S, tb + 5, s, LT, GR, sp, p, xl, x0, h, ybeg, yend, owns, lat, hcov, edge, c0, c1, c2, c3, tp, O, mine, HF);
        bsm_iter<V, R, 6, XE, HIST>(S, tb + 6, s, LT, GR, sp, p, xl, x0, h, ybeg, yend, owns, lat, hcov, edge, c0, c1, c2, c3, tp, O, mine, HF);
        bsm_iter<V, R, 7, XE, HIST>(S, tb + 7, s, LT, GR, sp, p, xl, x0, h, ybeg, yend, owns, lat, hcov, edge, c0, c1, c2, c3, tp, O, mine, HF);
        bsm_iter<V, R, 8, XE, HIST>(S, tb + 8, s, LT, GR, sp, p, xl, x0, h, ybeg, yend, owns, lat, hcov, edge, c0, c1, c2, c3, tp, O, mine, HF);
    }
    return S.tmax;
}

// ---- the contrast maximum in front of the pass (HIST mode): the sigma=1 low-pass (gConv2d<2>, akazed.cu:204-296 / 2922-2985) and its
// Scharr magnitude (akazed.cu:644-666 / 3208-3232) at the lattice points x % 16 == 0 && y % 16 == 0 that gFindMaxContrastU4's grid
// covers (hak_on_lattice) -- one thread per lattice point: 7 x 7 input pixels (reflect-101 on the input index, as everywhere),
// 7 rows x 3 columns of the row pass, 3 x 3 of the column pass, one magnitude, one atomicMax.  Same expressions, same order as the
// streaming pass and the tile kernels (sf_conv, bsm_mag): the value is bit-identical to the plane's.
template <typename V>
__global__ __launch_bounds__(256) void k_lattice_hmax(const typename BsT<V>::In* __restrict__ img, long img_stride, int sp, int w, int h,
                                                      SfTaps<V> tp, HakImgState* state, int nlx, int nly)
{
    const int im = blockIdx.y;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    V m = 0;
    if (idx < nlx * nly) {
        const int ly = idx / nlx, lx = idx - ly * nlx;
        const int x = 16 * lx, y = 16 * ly;
        const typename BsT<V>::In* s = img + (long)im * img_stride;
        // smooth is needed at the reflected neighbour coordinates the Scharr stencil reads (akazed.cu:655-662: abs / borderAdd)
        const int xs[3] = {hak_refl(x - 1, w), x, hak_refl(x + 1, w)};
        const int ys[3] = {hak_refl(y - 1, h), y, hak_refl(y + 1, h)};
        V sm[3][3];
#pragma unroll
        for (int j = 0; j < 3; j++) {
            V rp[5][3];                                             // row pass at rows ys[j] - 2 .. ys[j] + 2 (reflected), columns xs[0..2]
#pragma unroll
            for (int d = 0; d < 5; d++) {
                const typename BsT<V>::In* q = s + (long)hak_refl(ys[j] + d - 2, h) * sp;
#pragma unroll
                for (int i = 0; i < 3; i++) {
                    const int xc = xs[i];
                    rp[d][i] = sf_conv((V)q[xc], (V)q[hak_refl(xc - 1, w)], (V)q[hak_refl(xc + 1, w)], (V)q[hak_refl(xc - 2, w)],
                                       (V)q[hak_refl(xc + 2, w)], tp);
                }
            }
#pragma unroll
            for (int i = 0; i < 3; i++) sm[j][i] = sf_conv(rp[2][i], rp[1][i], rp[3][i], rp[0][i], rp[4][i], tp);
        }
        m = bsm_mag(sm[0][0], sm[0][1], sm[0][2], sm[1][0], sm[1][2], sm[2][0], sm[2][1], sm[2][2]);
    }
    for (int off = 32; off > 0; off >>= 1) m = bsm_max(m, __shfl_xor(m, off));
    if ((threadIdx.x & 63) == 0) {
        if constexpr (std::is_same<V, float>::value) {
            if (m > 0.f) atomicMax(&state[im].hmax_bits, __float_as_uint(m));        // (floored at 0.03f by the reset, akazed.cu:2413)
        } else {
            if (m > 1) atomicMax(&state[im].ihmax, m);                               // (floored at 1, akazed.cu:4101)
        }
    }
}

// grid: hak_xcd_grid(strips, groups of four row segments, images); a block's four waves take four consecutive segments
template <typename V, int R, bool HIST>
__global__ __launch_bounds__(256) void k_base_stream(const typename BsT<V>::In* __restrict__ img, long img_stride, int sp,
                                                     V* __restrict__ lt, V* __restrict__ grad, long stride, int w, int h, int p,
                                                     BsmTaps<V> tp, HakImgState* state, int ry, int nbx, int nby, int nimg)
{
    __shared__ int shist[HIST ? BS_HIST_COPIES * HAK_NBINS : 1];
    int bx, by, im;
    if (!hak_xcd_decode(nbx, nby, nimg, bx, by, im)) return;
    const typename BsT<V>::In* s = img + (long)im * img_stride;
    V* LT = lt + (long)im * stride;
    V* GR = HIST ? LT : grad + (long)im * stride;
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int x0 = bx * BS_XV - BS_HX + 4 * lane;           // first pixel of this lane (may lie outside the image)
    const int ybeg = (by * 4 + wv) * ry;
    BsHistF<V> HF{};
    if constexpr (HIST) {
        for (int i = threadIdx.x; i < BS_HIST_COPIES * HAK_NBINS; i += 256) shist[i] = 0;
        if constexpr (std::is_same<V, float>::value) HF.f = (double)(HAK_NBINS / __uint_as_float(state[im].hmax_bits));   // akazed.cu:2450
        else HF.f = (int)(HAK_NBINS / (float)state[im].ihmax * 65536 + 0.5f);                                          // akazed.cu:4133
        __syncthreads();
    }
    int* mine = shist + (HIST ? (lane & (BS_HIST_COPIES - 1)) * HAK_NBINS : 0);
    V m = 0;
    if (ybeg < h) {                                         // wave-uniform
        const int yend = min(ybeg + ry, h);
        const bool owns = lane >= 1 && lane < 63 && x0 < w;
        if (bx == 0 || (bx + 1) * BS_XV + BS_HX >= w) m = bsm_strip<V, R, true, HIST>(s, LT, GR, sp, p, w, h, x0, ybeg, yend, owns, tp, mine, HF);
        else m = bsm_strip<V, R, false, HIST>(s, LT, GR, sp, p, w, h, x0, ybeg, yend, owns, tp, mine, HF);
    }
    if constexpr (HIST) {
        __syncthreads();
        for (int i = threadIdx.x; i < HAK_NBINS; i += 256) {
            int sum = 0;
#pragma unroll
            for (int c = 0; c < BS_HIST_COPIES; c++) sum += shist[c * HAK_NBINS + i];
            if (sum) atomicAdd(&state[im].hist[i], sum);
        }
    } else {
        if (ybeg >= h) return;
        for (int off = 32; off > 0; off >>= 1) m = bsm_max(m, __shfl_xor(m, off));
        if (lane == 0) {
            if constexpr (std::is_same<V, float>::value) {
                if (m > 0.f) atomicMax(&state[im].hmax_bits, __float_as_uint(m));        // lattice maximum (akazed.cu:827-877)
            } else {
                if (m > 1) atomicMax(&state[im].ihmax, m);
            }
        }
    }
}

// grad != nullptr: pass A of the two-pass form (writes the gradient plane, finds the lattice maximum; the caller's histogram pass
// follows); grad == nullptr: k_lattice_hmax + the pass with the histogram inside (the caller only finishes the contrast factor)
template <typename V>
bool launch_base_stream(hipStream_t st, const typename BsT<V>::In* img, long img_stride, int sp, V* lt, V* grad, long stride, int w, int h,
                        int p, int nimg, const V* taps1, const V* taps_base, int R, HakImgState* state, int mode)
{
    using In = typename BsT<V>::In;
    constexpr long LA = 4;                                  // elements per aligned lane load: 4 floats / 4 bytes
    if (R < 2 || R > 4 || (w & 3) || w < 16 || h < 16) return false;
    if ((sp % LA) || (img_stride % LA) || (reinterpret_cast<uintptr_t>(img) % (LA * sizeof(In)))) return false;
    if (!hak_stream_pays(mode, w, h, nimg)) return false;
    if (grad && (lt < grad ? grad - lt : lt - grad) + (long)h * p >= (long)HAK_BUF_OOB / (long)sizeof(V)) return false;   // plane offset + plane size < marker
    if ((long)h * p >= (long)HAK_BUF_OOB / (long)sizeof(V)) return false;
    BsmTaps<V> tp;
    tp.a = SfTaps<V>{taps1[0], taps1[1], taps1[2]};
    for (int i = 0; i < 5; i++) tp.b[i] = i <= R ? taps_base[i] : V(0);
    const int gx = (w + BS_XV - 1) / BS_XV;
    // rows per wave: ~128-row segments of equal height amortise the 2R warm-up rows; shrink while the grid cannot fill the chip
    const int ry = hak_stream_rows(h, (long)gx * nimg, 16);
    const int gy = ((h + ry - 1) / ry + 3) / 4;
    const unsigned grid = hak_xcd_grid(gx, gy, nimg);
    if (!grad) {
        const int nlx = (hak_lattice_cov(w) + 15) / 16, nly = (hak_lattice_cov(h) + 15) / 16;
        k_lattice_hmax<V><<<dim3((nlx * nly + 255) / 256, nimg), 256, 0, st>>>(img, img_stride, sp, w, h, tp.a, state, nlx, nly);
        switch (R) {
        case 2: k_base_stream<V, 2, true><<<grid, 256, 0, st>>>(img, img_stride, sp, lt, lt, stride, w, h, p, tp, state, ry, gx, gy, nimg); break;
        case 3: k_base_stream<V, 3, true><<<grid, 256, 0, st>>>(img, img_stride, sp, lt, lt, stride, w, h, p, tp, state, ry, gx, gy, nimg); break;
        default: k_base_stream<V, 4, true><<<grid, 256, 0, st>>>(img, img_stride, sp, lt, lt, stride, w, h, p, tp, state, ry, gx, gy, nimg); break;
        }
        return true;
    }
    switch (R) {
    case 2: k_base_stream<V, 2, false><<<grid, 256, 0, st>>>(img, img_stride, sp, lt, grad, stride, w, h, p, tp, state, ry, gx, gy, nimg); break;
    case 3: k_base_stream<V, 3, false><<<grid, 256, 0, st>>>(img, img_stride, sp, lt, grad, stride, w, h, p, tp, state, ry, gx, gy, nimg); break;
    default: k_base_stream<V, 4, false><<<grid, 256, 0, st>>>(img, img_stride, sp, lt, grad, stride, w, h, p, tp, state, ry, gx, gy, nimg); break;
    }
    return true;
}

} // namespace

bool hak_launch_base_stream(hipStream_t st, const float* img, long img_stride, int sp, float* lt, float* grad, long stride, int w, int h,
                            int p, int nimg, const float* taps1, const float* taps_base, int R, HakImgState* state, int mode)
{
    return launch_base_stream<float>(st, img, img_stride, sp, lt, grad, stride, w, h, p, nimg, taps1, taps_base, R, state, mode);
}
bool hakf_launch_base_stream(hipStream_t st, const unsigned char* img, long img_stride, int sp, int* lt, int* grad, long stride, int w,
                             int h, int p, int nimg, const int* itaps1, const int* itaps_base, int R, HakImgState* state, int mode)
{
    return launch_base_stream<int>(st, img, img_stride, sp, lt, grad, stride, w, h, p, nimg, itaps1, itaps_base, R, state, mode);
}
