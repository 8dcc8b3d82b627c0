// kernels_level.hip -- one whole scale-space sublevel per launch, for launches too small to fill the chip.
//
//   akaze.cpp:369-421 for one (octave, sublevel):
//     j == 0: hDownWithSmooth(L(o-1,0) -> L, smooth)   akazed.cu:2389, 449      (HEAD)
//     j  > 0: hLowPass(L(o,j-1) -> smooth, var 1)      akazed.cu:2336, 204
//     hFlow(smooth -> g)                               akazed.cu:2487, 1068
//     naux x hNldStep(L, g -> L')                      akazed.cu:2509, 1241
//
// Why: a single-image call (the reference demo's call pattern, main.cpp:199-209) is bound by the NUMBER of launches, not by
// bytes -- the graph replay costs the host ~6.6 us per kernel node and a dependent kernel ~8 us on the GPU whatever its size
// (profiles/r03_single_*): 94 launches per 1080p image, 72 of them the tile-path's low-pass / conductivity / <= 4-step FED
// kernels of sixteen sublevels.  Here a 1024-thread block owns a TxT tile and keeps it in LDS with a halo as deep as the
// number of explicit steps (<= 36): raw L, sigma=1 low-pass, conductivity and ALL steps of the FED cycle run out of LDS, the
// valid window shrinking by one pixel per step; a thread works on four pixels of a row at a time (16-byte LDS accesses).  One launch per sublevel: 94 -> 43 launches per image.
//
// Bit-exactness: every expression is the reference's (and the oracle's) own -- no regrouping: the separable Gaussian through
// sf_conv, the Scharr / conductivity of k_smooth_flow, and  step = tE + tW + tS + tN  then  fma(stepfac, step, L)  with the
// neighbours taken at reflect-101 INDICES (abs / borderAdd, akazed.cu:1251-1254), never from mirrored halo cells: a mirrored
// cell would add its N and S terms in the other order.  The raw tile alone is loaded through mirrored indices (on the
// source extents for an octave head, akazed.cu:466-494), which is what the Gaussian's taps read.
#include <atomic>
#include "fed_common.h"

#define LV_MAX_STEPS 36
#define LV_NT 1024                    // threads per block: four waves per SIMD hide the LDS latency of the per-pixel loops
#define LV_LDS_FLOATS 38400          // 150 KB of the CU's 160 KB: three planes of (T + 2 HX) x (T + 2 (ns + 3)) elements

template <typename V> struct LvFacs { V f[LV_MAX_STEPS]; };

// The level's Hessian inside the same launch (hHessianDeterminant akazed.cu:2531 + gCalcExtremaMap 1334): the low-pass the
// derivatives are taken of sits in LDS anyway.  maps == nullptr: not fused (the caller launches the Hessian kernel).
template <typename V> struct LvHess {
    V* dxy;                         // the level's interleaved {Lx, Ly} plane of image 0
    V fac1, fac2;
    int S;                          // dilation (sigma_size)
    unsigned long long* maps; long map_stride;
    unsigned long long* cand; long cand_cap;
    HakImgState* state;
    int p0, octave, layer, psz;
    float border;
    V threshold;
};

namespace {

// source index of decimated coordinate d (tile coordinates may lie outside the image): 2d mirrored on the SOURCE extent
__device__ __forceinline__ int lv_src_index(int d, int sn)
{
    int i = 2 * d;
    i = i < 0 ? -i : i;
    i = i < sn ? i : sn + sn - 2 - i;                        // borderAdd on the source (akazed.cu:466, 490)
    i = i < 0 ? 0 : i;
    return i < sn ? i : sn - 1;                              // (beyond one reflection: never used)
}

template <typename V>
__device__ __forceinline__ float lv_conductivity(V dx, V dy, float ikc, int type)
{
    const float dif2 = sf_dif2(dx, dy, ikc);
    if (type == HAK_PM_G2) return 1.f / (1.f + dif2);
    if (type == HAK_PM_G1) return hak_expf(-dif2);
    if (type == HAK_WEICKERT) {
        const float d2 = dif2 * dif2;
        return 1.f - hak_expf(-3.315f / (d2 * d2));
    }
    return 1.f / sqrtf(1.f + dif2);
}

// ---- four pixels per thread: a thread owns one 16-byte group of a plane row per item, so a stencil row costs one ds_read_b128
// plus the two edge words instead of six ds_read_b32, and the index arithmetic is paid once per four pixels (the scalar version
// spent ~60 VALU instructions per pixel-step, most of them addressing)
template <typename V> struct Lv6 { V v[6]; };                // columns c-1 .. c+4 of one plane row
template <typename V>
__device__ __forceinline__ Lv6<V> lv_row6(const V* row, int c)
{
    using V4 = typename FedV<V>::V4;
    const V4 q = *reinterpret_cast<const V4*>(row + c);
    Lv6<V> o;
    o.v[0] = row[c - 1]; o.v[1] = q.x; o.v[2] = q.y; o.v[3] = q.z; o.v[4] = q.w; o.v[5] = row[c + 4];
    return o;
}
// reflect-101 in x on the image extents (abs(x - 1), borderAdd(x, 1, w)): pixel j of the group sits at image column x0 + j
template <typename V> __device__ __forceinline__ V lv_left(const Lv6<V>& r, int j, int x0) { return x0 + j == 0 ? r.v[j + 2] : r.v[j]; }
template <typename V> __device__ __forceinline__ V lv_right(const Lv6<V>& r, int j, int x0, int w) { return x0 + j == w - 1 ? r.v[j] : r.v[j + 2]; }

// HEAD: src = L(o-1, 0) of the previous octave (extents sw x sh, pitch sp); else src = L(o, s-1) of this octave.
// FIRST = false continues a cycle of more than LV_MAX_STEPS steps: L comes from src, the low-pass from `smooth`.
// Planes: EH = T + 2 (ns + 3) rows of EW = T + 2 HX elements, HX = ns + 5 rounded up to 4: image column X0 + c of plane column c
// is a multiple of 4 for every group, so interior groups load and store 16 bytes at once.
template <typename V, bool HEAD, bool FIRST>
__global__ __launch_bounds__(LV_NT) void k_level_tile(const V* __restrict__ src, V* __restrict__ smooth, V* __restrict__ dst,
                                                      long stride, int sw, int sh, int sp, int w, int h, int p,
                                                      SfTaps<V> t, int type, const HakImgState* __restrict__ state, int octave,
                                                      float fixed_ikc, LvFacs<V> fac, int ns, int T, int nbx, int nby, int nimg, LvHess<V> hs)
{
    using V4 = typename FedV<V>::V4;
    extern __shared__ __align__(16) unsigned char lv_lds_raw[];
    int bx, by, img;
    if (!hak_xcd_decode(nbx, nby, nimg, bx, by, img)) return;
    const int H = ns;                                        // FED halo
    // rows: raw region core +- (H + 3).  columns: the phases work on whole groups and skip the first and last one, so the smooth
    // plane is valid from plane column 4 and g from column 5: the FED halo needs HX - H >= 5, rounded up to whole groups
    const int HY = H + 3, HX = (H + 5 + 3) & ~3;
    const int EW = T + 2 * HX, EH = T + 2 * HY, NG = EW >> 2;
    V* A = reinterpret_cast<V*>(lv_lds_raw);                 // raw = L0, then L ping
    V* B = A + EW * EH;                                      // row pass, then g
    V* C = B + EW * EH;                                      // smooth, then L pong
    // Hessian scratch behind the three planes: Lx, Ly on core +- (S + 1), the determinant on core +- 1
    const bool hess = FIRST && hs.maps != nullptr;           // (uniform)
    const int XW = T + 2 * (hs.S + 1), DW = T + 2;
    V* HX_ = C + EW * EH;
    V* HY_ = HX_ + XW * XW;
    V* DT = HY_ + XW * XW;
    const V* s = src + (long)img * stride;
    V* osm = smooth + (long)img * stride;
    V* od = dst + (long)img * stride;
    const float ikc = state ? state[img].ikc[octave] : fixed_ikc;
    const int tid = threadIdx.x;
    const int X0 = bx * T - HX, Y0 = by * T - HY;            // image coordinates of plane element (0, 0)
    // idx / NG by one multiply-high: M = ceil(2^32 / NG) is exact for idx < 2^14 and NG <= 64 (idx * (M * NG - 2^32) < 2^32 / NG)
    const unsigned MG = 0xFFFFFFFFu / (unsigned)NG + 1u;
    const int nitems = EH * NG;

    // ---- raw plane (A): L0 with mirrored indices (on the source extents for an octave head)
    for (int idx = tid; idx < nitems; idx += LV_NT) {
        const int r = (int)__umulhi((unsigned)idx, MG), c = (idx - r * NG) << 2;
        const int x = X0 + c, y = Y0 + r;
        V4 v;
        if (HEAD && FIRST) {
            const V* row = s + (long)lv_src_index(y, sh) * sp;
            v = mk4(row[lv_src_index(x, sw)], row[lv_src_index(x + 1, sw)], row[lv_src_index(x + 2, sw)], row[lv_src_index(x + 3, sw)]);
        } else {
            const V* row = s + (long)hak_refl(y, h) * p;
            if (x >= 0 && x + 3 < w) v = *reinterpret_cast<const V4*>(row + x);
            else v = mk4(row[hak_refl(x, w)], row[hak_refl(x + 1, w)], row[hak_refl(x + 2, w)], row[hak_refl(x + 3, w)]);
        }
        *reinterpret_cast<V4*>(A + r * EW + c) = v;
    }
    if (FIRST) {
        hak_lds_barrier();
        // ---- row pass (akazed.cu:227-239 / 469-471) into B: all rows, groups 1 .. NG-2
        for (int idx = tid; idx < nitems; idx += LV_NT) {
            const int r = (int)__umulhi((unsigned)idx, MG), g = idx - r * NG, c = g << 2;
            if (g >= 1 && g < NG - 1) {
                const V* q = A + r * EW + c;
                const V4 l = *reinterpret_cast<const V4*>(q - 4), m = *reinterpret_cast<const V4*>(q), rr = *reinterpret_cast<const V4*>(q + 4);
                *reinterpret_cast<V4*>(B + r * EW + c) = mk4(sf_conv(m.x, l.w, m.y, l.z, m.z, t), sf_conv(m.y, m.x, m.z, l.w, m.w, t),
                                                            sf_conv(m.z, m.y, m.w, m.x, rr.x, t), sf_conv(m.w, m.z, rr.x, m.y, rr.y, t));
            }
        }
        hak_lds_barrier();
        // ---- column pass (akazed.cu:283-288 / 507-509) into C = smooth on core +- (H + 1); the core goes to HBM
        for (int idx = tid; idx < nitems; idx += LV_NT) {
            const int r = (int)__umulhi((unsigned)idx, MG), g = idx - r * NG, c = g << 2;
            if (r >= 2 && r < EH - 2 && g >= 1 && g < NG - 1) {
                const V* q = B + r * EW + c;
                const V4 m = *reinterpret_cast<const V4*>(q), u1 = *reinterpret_cast<const V4*>(q - EW), d1 = *reinterpret_cast<const V4*>(q + EW);
                const V4 u2 = *reinterpret_cast<const V4*>(q - 2 * EW), d2 = *reinterpret_cast<const V4*>(q + 2 * EW);
                const V4 ws = mk4(sf_conv(m.x, u1.x, d1.x, u2.x, d2.x, t), sf_conv(m.y, u1.y, d1.y, u2.y, d2.y, t),
                                  sf_conv(m.z, u1.z, d1.z, u2.z, d2.z, t), sf_conv(m.w, u1.w, d1.w, u2.w, d2.w, t));
                *reinterpret_cast<V4*>(C + r * EW + c) = ws;
                const int x = X0 + c, y = Y0 + r;
                if (r >= HY && r < HY + T && c >= HX && c < HX + T && y < h) {
                    V* o = osm + (long)y * p + x;
                    if (x + 3 < w) *reinterpret_cast<V4*>(o) = ws;
                    else {
                        if (x < w) o[0] = ws.x;
                        if (x + 1 < w) o[1] = ws.y;
                        if (x + 2 < w) o[2] = ws.z;
                    }
                }
            }
        }
    } else {
        // continuation: the low-pass of this sublevel was written by the first launch of the cycle
        for (int idx = tid; idx < nitems; idx += LV_NT) {
            const int r = (int)__umulhi((unsigned)idx, MG), c = (idx - r * NG) << 2;
            const int x = X0 + c;
            const V* row = osm + (long)hak_refl(Y0 + r, h) * p;
            *reinterpret_cast<V4*>(C + r * EW + c) = mk4(row[hak_refl(x, w)], row[hak_refl(x + 1, w)], row[hak_refl(x + 2, w)], row[hak_refl(x + 3, w)]);
        }
    }
    hak_lds_barrier();
    if (hess) {
        // ---- Lx, Ly (gDerivate akazed.cu:1267-1296) of the low-pass on core +- (S + 1), taps at reflect-101 indices; the core goes
        // to the interleaved derivative plane: element (y, x) = {Lx, Ly} at 2 * (y * p + x)
        const int S = hs.S;
        V* oxy = hs.dxy + (long)img * stride;
        const unsigned MX = 0xFFFFFFFFu / (unsigned)XW + 1u;
        for (int idx = tid; idx < XW * XW; idx += LV_NT) {
            const int r = (int)__umulhi((unsigned)idx, MX), c = idx - r * XW;
            const int x = bx * T - S - 1 + c, y = by * T - S - 1 + r;
            if (x < 0 || x >= w || y < 0 || y >= h) continue;
            const int c0 = hak_refl(x - S, w) - X0, c1 = x - X0, c2 = hak_refl(x + S, w) - X0;
            const int r0 = (hak_refl(y - S, h) - Y0) * EW, r1 = (y - Y0) * EW, r2 = (hak_refl(y + S, h) - Y0) * EW;
            const V ul = C[r0 + c0], uc = C[r0 + c1], ur = C[r0 + c2];
            const V cl = C[r1 + c0], cr = C[r1 + c2];
            const V ll = C[r2 + c0], lc = C[r2 + c1], lr = C[r2 + c2];
            const V vx = hs_d(hs.fac1, hs.fac2, ur + lr - ul - ll, cr - cl);          // akazed.cu:1294
            const V vy = hs_d(hs.fac1, hs.fac2, lr + ll - ur - ul, lc - uc);          // akazed.cu:1295
            HX_[idx] = vx;
            HY_[idx] = vy;
            if (r > S && r <= S + T && c > S && c <= S + T) {
                V* o = oxy + 2 * ((long)y * p + x);
                o[0] = vx;
                o[1] = vy;
            }
        }
    }
    // ---- conductivity (akazed.cu:1078-1106) into B on core +- H (whole groups), neighbours at reflect-101 indices.
    // (pixels outside the image or outside core +- H get values nobody reads)
    for (int idx = tid; idx < nitems; idx += LV_NT) {
        const int r = (int)__umulhi((unsigned)idx, MG), g = idx - r * NG, c = g << 2;
        const int x = X0 + c, y = Y0 + r;
        if (r >= 3 && r < EH - 3 && g >= 1 && g < NG - 1 && y >= 0 && y < h) {
            const Lv6<V> U = lv_row6<V>(C + (r + (y == 0 ? 1 : -1)) * EW, c);
            const Lv6<V> M = lv_row6<V>(C + r * EW, c);
            const Lv6<V> D = lv_row6<V>(C + (r + (y == h - 1 ? -1 : 1)) * EW, c);
            V gq[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const V ul = lv_left(U, j, x), uc = U.v[j + 1], ur = lv_right(U, j, x, w);
                const V cl = lv_left(M, j, x), cr = lv_right(M, j, x, w);
                const V ll = lv_left(D, j, x), lc = D.v[j + 1], lr = lv_right(D, j, x, w);
                const V dx = 10 * (cr - cl) + 3 * (ur + lr - ul - ll);
                const V dy = 10 * (lc - uc) + 3 * (ll + lr - ul - ur);
                gq[j] = sf_g_as<V>(lv_conductivity<V>(dx, dy, ikc, type));
            }
            *reinterpret_cast<V4*>(B + r * EW + c) = mk4(gq[0], gq[1], gq[2], gq[3]);
        }
    }
    hak_lds_barrier();
    // ---- ns explicit steps (akazed.cu:1241-1264), ping-pong A <-> C; step k is valid on core +- (ns - k).  A step works on the
    // whole groups that cover its window: the few pixels beyond it are computed from values that are no longer valid and
    // written where no valid pixel of a later step looks (window k+1 +- 1 lies inside window k).
    if (hess) {
        // ---- determinant (gHessianDeterminant akazed.cu:1299-1331) on core +- 1 from the Lx / Ly scratch
        const int S = hs.S;
        const unsigned MD = 0xFFFFFFFFu / (unsigned)DW + 1u;
        for (int idx = tid; idx < DW * DW; idx += LV_NT) {
            const int r = (int)__umulhi((unsigned)idx, MD), c = idx - r * DW;
            const int x = bx * T - 1 + c, y = by * T - 1 + r;
            if (x < 0 || x >= w || y < 0 || y >= h) continue;
            const int xo = bx * T - S - 1, yo = by * T - S - 1;                       // image coordinates of scratch element (0, 0)
            const int c0 = hak_refl(x - S, w) - xo, c1 = x - xo, c2 = hak_refl(x + S, w) - xo;
            const int r0 = (hak_refl(y - S, h) - yo) * XW, r1 = (y - yo) * XW, r2 = (hak_refl(y + S, h) - yo) * XW;
            const V xul = HX_[r0 + c0], xuc = HX_[r0 + c1], xur = HX_[r0 + c2], xcl = HX_[r1 + c0], xcr = HX_[r1 + c2];
            const V xll = HX_[r2 + c0], xlc = HX_[r2 + c1], xlr = HX_[r2 + c2];
            const V yul = HY_[r0 + c0], yuc = HY_[r0 + c1], yur = HY_[r0 + c2];
            const V yll = HY_[r2 + c0], ylc = HY_[r2 + c1], ylr = HY_[r2 + c2];
            const V dxx = hs_d(hs.fac1, hs.fac2, xur + xlr - xul - xll, xcr - xcl);   // akazed.cu:1326-1328
            const V dxy_ = hs_d(hs.fac1, hs.fac2, xlr + xll - xur - xul, xlc - xuc);
            const V dyy = hs_d(hs.fac1, hs.fac2, ylr + yll - yur - yul, ylc - yuc);
            DT[idx] = hs_det(dxx, dyy, dxy_);                                          // akazed.cu:1330
        }
    }
    V* cur = A;
    V* nxt = C;
    for (int k = 1; k <= ns; k++) {
        const int r0 = HY - (ns - k), nr = T + 2 * (ns - k);                          // rows of this step's window
        const int g0 = (HX - (ns - k)) >> 2, g1 = (HX + T + (ns - k) + 3) >> 2;       // groups covering its columns
        const int ng = g1 - g0;
        const V f = fac.f[k - 1];
        const unsigned MN = 0xFFFFFFFFu / (unsigned)ng + 1u;
        for (int idx = tid; idx < nr * ng; idx += LV_NT) {
            const int rr = (int)__umulhi((unsigned)idx, MN), gg = idx - rr * ng;
            const int r = r0 + rr, c = (g0 + gg) << 2, x = X0 + c, y = Y0 + r;
            if (y < 0 || y >= h) continue;
            const int rn = (r + (y == 0 ? 1 : -1)) * EW, rs = (r + (y == h - 1 ? -1 : 1)) * EW, ro = r * EW;
            const Lv6<V> Lm = lv_row6<V>(cur + ro, c), Gm = lv_row6<V>(B + ro, c);
            const V4 Ln = *reinterpret_cast<const V4*>(cur + rn + c), Gn = *reinterpret_cast<const V4*>(B + rn + c);
            const V4 Ls = *reinterpret_cast<const V4*>(cur + rs + c), Gs = *reinterpret_cast<const V4*>(B + rs + c);
            const V ln[4] = {Ln.x, Ln.y, Ln.z, Ln.w}, gn[4] = {Gn.x, Gn.y, Gn.z, Gn.w};
            const V ls[4] = {Ls.x, Ls.y, Ls.z, Ls.w}, gs[4] = {Gs.x, Gs.y, Gs.z, Gs.w};
            V o[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const V L = Lm.v[j + 1], g = Gm.v[j + 1];
                const V tE = vmul(vadd(g, lv_right(Gm, j, x, w)), vsub(lv_right(Lm, j, x, w), L));
                const V tW = vmul(vadd(g, lv_left(Gm, j, x)), vsub(lv_left(Lm, j, x), L));
                const V tS = vmul(vadd(g, gs[j]), vsub(ls[j], L));
                const V tN = vmul(vadd(g, gn[j]), vsub(ln[j], L));
                o[j] = vstep(f, vadd(vadd(vadd(tE, tW), tS), tN), L);                     // akazed.cu:1259-1263
            }
            *reinterpret_cast<V4*>(nxt + ro + c) = mk4(o[0], o[1], o[2], o[3]);
        }
        hak_lds_barrier();
        V* tmp = cur; cur = nxt; nxt = tmp;
        if (k == 1 && hess) {
            // ---- extrema of the level on the core (gCalcExtremaMap akazed.cu:1346-1373): border filter, threshold, strict 3 x 3
            // maximum; winners raise the full-resolution key map and join the image's candidate list (as k_extrema does)
            const int lane = tid & 63;
            for (int base = 0; base < T * T; base += LV_NT) {                         // (uniform trip count: ballots inside)
                const int idx = base + tid;
                const int r = idx / T, c = idx - r * T;
                const int x = bx * T + c, y = by * T + r;
                bool hit = false;
                V v = V(0);
                if (idx < T * T && x >= hs.psz && x < w && y >= hs.psz && y < h &&
                    (int)(x - hs.border + 0.5f) - 1 >= 0 && (int)(x + hs.border + 0.5f) + 1 < w &&
                    (int)(y - hs.border + 0.5f) - 1 >= 0 && (int)(y + hs.border + 0.5f) + 1 < h) {
                    const V* vp = DT + (r + 1) * DW + c + 1;
                    v = *vp;
                    hit = v > hs.threshold && v > vp[-DW] && v > vp[DW] && v > vp[-1] && v > vp[1] &&
                          v > vp[-DW - 1] && v > vp[-DW + 1] && v > vp[DW - 1] && v > vp[DW + 1];
                }
                const unsigned long long m = __ballot(hit);
                if (m) {
                    int cbase = 0;
                    if (lane == 0) cbase = atomicAdd(&hs.state[img].ncand, __popcll(m));
                    cbase = __builtin_amdgcn_readfirstlane(cbase);
                    if (hit) {
                        const int fx = x << hs.octave, fy = y << hs.octave;
                        const unsigned long long key = ((unsigned long long)hs_key_bits(v) << 32) | (0xFFFFFFFFu - (unsigned)hs.layer);
                        atomicMax(&hs.maps[(long)img * hs.map_stride + (long)fy * hs.p0 + fx], key);
                        const long slot = cbase + __popcll(m & ((1ull << lane) - 1ull));
                        if (slot < hs.cand_cap)
                            hs.cand[(long)img * hs.cand_cap + slot] = ((unsigned long long)hs.layer << 32) | ((unsigned)fy << 16) | (unsigned)fx;
                    }
                }
            }
        }
    }
    // ---- the core of the last step -> HBM
    const int tg = T >> 2;
    const unsigned MT = 0xFFFFFFFFu / (unsigned)tg + 1u;
    for (int idx = tid; idx < T * tg; idx += LV_NT) {
        const int r = (int)__umulhi((unsigned)idx, MT), c = (idx - r * tg) << 2;
        const int x = bx * T + c, y = by * T + r;
        if (y >= h || x >= w) continue;
        const V4 v = *reinterpret_cast<const V4*>(cur + (r + HY) * EW + c + HX);
        V* o = od + (long)y * p + x;
        if (x + 3 < w) *reinterpret_cast<V4*>(o) = v;
        else {
            o[0] = v.x;
            if (x + 1 < w) o[1] = v.y;
            if (x + 2 < w) o[2] = v.z;
        }
    }
}

template <typename V, bool HEAD, bool FIRST>
void launch_level(hipStream_t st, const V* src, V* smooth, V* dst, long stride, HakOct so, int w, int h, int p, int nimg,
                  SfTaps<V> t, int diffusivity, const HakImgState* state, int octave, float fixed_ikc, const float* tau, int ns,
                  const LvHess<V>& hs)
{
    LvFacs<V> fac;
    for (int k = 0; k < LV_MAX_STEPS; k++) {
        const float tk = k < ns ? tau[k] : 0.f;
        if constexpr (std::is_same<V, float>::value) fac.f[k] = 0.5f * tk;              // akazed.cu:2515
        else fac.f[k] = (int)(0.5f * tk * 65536 + 0.5f);                                // akazed.cu:4235
    }
    // the largest tile (multiple of 8, at most 64) whose three planes fit the LDS budget
    const int HX = (ns + 5 + 3) & ~3;                      // (as in the kernel)
    const bool hess = FIRST && hs.maps != nullptr;
    auto plane = [&](int t) { return (long)(t + 2 * HX) * (t + 2 * (ns + 3)); };
    auto total = [&](int t) { return 3L * plane(t) + (hess ? 2L * (t + 2 * hs.S + 2) * (t + 2 * hs.S + 2) + (long)(t + 2) * (t + 2) : 0L); };
    int T = 64;
    while (T > 8 && total(T) > LV_LDS_FLOATS) T -= 8;
    // ... but not so large that a small plane runs on a handful of CUs.  Smaller tiles mean more halo work in total (the planes are
    // (T + 2 ns + 6)^2): ~100 blocks keep a block short without multiplying the work of the octaves that run beside the critical chain
    static const long min_blocks = [] { const char* e = getenv("HAK_LEVEL_MIN_BLOCKS"); const long v = e ? atol(e) : 96; return v < 1 ? 1 : v; }();
    while (T > 16 && (long)((w + T - 1) / T) * ((h + T - 1) / T) * nimg < min_blocks) T -= 8;
    const size_t lds = sizeof(V) * (size_t)total(T);
    // the 150 KB dynamic-LDS opt-in is a per-DEVICE attribute of the function: once per device and instantiation (contexts on
    // several devices may live in one process, and two threads may create contexts at once)
    static std::atomic<unsigned long long> attr_done{0};     // bit d: done on device d (per instantiation: each has its own static)
    int dev = 0;
    (void)hipGetDevice(&dev);
    const unsigned long long bit = 1ull << (dev & 63);
    if (!(attr_done.load(std::memory_order_acquire) & bit)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_level_tile<V, HEAD, FIRST>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  LV_LDS_FLOATS * 4);
        attr_done.fetch_or(bit, std::memory_order_release);
    }
    const int nbx = (w + T - 1) / T, nby = (h + T - 1) / T;
    k_level_tile<V, HEAD, FIRST><<<hak_xcd_grid(nbx, nby, nimg), LV_NT, lds, st>>>(src, smooth, dst, stride, so.w, so.h, so.p, w, h, p, t, diffusivity,
                                                                                 state, octave, fixed_ikc, fac, ns, T, nbx, nby, nimg, hs);
}

template <typename V>
int level_steps(hipStream_t st, const V* src, HakOct so, bool head, V* smooth, V* dst, V* tmp, long stride, HakOct dd, int nimg,
                const V* taps, int diffusivity, const float* tau, int n, const HakImgState* state, int octave, float fixed_ikc,
                const LvHess<V>& hs)
{
    const SfTaps<V> t{taps[0], taps[1], taps[2]};
    const int G = (n + LV_MAX_STEPS - 1) / LV_MAX_STEPS;     // launches of this cycle (1 for every BASELINE configuration but 4K octave 4)
    int done = 0;
    const V* cur = src;
    for (int g = 0; g < G; g++) {
        const int ns = hak_fed_group_size(n, G, g);
        V* out = ((G - g) % 2 == 1) ? dst : tmp;             // ping-pong so that the last launch lands in dst
        if (g == 0 && head) launch_level<V, true, true>(st, cur, smooth, out, stride, so, dd.w, dd.h, dd.p, nimg, t, diffusivity, state, octave, fixed_ikc, tau, ns, hs);
        else if (g == 0) launch_level<V, false, true>(st, cur, smooth, out, stride, dd, dd.w, dd.h, dd.p, nimg, t, diffusivity, state, octave, fixed_ikc, tau, ns, hs);
        else launch_level<V, false, false>(st, cur, smooth, out, stride, dd, dd.w, dd.h, dd.p, nimg, t, diffusivity, state, octave, fixed_ikc, tau + done, ns, hs);
        done += ns;
        cur = out;
    }
    return G;
}


template <typename V>
LvHess<V> level_hess(V* dxy, int step, int first_ns, const HakBatch* b, const HakLayout* L, const HakTables* htab, int octave, int sub, V threshold, bool* fused)
{
    LvHess<V> hs{};
    // the derivatives reach 2 S + 1 pixels beyond the core, the low-pass in LDS ns + 1: longer cycles only (every level the
    // size rule sends here, at the demo schedule); the caller launches the Hessian kernel otherwise
    *fused = b != nullptr && dxy != nullptr && step >= 1 && first_ns >= 2 * step;
    if (!*fused) return hs;
    float f1, f2;
    hak_deriv_factors(&f1, &f2);
    if constexpr (std::is_same<V, float>::value) { hs.fac1 = f1; hs.fac2 = f2; }
    else { hs.fac1 = (int)(f1 * 65536 + 0.5f); hs.fac2 = (int)(f2 * 65536 + 0.5f); }     // akazed.cu:4183-4184
    const int layer = octave * L->ms + sub;
    hs.dxy = dxy; hs.S = step;
    hs.maps = b->maps; hs.map_stride = b->map_stride; hs.cand = b->cand; hs.cand_cap = b->cand_cap; hs.state = b->state;
    hs.p0 = L->oct[0].p; hs.octave = octave; hs.layer = layer;
    hs.psz = (int)htab->borders[octave * L->ms]; hs.border = htab->borders[layer]; hs.threshold = threshold;
    return hs;
}

} // namespace

// One sublevel in ceil(n / 36) launches.  src: L(o, s-1), or (head) L(o-1, 0) with extents `so`.  dst receives L(o, s);
// `tmp` is a scratch plane of the octave (used only when the cycle needs more than one launch); `smooth` receives the
// sigma=1 low-pass (the Hessian's input).  With b != nullptr the level's Hessian (derivative plane dxy, extrema into the batch's
// key map and candidate list) runs inside the first launch when the cycle is long enough; *hess_done says whether it did.
// Returns the number of launches.
int hak_launch_level_tile(hipStream_t st, const float* src, HakOct so, bool head, float* smooth, float* dst, float* tmp, long stride,
                          HakOct dd, int nimg, const float* taps, int diffusivity, const float* tau, int n,
                          const HakImgState* state, int octave, float fixed_ikc,
                          float* dxy, int step, const HakBatch* b, const HakLayout* L, const HakTables* htab, int sub, float dthreshold, bool* hess_done)
{
    const int G = (n + LV_MAX_STEPS - 1) / LV_MAX_STEPS;
    bool fused = false;
    const LvHess<float> hs = level_hess<float>(dxy, step, hak_fed_group_size(n, G, 0), b, L, htab, octave, sub, dthreshold, &fused);
    if (hess_done) *hess_done = fused;
    return level_steps<float>(st, src, so, head, smooth, dst, tmp, stride, dd, nimg, taps, diffusivity, tau, n, state, octave, fixed_ikc, hs);
}

int hakf_launch_level_tile(hipStream_t st, const int* src, HakOct so, bool head, int* smooth, int* dst, int* tmp, long stride,
                           HakOct dd, int nimg, const int* itaps, int diffusivity, const float* tau, int n,
                           const HakImgState* state, int octave,
                           int* dxy, int step, const HakBatch* b, const HakLayout* L, const HakTables* htab, int sub, int idthreshold, bool* hess_done)
{
    const int G = (n + LV_MAX_STEPS - 1) / LV_MAX_STEPS;
    bool fused = false;
    const LvHess<int> hs = level_hess<int>(dxy, step, hak_fed_group_size(n, G, 0), b, L, htab, octave, sub, idthreshold, &fused);
    if (hess_done) *hess_done = fused;
    return level_steps<int>(st, src, so, head, smooth, dst, tmp, stride, dd, nimg, itaps, diffusivity, tau, n, state, octave, 0.f, hs);
}
