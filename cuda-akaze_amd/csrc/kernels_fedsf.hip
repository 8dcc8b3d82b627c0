// kernels_fedsf.hip -- sigma=1 low-pass + PM_G2 conductivity + the first NS FED steps of a sublevel in ONE streaming pass.
//
//   hLowPass(Lt(o,s-1) -> smooth, var 1, ksz 5)   akaze.cpp:403, akazed.cu:2336, 204
//   hFlow(smooth -> flow)                         akaze.cpp:404, akazed.cu:2487, 1068
//   hNldStep x NS (Lt(o,s-1), flow -> ...)        akaze.cpp:408-420, akazed.cu:2509, 1241
//
// The unfused sequence moves 12 B/px in k_smooth_flow (read L, write smooth, write g) and 12 B/px in the first
// k_fed_multi launch (read L, read g, write L').  Here a wave streams down its 256-px strip exactly like
// k_fed_multi, but the only input is L: when row t arrives,
//     row pass  rp(t)      of the separable Gaussian (x neighbours at +-1, +-2 by DPP wave shifts),
//     smooth    row t-2    = column pass over rp(t-4 .. t)            -> HBM (the Hessian kernel reads it),
//     g         row t-3    = conductivity of the Scharr gradient of smooth rows t-4, t-3, t-2
//                            -> HBM only when later FED launches of this cycle still need it,
//     FED level k          produces row t-3-k from level k-1 (as in kernels_fed.hip, delayed by three rows),
// so the pass reads 4 B/px and writes 8 (or 12) B/px instead of 24.  Rings (static slots, loop unrolled by 6):
// L 6 rows, rp 6 rows, smooth 3 rows (+ its two x-neighbour columns), the g-sum rings and per-level windows of
// the FED kernel.  No LDS, no barriers.
//
// Bit-exactness: every stage evaluates the reference expression in the reference order (Gaussian: c*k0, then
// += k1*(l1+r1), then += k2*(l2+r2), akazed.cu:227-239 / 283-288; Scharr / conductivity akazed.cu:1088-1106).
// Reflect-101: in x by selects in the lane that holds image column 0 / w-1; in y by writing a row that arrives at
// 1 (2) also into the slot of row -1 (-2) and by copying rows h-2 (h-3) into the virtual rows h (h+1) -- for the
// rp ring (the Gaussian is symmetric with commutative tap pairs, so the smooth value computed at a mirrored
// position IS the smooth value at the reflected index, which is what gFlowNaive reads), for the smooth ring and
// for the FED rings as before.
//
// Both element types (float: akaze, int: fastakaze 16.16), PM_G2, w % 4 == 0 only; everything else takes
// k_smooth_flow + k_fed_multi.
#include "fed_common.h"

#ifndef HAK_FS_PD
#define HAK_FS_PD 3
#endif
#ifndef HAK_FS_PD_MAXNS
#define HAK_FS_PD_MAXNS 4
#endif

namespace {

template <typename V, int NS>
struct FsState {
    using V4 = typename FedV<V>::V4;
    static constexpr int GS = 6;
    static constexpr int PD = NS <= HAK_FS_PD_MAXNS ? HAK_FS_PD : 3;     // input rows in flight (divides the unroll factor 6)
    V4 Lr[6];                           // L rows t-5 .. t                      slot = iteration mod 6
    V4 Rp[6];                           // row-pass rows t-5 .. t
    V4 Sm[3];                           // smooth rows a-2 .. a (a = t-2)       slot = iteration mod 3
    V SmL[3], SmR[3];                   // smooth at columns x0-1 and x0+4 of those rows
    V4 Lw[NS][2];                       // FED level k >= 1: its two newest rows ([0] unused: level 0 reads Lr)
    V4 Qw[NS][2];                       // vertical flux products of level j (fed_common.h fed_row), slot = row mod 2
    GHrow<V> GH[GS];
    V4 GV[GS];
    V4 gprev;
    V4 Lq[PD];                          // prefetch ring
};

// row `row` of the kernel's input L, columns xl .. xl+3.  DEC = false: L is the plane itself.  DEC = true (octave head,
// hDownWithSmooth akazed.cu:449-511): L is Lt(o-1,0) and the input is its 2x decimation, L(x, y) = src(2x, 2y); the two
// 16-byte loads are never copied, the even components are simply the registers the consumers read.
template <bool DEC, typename V, typename V4>
__device__ __forceinline__ V4 fs_fetch(const V* __restrict__ L, const int row, const int lp, const int xl, const int sh)
{
    if constexpr (!DEC) return hak_load_stream(reinterpret_cast<const V4*>(L + (long)row * lp + xl));
    else {
        const V* r = L + (long)min(2 * row, sh - 1) * lp + min(2 * xl, lp - 8);
        const V4 a = hak_load_stream(reinterpret_cast<const V4*>(r)), b = hak_load_stream(reinterpret_cast<const V4*>(r + 4));
        return mk4(a.x, a.z, b.x, b.z);
    }
}

// smooth and g as one raw buffer (lower pointer): unconditional, countable stores (fed_common.h).  smo / go: per-lane byte
// offsets (column + plane) or HAK_BUF_OOB for lanes that own nothing
struct FsOut { __amdgpu_buffer_rsrc_t r; unsigned smo, go; };

// lp / sh: pitch and row count of the plane L points to (= p, h unless DEC)
template <typename V, int NS, int U, bool YEDGE, bool XE, bool WRITE_G, bool DEC, bool WRITE_SM>
__device__ __forceinline__ void fs_iter(FsState<V, NS>& S, const int t, const V* __restrict__ L, V* __restrict__ SMO,
                                        V* __restrict__ GO, V* __restrict__ D, const int p, const int xl,
                                        const int x0, const int w, const int h, const int ybeg, const int yend,
                                        const bool owns, const FedFacs<V, NS>& fac, const SfTaps<V> kk, const float ikc,
                                        const int lp, const int sh, const FsOut& O)
{
    using V4 = typename FedV<V>::V4;
    constexpr int GS = FsState<V, NS>::GS, PD = FsState<V, NS>::PD;
    const bool le = x0 == 0, re = x0 + 3 == w - 1;
    // ---- L row t arrives; request row t + PD (clamped: rows past the image are never used)
    {
        const V4 Lc = S.Lq[pmod(U, PD)];
        S.Lr[pmod(U, 6)] = Lc;
        S.Lq[pmod(U, PD)] = fs_fetch<DEC, V, V4>(L, min(t + PD, h - 1), lp, xl, sh);
    }
    // ---- row pass of the Gaussian on row t (akazed.cu:227-239)
    {
        const V4 c = S.Lr[pmod(U, 6)];
        const V sl1 = wave_shr1(c.w), sl2 = wave_shr1(c.z), sr1 = wave_shl1(c.x), sr2 = wave_shl1(c.y);
        V4 l1 = mk4(sl1, c.x, c.y, c.z), l2 = mk4(sl2, sl1, c.x, c.y);
        V4 r1 = mk4(c.y, c.z, c.w, sr1), r2 = mk4(c.z, c.w, sr1, sr2);
        if (XE) {
            l1.x = le ? c.y : l1.x;                         // column -1 -> 1
            l2.x = le ? c.z : l2.x;                         // column -2 -> 2
            l2.y = le ? c.y : l2.y;                         // column -1 -> 1
            if constexpr (!DEC) {
                r1.w = re ? c.z : r1.w;                     // column w   -> w-2
                r2.z = re ? c.z : r2.z;                     // column w   -> w-2
                r2.w = re ? c.y : r2.w;                     // column w+1 -> w-3
            } else {
                // octave head: the mirror is taken on the SOURCE extents (akazed.cu:466, 477-494); with an even source
                // width that is column w -> w-1, w+1 -> w-2 on the decimated lattice
                r1.w = re ? c.w : r1.w;
                r2.z = re ? c.w : r2.z;
                r2.w = re ? c.z : r2.w;
            }
        }
        V4 rp;
        rp.x = sf_conv(c.x, l1.x, r1.x, l2.x, r2.x, kk);
        rp.y = sf_conv(c.y, l1.y, r1.y, l2.y, r2.y, kk);
        rp.z = sf_conv(c.z, l1.z, r1.z, l2.z, r2.z, kk);
        rp.w = sf_conv(c.w, l1.w, r1.w, l2.w, r2.w, kk);
        S.Rp[pmod(U, 6)] = rp;
        if (YEDGE) {
            // selects on values, not conditional stores: two adjacent `if (t == ..) ring[slot] = rp` are merged by the optimiser
            // into ONE store through a pointer phi, which keeps the whole ring in scratch memory
            S.Rp[pmod(U - 2, 6)] = vsel4(t == 1, rp, S.Rp[pmod(U - 2, 6)]);          // row -1 := row 1
            S.Rp[pmod(U - 4, 6)] = vsel4(t == 2, rp, S.Rp[pmod(U - 4, 6)]);          // row -2 := row 2
            if constexpr (!DEC) {
                S.Rp[pmod(U, 6)] = vsel4(t == h, S.Rp[pmod(U - 2, 6)], S.Rp[pmod(U, 6)]);        // row h   := row h-2
                S.Rp[pmod(U, 6)] = vsel4(t == h + 1, S.Rp[pmod(U - 4, 6)], S.Rp[pmod(U, 6)]);    // row h+1 := row h-3
            } else {                                                                 // source-extent mirror, even source height
                S.Rp[pmod(U, 6)] = vsel4(t == h, S.Rp[pmod(U - 1, 6)], S.Rp[pmod(U, 6)]);        // row h   := row h-1
                S.Rp[pmod(U, 6)] = vsel4(t == h + 1, S.Rp[pmod(U - 3, 6)], S.Rp[pmod(U, 6)]);    // row h+1 := row h-2
            }
        }
    }
    // ---- column pass -> smooth row a = t - 2 (akazed.cu:283-288)
    {
        const int a = t - 2;
        const V4 c = S.Rp[pmod(U - 2, 6)], u1 = S.Rp[pmod(U - 3, 6)], d1 = S.Rp[pmod(U - 1, 6)];
        const V4 u2 = S.Rp[pmod(U - 4, 6)], d2 = S.Rp[pmod(U, 6)];
        V4 sm;
        sm.x = sf_conv(c.x, u1.x, d1.x, u2.x, d2.x, kk);
        sm.y = sf_conv(c.y, u1.y, d1.y, u2.y, d2.y, kk);
        sm.z = sf_conv(c.z, u1.z, d1.z, u2.z, d2.z, kk);
        sm.w = sf_conv(c.w, u1.w, d1.w, u2.w, d2.w, kk);
        if constexpr (WRITE_SM)
            hak_buf_store_nt(O.r, O.smo + (a >= ybeg && a < yend ? (unsigned)(a * p) * (unsigned)sizeof(V) : HAK_BUF_OOB), sm);
        V sl = wave_shr1(sm.w), sr = wave_shl1(sm.x);
        if (XE) {
            sl = le ? sm.y : sl;                            // abs(x-1) = 1
            sr = re ? sm.z : sr;                            // borderAdd(x,1,w) = w-2
        }
        S.Sm[pmod(U, 3)] = sm; S.SmL[pmod(U, 3)] = sl; S.SmR[pmod(U, 3)] = sr;
        if (YEDGE && a == 1) { S.Sm[pmod(U - 2, 3)] = sm; S.SmL[pmod(U - 2, 3)] = sl; S.SmR[pmod(U - 2, 3)] = sr; }
        if (YEDGE && a == h) {
            S.Sm[pmod(U, 3)] = S.Sm[pmod(U - 2, 3)]; S.SmL[pmod(U, 3)] = S.SmL[pmod(U - 2, 3)]; S.SmR[pmod(U, 3)] = S.SmR[pmod(U - 2, 3)];
        }
    }
    // ---- conductivity row b = t - 3 (akazed.cu:1088-1098, PM_G2) and the FED rings' level-0 bookkeeping for that row
    const int tf = t - 3;                                   // the FED part runs three rows behind the input
    {
        const V4 su = S.Sm[pmod(U - 2, 3)], sc = S.Sm[pmod(U - 1, 3)], sd = S.Sm[pmod(U, 3)];
        const V uL = S.SmL[pmod(U - 2, 3)], uR = S.SmR[pmod(U - 2, 3)];
        const V cL = S.SmL[pmod(U - 1, 3)], cR = S.SmR[pmod(U - 1, 3)];
        const V dL = S.SmL[pmod(U, 3)], dR = S.SmR[pmod(U, 3)];
        V4 g;
        float4 den;                                         // 1 + dif2 of the four pixels
#define FS_G(k, ul, uc, ur, cl, cr, ll, lc, lr)                                             \
        {                                                                                   \
            const V dx = 10 * ((cr) - (cl)) + 3 * ((ur) + (lr) - (ul) - (ll));              \
            const V dy = 10 * ((lc) - (uc)) + 3 * ((ll) + (lr) - (ul) - (ur));              \
            den.k = 1.f + sf_dif2(dx, dy, ikc);                                             \
        }
        FS_G(x, uL, su.x, su.y, cL, sc.y, dL, sd.x, sd.y)
        FS_G(y, su.x, su.y, su.z, sc.x, sc.z, sd.x, sd.y, sd.z)
        FS_G(z, su.y, su.z, su.w, sc.y, sc.w, sd.y, sd.z, sd.w)
        FS_G(w, su.z, su.w, uR, sc.z, cR, sd.z, sd.w, dR)
#undef FS_G
        // g = 1 / den: the 3-instruction reciprocal is bit-identical to the IEEE division on [1, 2^64) (fed_common.h); a wave
        // with any value outside that range (NaN / inf from a degenerate contrast factor) takes the division for all lanes
        const bool fast = den.x < 0x1p64f && den.y < 0x1p64f && den.z < 0x1p64f && den.w < 0x1p64f;
        if (__ballot(!fast) == 0ull) {
            g = mk4(sf_g_as<V>(hak_rcp_newton(den.x)), sf_g_as<V>(hak_rcp_newton(den.y)), sf_g_as<V>(hak_rcp_newton(den.z)),
                    sf_g_as<V>(hak_rcp_newton(den.w)));
        } else {
            g = mk4(sf_g_as<V>(1.f / den.x), sf_g_as<V>(1.f / den.y), sf_g_as<V>(1.f / den.z), sf_g_as<V>(1.f / den.w));
        }
        if (WRITE_G) hak_buf_store_nt(O.r, O.go + (tf >= ybeg && tf < yend ? (unsigned)(tf * p) * (unsigned)sizeof(V) : HAK_BUF_OOB), g);
        const V gr = wave_shl1(g.x);
        S.GH[pmod(U, GS)] = GHrow<V>{vadd(g.x, g.y), vadd(g.y, g.z), vadd(g.z, g.w), vadd(g.w, gr)};
        S.GV[pmod(U - 1, GS)] = mk4(vadd(S.gprev.x, g.x), vadd(S.gprev.y, g.y), vadd(S.gprev.z, g.z), vadd(S.gprev.w, g.w));
        S.gprev = g;
    }
    // ---- FED levels 1..NS: level k produces row rho = tf-k from level k-1's rows rho, rho+1 and its flux rows Q[rho-1], Q[rho]
    // (fed_common.h fed_row; reflect-101 in y = sign flips of the flux rows at rows 0 and h-1)
#pragma unroll
    for (int k = 1; k <= NS; k++) {
        const int rho = tf - k;
        // level 0 = the L ring (row tf-j lives in slot U-3-j); levels >= 1 = their two newest rows
        const V4 Lc = k == 1 ? S.Lr[pmod(U - 3 - 1, 6)] : S.Lw[k - 1][pmod(U - k, 2)];
        const V4 Ls = k == 1 ? S.Lr[pmod(U - 3, 6)] : S.Lw[k - 1][pmod(U - k + 1, 2)];
        V4 Qn = fed_q<V, V4>(S.GV[pmod(U - k, GS)], Ls, Lc);
        V4 Qp = S.Qw[k - 1][pmod(U - k - 1, 2)];
        if (YEDGE) {                                        // (selects on values: a branch here keeps the rings out of registers)
            Qp = vsel4(rho == 0, vneg4(Qn), Qp);            // abs(y-1) = 1
            Qn = vsel4(rho == h - 1, vneg4(Qp), Qn);        // borderAdd(y,1,h) = h-2
        }
        S.Qw[k - 1][pmod(U - k, 2)] = Qn;
        const V4 out = fed_row<XE, V, V4>(Lc, S.GH[pmod(U - k, GS)], Qn, Qp, x0, w, fac.f[k - 1]);
        if (k < NS) {
            S.Lw[k < NS ? k : 0][pmod(U - k, 2)] = out;
        } else if (rho >= ybeg && rho < yend && owns) {
            // kept conditional: an unconditional store means evaluating the last level for every row, which costs ~100 VGPRs
            hak_store_nt(reinterpret_cast<V4*>(D + (long)rho * p + x0), out);
        }
    }
}

template <typename V, int NS, bool XE, bool WRITE_G, bool DEC, bool WRITE_SM>
__device__ __forceinline__ void fs_strip(const V* __restrict__ L, V* __restrict__ SMO, V* __restrict__ GO,
                                         V* __restrict__ D, int w, int h, int p, const FedFacs<V, NS>& fac,
                                         const SfTaps<V> kk, const float ikc, int x0, int ybeg, int yend, bool owns,
                                         const int lp, const int sh)
{
    // WRITE_SM = false (SMO == nullptr): the low-pass has no reader -- the level's Hessian low-passes Lt itself -- and is not stored
    using V4 = typename FedV<V>::V4;
    const int xl = min(max(x0, 0), p - 4);                  // keep every lane's loads inside the plane
    FsOut O;
    {
        V* lo = SMO ? SMO : GO;
        if (WRITE_G && SMO) lo = GO < lo ? GO : lo;
        O.r = hak_buf_rsrc(lo);
        const unsigned xb = (unsigned)x0 * (unsigned)sizeof(V);
        O.smo = owns && SMO ? xb + (unsigned)((SMO - lo) * (long)sizeof(V)) : HAK_BUF_OOB;
        O.go = WRITE_G && owns ? xb + (unsigned)((GO - lo) * (long)sizeof(V)) : HAK_BUF_OOB;
    }
    const int t0 = max(0, ybeg - NS - 4);                   // rp from t0, smooth from t0+2, g from t0+3, level k from t0+3+k
    const int tend = min(yend - 1, h - 1) + NS + 3;         // iteration that emits the strip's last L' row
    FsState<V, NS> S;
    const V z = 0;
    const V4 z4 = mk4(z, z, z, z);
#pragma unroll
    for (int i = 0; i < 6; i++) { S.Lr[i] = z4; S.Rp[i] = z4; }
#pragma unroll
    for (int i = 0; i < 3; i++) { S.Sm[i] = z4; S.SmL[i] = z; S.SmR[i] = z; }
#pragma unroll
    for (int k = 0; k < NS; k++) S.Lw[k][0] = S.Lw[k][1] = S.Qw[k][0] = S.Qw[k][1] = z4;
#pragma unroll
    for (int i = 0; i < FsState<V, NS>::GS; i++) {
        S.GH[i] = GHrow<V>{z, z, z, z};
        S.GV[i] = z4;
    }
    S.gprev = z4;
#pragma unroll
    for (int i = 0; i < FsState<V, NS>::PD; i++)
        S.Lq[i] = fs_fetch<DEC, V, V4>(L, min(t0 + i, h - 1), lp, xl, sh);
    for (int tb = t0; tb <= tend; tb += 6) {
        // reflect injections fire while some stage is at rows 1..2 (t <= NS + 4) or at the virtual rows past h-1
        if (tb <= NS + 4 || tb + 5 >= h) {
            fs_iter<V, NS, 0, true, XE, WRITE_G, DEC, WRITE_SM>(S, tb + 0, L, SMO, GO, D, p, xl, x0, w, h, ybeg, yend, owns, fac, kk, ikc, lp, sh, O);
            fs_iter<V, NS, 1, true, XE, WRITE_G, DEC, WRITE_SM>(S, tb + 1, L, SMO, GO, D, p, xl, x0, w, h, ybeg, yend, owns, fac, kk, ikc, lp, sh, O);
            fs_iter<V, NS, 2, true, XE, WRITE_G, DEC, WRITE_SM>(S, tb + 2, L, SMO, GO, D, p, xl, x0, w, h, ybeg, yend, owns, fac, kk, ikc, lp, sh, O);
            fs_iter<V, NS, 3, true, XE, WRITE_G, DEC, WRITE_SM>(S, tb + 3, L, SMO, GO, D, p, xl, x0, w, h, ybeg, yend, owns, fac, kk, ikc, lp, sh, O);
            fs_iter<V, NS, 4, true, XE, WRITE_G, DEC, WRITE_SM>(S, tb + 4, L, SMO, GO, D, p, xl, x0, w, h, ybeg, yend, owns, fac, kk, ikc, lp, sh, O);
            fs_iter<V, NS, 5, true, XE, WRITE_G, DEC, WRITE_SM>(S, tb + 5, L, SMO, GO, D, p, xl, x0, w, h, ybeg, yend, owns, fac, kk, ikc, lp, sh, O);
        } else {
            fs_iter<V, NS, 0, false, XE, WRITE_G, DEC, WRITE_SM>(S, tb + 0, L, SMO, GO, D, p, xl, x0, w, h, ybeg, yend, owns, fac, kk, ikc, lp, sh, O);
            fs_iter<V, NS, 1, false, XE, WRITE_G, DEC, WRITE_SM>(S, tb + 1, L, SMO, GO, D, p, xl, x0, w, h, ybeg, yend, owns, fac, kk, ikc, lp, sh, O);
            fs_iter<V, NS, 2, false, XE, WRITE_G, DEC, WRITE_SM>(S, tb + 2, L, SMO, GO, D, p, xl, x0, w, h, ybeg, yend, owns, fac, kk, ikc, lp, sh, O);
            fs_iter<V, NS, 3, false, XE, WRITE_G, DEC, WRITE_SM>(S, tb + 3, L, SMO, GO, D, p, xl, x0, w, h, ybeg, yend, owns, fac, kk, ikc, lp, sh, O);
            fs_iter<V, NS, 4, false, XE, WRITE_G, DEC, WRITE_SM>(S, tb + 4, L, SMO, GO, D, p, xl, x0, w, h, ybeg, yend, owns, fac, kk, ikc, lp, sh, O);
            fs_iter<V, NS, 5, false, XE, WRITE_G, DEC, WRITE_SM>(S, tb + 5, L, SMO, GO, D, p, xl, x0, w, h, ybeg, yend, owns, fac, kk, ikc, lp, sh, O);
        }
    }
}

constexpr int FS_HX = 8;                                    // x halo: 2 (Gaussian) + 1 (Scharr) + NS (FED) <= 7, multiple of 4
constexpr int FS_XV = 256 - 2 * FS_HX;

// grid: hak_xcd_grid(strips, strip-row groups, images); a block's four waves take four consecutive row segments
template <typename V, int NS, bool WRITE_G, bool DEC, bool WRITE_SM>
__global__ __launch_bounds__(256) void k_fed_sf(const V* __restrict__ src, V* __restrict__ smooth, V* __restrict__ flow,
                                                V* __restrict__ dst, long stride, int w, int h, int p,
                                                FedFacs<V, NS> fac, SfTaps<V> kk, const HakImgState* __restrict__ state, int octave,
                                                float fixed_ikc, int ry, int nbx, int nby, int nimg, int lp, int sh)
{
    int bx, by, img;
    if (!hak_xcd_decode(nbx, nby, nimg, bx, by, img)) return;
    const V* L = src + (long)img * stride;
    V* SMO = smooth ? smooth + (long)img * stride : nullptr;
    V* GO = flow + (long)img * stride;
    V* D = dst + (long)img * stride;
    const float ikc = state ? state[img].ikc[octave] : fixed_ikc;
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int x0 = bx * FS_XV - FS_HX + 4 * lane;           // first pixel of this lane (may lie outside the image)
    const int ybeg = (by * 4 + wv) * ry;
    if (ybeg >= h) return;                                  // wave-uniform
    const int yend = min(ybeg + ry, h);
    const bool owns = 4 * lane >= FS_HX && 4 * lane < FS_HX + FS_XV && x0 < w && x0 >= 0;
    if (bx == 0 || (bx + 1) * FS_XV + FS_HX >= w) fs_strip<V, NS, true, WRITE_G, DEC, WRITE_SM>(L, SMO, GO, D, w, h, p, fac, kk, ikc, x0, ybeg, yend, owns, lp, sh);
    else fs_strip<V, NS, false, WRITE_G, DEC, WRITE_SM>(L, SMO, GO, D, w, h, p, fac, kk, ikc, x0, ybeg, yend, owns, lp, sh);
}

// sp > 0: octave head -- `src` is Lt(o-1,0) with pitch sp and sh rows, the kernel's input is its 2x decimation
template <typename V, int NS>
void launch_fs(hipStream_t st, const V* src, V* smooth, V* flow, V* dst, long stride, int w, int h, int p,
               int nimg, SfTaps<V> kk, const float* tau, const HakImgState* state, int octave, float fixed_ikc, bool write_g,
               int sp = 0, int sh = 0)
{
    FedFacs<V, NS> fac;
    for (int k = 0; k < NS; k++) {
        if constexpr (std::is_same<V, float>::value) fac.f[k] = 0.5f * tau[k];      // akazed.cu:2515
        else fac.f[k] = (int)(0.5f * tau[k] * 65536 + 0.5f);                        // akazed.cu:4235
    }
    const int gx = (w + FS_XV - 1) / FS_XV;
    // rows per wave: tall segments amortise the NS+4 warm-up rows; shrink while the grid cannot fill the chip
    const int ry = hak_stream_rows(h, (long)gx * nimg, 8);
    const int gy = (h + 4 * ry - 1) / (4 * ry);
    const unsigned grid = hak_xcd_grid(gx, gy, nimg);
    if (!smooth) {                                          // (sublevels only; float only: launch_fs_any)
        if constexpr (std::is_same<V, float>::value) {
            if (write_g) k_fed_sf<V, NS, true, false, false><<<grid, 256, 0, st>>>(src, smooth, flow, dst, stride, w, h, p, fac, kk, state, octave, fixed_ikc, ry, gx, gy, nimg, p, h);
            else k_fed_sf<V, NS, false, false, false><<<grid, 256, 0, st>>>(src, smooth, flow, dst, stride, w, h, p, fac, kk, state, octave, fixed_ikc, ry, gx, gy, nimg, p, h);
        }
    } else if (sp > 0) {
        if (write_g) k_fed_sf<V, NS, true, true, true><<<grid, 256, 0, st>>>(src, smooth, flow, dst, stride, w, h, p, fac, kk, state, octave, fixed_ikc, ry, gx, gy, nimg, sp, sh);
        else k_fed_sf<V, NS, false, true, true><<<grid, 256, 0, st>>>(src, smooth, flow, dst, stride, w, h, p, fac, kk, state, octave, fixed_ikc, ry, gx, gy, nimg, sp, sh);
    } else {
        if (write_g) k_fed_sf<V, NS, true, false, true><<<grid, 256, 0, st>>>(src, smooth, flow, dst, stride, w, h, p, fac, kk, state, octave, fixed_ikc, ry, gx, gy, nimg, p, h);
        else k_fed_sf<V, NS, false, false, true><<<grid, 256, 0, st>>>(src, smooth, flow, dst, stride, w, h, p, fac, kk, state, octave, fixed_ikc, ry, gx, gy, nimg, p, h);
    }
}

template <typename V>
bool launch_fs_any(hipStream_t st, const V* src, V* smooth, V* flow, V* dst, long stride, int w, int h, int p, int nimg,
                   SfTaps<V> kk, int diffusivity, const float* tau, int ns, const HakImgState* state, int octave, float fixed_ikc,
                   bool write_g, int sp = 0, int sh = 0)
{
    if (diffusivity != HAK_PM_G2 || (w & 3) || w < 16 || h < 8 || ns < 1 || ns > 4) return false;
    if (!smooth && (sp > 0 || !std::is_same<V, float>::value)) return false;       // not stored: float sublevels only
    // smooth and g are addressed as 32-bit byte offsets from the lower of them: plane offset + plane size < the marker
    if ((write_g && smooth ? (flow < smooth ? smooth - flow : flow - smooth) : 0L) + (long)h * p >= (long)HAK_BUF_OOB / (long)sizeof(V)) return false;
    switch (ns) {
    case 1: launch_fs<V, 1>(st, src, smooth, flow, dst, stride, w, h, p, nimg, kk, tau, state, octave, fixed_ikc, write_g, sp, sh); break;
    case 2: launch_fs<V, 2>(st, src, smooth, flow, dst, stride, w, h, p, nimg, kk, tau, state, octave, fixed_ikc, write_g, sp, sh); break;
    case 3: launch_fs<V, 3>(st, src, smooth, flow, dst, stride, w, h, p, nimg, kk, tau, state, octave, fixed_ikc, write_g, sp, sh); break;
    default: launch_fs<V, 4>(st, src, smooth, flow, dst, stride, w, h, p, nimg, kk, tau, state, octave, fixed_ikc, write_g, sp, sh); break;
    }
    return true;
}

__global__ __launch_bounds__(256) void k_rcp_check(unsigned lo, unsigned hi, unsigned long long* bad)
{
    unsigned long long n = 0;
    for (unsigned long long b = lo + blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x; b < hi;
         b += (unsigned long long)gridDim.x * blockDim.x) {
        const float d = __uint_as_float((unsigned)b);
        n += __float_as_uint(hak_rcp_newton(d)) != __float_as_uint(1.0f / d);
    }
    if (n) atomicAdd(bad, n);
}

}   // namespace

// number of floats with bit patterns in [lo, hi) for which hak_rcp_newton(d) differs from the IEEE quotient 1.0f / d
int hak_launch_rcp_check(unsigned lo, unsigned hi, unsigned long long* d_bad)
{
    k_rcp_check<<<4096, 256>>>(lo, hi, d_bad);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}

// smooth = G1(src); g = PM_G2(smooth) (written to `flow` only when write_g); dst = ns FED steps of src under g.
// Returns false when the case is not covered (caller: hak_launch_smooth_flow + hak_launch_fed_group).
bool hak_launch_fed_sf(hipStream_t st, const float* src, float* smooth, float* flow, float* dst, long stride,
                       int w, int h, int p, int nimg, const float* taps, int diffusivity, const float* tau, int ns,
                       const HakImgState* state, int octave, float fixed_ikc, bool write_g, bool store_smooth)
{
    return launch_fs_any<float>(st, src, store_smooth ? smooth : nullptr, flow, dst, stride, w, h, p, nimg, SfTaps<float>{taps[0], taps[1], taps[2]},
                                diffusivity, tau, ns, state, octave, fixed_ikc, write_g);
}

// the integer FAST path's sublevel head (akaze.cpp:664-695): itaps = (int)(tap * 65536 + 0.5f)
bool hakf_launch_fed_sf(hipStream_t st, const int* src, int* smooth, int* flow, int* dst, long stride,
                        int w, int h, int p, int nimg, const int* itaps, int diffusivity, const float* tau, int ns,
                        const HakImgState* state, int octave, bool write_g)
{
    return launch_fs_any<int>(st, src, smooth, flow, dst, stride, w, h, p, nimg, SfTaps<int>{itaps[0], itaps[1], itaps[2]},
                              diffusivity, tau, ns, state, octave, 0.f, write_g);
}

// octave head (akaze.cpp:369-392): src = Lt(o-1,0) of the previous octave (so = its geometry); smooth = G1(decimated src) with
// the source-extent mirror of hDownWithSmooth, g = PM_G2(smooth), dst = ns FED steps of the decimated plane.  The decimated
// plane itself is never written.  Covered for even source extents only (odd ones mirror onto source pixels that are not on
// the decimated lattice): otherwise returns false (caller: k_down_smooth + k_flow + k_fed_multi).
bool hak_launch_fed_sf_head(hipStream_t st, const float* src, HakOct so, float* smooth, float* flow, float* dst, long stride,
                            HakOct dd, int nimg, const float* taps, int diffusivity, const float* tau, int ns,
                            const HakImgState* state, int octave, bool write_g)
{
    if ((so.w & 1) || (so.h & 1) || so.p < 2 * 8 || dd.w != so.w / 2 || dd.h != so.h / 2) return false;
    return launch_fs_any<float>(st, src, smooth, flow, dst, stride, dd.w, dd.h, dd.p, nimg, SfTaps<float>{taps[0], taps[1], taps[2]},
                                diffusivity, tau, ns, state, octave, 0.f, write_g, so.p, so.h);
}

// the FAST path's octave head (fastakaze::gDownWithSmooth akazed.cu:3143-3205 uses the same source-extent mirror)
bool hakf_launch_fed_sf_head(hipStream_t st, const int* src, HakOct so, int* smooth, int* flow, int* dst, long stride,
                             HakOct dd, int nimg, const int* itaps, int diffusivity, const float* tau, int ns,
                             const HakImgState* state, int octave, bool write_g)
{
    if ((so.w & 1) || (so.h & 1) || so.p < 2 * 8 || dd.w != so.w / 2 || dd.h != so.h / 2) return false;
    return launch_fs_any<int>(st, src, smooth, flow, dst, stride, dd.w, dd.h, dd.p, nimg, SfTaps<int>{itaps[0], itaps[1], itaps[2]},
                              diffusivity, tau, ns, state, octave, 0.f, write_g, so.p, so.h);
}
