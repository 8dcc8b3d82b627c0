// kernels_fed.hip -- the FED hot loop: NS explicit diffusion steps fused in one launch.
//
//   hNldStep/gNldStepNaive  akazed.cu:2509, 1241   (one step:  L' = fma(0.5*tau, sum_{E,W,S,N}(g+g_n)(L_n-L), L))
//   loop over tau[k]        akaze.cpp:381-391, 408-420
//
// The conductivity g is fixed for all steps of a sublevel (akaze.cpp:379, 404), so the n steps of a
// FED cycle are pipelined in time.  A wave owns a 256-px-wide strip (one float4 per lane, 1 KiB
// contiguous per row) and streams down the rows.  For each fused level k < NS it keeps a 3-row
// register window; when input row t arrives, level k produces row t-k from level k-1's window, so
// L and g are read from HBM once and L' written once per NS steps (12 B/px per launch instead of
// 12 B/px per step).  East/west neighbours come from the adjacent lane with one DPP wave shift;
// no LDS, no barriers.  The strip's outer HX columns and the NS warm-up rows above/below a strip are
// recomputed by the neighbouring wave (halo): efficiency (256-2HX)/256 x RY/(RY+2NS).
//
// The kernel is VALU-issue-bound (rocprof: 4 waves/SIMD x 26 % VALU-active), so the inner loop is
// written for instruction count:
//   * the pair sums (g+gE), (g+gS) are formed ONCE per g row (GH: 4 horizontal sums per lane,
//     GV: 4 vertical sums) and reused by all NS levels and by both pixels that share the pair
//     -- (g+gE) of pixel x and (g+gW) of pixel x+1 are the same IEEE sum;
//   * the flux products are shared in both directions (fed_common.h fed_row): horizontally
//     P = (g[x-1]+g[x]) * (L[x]-L[x-1]) gives term_E(x-1) = P, term_W(x) = -P exactly, and a lane's P[0] is its left
//     neighbour's P[4] (one wave shift); vertically Q = (g[y]+g[y+1]) * (L[y+1]-L[y]) gives term_S(y) = Q,
//     term_N(y+1) = -Q, so a level keeps ONE L row and ONE Q row between iterations: 8.5 VALU per pixel and step
//     instead of the 16 of the per-pixel expression;
//   * the register windows rotate statically (loop unrolled by 6 with compile-time slots) instead
//     of being shifted with v_mov;
//   * border selects (reflect-101) are applied where a row / column IS a border, not per pixel.
//
// Bit-exactness: every level evaluates the reference's per-pixel expression in the reference's
// order  ((tE + tW) + tS) + tN  then fma(stepfac, sum, L), and the reflect-101 rule (abs /
// borderAdd, akazed.cu:1251-1254) is applied at EVERY level, as sign flips of the flux products at the
// image border (mirroring level-0 rows instead would swap the S and N terms and change roundings).
#include "fed_common.h"

#ifndef HAK_FM_PD
#define HAK_FM_PD 3
#endif
#ifndef HAK_FM_PD_MAXNS
#define HAK_FM_PD_MAXNS 4
#endif

template <typename V, int NS>
struct FedState {
    using V4 = typename FedV<V>::V4;
    static constexpr int GS = 6;        // slots of the g-sum rings (needs NS + 1 <= GS; divides the unroll factor)
    V4 Lw[NS][2];                       // level j (0 = input): its two newest rows, slot = (row - origin) mod 2
    V4 Qw[NS][2];                       // vertical flux products of level j: Q[r] at slot (r - origin) mod 2
    GHrow<V> GH[GS];                    // ring: horizontal sums of g row r        at slot (r - origin) mod GS
    V4 GV[GS];                          // ring: g[r] + g[r+1]                     at slot (r - origin) mod GS
    V4 gprev;                           // g row t-1
    static constexpr int PD = NS <= HAK_FM_PD_MAXNS ? HAK_FM_PD : 3;        // prefetch distance in rows (divides the unroll factor 6)
    V4 Lq[PD], Gq[PD];                  // software prefetch ring: rows t .. t+PD-1 in flight
};

// One row-iteration.  Everything is computed unconditionally: a level-k row outside the range this
// strip can produce exactly is garbage that no valid row ever reads (validity shrinks one row per
// level exactly like the halo), so the body is branch-free except for the final store and the
// reflect-101 rule in y, which only touches the flux rows:  Q[-1] := -Q[0] when a level is at row 0 and
// Q[h-1] := -Q[h-2] when it is at row h-1 (fed_common.h) -- no mirrored L or g rows are ever needed.
template <typename V, int NS, int U, bool YEDGE, bool XE>
__device__ __forceinline__ void fed_iter(FedState<V, NS>& S, const int t, const V* __restrict__ L,
                                         const V* __restrict__ G, V* __restrict__ D, const int p, const int xl,
                                         const int x0, const int w, const int h,
                                         const int ybeg, const int yend, const bool owns, const FedFacs<V, NS>& fac)
{
    using V4 = typename FedV<V>::V4;
    constexpr int GS = FedState<V, NS>::GS;
    // ---- level 0: input row t arrives (prefetched); request row t+PD (clamped: rows past the image are never used)
    {
        constexpr int PD = FedState<V, NS>::PD;
        const V4 g = S.Gq[pmod(U, PD)];
        S.Lw[0][pmod(U, 2)] = S.Lq[pmod(U, PD)];
        // bytes in flight, not issue rate, bound this kernel (one row ahead = 2 KB per wave < latency x bandwidth):
        // keep PD rows of L and g outstanding per wave
        const long nrow = (long)min(t + PD, h - 1) * p + xl;
        S.Lq[pmod(U, PD)] = hak_load_stream(reinterpret_cast<const V4*>(L + nrow));
        S.Gq[pmod(U, PD)] = hak_load_stream(reinterpret_cast<const V4*>(G + nrow));
        const V gr = wave_shl1(g.x);
        S.GH[pmod(U, GS)] = GHrow<V>{vadd(g.x, g.y), vadd(g.y, g.z), vadd(g.z, g.w), vadd(g.w, gr)};
        S.GV[pmod(U - 1, GS)] = mk4(vadd(S.gprev.x, g.x), vadd(S.gprev.y, g.y), vadd(S.gprev.z, g.z), vadd(S.gprev.w, g.w));
        S.gprev = g;
    }
    // ---- levels 1..NS: level k produces row rho = t-k from level k-1's rows rho, rho+1 and its flux rows Q[rho-1], Q[rho]
#pragma unroll
    for (int k = 1; k <= NS; k++) {
        const int rho = t - k;
        const V4 Lc = S.Lw[k - 1][pmod(U - k, 2)];                              // row rho   of level k-1
        V4 Qn = fed_q<V, V4>(S.GV[pmod(U - k, GS)], S.Lw[k - 1][pmod(U - k + 1, 2)], Lc);    // Q[rho] (row rho+1 is this iteration's)
        V4 Qp = S.Qw[k - 1][pmod(U - k - 1, 2)];                                // Q[rho-1]
        if (YEDGE) {                                        // (selects on values: a branch here keeps the rings out of registers)
            Qp = vsel4(rho == 0, vneg4(Qn), Qp);            // abs(y-1) = 1
            Qn = vsel4(rho == h - 1, vneg4(Qp), Qn);        // borderAdd(y,1,h) = h-2
        }
        S.Qw[k - 1][pmod(U - k, 2)] = Qn;
        const V4 out = fed_row<XE, V, V4>(Lc, S.GH[pmod(U - k, GS)], Qn, Qp, x0, w, fac.f[k - 1]);
        if (k < NS) {
            S.Lw[k < NS ? k : 0][pmod(U - k, 2)] = out;
        } else if (rho >= ybeg && rho < yend && owns) {
            hak_store_nt(reinterpret_cast<V4*>(D + (long)rho * p + x0), out);
        }
    }
}

// requires w % 4 == 0 (true for every octave of BASELINE's configs); other widths: k_fed_generic
template <typename V, int NS, bool XE>
__device__ __forceinline__ void fed_strip(const V* __restrict__ L, const V* __restrict__ G,
                                          V* __restrict__ D, int w, int h, int p, const FedFacs<V, NS>& fac,
                                          int x0, int ybeg, int yend, bool owns)
{
    using V4 = typename FedV<V>::V4;
    const int xl = min(max(x0, 0), p - 4);                  // keep every lane's loads inside the plane
    const int t0 = max(0, ybeg - NS);                       // first input row; level k is exact from row t0 + k (or 0)
    const int tend = min(yend - 1, h - 1) + NS;             // iteration that emits the strip's last output row
    FedState<V, NS> S;
    const V z = 0;
#pragma unroll
    for (int k = 0; k < NS; k++) S.Lw[k][0] = S.Lw[k][1] = S.Qw[k][0] = S.Qw[k][1] = mk4(z, z, z, z);
#pragma unroll
    for (int i = 0; i < FedState<V, NS>::GS; i++) {
        S.GH[i] = GHrow<V>{z, z, z, z};
        S.GV[i] = mk4(z, z, z, z);
    }
    S.gprev = mk4(z, z, z, z);
#pragma unroll
    for (int i = 0; i < FedState<V, NS>::PD; i++) {
        const long row = (long)min(t0 + i, h - 1) * p + xl;
        S.Lq[i] = hak_load_stream(reinterpret_cast<const V4*>(L + row));
        S.Gq[i] = hak_load_stream(reinterpret_cast<const V4*>(G + row));
    }
    for (int tb = t0; tb <= tend; tb += 6) {                // ring slot = (row - t0) mod 2 / mod 6: static per unrolled body
        // the reflect rule can only fire while some level is at row 0 (t <= NS) or at row h-1
        if (tb <= NS || tb + 5 >= h) {
            fed_iter<V, NS, 0, true, XE>(S, tb + 0, L, G, D, p, xl, x0, w, h, ybeg, yend, owns, fac);
            fed_iter<V, NS, 1, true, XE>(S, tb + 1, L, G, D, p, xl, x0, w, h, ybeg, yend, owns, fac);
            fed_iter<V, NS, 2, true, XE>(S, tb + 2, L, G, D, p, xl, x0, w, h, ybeg, yend, owns, fac);
            fed_iter<V, NS, 3, true, XE>(S, tb + 3, L, G, D, p, xl, x0, w, h, ybeg, yend, owns, fac);
            fed_iter<V, NS, 4, true, XE>(S, tb + 4, L, G, D, p, xl, x0, w, h, ybeg, yend, owns, fac);
            fed_iter<V, NS, 5, true, XE>(S, tb + 5, L, G, D, p, xl, x0, w, h, ybeg, yend, owns, fac);
        } else {
            fed_iter<V, NS, 0, false, XE>(S, tb + 0, L, G, D, p, xl, x0, w, h, ybeg, yend, owns, fac);
            fed_iter<V, NS, 1, false, XE>(S, tb + 1, L, G, D, p, xl, x0, w, h, ybeg, yend, owns, fac);
            fed_iter<V, NS, 2, false, XE>(S, tb + 2, L, G, D, p, xl, x0, w, h, ybeg, yend, owns, fac);
            fed_iter<V, NS, 3, false, XE>(S, tb + 3, L, G, D, p, xl, x0, w, h, ybeg, yend, owns, fac);
            fed_iter<V, NS, 4, false, XE>(S, tb + 4, L, G, D, p, xl, x0, w, h, ybeg, yend, owns, fac);
            fed_iter<V, NS, 5, false, XE>(S, tb + 5, L, G, D, p, xl, x0, w, h, ybeg, yend, owns, fac);
        }
    }
}

// grid: hak_xcd_grid(strips, strip-row groups, images); a block's four waves take four consecutive row segments
template <typename V, int NS>
__global__ __launch_bounds__(256) void k_fed_multi(const V* __restrict__ src, const V* __restrict__ flow,
                                                   V* __restrict__ dst, long stride, int w, int h, int p,
                                                   FedFacs<V, NS> fac, int ry, int xv, int hx, int nbx, int nby, int nimg)
{
    int bx, by, img;
    if (!hak_xcd_decode(nbx, nby, nimg, bx, by, img)) return;
    const V* L = src + (long)img * stride;
    const V* G = flow + (long)img * stride;
    V* D = dst + (long)img * stride;
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);     // wave-uniform -> row bookkeeping in SGPRs
    const int x0 = bx * xv - hx + 4 * lane;                 // first pixel of this lane (may lie outside the image)
    const int ybeg = (by * 4 + wv) * ry;
    if (ybeg >= h) return;                                  // wave-uniform
    const int yend = min(ybeg + ry, h);
    const bool owns = 4 * lane >= hx && 4 * lane < hx + xv && x0 < w && x0 >= 0;
    if (bx == 0 || (bx + 1) * xv + hx >= w) fed_strip<V, NS, true>(L, G, D, w, h, p, fac, x0, ybeg, yend, owns);
    else fed_strip<V, NS, false>(L, G, D, w, h, p, fac, x0, ybeg, yend, owns);
}

// any width (w % 4 != 0): ONE step per launch with the per-pixel form of the reference expression;
// only odd-sized test images take this path
__global__ __launch_bounds__(256) void k_fed_generic(const float* __restrict__ src, const float* __restrict__ flow,
                                                     float* __restrict__ dst, long stride, int w, int h, int p,
                                                     float stepfac, int ry)
{
    const float* L = src + (long)blockIdx.z * stride;
    const float* G = flow + (long)blockIdx.z * stride;
    float* D = dst + (long)blockIdx.z * stride;
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);     // wave-uniform -> row bookkeeping in SGPRs
    const int x0 = blockIdx.x * 248 - 4 + 4 * lane;
    const int ybeg = (blockIdx.y * 4 + wv) * ry;
    if (ybeg >= h) return;
    const int yend = min(ybeg + ry, h);
    const int xl = min(max(x0, 0), p - 4);
    const bool owns = lane >= 1 && lane < 63 && x0 < w && x0 >= 0;
    for (int y = ybeg; y < yend; y++) {
        const int yn = y == 0 ? 1 : y - 1, ys = y == h - 1 ? h - 2 : y + 1;
        const float4 Lc = *reinterpret_cast<const float4*>(L + (long)y * p + xl);
        const float4 Gc = *reinterpret_cast<const float4*>(G + (long)y * p + xl);
        const float4 Ln = *reinterpret_cast<const float4*>(L + (long)yn * p + xl);
        const float4 Gn = *reinterpret_cast<const float4*>(G + (long)yn * p + xl);
        const float4 Ls = *reinterpret_cast<const float4*>(L + (long)ys * p + xl);
        const float4 Gs = *reinterpret_cast<const float4*>(G + (long)ys * p + xl);
        const float Ll = wave_shr1(Lc.w), Lr = wave_shl1(Lc.x), Gl = wave_shr1(Gc.w), Gr = wave_shl1(Gc.x);
        const float l[6] = {Ll, Lc.x, Lc.y, Lc.z, Lc.w, Lr}, g[6] = {Gl, Gc.x, Gc.y, Gc.z, Gc.w, Gr};
        const float ln[4] = {Ln.x, Ln.y, Ln.z, Ln.w}, gn[4] = {Gn.x, Gn.y, Gn.z, Gn.w};
        const float ls[4] = {Ls.x, Ls.y, Ls.z, Ls.w}, gs[4] = {Gs.x, Gs.y, Gs.z, Gs.w};
        float o[4];
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const int x = x0 + e;
            float LW = l[e], GW = g[e], LE = l[e + 2], GE = g[e + 2];
            if (x == 0) { LW = LE; GW = GE; }               // abs(x-1) = 1
            if (x == w - 1) { LE = l[e]; GE = g[e]; }        // borderAdd(x,1,w) = w-2
            const float Lx = l[e + 1], Gx = g[e + 1];
            const float step = (Gx + GE) * (LE - Lx) + (Gx + GW) * (LW - Lx) + (Gx + gs[e]) * (ls[e] - Lx) +
                               (Gx + gn[e]) * (ln[e] - Lx);                     // akazed.cu:1259-1262
            o[e] = fmaf(stepfac, step, Lx);                                     // akazed.cu:1263
        }
        if (owns) {
            float* drow = D + (long)y * p + x0;
            if (x0 + 3 < w) *reinterpret_cast<float4*>(drow) = make_float4(o[0], o[1], o[2], o[3]);
            else {
                if (x0 < w) drow[0] = o[0];
                if (x0 + 1 < w) drow[1] = o[1];
                if (x0 + 2 < w) drow[2] = o[2];
            }
        }
    }
}

template <typename V, int NS>
static void launch_multi(hipStream_t st, const V* src, const V* flow, V* dst, long stride,
                         int w, int h, int p, int nimg, const float* tau)
{
    FedFacs<V, NS> fac;
    for (int k = 0; k < NS; k++) {
        if constexpr (std::is_same<V, float>::value) fac.f[k] = 0.5f * tau[k];      // akazed.cu:2515
        else fac.f[k] = (int)(0.5f * tau[k] * 65536 + 0.5f);                        // akazed.cu:4235
    }
    const int hx = 4;                                               // x halo >= NS, multiple of 4 (16-byte alignment)
    const int xv = 256 - 2 * hx;
    const int gx = (w + xv - 1) / xv;
    // rows per wave: tall strips amortise the 2*NS warm-up rows; shrink while the grid cannot fill the chip
    const int ry = hak_stream_rows(h, (long)gx * nimg, 8);
    const int gy = (h + 4 * ry - 1) / (4 * ry);
    k_fed_multi<V, NS><<<hak_xcd_grid(gx, gy, nimg), 256, 0, st>>>(src, flow, dst, stride, w, h, p, fac, ry, xv, hx, gx, gy, nimg);
}

// launches needed for n steps at width w when at most max_fuse steps are fused per launch
int hak_fed_groups(int n, int max_fuse, int w)
{
    if (max_fuse < 1 || w % 4 != 0) max_fuse = 1;                   // odd widths: one step per launch
    if (max_fuse > HAK_FED_MAX_FUSE) max_fuse = HAK_FED_MAX_FUSE;
    return (n + max_fuse - 1) / max_fuse;
}

// steps of group g when n steps are split into G balanced groups (sizes differ by at most one)
int hak_fed_group_size(int n, int G, int g)
{
    int done = 0, ns = 0;
    for (int i = 0; i <= g; i++) { ns = (n - done + (G - i) - 1) / (G - i); done += ns; }
    return ns;
}

// ns fused steps src -> dst (dst != src), tau[0..ns); ns must be 1 when w % 4 != 0
void hak_launch_fed_group(hipStream_t st, const float* src, const float* flow, float* dst, long stride,
                          int w, int h, int p, int nimg, const float* tau, int ns)
{
    if (w % 4 != 0) {
        const int gx = (w + 247) / 248;
        int ry = 16;
        while (ry > 2 && (long)gx * ((h + ry - 1) / ry) * nimg < 2048) ry >>= 1;
        dim3 grid(gx, (h + 4 * ry - 1) / (4 * ry), nimg);
        k_fed_generic<<<grid, 256, 0, st>>>(src, flow, dst, stride, w, h, p, 0.5f * tau[0], ry);
        return;
    }
    switch (ns) {
    case 1: launch_multi<float, 1>(st, src, flow, dst, stride, w, h, p, nimg, tau); break;
    case 2: launch_multi<float, 2>(st, src, flow, dst, stride, w, h, p, nimg, tau); break;
    case 3: launch_multi<float, 3>(st, src, flow, dst, stride, w, h, p, nimg, tau); break;
    default: launch_multi<float, 4>(st, src, flow, dst, stride, w, h, p, nimg, tau); break;
    }
}

// integer FAST path (fastakaze::hNldStep, akazed.cu:4231-4238): ns fused 16.16 steps src -> dst; requires w % 4 == 0
// (other widths: the one-step kf_nld_step of kernels_fast.hip)
void hakf_launch_fed_group(hipStream_t st, const int* src, const int* flow, int* dst, long stride,
                           int w, int h, int p, int nimg, const float* tau, int ns)
{
    switch (ns) {
    case 1: launch_multi<int, 1>(st, src, flow, dst, stride, w, h, p, nimg, tau); break;
    case 2: launch_multi<int, 2>(st, src, flow, dst, stride, w, h, p, nimg, tau); break;
    case 3: launch_multi<int, 3>(st, src, flow, dst, stride, w, h, p, nimg, tau); break;
    default: launch_multi<int, 4>(st, src, flow, dst, stride, w, h, p, nimg, tau); break;
    }
}
