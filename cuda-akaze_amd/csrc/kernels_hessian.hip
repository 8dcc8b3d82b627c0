// kernels_hessian.hip -- derivatives + Hessian determinant + per-level extrema, one fused kernel.
//
//   hHessianDeterminant/gDerivate/gHessianDeterminant   akazed.cu:2531, 1267, 1299
//   hCalcExtremaMap/gCalcExtremaMap                     akazed.cu:2563, 1334
//
// For dilation S in {1..4}:  smooth tile (+2S+1 halo) -> Lx, Ly tile (+S+1 halo, kept in LDS,
// centre written out) -> det tile (+1 halo, in LDS, centre written out) -> 3x3 strict-maximum test
// + threshold + border test -> atomicMax into the full-resolution key map and an entry in the
// image's candidate list (the NMS then visits candidates instead of scanning the dense map).
// HBM traffic: read smooth once (4 B/px), write the interleaved {Lx, Ly} plane (8 B/px); nothing is re-read.  The
// determinant stays in LDS (it is written out only for the stage tests): the refinement re-evaluates it (hak_det_at).
//
// The kernel is bytes-in-flight-bound when a block loads one tile and then computes with nothing
// outstanding (rocprof: ~4 % VALU-active, waves waiting on memory), so a block is PERSISTENT over a
// vertical run of tiles and prefetches the next tile's smooth values into registers while it works
// on the current tile from LDS.  Mapping: lane = tile column, wave = tile row (mod 4): row indices
// are wave-uniform (SALU), column indices are computed once per thread; INTERIOR tiles (no
// reflection anywhere in the halo) address LDS with compile-time offsets only.
//
// Same per-pixel expressions and reflect-101 index rule as the reference (akazed.cu:1284-1295,
// 1326-1330, 1346-1373).
#include "fed_common.h"
#include <cstdlib>
#include <type_traits>

#define HF_TX 64
#define HF_CBUF 256                               // staged candidates per block
#define HF_NW 4                                  // waves per block (8 measured slower: 4.28 vs 3.55 ms)
#define HF_E 1                                   // extra halo so the det tile has its 3x3 neighbourhood
template <int S> struct HessGeo {
    static constexpr int TY = S == 4 ? 28 : 32;                            // keeps LDS <= 40 KB: 4 blocks / CU
    // tile extents (W* = columns in use); the LDS pitches SW / DW / EW are the widths rounded up to odd
    static constexpr int WS = HF_TX + 2 * HF_E + 4 * S, SH = TY + 2 * HF_E + 4 * S;   // smooth tile
    static constexpr int WD = HF_TX + 2 * HF_E + 2 * S, DH = TY + 2 * HF_E + 2 * S;   // Lx / Ly tile
    static constexpr int WE = HF_TX + 2 * HF_E, EH = TY + 2 * HF_E;                   // det tile
    static constexpr int SW = WS | 1, DW = WD | 1, EW = WE | 1;
    static constexpr int NR = (SH + HF_NW - 1) / HF_NW;                                          // smooth rows per wave
};

template <typename V, int S>
struct HessPrefetch { V a[HessGeo<S>::NR], b[HessGeo<S>::NR]; };                 // columns lane, 64+lane

template <typename V, int S>
__device__ __forceinline__ void hess_fetch(HessPrefetch<V, S>& P, const V* __restrict__ s, int w, int h, int p,
                                           int x0, int y0, int lane, int wv)
{
    using G = HessGeo<S>;
    const int sx0 = x0 - HF_E - 2 * S, sy0 = y0 - HF_E - 2 * S;
    const int ca = hak_refl(sx0 + lane, w), cb = hak_refl(sx0 + 64 + lane, w);
#pragma unroll
    for (int i = 0; i < G::NR; i++) {
        const int r = wv + HF_NW * i;
        if (r < G::SH) {
            const V* row = s + (long)hak_refl(sy0 + r, h) * p;
            P.a[i] = row[ca];
            if (lane < G::WS - 64) P.b[i] = row[cb];
        }
    }
}

template <typename V, int S>
__device__ __forceinline__ void hess_commit(const HessPrefetch<V, S>& P, V* sm, int lane, int wv)
{
    using G = HessGeo<S>;
#pragma unroll
    for (int i = 0; i < G::NR; i++) {
        const int r = wv + HF_NW * i;
        if (r < G::SH) {
            sm[r * G::SW + lane] = P.a[i];
            if (lane < G::WS - 64) sm[r * G::SW + 64 + lane] = P.b[i];
        }
    }
}

template <typename V>
struct HakExtremaArgs {
    unsigned long long* maps;       // [nimg][map_stride]
    long map_stride;
    unsigned long long* cand;       // [nimg][cand_cap]
    long cand_cap;
    HakImgState* state;
    int p0;                         // pitch of the full-resolution map
    int octave, layer;
    int psz;                        // (int)borders[octave*ms]     akazed.cu:2572
    float border;
    V threshold;
};

template <typename V, int S, bool INTERIOR>
__device__ __forceinline__ void hessian_tile(V* __restrict__ oxy, V* __restrict__ od,
                                             int w, int h, int p, int x0, int y0, V fac1, V fac2,
                                             V* __restrict__ sm, V* __restrict__ sx, V* __restrict__ sy, int lane, int wv,
                                             const HakExtremaArgs<V>& ex, int img, unsigned long long* cbuf, int* ccnt, const int ccap)
{
    using G = HessGeo<S>;
    constexpr int SW = G::SW, DW = G::DW, DH = G::DH, EW = G::EW, EH = G::EH, TY = G::TY;
    const int sx0 = x0 - HF_E - 2 * S, sy0 = y0 - HF_E - 2 * S;     // image coordinates of sm[0][0]
    const int dx0 = x0 - HF_E - S, dy0 = y0 - HF_E - S;             // image coordinates of sx[0][0]
    const int ex0 = x0 - HF_E, ey0 = y0 - HF_E;                     // image coordinates of the det tile
    // Work mapping (the kernel was VALU-issue-bound on index arithmetic when tile positions were flattened
    // over the block): every pass keeps ONE coordinate wave-uniform.
    //   main pass  -- the 64 output columns: lane = column, rows dealt round-robin to the four waves; row
    //                 indices, the reflect rule in y and the global row pointers are SALU work and an
    //                 INTERIOR tile addresses LDS with one per-thread base plus immediates;
    //   halo pass  -- the few columns either side that only feed the next stage: lane = ROW, one column per
    //                 wave iteration (dealt from wave 3 downwards: those waves have the shorter row share).
    // Odd LDS pitches (HessGeo) keep the lane = row accesses conflict-free.
    const int x = x0 + lane;
    const bool xin = INTERIOR || x < w;
    // ---- Lx, Ly on the derivative tile, centre -> HBM
    {
        constexpr int CM = S + HF_E;                                // derivative-tile column of output column 0
        const int c1 = lane + CM + S;                               // sm column of x
        const int c0 = INTERIOR ? c1 - S : hak_refl(x - S, w) - sx0;
        const int c2 = INTERIOR ? c1 + S : hak_refl(x + S, w) - sx0;
#pragma unroll 2
        for (int i = 0; i < (DH + HF_NW - 1) / HF_NW; i++) {
            const int r = wv + HF_NW * i;
            if (r >= DH) break;
            const int y = dy0 + r;
            if (!INTERIOR && (y < 0 || y >= h)) continue;
            const int r1 = (r + S) * SW;
            const int r0 = INTERIOR ? r * SW : (hak_refl(y - S, h) - sy0) * SW;
            const int r2 = INTERIOR ? (r + 2 * S) * SW : (hak_refl(y + S, h) - sy0) * SW;
            if (xin) {
                const V ul = sm[r0 + c0], uc = sm[r0 + c1], ur = sm[r0 + c2];
                const V cl = sm[r1 + c0], cr = sm[r1 + c2];
                const V ll = sm[r2 + c0], lc = sm[r2 + c1], lr = sm[r2 + c2];
                const V vx = hs_d(fac1, fac2, ur + lr - ul - ll, cr - cl);       // akazed.cu:1294
                const V vy = hs_d(fac1, fac2, lr + ll - ur - ul, lc - uc);       // akazed.cu:1295
                sx[r * DW + CM + lane] = vx;
                sy[r * DW + CM + lane] = vy;
                if (r >= S + HF_E && r < S + HF_E + TY) {
                    using V2 = typename std::conditional<std::is_same<V, float>::value, float2, int2>::type;
                    V2 pr;
                    pr.x = vx; pr.y = vy;
                    reinterpret_cast<V2*>(oxy + 2 * ((long)y * p + x0))[lane] = pr;      // interleaved {Lx, Ly}
                }
            }
        }
        // halo columns: lane = derivative-tile row
        const int yr = dy0 + lane;
        const bool rin = lane < DH && (INTERIOR || (yr >= 0 && yr < h));
        const int q1 = (lane + S) * SW;
        const int q0 = INTERIOR ? lane * SW : (hak_refl(yr - S, h) - sy0) * SW;
        const int q2 = INTERIOR ? (lane + 2 * S) * SW : (hak_refl(yr + S, h) - sy0) * SW;
        for (int k = HF_NW - 1 - wv; k < 2 * CM; k += HF_NW) {
            const int c = k < CM ? k : k + HF_TX;
            const int xx = dx0 + c;
            if (!INTERIOR && (xx < 0 || xx >= w)) continue;
            const int h1 = c + S;
            const int h0 = INTERIOR ? c : hak_refl(xx - S, w) - sx0;
            const int h2 = INTERIOR ? c + 2 * S : hak_refl(xx + S, w) - sx0;
            if (rin) {
                const V ul = sm[q0 + h0], uc = sm[q0 + h1], ur = sm[q0 + h2];
                const V cl = sm[q1 + h0], cr = sm[q1 + h2];
                const V ll = sm[q2 + h0], lc = sm[q2 + h1], lr = sm[q2 + h2];
                sx[lane * DW + c] = hs_d(fac1, fac2, ur + lr - ul - ll, cr - cl);
                sy[lane * DW + c] = hs_d(fac1, fac2, lr + ll - ur - ul, lc - uc);
            }
        }
    }
    hak_lds_barrier();                                                // sm is dead from here: the det tile reuses it
    // ---- determinant on the det tile, centre -> HBM
    V* dt = sm;
    {
        const int c1 = lane + HF_E + S;                             // sx column of x
        const int c0 = INTERIOR ? c1 - S : hak_refl(x - S, w) - dx0;
        const int c2 = INTERIOR ? c1 + S : hak_refl(x + S, w) - dx0;
#pragma unroll 2
        for (int i = 0; i < (EH + HF_NW - 1) / HF_NW; i++) {
            const int r = wv + HF_NW * i;
            if (r >= EH) break;
            const int y = ey0 + r;
            if (!INTERIOR && (y < 0 || y >= h)) continue;
            const int r1 = (r + S) * DW;
            const int r0 = INTERIOR ? r * DW : (hak_refl(y - S, h) - dy0) * DW;
            const int r2 = INTERIOR ? (r + 2 * S) * DW : (hak_refl(y + S, h) - dy0) * DW;
            if (xin) {
                const V xul = sx[r0 + c0], xuc = sx[r0 + c1], xur = sx[r0 + c2];
                const V xcl = sx[r1 + c0], xcr = sx[r1 + c2];
                const V xll = sx[r2 + c0], xlc = sx[r2 + c1], xlr = sx[r2 + c2];
                const V yul = sy[r0 + c0], yuc = sy[r0 + c1], yur = sy[r0 + c2];
                const V yll = sy[r2 + c0], ylc = sy[r2 + c1], ylr = sy[r2 + c2];
                const V dxx = hs_d(fac1, fac2, xur + xlr - xul - xll, xcr - xcl);
                const V dxy = hs_d(fac1, fac2, xlr + xll - xur - xul, xlc - xuc);
                const V dyy = hs_d(fac1, fac2, ylr + yll - yur - yul, ylc - yuc);
                const V d = hs_det(dxx, dyy, dxy);                                // akazed.cu:1330
                dt[r * EW + HF_E + lane] = d;
                if (od != nullptr && r >= HF_E && r < HF_E + TY) {                 // (uniform) stage tests only
                    V* rd = od + (long)y * p + x0;
                    rd[lane] = d;
                }
            }
        }
        // the two halo columns of the det tile: lane = det-tile row
        const int yr = ey0 + lane;
        const bool rin = lane < EH && (INTERIOR || (yr >= 0 && yr < h));
        const int q1 = (lane + S) * DW;
        const int q0 = INTERIOR ? lane * DW : (hak_refl(yr - S, h) - dy0) * DW;
        const int q2 = INTERIOR ? (lane + 2 * S) * DW : (hak_refl(yr + S, h) - dy0) * DW;
        for (int k = HF_NW - 1 - wv; k < 2 * HF_E; k += HF_NW) {
            const int c = k < HF_E ? k : k + HF_TX;
            const int xx = ex0 + c;
            if (!INTERIOR && (xx < 0 || xx >= w)) continue;
            const int h1 = c + S;
            const int h0 = INTERIOR ? c : hak_refl(xx - S, w) - dx0;
            const int h2 = INTERIOR ? c + 2 * S : hak_refl(xx + S, w) - dx0;
            if (rin) {
                const V xul = sx[q0 + h0], xuc = sx[q0 + h1], xur = sx[q0 + h2];
                const V xcl = sx[q1 + h0], xcr = sx[q1 + h2];
                const V xll = sx[q2 + h0], xlc = sx[q2 + h1], xlr = sx[q2 + h2];
                const V yul = sy[q0 + h0], yuc = sy[q0 + h1], yur = sy[q0 + h2];
                const V yll = sy[q2 + h0], ylc = sy[q2 + h1], ylr = sy[q2 + h2];
                const V dxx = hs_d(fac1, fac2, xur + xlr - xul - xll, xcr - xcl);
                const V dxy = hs_d(fac1, fac2, xlr + xll - xur - xul, xlc - xuc);
                const V dyy = hs_d(fac1, fac2, ylr + yll - yur - yul, ylc - yuc);
                dt[lane * EW + c] = hs_det(dxx, dyy, dxy);
            }
        }
    }
    if (ex.maps == nullptr) return;                                 // (uniform) determinant only
    hak_lds_barrier();
    // ---- extrema of this level on the output tile (akazed.cu:1346-1373)
    const bool xok = x >= ex.psz && (int)(x - ex.border + 0.5f) - 1 >= 0 && (int)(x + ex.border + 0.5f) + 1 < w;
    for (int rr = wv; rr < TY; rr += HF_NW) {
        const int y = y0 + rr;
        bool hit = false;
        // threshold first: almost no pixel passes it, so the wave usually skips the neighbourhood test
        const V* vp = dt + (rr + HF_E) * EW + lane + HF_E;
        const V v = *vp;
        if (__ballot(v > ex.threshold) == 0ull) continue;
        if (v > ex.threshold && xok && y >= ex.psz && (int)(y - ex.border + 0.5f) - 1 >= 0 && (int)(y + ex.border + 0.5f) + 1 < h) {
            hit = v > vp[-EW] && v > vp[EW] && v > vp[-1] && v > vp[1] &&
                  v > vp[-EW - 1] && v > vp[-EW + 1] && v > vp[EW - 1] && v > vp[EW + 1];
        }
        const unsigned long long m = __ballot(hit);
        if (m) {
            // list slots: reserving them in the global counter needs an atomic WITH return, and waiting for it drains the
            // wave's outstanding stores and prefetch loads (s_waitcnt vmcnt(0)).  Candidates are staged in a per-block LDS
            // buffer (LDS atomic: a short lgkmcnt wait) and flushed by the tile loop; only an overflow goes direct.
            // The reservation never publishes a count it cannot back: a compare-and-swap loop either claims [base, base + n)
            // inside the buffer or leaves ccnt untouched and sends this row's candidates straight to the global list
            // (an add followed by a compensating subtract let a third wave claim slots above the final count).
            const int n = __popcll(m);
            int base = 0, direct = 0;
            if (lane == 0) {
                int old = *(volatile int*)ccnt;
                for (;;) {
                    if (old + n > ccap) { direct = 1; break; }
                    const int prev = atomicCAS(ccnt, old, old + n);
                    if (prev == old) break;
                    old = prev;
                }
                base = direct ? atomicAdd(&ex.state[img].ncand, n) : old;
            }
            base = __builtin_amdgcn_readfirstlane(base);
            direct = __builtin_amdgcn_readfirstlane(direct);                  // wave-uniform
            if (hit) {
                const int fx = x << ex.octave, fy = y << ex.octave;
                const unsigned long long key = ((unsigned long long)hs_key_bits(v) << 32) | (0xFFFFFFFFu - (unsigned)ex.layer);
                atomicMax(&ex.maps[(long)img * ex.map_stride + (long)fy * ex.p0 + fx], key);
                const unsigned long long entry = ((unsigned long long)ex.layer << 32) | ((unsigned)fy << 16) | (unsigned)fx;
                const long slot = base + __popcll(m & ((1ull << lane) - 1ull));
                if (!direct) cbuf[slot] = entry;
                else if (slot < ex.cand_cap) ex.cand[(long)img * ex.cand_cap + slot] = entry;
            }
        }
    }
}

// grid: (x tiles, y tile groups, images); a block walks `tiles_per_block` tiles downwards
template <typename V, int S>
__global__ __launch_bounds__(64 * HF_NW) void k_hessian_fused(const V* __restrict__ src, V* __restrict__ dxy,
                                                       V* __restrict__ det, long stride,
                                                       int w, int h, int p, V fac1, V fac2, int tiles_per_block,
                                                       int ntx, int nby, int nimg, HakExtremaArgs<V> ex, int ccap)
{
    using G = HessGeo<S>;
    __shared__ V sm[G::SH * G::SW];
    __shared__ V sx[G::DH * G::DW];
    __shared__ V sy[G::DH * G::DW];
    int bx, by, img;                                              // XCD-aware block order (hak_internal.h)
    if (!hak_xcd_decode(ntx, nby, nimg, bx, by, img)) return;
    const V* s = src + (long)img * stride;
    V* oxy = dxy + (long)img * stride;
    V* od = det ? det + (long)img * stride : nullptr;
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int x0 = bx * HF_TX;
    const int ty0 = by * tiles_per_block;
    const int ntiles = (h + G::TY - 1) / G::TY;
    const int ty1 = min(ty0 + tiles_per_block, ntiles);
    constexpr int HALO = HF_E + 2 * S;
    __shared__ unsigned long long cbuf[HF_CBUF];
    __shared__ int ccnt, cbase;
    if (threadIdx.x == 0) ccnt = 0;
    // staged candidates -> the image's list: one global slot reservation for the whole buffer.  Called by all threads
    // between two barriers that order it against the extrema passes before and after.
    auto flush = [&]() {
        const int n = ccnt;                                           // (uniform: read after a barrier)
        if (n > 0) {
            if (threadIdx.x == 0) cbase = atomicAdd(&ex.state[img].ncand, n);
            __syncthreads();
            const long gb = cbase;
            for (int i = threadIdx.x; i < n; i += 64 * HF_NW)
                if (gb + i < ex.cand_cap) ex.cand[(long)img * ex.cand_cap + gb + i] = cbuf[i];
            __syncthreads();
            if (threadIdx.x == 0) ccnt = 0;
        }
    };
    HessPrefetch<V, S> P;
    if (ty0 < ty1) hess_fetch<V, S>(P, s, w, h, p, x0, ty0 * G::TY, lane, wv);
    for (int ty = ty0; ty < ty1; ty++) {
        const int y0 = ty * G::TY;
        hak_lds_barrier();                                        // previous tile's readers of sm / sx / sy are done
        if (ex.maps != nullptr && ccnt > ccap / 2) flush();    // (ccnt is stable here: every wave passed the barrier)
        hess_commit<V, S>(P, sm, lane, wv);
        hak_lds_barrier();
        if (ty + 1 < ty1) hess_fetch<V, S>(P, s, w, h, p, x0, y0 + G::TY, lane, wv);   // in flight during the compute below
        const bool interior = x0 - HALO >= 0 && x0 + HF_TX + HALO <= w && y0 - HALO >= 0 && y0 + G::TY + HALO <= h;
        if (interior) hessian_tile<V, S, true>(oxy, od, w, h, p, x0, y0, fac1, fac2, sm, sx, sy, lane, wv, ex, img, cbuf, &ccnt, ccap);
        else hessian_tile<V, S, false>(oxy, od, w, h, p, x0, y0, fac1, fac2, sm, sx, sy, lane, wv, ex, img, cbuf, &ccnt, ccap);
    }
    if (ex.maps != nullptr) {
        __syncthreads();
        flush();
    }
}

static void deriv_factors(float& fac1, float& fac2) { hak_deriv_factors(&fac1, &fac2); }

// HakKnobs::hess_stream: 0 never / 1 by size / 2 always.  HakKnobs::hess_cbuf: staged candidates per block actually used
// (1..HF_CBUF); HAK_HESS_CBUF shrinks it so that the tests can drive the overflow path of the staging buffer with ordinary images.
static HakKnobs knobs_of(const HakBatch* b) { return (b && b->knobs) ? *b->knobs : hak_knobs_from_env(); }

template <typename V, int S>
static void launch_fused(hipStream_t st, const V* src, V* dxy, V* det, long stride,
                         int w, int h, int p, int nimg, const HakExtremaArgs<V>& ex, int cbuf_cap)
{
    float f1, f2;
    deriv_factors(f1, f2);
    V v1, v2;
    if constexpr (std::is_same<V, float>::value) { v1 = f1; v2 = f2; }
    else { v1 = (int)(f1 * 65536 + 0.5f); v2 = (int)(f2 * 65536 + 0.5f); }      // akazed.cu:4183-4184
    const int ntx = (w + HF_TX - 1) / HF_TX, nty = (h + HessGeo<S>::TY - 1) / HessGeo<S>::TY;
    // tiles per persistent block: long runs while the grid still covers the chip several times
    int tpb = 8;
    while (tpb > 1 && (long)ntx * ((nty + tpb - 1) / tpb) * nimg < 4096) tpb >>= 1;
    const int nby = (nty + tpb - 1) / tpb;
    k_hessian_fused<V, S><<<hak_xcd_grid(ntx, nby, nimg), 64 * HF_NW, 0, st>>>(src, dxy, det, stride, w, h, p, v1, v2, tpb, ntx, nby, nimg, ex, cbuf_cap < 1 ? 1 : (cbuf_cap > HF_CBUF ? HF_CBUF : cbuf_cap));
}

template <typename V>
static HakExtremaArgs<V> extrema_args(const HakBatch* b, const HakLayout* L, const HakTables* htab, int octave, int sub, V threshold)
{
    HakExtremaArgs<V> ex{};
    if (b) {
        const int layer = octave * L->ms + sub;
        ex.maps = b->maps; ex.map_stride = b->map_stride; ex.cand = b->cand; ex.cand_cap = b->cand_cap;
        ex.state = b->state; ex.p0 = L->oct[0].p; ex.octave = octave; ex.layer = layer;
        ex.psz = (int)htab->borders[octave * L->ms]; ex.border = htab->borders[layer]; ex.threshold = threshold;
    }
    return ex;
}


// derivate + determinant (+ extrema when b != nullptr) of one level.  Returns true when the
// extrema were handled here; false means the caller must run the stand-alone extrema kernel on `det`.
// The fused kernels write `det` only when store_det is set; the dilation > 4 fallback always fills it.
bool hak_launch_hessian_level(hipStream_t st, const float* src, float* dxy, float* det, bool store_det, long stride,
                              int w, int h, int p, int nimg, int step,
                              const HakBatch* b, const HakLayout* L, const HakTables* htab, int octave, int sub, float dthreshold,
                              const float* lp_taps)
{
    // register-streaming kernel (kernels_hessian_stream.hip) when it covers the case; HAK_HESS_STREAM=0 forces the tile kernel
    const HakKnobs kn = knobs_of(b);
    if (lp_taps || hak_stream_pays(kn.hess_stream, w, h, nimg)) {
        float f1, f2;
        deriv_factors(f1, f2);
        if (hak_launch_hessian_stream(st, src, dxy, det, store_det, stride, w, h, p, nimg, step, f1, f2, b, L, htab, octave, sub, dthreshold,
                                      lp_taps))
            return true;
    }
    if (lp_taps) {
        // the caller asked hak_hessian_stream_covers first, so this is not reached; should the two predicates ever drift apart the
        // CALL fails (enqueue_detect returns the error) -- a drop-in library does not end its host process
        hak_note_launch_error("LP Hessian requested for a level the streaming kernel does not cover");
        return true;
    }
    const HakExtremaArgs<float> ex = extrema_args<float>(b, L, htab, octave, sub, dthreshold);
    float* od = store_det ? det : nullptr;
    switch (step) {
    case 1: launch_fused<float, 1>(st, src, dxy, od, stride, w, h, p, nimg, ex, kn.hess_cbuf); return true;
    case 2: launch_fused<float, 2>(st, src, dxy, od, stride, w, h, p, nimg, ex, kn.hess_cbuf); return true;
    case 3: launch_fused<float, 3>(st, src, dxy, od, stride, w, h, p, nimg, ex, kn.hess_cbuf); return true;
    case 4: launch_fused<float, 4>(st, src, dxy, od, stride, w, h, p, nimg, ex, kn.hess_cbuf); return true;
    default: break;
    }
    hak_launch_derivate(st, src, dxy, stride, w, h, p, nimg, step);        // dilation > 4: two direct passes
    hak_launch_hessian(st, dxy, det, stride, w, h, p, nimg, step);
    return false;
}

// the integer FAST path's level (fastakaze::hHessianDeterminant + hCalcExtremaMap, akazed.cu:4175-4195, 4260-4285):
// same kernel on int32 planes.  Returns false for dilation > 4 (caller: kf_derivate / kf_hessian / kf_extrema).
bool hakf_launch_hessian_level(hipStream_t st, const int* src, int* dxy, int* det, bool store_det, long stride,
                               int w, int h, int p, int nimg, int step,
                               const HakBatch* b, const HakLayout* L, const HakTables* htab, int octave, int sub, int idthreshold)
{
    const HakKnobs kn = knobs_of(b);
    if (hak_stream_pays(kn.hess_stream, w, h, nimg)) {
        float f1, f2;
        deriv_factors(f1, f2);
        if (hakf_launch_hessian_stream(st, src, dxy, det, store_det, stride, w, h, p, nimg, step, (int)(f1 * 65536 + 0.5f),
                                       (int)(f2 * 65536 + 0.5f), b, L, htab, octave, sub, idthreshold))
            return true;
    }
    const HakExtremaArgs<int> ex = extrema_args<int>(b, L, htab, octave, sub, idthreshold);
    int* od = store_det ? det : nullptr;
    switch (step) {
    case 1: launch_fused<int, 1>(st, src, dxy, od, stride, w, h, p, nimg, ex, kn.hess_cbuf); return true;
    case 2: launch_fused<int, 2>(st, src, dxy, od, stride, w, h, p, nimg, ex, kn.hess_cbuf); return true;
    case 3: launch_fused<int, 3>(st, src, dxy, od, stride, w, h, p, nimg, ex, kn.hess_cbuf); return true;
    case 4: launch_fused<int, 4>(st, src, dxy, od, stride, w, h, p, nimg, ex, kn.hess_cbuf); return true;
    default: return false;
    }
}
