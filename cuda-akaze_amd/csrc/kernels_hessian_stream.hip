// kernels_hessian_stream.hip -- derivatives + Hessian determinant + per-level extrema as a register-streaming
// kernel: no LDS, no barriers (same skeleton as kernels_fed.hip / kernels_fedsf.hip).
//
//   hHessianDeterminant/gDerivate/gHessianDeterminant   akazed.cu:2531, 1267, 1299
//   hCalcExtremaMap/gCalcExtremaMap                     akazed.cu:2563, 1334
//
// A wave owns a 256-px-wide strip (one float4 per lane) and streams down the rows of its segment.  For dilation S
// it keeps rings of the last 2S+1 rows of the smoothed input and of Lx in registers (plus the slots of the rows being
// loaded), the Ly ring in per-wave LDS, and the last three rows of the determinant.  When input row t arrives,
//     Lx, Ly  of row b = t - S      come from input rows b-S, b, b+S      (the ring's oldest / middle / newest),
//     det     of row c = t - 2S     comes from Lx / Ly rows c-S, c, c+S,
//     extrema of row e = t - 2S - 1 come from det rows e-1, e, e+1,
// so the input is read once (4 B/px) and the derivatives are written once, INTERLEAVED as {Lx, Ly} pairs (8 B/px, two
// 16-byte non-temporal stores per lane and row: the plane is next touched sparsely, by the descriptor stage, which then
// fetches both derivatives of a sample with one 8-byte gather from one sector).  The determinant is NOT written (only when
// a stage test asks for it): its consumers are the extrema test below and the refinement of the few keypoints, which
// re-evaluates it from the derivative plane (hak_det_at).  Horizontal
// neighbours at distance S are the adjacent lane's components: S DPP wave shifts per direction per row.  The rings
// rotate statically (row loop unrolled by R); the strip's outer M columns and the 2S+1 warm-up rows above / below a
// segment are recomputed by the neighbouring wave.
//
// Reflect-101 (akazed.cu:1284-1291, 1318-1325): in x the reference reflects the INDEX it reads the plane at, so the
// lane that holds image column 0 (w-1) replaces the shifted-in values by its own / its neighbour's components; in y
// a row arriving at 1..S is also written to the ring slot of row -1..-S and the virtual rows h..h+S-1 are copies of
// rows h-2..h-1-S, for each of the three rings.  Per-pixel expressions and their order are those of the tile kernel
// (kernels_hessian.hip) and the reference.
//
// LP variant (round 3): the kernel's input is Lt(o,s-1) itself and the sigma=1 low-pass the level differentiates
// (hLowPass akaze.cpp:403, akazed.cu:2336/204) is evaluated on the way in -- row pass of raw row t+2 by DPP shifts, column pass
// over the five newest row-pass rows -> smooth row t, exactly as k_fed_sf does it (same expressions, same reflect rules).  The
// `smooth` plane then has no reader left and k_fed_sf stops writing it: one 4 B/px store per sublevel less.
//
// Requires w % 4 == 0 and 1 <= S <= 4; everything else takes the LDS tile kernel.
#include "fed_common.h"
#include <utility>

namespace {

template <int S, bool LP> struct HsGeo {
    // input rows in flight ahead of the current one.  Two everywhere but S = 3, which sits at the 168-VGPR cap of its third wave:
    // with two rows in flight it spilled (1 dword in the interior strips, 7 in the edge strips); one row in flight frees a ring slot
    // of both register rings (165 VGPRs, no spill) and is 2-3 % faster for the class (A/B on one box, twice: 14.77 / 14.60 ->
    // 14.26 / 14.30 ms per 512 images).  One row in flight at S = 2 or 4 is slower (+1 %), a third row at any S was (+3 .. +9 %).
    static constexpr int PD = S == 3 ? 1 : 2;
    static constexpr int R = 2 * S + 1 + PD;                        // ring slots = unroll factor: 2S+1 live rows + PD rows being loaded
    // strip margin: multiple of 4, >= 2S+1 (+2 for the low-pass taps of the LP variant)
    static constexpr int M = LP ? (S <= 2 ? 8 : 12) : (S == 1 ? 4 : S == 4 ? 12 : 8);
    static constexpr int XV = 256 - 2 * M;                          // columns a wave stores
    // waves per SIMD the register allocator must make room for (LP: the row-pass ring adds ~24 VGPRs)
    static constexpr int MINW = LP ? (S <= 1 ? 3 : 2) : (S <= 3 ? 3 : 2);
};

template <typename V> struct HsV2;
template <> struct HsV2<float> { using T = float2; };
template <> struct HsV2<int> { using T = int2; };

#define HS_CBUF 256
struct HsCand { unsigned long long* buf; int n; };      // staged candidates: this wave's HS_CBUF LDS entries; n: wave-uniform fill count

template <typename V, int S, bool LP> struct HsState {
    using V4 = typename FedV<V>::V4;
    static constexpr int R = HsGeo<S, LP>::R;
    V4 A[R], X[R];                                                  // slot = iteration index mod R
    // LP only (static slots: the allocator keeps just the live ones): raw rows in flight, row t+2 in slot (U+2) mod R; row-pass
    // rows t-2 .. t+2
    V4 W[LP ? R : 1], Rp[LP ? R : 1];
    V4* Y;                                                      // Ly ring: this wave's private LDS rows [R][64] (written once, read back once)
    V4* XS;                                                     // one staging row for Lx (interleaved store below)
    V4 Dm, Dc, Dp;                                                  // det rows e-1, e, e+1 (rotated by moves)
    HsCand cb;                                                      // staged candidates
};

template <int I, typename V4> __device__ __forceinline__ auto hs_c(const V4& r)
{
    if constexpr (I == 0) return r.x;
    else if constexpr (I == 1) return r.y;
    else if constexpr (I == 2) return r.z;
    else return r.w;
}
// value at column x0 + K - S of the row held as r (columns x0 .. x0+3): the own component, or the left lane's through one DPP
// shift.  One scalar at a time so that only the values of the component being evaluated are live.  No reflect logic here: in
// the strips that contain image column 0 / w-1 the lane just OUTSIDE the image holds the reflected columns (hs_patch).
template <int S, int K, typename V4>
__device__ __forceinline__ auto hs_l(const V4& r)
{
    if constexpr (K - S >= 0) return hs_c<K - S>(r);
    else return wave_shr1(hs_c<4 + K - S>(r));
}
// value at column x0 + K + S
template <int S, int K, typename V4>
__device__ __forceinline__ auto hs_r(const V4& r)
{
    if constexpr (K + S <= 3) return hs_c<K + S>(r);
    else return wave_shl1(hs_c<K + S - 4>(r));
}
// Reflect-101 in x, once per ring row instead of once per tap (round 4).  The reference reflects the INDEX it reads a plane at
// (akazed.cu:1284-1291, 1318-1325): column -k reads column k, column w-1+k reads column w-1-k -- in EVERY plane (smooth, Lx, Ly)
// separately.  So when a row enters a ring, the lane that holds columns -4 .. -1 (lm1) takes columns 4, 3, 2, 1 from its two right
// neighbours and the lane that holds w .. w+3 (rp1) takes w-2 .. w-5 from its two left neighbours; every dilated tap (|offset| <= 4)
// of an image column then finds the reflected value through the ordinary wave shift.  XE: bit 0 = the strip contains column 0,
// bit 1 = it contains column w-1.  10 vector instructions per side and ring row against a select on every shifted tap before
// (S = 3: 307 -> ~250 instructions per row in the edge strips, a third of the class's pixels).
template <int XE, typename V4>
__device__ __forceinline__ V4 hs_patch(const V4 r, const bool lm1, const bool rp1)
{
    V4 o = r;
    if constexpr ((XE & 1) != 0) {
        const auto c1 = wave_shl1(r.y), c2 = wave_shl1(r.z), c3 = wave_shl1(r.w);      // the right neighbour's columns 1, 2, 3
        const auto c4 = wave_shl1(wave_shl1(r.x));                                     // column 4: two lanes to the right
        o.x = lm1 ? c4 : o.x; o.y = lm1 ? c3 : o.y; o.z = lm1 ? c2 : o.z; o.w = lm1 ? c1 : o.w;
    }
    if constexpr ((XE & 2) != 0) {
        const auto m2 = wave_shr1(r.z), m3 = wave_shr1(r.y), m4 = wave_shr1(r.x);      // the left neighbour's columns w-2, w-3, w-4
        const auto m5 = wave_shr1(wave_shr1(r.w));                                     // column w-5: two lanes to the left
        o.x = rp1 ? m2 : o.x; o.y = rp1 ? m3 : o.y; o.z = rp1 ? m4 : o.z; o.w = rp1 ? m5 : o.w;
    }
    return o;
}

__device__ __forceinline__ float hs_max3(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }
__device__ __forceinline__ int hs_max3(int a, int b, int c) { return max(max(a, b), c); }

// dilated Scharr pair on one component (akazed.cu:1294-1295)
#define HS_DX(ul, ur, cl, cr, ll, lr) hs_d(fac1, fac2, (ur) + (lr) - (ul) - (ll), (cr) - (cl))
#define HS_DY(ul, uc, ur, ll, lc, lr) hs_d(fac1, fac2, (lr) + (ll) - (ur) - (ul), (lc) - (uc))

template <typename V> struct HsArgs {
    const V* src; V* dxy; V* det;
    SfTaps<V> kk;                                   // LP: the sigma=1 taps (float) / their 16.16 values (int)
    V* obase; unsigned off_dxy, off_det;            // dxy / det as byte offsets from the lower of the two (buffer stores); off_det =
                                                    // HAK_BUF_OOB: the determinant is not stored (the hardware drops the store)
    int w, h, p;
    V fac1, fac2;
    // extrema (maps == nullptr: determinant only)
    unsigned long long* maps; unsigned long long* cand; long cand_cap; HakImgState* st;
    int p0, octave, layer, psz; float border; V threshold;
};

// Arguments that only the (rare) candidate emission needs.  As kernel arguments they would sit in SGPRs for the whole row
// loop -- the loop is short of scalar registers (10 v_readlane spill reloads per row at S = 3) -- so the block parks them in
// LDS once and the emission path reads them back.
struct HsCold { unsigned long long* maps; unsigned long long* cand; long cand_cap; HakImgState* st; int p0, octave, layer, pad; };
// Pointers that come back from LDS are generic to the compiler: it would use FLAT atomics / stores for them, and a FLAT
// operation in the row loop makes every later wait a full `vmcnt(0) lgkmcnt(0)` drain (measured: Hessian +12 %).  The casts
// below tell it the truth -- these are global addresses.
typedef unsigned long long __attribute__((address_space(1))) * hs_gu64p;
typedef int __attribute__((address_space(1))) * hs_gi32p;

// Candidate emission.  Reserving list slots needs an atomic WITH return, and waiting for it drains every outstanding
// store and prefetch of the wave (s_waitcnt vmcnt(0)) -- with a candidate in roughly every third row that stalled the
// stream for a full memory round trip again and again.  Candidates are therefore staged in a 256-entry per-wave LDS
// buffer and flushed with ONE slot reservation when it runs full (and at the end of the segment); the key-map update is a
// return-less atomic and stays inline.

__device__ __forceinline__ void hs_flush(HsCand& cb, const HsCold* cold, const int lane)
{
    if (cb.n > 0) {
        const HsCold a = *cold;
        int base = 0;
        if (lane == 0) base = __hip_atomic_fetch_add((hs_gi32p)&a.st->ncand, cb.n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        base = __builtin_amdgcn_readfirstlane(base);
        const hs_gu64p gcand = (hs_gu64p)a.cand;
        for (int i = lane; i < cb.n; i += 64) {
            const long slot = (long)base + i;
            if (slot < a.cand_cap) gcand[slot] = cb.buf[i];
        }
        cb.n = 0;
    }
}

template <typename V>
__device__ __forceinline__ void hs_emit(const bool hit, const V v, const int x, const int e, const HsCold& a, const int lane,
                                        HsCand& cb)
{
    const unsigned long long m = __ballot(hit);
    if (m) {
        const int cnt = __popcll(m);
        if (hit) {
            const int fx = x << a.octave, fy = e << a.octave;
            const unsigned long long key = ((unsigned long long)hs_key_bits(v) << 32) | (0xFFFFFFFFu - (unsigned)a.layer);
            (void)__hip_atomic_fetch_max((hs_gu64p)a.maps + ((long)fy * a.p0 + fx), key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            cb.buf[cb.n + __popcll(m & ((1ull << lane) - 1ull))] =
                ((unsigned long long)a.layer << 32) | ((unsigned)fy << 16) | (unsigned)fx;
        }
        cb.n += cnt;
    }
}

template <typename V, int S, int U, int XE, bool YEDGE, bool LP>
__device__ __forceinline__ void hs_iter(HsState<V, S, LP>& T, const int t, const HsArgs<V>& a, const int xl, const int x0,
                                        const int ybeg, const int yend, const bool owns, const unsigned xok, const int lane,
                                        const __amdgpu_buffer_rsrc_t orsrc, const unsigned (&ovoff)[3], const int er0, const int er1,
                                        const HsCold* cold)
{
    using V4 = typename FedV<V>::V4;
    constexpr int R = HsGeo<S, LP>::R;
    constexpr int PD = HsGeo<S, LP>::PD;
    const V fac1 = a.fac1, fac2 = a.fac2;
    const int w = a.w, h = a.h, p = a.p;
    [[maybe_unused]] const bool le = x0 == 0, re = x0 + 3 == w - 1;      // (LP row pass only)
    const bool lm1 = x0 == -4, rp1 = x0 == w;                            // the lanes just outside the image: hs_patch
    // ---- input row t arrived in its ring slot (requested PD iterations ago); request row t + PD straight into the slot it
    // will occupy (the row that slot held, t - 2S - 1, is dead).  No register is ever copied while its load is in flight:
    // rotating a prefetch queue by moves made the compiler wait for vmcnt(0) every iteration.
    if constexpr (!LP) {
        T.A[pmod(U + PD, R)] = hak_load_stream(reinterpret_cast<const V4*>(a.src + (unsigned)(min(t + PD, h - 1) * p + xl)));
    } else {
        // ---- LP: raw row r = t + 2 arrived in W[(U+2) mod R]; request row r + PD into the slot it will be consumed from
        const int r = t + 2;
        T.W[pmod(U + 2 + PD, R)] = hak_load_stream(reinterpret_cast<const V4*>(a.src + (unsigned)(min(r + PD, h - 1) * p + xl)));
        const SfTaps<V> kk = a.kk;
        {   // row pass of the Gaussian on raw row r (akazed.cu:227-239; kernels_fedsf.hip fs_iter)
            const V4 c = T.W[pmod(U + 2, R)];
            const V sl1 = wave_shr1(c.w), sl2 = wave_shr1(c.z), sr1 = wave_shl1(c.x), sr2 = wave_shl1(c.y);
            V4 l1 = mk4(sl1, c.x, c.y, c.z), l2 = mk4(sl2, sl1, c.x, c.y);
            V4 r1 = mk4(c.y, c.z, c.w, sr1), r2 = mk4(c.z, c.w, sr1, sr2);
            if (XE) {
                l1.x = le ? c.y : l1.x;                         // column -1 -> 1
                l2.x = le ? c.z : l2.x;                         // column -2 -> 2
                l2.y = le ? c.y : l2.y;                         // column -1 -> 1
                r1.w = re ? c.z : r1.w;                         // column w   -> w-2
                r2.z = re ? c.z : r2.z;                         // column w   -> w-2
                r2.w = re ? c.y : r2.w;                         // column w+1 -> w-3
            }
            V4 rp;
            rp.x = sf_conv(c.x, l1.x, r1.x, l2.x, r2.x, kk);
            rp.y = sf_conv(c.y, l1.y, r1.y, l2.y, r2.y, kk);
            rp.z = sf_conv(c.z, l1.z, r1.z, l2.z, r2.z, kk);
            rp.w = sf_conv(c.w, l1.w, r1.w, l2.w, r2.w, kk);
            T.Rp[pmod(U + 2, R)] = rp;
            if (YEDGE) {                                        // (selects on values: see kernels_fedsf.hip)
                T.Rp[pmod(U, R)] = vsel4(r == 1, rp, T.Rp[pmod(U, R)]);                              // row -1 := row 1
                T.Rp[pmod(U - 2, R)] = vsel4(r == 2, rp, T.Rp[pmod(U - 2, R)]);                      // row -2 := row 2
                T.Rp[pmod(U + 2, R)] = vsel4(r == h, T.Rp[pmod(U, R)], T.Rp[pmod(U + 2, R)]);        // row h   := row h-2
                T.Rp[pmod(U + 2, R)] = vsel4(r == h + 1, T.Rp[pmod(U - 2, R)], T.Rp[pmod(U + 2, R)]);// row h+1 := row h-3
            }
        }
        {   // column pass -> smooth row t (akazed.cu:283-288)
            const V4 c = T.Rp[pmod(U, R)], u1 = T.Rp[pmod(U - 1, R)], d1 = T.Rp[pmod(U + 1, R)];
            const V4 u2 = T.Rp[pmod(U - 2, R)], d2 = T.Rp[pmod(U + 2, R)];
            V4 sm;
            sm.x = sf_conv(c.x, u1.x, d1.x, u2.x, d2.x, kk);
            sm.y = sf_conv(c.y, u1.y, d1.y, u2.y, d2.y, kk);
            sm.z = sf_conv(c.z, u1.z, d1.z, u2.z, d2.z, kk);
            sm.w = sf_conv(c.w, u1.w, d1.w, u2.w, d2.w, kk);
            T.A[pmod(U, R)] = sm;
        }
    }
    if constexpr (XE != 0) T.A[pmod(U, R)] = hs_patch<XE>(T.A[pmod(U, R)], lm1, rp1);
    if (YEDGE) {
#pragma unroll
        for (int j = 1; j <= S; j++) {
            if (t == j) T.A[pmod(U - 2 * j, R)] = T.A[pmod(U, R)];                   // row -j := row j
            if (t == h - 1 + j) T.A[pmod(U, R)] = T.A[pmod(U - 2 * j, R)];           // row h-1+j := row h-1-j
        }
    }
    // ---- Lx, Ly of row b = t - S
    V4 vy_row;                                                  // Ly row b (= the determinant's row c+S below)
    {
        const int b = t - S;
        const V4 ru = T.A[pmod(U - 2 * S, R)], rc = T.A[pmod(U - S, R)], rl = T.A[pmod(U, R)];
        V4 vx, vy;
#define HS_S1(k, K)                                                                                         \
        {                                                                                                   \
            const V ul = hs_l<S, K>(ru), ur = hs_r<S, K>(ru);                                               \
            const V cl = hs_l<S, K>(rc), cr = hs_r<S, K>(rc);                                               \
            const V ll = hs_l<S, K>(rl), lr = hs_r<S, K>(rl);                                               \
            vx.k = HS_DX(ul, ur, cl, cr, ll, lr);                                                           \
            vy.k = HS_DY(ul, ru.k, ur, ll, rl.k, lr);                                                       \
        }
        // scheduling fences keep one component's shifted values live at a time (the scheduler otherwise hoists every
        // DPP shift of the iteration to the top and needs > 250 VGPRs)
        HS_S1(x, 0) __builtin_amdgcn_sched_barrier(0); HS_S1(y, 1) __builtin_amdgcn_sched_barrier(0);
        HS_S1(z, 2) __builtin_amdgcn_sched_barrier(0); HS_S1(w, 3) __builtin_amdgcn_sched_barrier(0);
#undef HS_S1
        if constexpr (XE != 0) { vx = hs_patch<XE>(vx, lm1, rp1); vy = hs_patch<XE>(vy, lm1, rp1); }
        T.X[pmod(U, R)] = vx;
        T.Y[pmod(U, R) * 64 + lane] = vy;
        {
            // Interleaved plane: pixel x of row b lives at 2 * (b * p + x).  Storing a lane's own four pixels would be two
            // 16-byte pieces at a 32-byte lane stride -- every store instruction half-fills sixteen 128-byte lines (measured:
            // +0.7 ms per 256 images against dense stores).  The row is therefore turned through LDS: Lx goes to a staging
            // row (Ly sits in its ring slot already), lane j reads back the pixel PAIR 2j, 2j+1 of both (two conflict-free
            // ds_read_b64) and stores {Lx, Ly, Lx, Ly} -- one dense kilobyte per instruction; a second pair of reads serves
            // pixels 128 + 2j.  Same wave, LDS operations complete in order: no barrier.
            // Unconditional buffer stores; rows outside the segment and pixel pairs the strip does not own carry the
            // out-of-range bit.
            using V2 = typename HsV2<V>::T;
            __builtin_amdgcn_wave_barrier();                        // (no instruction: orders the cross-lane hand-over for the compiler)
            T.XS[lane] = vx;
            __builtin_amdgcn_wave_barrier();
            const V2* xs2 = reinterpret_cast<const V2*>(T.XS);
            const V2* ys2 = reinterpret_cast<const V2*>(T.Y + pmod(U, R) * 64);
            const V2 xa = xs2[lane], ya = ys2[lane], xb2 = xs2[64 + lane], yb2 = ys2[64 + lane];
            const unsigned roff = b >= ybeg && b < yend ? (unsigned)(b * p) * 2u * (unsigned)sizeof(V) : HAK_BUF_OOB;
            hak_buf_store_nt(orsrc, ovoff[0] + roff, mk4(xa.x, ya.x, xa.y, ya.y));
            hak_buf_store_nt(orsrc, ovoff[2] + roff, mk4(xb2.x, yb2.x, xb2.y, yb2.y));
        }
        if (YEDGE) {
#pragma unroll
            for (int j = 1; j <= S; j++) {
                if (b == j) {
                    T.X[pmod(U - 2 * j, R)] = T.X[pmod(U, R)];
                    T.Y[pmod(U - 2 * j, R) * 64 + lane] = vy;
                }
                if (b == h - 1 + j) {
                    T.X[pmod(U, R)] = T.X[pmod(U - 2 * j, R)];
                    vy = T.Y[pmod(U - 2 * j, R) * 64 + lane];
                    T.Y[pmod(U, R) * 64 + lane] = vy;
                }
            }
        }
        vy_row = vy;
    }
    // ---- determinant of row c = t - 2S
    {
        const int c = t - 2 * S;
        const V4 xu = T.X[pmod(U - 2 * S, R)], xc = T.X[pmod(U - S, R)], xd = T.X[pmod(U, R)];
        const V4 yu = T.Y[pmod(U - 2 * S, R) * 64 + lane], yd = vy_row;
        V4 d;
#define HS_DET(k, K)                                                                                        \
        {                                                                                                   \
            const V xul = hs_l<S, K>(xu), xur = hs_r<S, K>(xu);                                             \
            const V xcl = hs_l<S, K>(xc), xcr = hs_r<S, K>(xc);                                             \
            const V xll = hs_l<S, K>(xd), xlr = hs_r<S, K>(xd);                                             \
            const V yul = hs_l<S, K>(yu), yur = hs_r<S, K>(yu);                                             \
            const V yll = hs_l<S, K>(yd), ylr = hs_r<S, K>(yd);                                             \
            const V dxx = HS_DX(xul, xur, xcl, xcr, xll, xlr);                                          \
            const V dxy = HS_DY(xul, xu.k, xur, xll, xd.k, xlr);                                        \
            const V dyy = HS_DY(yul, yu.k, yur, yll, yd.k, ylr);                                        \
            d.k = hs_det(dxx, dyy, dxy);                                                                    \
        }
        HS_DET(x, 0) __builtin_amdgcn_sched_barrier(0); HS_DET(y, 1) __builtin_amdgcn_sched_barrier(0);
        HS_DET(z, 2) __builtin_amdgcn_sched_barrier(0); HS_DET(w, 3) __builtin_amdgcn_sched_barrier(0);
#undef HS_DET
        T.Dm = T.Dc; T.Dc = T.Dp; T.Dp = d;
        hak_buf_store_nt(orsrc, ovoff[1] + (c >= ybeg && c < yend ? (unsigned)(c * p) * (unsigned)sizeof(V) : HAK_BUF_OOB), d);
    }
    // ---- extrema of row e = t - 2S - 1 (akazed.cu:1346-1373)
    {
        const int e = t - 2 * S - 1;
        const V thr = a.threshold;
        const V4 v = T.Dc;
        // threshold first: almost no pixel passes it, so the wave almost always skips the neighbourhood test.  [er0, er1) = the
        // rows of this segment that pass the reference's border test (akazed.cu:1351-1356), worked out once per strip; empty
        // when the launch wants the determinant only
        const bool any = owns && (v.x > thr || v.y > thr || v.z > thr || v.w > thr);
        if (e >= er0 && e < er1 && __ballot(any) != 0ull) {
            const V4 up = T.Dm, dn = T.Dp;
            const V vl = wave_shr1(v.w), vr = wave_shl1(v.x);
            const V ul = wave_shr1(up.w), ur = wave_shl1(up.x);
            const V dl = wave_shr1(dn.w), dr = wave_shl1(dn.x);
            // strict maximum over the threshold and the 8 neighbours: v > max(all nine) -- four v_max3 and one compare per
            // pixel instead of nine compares and their scalar mask chain (ordered compares: a NaN neighbour can only appear
            // for non-finite input images)
            const V mx = hs_max3(hs_max3(thr, up.x, dn.x), hs_max3(vl, v.y, ul), hs_max3(up.y, dl, dn.y));
            const V my = hs_max3(hs_max3(thr, up.y, dn.y), hs_max3(v.x, v.z, up.x), hs_max3(up.z, dn.x, dn.z));
            const V mz = hs_max3(hs_max3(thr, up.z, dn.z), hs_max3(v.y, v.w, up.y), hs_max3(up.w, dn.y, dn.w));
            const V mw = hs_max3(hs_max3(thr, up.w, dn.w), hs_max3(v.z, vr, up.z), hs_max3(ur, dn.z, dr));
            const bool hx = owns && (xok & 1u) && v.x > mx;
            const bool hy = owns && (xok & 2u) && v.y > my;
            const bool hz = owns && (xok & 4u) && v.z > mz;
            const bool hw = owns && (xok & 8u) && v.w > mw;
            if (__ballot(hx || hy || hz || hw) != 0ull) {
                if (T.cb.n > HS_CBUF - 128) hs_flush(T.cb, cold, lane);    // a row holds at most 128 strict 3x3 maxima per wave
                const HsCold ca = *cold;                                   // one LDS read of the parked arguments per row with a hit
                hs_emit(hx, v.x, x0, e, ca, lane, T.cb);
                hs_emit(hy, v.y, x0 + 1, e, ca, lane, T.cb);
                hs_emit(hz, v.z, x0 + 2, e, ca, lane, T.cb);
                hs_emit(hw, v.w, x0 + 3, e, ca, lane, T.cb);
            }
        }
    }
}

template <typename V, int S, int XE, bool YEDGE, bool LP, int... U>
__device__ __forceinline__ void hs_group(std::integer_sequence<int, U...>, HsState<V, S, LP>& T, const int tb, const HsArgs<V>& a,
                                         const int xl, const int x0, const int ybeg, const int yend, const bool owns,
                                         const unsigned xok, const int lane, const __amdgpu_buffer_rsrc_t orsrc, const unsigned (&ovoff)[3],
                                         const int er0, const int er1, const HsCold* cold)
{
    (hs_iter<V, S, U, XE, YEDGE, LP>(T, tb + U, a, xl, x0, ybeg, yend, owns, xok, lane, orsrc, ovoff, er0, er1, cold), ...);
}

template <typename V, int S, int XE, bool LP>
__device__ __forceinline__ void hs_strip(const HsArgs<V>& a, const int x0, const int ybeg, const int yend, const bool owns,
                                         const int lane, typename FedV<V>::V4* yring, typename FedV<V>::V4* xstage, unsigned long long* cbuf,
                                         const HsCold* cold)
{
    using G = HsGeo<S, LP>;
    using V4 = typename FedV<V>::V4;
    const int h = a.h, w = a.w;
    const int xl = min(max(x0, 0), a.p - 4);                    // keep every lane's loads inside the plane
    // first input row of the derivative pipeline.  LP: its smooth row needs row-pass rows t0-2 .. t0+2, i.e. four more
    // iterations in front (raw row t+2 enters at iteration t; rows -2, -1 are written when rows 2, 1 arrive: from t = -2)
    const int t0 = LP ? max(max(0, ybeg - 1 - 2 * S) - 4, -2) : max(0, ybeg - 1 - 2 * S);
    const int tend = yend + 2 * S;                              // iteration that tests the segment's last row for extrema
    // per-component x range of the extrema test (akazed.cu:1351-1356), constant along the strip
    unsigned xok = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int x = x0 + k;
        if (x >= a.psz && (int)(x - a.border + 0.5f) - 1 >= 0 && (int)(x + a.border + 0.5f) + 1 < w) xok |= 1u << k;
    }
    // rows of this segment whose extrema are wanted: the reference's border test is monotone in the row, so it is a range
    int er0 = 0, er1 = 0;
    if (a.maps != nullptr) {
        auto row_ok = [&](int e) { return e >= a.psz && (int)(e - a.border + 0.5f) - 1 >= 0 && (int)(e + a.border + 0.5f) + 1 < h; };
        er0 = ybeg; er1 = yend;
        while (er0 < er1 && !row_ok(er0)) er0++;
        while (er1 > er0 && !row_ok(er1 - 1)) er1--;
    }
    const __amdgpu_buffer_rsrc_t orsrc = hak_buf_rsrc(a.obase);
    // per-plane lane offsets (column + plane); lanes that own nothing carry the out-of-range marker
    const unsigned xb = (unsigned)x0 * (unsigned)sizeof(V);
    // [0], [2]: the interleaved derivative stores of pixel pairs 2 * lane and 128 + 2 * lane of the strip (ownership is a
    // property of the PIXEL: margins and widths are multiples of 4, so both pixels of a pair share it); [1]: det (lane's own pixels)
    const int sx = x0 - 4 * lane;                               // image column of the strip's pixel 0
    auto pair_off = [&](int q) -> unsigned {
        const int x = sx + q;
        return q >= G::M && q < G::M + G::XV && x >= 0 && x < w ? (unsigned)x * 2u * (unsigned)sizeof(V) + a.off_dxy : HAK_BUF_OOB;
    };
    const unsigned ovoff[3] = {pair_off(2 * lane), owns && a.off_det != HAK_BUF_OOB ? xb + a.off_det : HAK_BUF_OOB, pair_off(128 + 2 * lane)};
    HsState<V, S, LP> T;
    T.Y = yring;
    T.XS = xstage;
    T.cb.buf = cbuf;
    T.cb.n = 0;
    const V zz = 0;
    const V4 z4 = mk4(zz, zz, zz, zz);
#pragma unroll
    for (int i = 0; i < G::R; i++) { T.A[i] = T.X[i] = z4; T.Y[i * 64 + lane] = z4; }
    T.Dm = T.Dc = T.Dp = z4;
    if constexpr (LP) {
#pragma unroll
        for (int i = 0; i < G::R; i++) T.W[i] = T.Rp[i] = z4;
#pragma unroll
        for (int i = 0; i < G::PD; i++)                         // raw rows t0+2 .. : consumed from slot (U+2) mod R at U = i
            T.W[pmod(2 + i, G::R)] = hak_load_stream(reinterpret_cast<const V4*>(a.src + (unsigned)(min(t0 + 2 + i, h - 1) * a.p + xl)));
    } else {
#pragma unroll
        for (int i = 0; i < G::PD; i++) T.A[i] = hak_load_stream(reinterpret_cast<const V4*>(a.src + (unsigned)(min(t0 + i, h - 1) * a.p + xl)));
    }
    for (int tb = t0; tb <= tend; tb += G::R) {
        // reflect injections fire while a ring is at rows 1..S (t <= 2S) or at the virtual rows past h-1 (LP: the raw row of
        // iteration t is t+2)
        if (tb <= 2 * S || tb + G::R - 1 >= h - (LP ? 2 : 0))
            hs_group<V, S, XE, true, LP>(std::make_integer_sequence<int, G::R>{}, T, tb, a, xl, x0, ybeg, yend, owns, xok, lane, orsrc, ovoff, er0, er1, cold);
        else
            hs_group<V, S, XE, false, LP>(std::make_integer_sequence<int, G::R>{}, T, tb, a, xl, x0, ybeg, yend, owns, xok, lane, orsrc, ovoff, er0, er1, cold);
    }
    if (a.maps != nullptr) hs_flush(T.cb, cold, lane);
}

// grid: hak_xcd_grid(strips, segment groups of 4, images); wave wv of a block takes segment by*4 + wv
template <typename V, int S, bool LP>
__global__ __launch_bounds__(256, (HsGeo<S, LP>::MINW)) void k_hessian_stream(HsArgs<V> a, long stride, long map_stride, int ry, int nbx, int nby, int nimg)
{
    using G = HsGeo<S, LP>;
    __shared__ typename FedV<V>::V4 yring[4 * G::R * 64];                     // per-wave private Ly rings: no barrier ever needed
    __shared__ typename FedV<V>::V4 xstage[4 * 64];                           // per-wave Lx staging row of the interleaved store
    __shared__ unsigned long long cbuf[4 * HS_CBUF];            // per-wave candidate staging
    __shared__ HsCold cold;
    int bx, by, img;
    if (!hak_xcd_decode(nbx, nby, nimg, bx, by, img)) return;
    a.src += (long)img * stride; a.obase += (long)img * stride;
    if (a.maps) {
        if (threadIdx.x == 0)
            cold = HsCold{a.maps + (long)img * map_stride, a.cand + (long)img * a.cand_cap, a.cand_cap, a.st + img, a.p0, a.octave, a.layer, 0};
        __syncthreads();                                        // (block-uniform; the only barrier of the kernel)
    }
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int ybeg = (by * 4 + wv) * ry;
    if (ybeg >= a.h) return;                                    // wave-uniform
    const int yend = min(ybeg + ry, a.h);
    const int x0 = bx * G::XV - G::M + 4 * lane;                // first pixel of this lane (may lie outside the image)
    const bool owns = 4 * lane >= G::M && 4 * lane < G::M + G::XV && x0 >= 0 && x0 < a.w;
    // only the strips that contain image column 0 or w-1 pay for the reflect patch, and only for their own side (block-uniform)
    const bool sl = bx == 0, sr = (bx + 1) * G::XV + G::M >= a.w;
#define HS_GO(XE) hs_strip<V, S, XE, LP>(a, x0, ybeg, yend, owns, lane, yring + wv * G::R * 64, xstage + wv * 64, cbuf + wv * HS_CBUF, &cold)
    if (sl && sr) HS_GO(3);
    else if (sl) HS_GO(1);
    else if (sr) HS_GO(2);
    else HS_GO(0);
#undef HS_GO
}

template <typename V, int S, bool LP>
void launch_stream(hipStream_t st, HsArgs<V> a, long stride, long map_stride, int nimg)
{
    using G = HsGeo<S, LP>;
    const int gx = (a.w + G::XV - 1) / G::XV;
    // rows per wave: tall segments amortise the 4S+2 warm-up rows; shrink while the grid cannot fill the chip
    const int ry = hak_stream_rows(a.h, (long)gx * nimg, 16);
    const int gy = (a.h + 4 * ry - 1) / (4 * ry);
    k_hessian_stream<V, S, LP><<<hak_xcd_grid(gx, gy, nimg), 256, 0, st>>>(a, stride, map_stride, ry, gx, gy, nimg);
}


template <typename V>
bool launch_stream_any(hipStream_t st, const V* src, V* dxy, V* det, bool store_det, long stride, int w, int h, int p, int nimg, int step,
                       V fac1, V fac2, const HakBatch* b, const HakLayout* L, const HakTables* htab, int octave, int sub, V threshold,
                       const SfTaps<V>* lp)
{
    if (!hak_hessian_stream_covers(w, h, step, lp != nullptr)) return false;
    HsArgs<V> a{};
    if (lp) a.kk = *lp;
    V* lo = dxy;
    if (store_det && det < lo) lo = det;
    {   // plane offset + plane size must stay below the out-of-range marker
        const long room = (long)HAK_BUF_OOB / (long)sizeof(V) - 2L * h * p;
        if ((dxy - lo) >= room || (store_det && (det - lo) >= room)) return false;
    }
    a.obase = lo; a.off_dxy = (unsigned)((dxy - lo) * sizeof(V));
    a.off_det = store_det ? (unsigned)((det - lo) * sizeof(V)) : HAK_BUF_OOB;
    a.src = src; a.dxy = dxy; a.det = det; a.w = w; a.h = h; a.p = p; a.fac1 = fac1; a.fac2 = fac2;
    long map_stride = 0;
    if (b) {
        const int layer = octave * L->ms + sub;
        a.maps = b->maps; map_stride = b->map_stride; a.cand = b->cand; a.cand_cap = b->cand_cap; a.st = b->state;
        a.p0 = L->oct[0].p; a.octave = octave; a.layer = layer;
        a.psz = (int)htab->borders[octave * L->ms]; a.border = htab->borders[layer]; a.threshold = threshold;
    }
    if (lp) {
        switch (step) {
        case 1: launch_stream<V, 1, true>(st, a, stride, map_stride, nimg); break;
        case 2: launch_stream<V, 2, true>(st, a, stride, map_stride, nimg); break;
        case 3: launch_stream<V, 3, true>(st, a, stride, map_stride, nimg); break;
        default: launch_stream<V, 4, true>(st, a, stride, map_stride, nimg); break;
        }
        return true;
    }
    switch (step) {
    case 1: launch_stream<V, 1, false>(st, a, stride, map_stride, nimg); break;
    case 2: launch_stream<V, 2, false>(st, a, stride, map_stride, nimg); break;
    case 3: launch_stream<V, 3, false>(st, a, stride, map_stride, nimg); break;
    default: launch_stream<V, 4, false>(st, a, stride, map_stride, nimg); break;
    }
    return true;
}

}   // namespace

// return false when this kernel does not cover the case (caller falls back to the LDS tile kernel)
bool hak_launch_hessian_stream(hipStream_t st, const float* src, float* dxy, float* det, bool store_det, long stride,
                               int w, int h, int p, int nimg, int step, float fac1, float fac2,
                               const HakBatch* b, const HakLayout* L, const HakTables* htab, int octave, int sub, float dthreshold,
                               const float* lp_taps)
{
    const SfTaps<float> kk = lp_taps ? SfTaps<float>{lp_taps[0], lp_taps[1], lp_taps[2]} : SfTaps<float>{};
    return launch_stream_any<float>(st, src, dxy, det, store_det, stride, w, h, p, nimg, step, fac1, fac2, b, L, htab, octave, sub, dthreshold,
                                    lp_taps ? &kk : nullptr);
}

bool hakf_launch_hessian_stream(hipStream_t st, const int* src, int* dxy, int* det, bool store_det, long stride,
                                int w, int h, int p, int nimg, int step, int fac1, int fac2,
                                const HakBatch* b, const HakLayout* L, const HakTables* htab, int octave, int sub, int idthreshold)
{
    return launch_stream_any<int>(st, src, dxy, det, store_det, stride, w, h, p, nimg, step, fac1, fac2, b, L, htab, octave, sub, idthreshold,
                                  nullptr);
}
