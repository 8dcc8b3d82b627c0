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
// valid window shrinking by one pixel per step.  One launch per sublevel: 94 -> 43 launches per image.
//
// Bit-exactness: every expression is the reference's (and the oracle's) own -- no regrouping: the separable Gaussian through
// sf_conv, the Scharr / conductivity of k_smooth_flow, and  step = tE + tW + tS + tN  then  fma(stepfac, step, L)  with the
// neighbours taken at reflect-101 INDICES (abs / borderAdd, akazed.cu:1251-1254), never from mirrored halo cells: a mirrored
// cell would add its N and S terms in the other order.  The raw tile alone is loaded through mirrored indices (on the
// source extents for an octave head, akazed.cu:466-494), which is what the Gaussian's taps read.
#include "fed_common.h"

#define LV_MAX_STEPS 36
#define LV_NT 1024                    // threads per block: four waves per SIMD hide the LDS latency of the per-pixel loops
#define LV_LDS_FLOATS 38400          // 150 KB of the CU's 160 KB: three planes of (T + 2 * (ns + 3))^2 elements

template <typename V> struct LvFacs { V f[LV_MAX_STEPS]; };

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

// HEAD: src = L(o-1, 0) of the previous octave (extents sw x sh, pitch sp); else src = L(o, s-1) of this octave.
// FIRST = false continues a cycle of more than LV_MAX_STEPS steps: L comes from src, the low-pass from `smooth`.
template <typename V, bool HEAD, bool FIRST>
__global__ __launch_bounds__(LV_NT) void k_level_tile(const V* __restrict__ src, V* __restrict__ smooth, V* __restrict__ dst,
                                                    long stride, int sw, int sh, int sp, int w, int h, int p,
                                                    SfTaps<V> t, int type, const HakImgState* __restrict__ state, int octave,
                                                    float fixed_ikc, LvFacs<V> fac, int ns, int T, int nbx, int nby, int nimg)
{
    extern __shared__ __align__(16) unsigned char lv_lds_raw[];
    int bx, by, img;
    if (!hak_xcd_decode(nbx, nby, nimg, bx, by, img)) return;
    const int H = ns;                                        // FED halo
    const int E = T + 2 * (H + 3);                           // plane edge (raw region: core +- (H + 3))
    V* A = reinterpret_cast<V*>(lv_lds_raw);                 // raw = L0, then L ping
    V* B = A + E * E;                                        // row pass, then g
    V* C = B + E * E;                                        // smooth, then L pong
    const V* s = src + (long)img * stride;
    V* osm = smooth + (long)img * stride;
    V* od = dst + (long)img * stride;
    const float ikc = state ? state[img].ikc[octave] : fixed_ikc;
    const int tid = threadIdx.x;
    const int X0 = bx * T - (H + 3), Y0 = by * T - (H + 3);  // image coordinates of plane element (0, 0)
    // idx / E by one multiply-high: M = ceil(2^32 / E) is exact for idx < 2^14 and E <= 128 (idx * (M * E - 2^32) < 2^32 / E)
    const unsigned ME = 0xFFFFFFFFu / (unsigned)E + 1u;

    // ---- raw plane (A): L0 with mirrored indices
    for (int idx = tid; idx < E * E; idx += LV_NT) {
        const int r = (int)__umulhi((unsigned)idx, ME), c = idx - r * E;
        V v;
        if (HEAD && FIRST) v = s[(long)lv_src_index(Y0 + r, sh) * sp + lv_src_index(X0 + c, sw)];
        else v = s[(long)hak_refl(Y0 + r, h) * p + hak_refl(X0 + c, w)];
        A[idx] = v;
    }
    if (FIRST) {
        hak_lds_barrier();
        // ---- row pass (akazed.cu:227-239 / 469-471) into B: all rows, columns 2 .. E-3
        for (int idx = tid; idx < E * E; idx += LV_NT) {
            const int r = (int)__umulhi((unsigned)idx, ME), c = idx - r * E;
            if (c >= 2 && c < E - 2) {
                const V* q = A + idx;
                B[idx] = sf_conv(q[0], q[-1], q[1], q[-2], q[2], t);
            }
        }
        hak_lds_barrier();
        // ---- column pass (akazed.cu:283-288 / 507-509) into C = smooth on core +- (H + 1); the core goes to HBM
        for (int idx = tid; idx < E * E; idx += LV_NT) {
            const int r = (int)__umulhi((unsigned)idx, ME), c = idx - r * E;
            if (r >= 2 && r < E - 2 && c >= 2 && c < E - 2) {
                const V* q = B + idx;
                const V ws = sf_conv(q[0], q[-E], q[E], q[-2 * E], q[2 * E], t);
                C[idx] = ws;
                const int x = X0 + c, y = Y0 + r;
                if (r >= H + 3 && r < H + 3 + T && c >= H + 3 && c < H + 3 + T && x < w && y < h) osm[(long)y * p + x] = ws;
            }
        }
    } else {
        // continuation: the low-pass of this sublevel was written by the first launch of the cycle
        for (int idx = tid; idx < E * E; idx += LV_NT) {
            const int r = (int)__umulhi((unsigned)idx, ME), c = idx - r * E;
            C[idx] = osm[(long)hak_refl(Y0 + r, h) * p + hak_refl(X0 + c, w)];
        }
    }
    hak_lds_barrier();
    // ---- conductivity (akazed.cu:1078-1106) into B on core +- H, in-image pixels only, neighbours at reflect-101 indices
    for (int idx = tid; idx < E * E; idx += LV_NT) {
        const int r = (int)__umulhi((unsigned)idx, ME), c = idx - r * E;
        const int x = X0 + c, y = Y0 + r;
        if (r >= 3 && r < E - 3 && c >= 3 && c < E - 3 && x >= 0 && x < w && y >= 0 && y < h) {
            const int cl_ = c + (x == 0 ? 1 : -1), cr_ = c + (x == w - 1 ? -1 : 1);       // abs(x - 1), borderAdd(x, 1, w)
            const int ru = (r + (y == 0 ? 1 : -1)) * E, rl = (r + (y == h - 1 ? -1 : 1)) * E, rc = r * E;
            const V ul = C[ru + cl_], uc = C[ru + c], ur = C[ru + cr_];
            const V cl = C[rc + cl_], cr = C[rc + cr_];
            const V ll = C[rl + cl_], lc = C[rl + c], lr = C[rl + cr_];
            const V dx = 10 * (cr - cl) + 3 * (ur + lr - ul - ll);
            const V dy = 10 * (lc - uc) + 3 * (ll + lr - ul - ur);
            B[idx] = sf_g_as<V>(lv_conductivity<V>(dx, dy, ikc, type));
        }
    }
    hak_lds_barrier();
    // ---- ns explicit steps (akazed.cu:1241-1264), ping-pong A <-> C; step k is valid on core +- (ns - k)
    V* cur = A;
    V* nxt = C;
    for (int k = 1; k <= ns; k++) {
        const int m = H + 3 - (ns - k);                      // first plane row / column of this step's window
        const int n = T + 2 * (ns - k);                      // window edge
        const V f = fac.f[k - 1];
        const unsigned MN = 0xFFFFFFFFu / (unsigned)n + 1u;
        for (int idx = tid; idx < n * n; idx += LV_NT) {
            const int rr = (int)__umulhi((unsigned)idx, MN), cc = idx - rr * n;
            const int r = m + rr, c = m + cc, x = X0 + c, y = Y0 + r;
            if (x < 0 || x >= w || y < 0 || y >= h) continue;
            const int ro = r * E;
            const int rn = (r + (y == 0 ? 1 : -1)) * E, rs = (r + (y == h - 1 ? -1 : 1)) * E;
            const int cw = c + (x == 0 ? 1 : -1), ce = c + (x == w - 1 ? -1 : 1);
            const V L = cur[ro + c], g = B[ro + c];
            const V tE = vmul(vadd(g, B[ro + ce]), vsub(cur[ro + ce], L));
            const V tW = vmul(vadd(g, B[ro + cw]), vsub(cur[ro + cw], L));
            const V tS = vmul(vadd(g, B[rs + c]), vsub(cur[rs + c], L));
            const V tN = vmul(vadd(g, B[rn + c]), vsub(cur[rn + c], L));
            nxt[ro + c] = vstep(f, vadd(vadd(vadd(tE, tW), tS), tN), L);                  // akazed.cu:1259-1263
        }
        hak_lds_barrier();
        V* tmp = cur; cur = nxt; nxt = tmp;
    }
    // ---- the core of the last step -> HBM
    const unsigned MT = 0xFFFFFFFFu / (unsigned)T + 1u;
    for (int idx = tid; idx < T * T; idx += LV_NT) {
        const int r = (int)__umulhi((unsigned)idx, MT), c = idx - r * T;
        const int x = bx * T + c, y = by * T + r;
        if (x < w && y < h) od[(long)y * p + x] = cur[(r + H + 3) * E + c + H + 3];
    }
}

template <typename V, bool HEAD, bool FIRST>
void launch_level(hipStream_t st, const V* src, V* smooth, V* dst, long stride, HakOct so, int w, int h, int p, int nimg,
                  SfTaps<V> t, int diffusivity, const HakImgState* state, int octave, float fixed_ikc, const float* tau, int ns)
{
    LvFacs<V> fac;
    for (int k = 0; k < LV_MAX_STEPS; k++) {
        const float tk = k < ns ? tau[k] : 0.f;
        if constexpr (std::is_same<V, float>::value) fac.f[k] = 0.5f * tk;              // akazed.cu:2515
        else fac.f[k] = (int)(0.5f * tk * 65536 + 0.5f);                                // akazed.cu:4235
    }
    // the largest tile (multiple of 8, at most 64) whose three planes fit the LDS budget
    int T = 64;
    while (T > 8 && 3L * (T + 2 * (ns + 3)) * (T + 2 * (ns + 3)) > LV_LDS_FLOATS) T -= 8;
    // ... but not so large that a small plane runs on a handful of CUs.  Smaller tiles mean more halo work in total (the planes are
    // (T + 2 ns + 6)^2): ~100 blocks keep a block short without multiplying the work of the octaves that run beside the critical chain
    while (T > 16 && (long)((w + T - 1) / T) * ((h + T - 1) / T) * nimg < 96) T -= 8;
    const int E = T + 2 * (ns + 3);
    const size_t lds = sizeof(V) * 3 * (size_t)E * E;
    static bool attr_done = false;                           // (per instantiation: each has its own static)
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_level_tile<V, HEAD, FIRST>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  LV_LDS_FLOATS * 4);
        attr_done = true;
    }
    const int nbx = (w + T - 1) / T, nby = (h + T - 1) / T;
    k_level_tile<V, HEAD, FIRST><<<hak_xcd_grid(nbx, nby, nimg), LV_NT, lds, st>>>(src, smooth, dst, stride, so.w, so.h, so.p, w, h, p, t, diffusivity,
                                                                                 state, octave, fixed_ikc, fac, ns, T, nbx, nby, nimg);
}

template <typename V>
int level_steps(hipStream_t st, const V* src, HakOct so, bool head, V* smooth, V* dst, V* tmp, long stride, HakOct dd, int nimg,
                const V* taps, int diffusivity, const float* tau, int n, const HakImgState* state, int octave, float fixed_ikc)
{
    const SfTaps<V> t{taps[0], taps[1], taps[2]};
    const int G = (n + LV_MAX_STEPS - 1) / LV_MAX_STEPS;     // launches of this cycle (1 for every BASELINE configuration but 4K octave 4)
    int done = 0;
    const V* cur = src;
    for (int g = 0; g < G; g++) {
        const int ns = hak_fed_group_size(n, G, g);
        V* out = ((G - g) % 2 == 1) ? dst : tmp;             // ping-pong so that the last launch lands in dst
        if (g == 0 && head) launch_level<V, true, true>(st, cur, smooth, out, stride, so, dd.w, dd.h, dd.p, nimg, t, diffusivity, state, octave, fixed_ikc, tau, ns);
        else if (g == 0) launch_level<V, false, true>(st, cur, smooth, out, stride, dd, dd.w, dd.h, dd.p, nimg, t, diffusivity, state, octave, fixed_ikc, tau, ns);
        else launch_level<V, false, false>(st, cur, smooth, out, stride, dd, dd.w, dd.h, dd.p, nimg, t, diffusivity, state, octave, fixed_ikc, tau + done, ns);
        done += ns;
        cur = out;
    }
    return G;
}

} // namespace

// One sublevel in ceil(n / 36) launches.  src: L(o, s-1), or (head) L(o-1, 0) with extents `so`.  dst receives L(o, s);
// `tmp` is a scratch plane of the octave (used only when the cycle needs more than one launch); `smooth` receives the
// sigma=1 low-pass (the Hessian's input).  Returns the number of launches.
int hak_launch_level_tile(hipStream_t st, const float* src, HakOct so, bool head, float* smooth, float* dst, float* tmp, long stride,
                          HakOct dd, int nimg, const float* taps, int diffusivity, const float* tau, int n,
                          const HakImgState* state, int octave, float fixed_ikc)
{
    return level_steps<float>(st, src, so, head, smooth, dst, tmp, stride, dd, nimg, taps, diffusivity, tau, n, state, octave, fixed_ikc);
}

int hakf_launch_level_tile(hipStream_t st, const int* src, HakOct so, bool head, int* smooth, int* dst, int* tmp, long stride,
                           HakOct dd, int nimg, const int* itaps, int diffusivity, const float* tau, int n,
                           const HakImgState* state, int octave)
{
    return level_steps<int>(st, src, so, head, smooth, dst, tmp, stride, dd, nimg, itaps, diffusivity, tau, n, state, octave, 0.f);
}
