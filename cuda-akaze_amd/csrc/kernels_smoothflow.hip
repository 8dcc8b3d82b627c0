// kernels_smoothflow.hip -- sigma=1 low-pass + conductivity in one pass over Lt(o, s-1).
//
//   hLowPass(oldnld -> smooth, var 1, ksz 5)   akaze.cpp:403,  akazed.cu:2336, 204  (gConv2d<2>)
//   hFlow(smooth -> flow)                      akaze.cpp:404,  akazed.cu:2487, 1068 (gFlowNaive)
//
// Persistent tile kernel (same skeleton as kernels_hessian.hip): a block walks a vertical run of
// 64x32 tiles; the next tile's input (+3 halo) is prefetched into registers while the current tile
// goes  raw -> row pass -> column pass (smooth, +1 halo) -> Scharr + conductivity  through LDS.
// HBM traffic: read Lt once (4 B/px), write smooth and g (8 B/px); the unfused pair moves 16 B/px.
//
// The raw tile is loaded with reflect-101 indices, after which every stage uses plain tile
// indexing: the Gaussian is symmetric and each tap pair is summed as one commutative add
// (akazed.cu:237, 286), so the smooth value computed at a mirrored position equals the smooth value
// at the reflected coordinate bit for bit -- which is exactly what gFlowNaive reads at the border
// (abs / borderAdd, akazed.cu:1078-1081).
#include "fed_common.h"

#define SF_TX 64
#define SF_TY 32
#define SF_RW (SF_TX + 6)            // raw tile   (halo 3)
#define SF_RH (SF_TY + 6)
#define SF_PW (SF_TX + 2)            // row-pass / smooth tile width (halo 1)
#define SF_SH (SF_TY + 2)            // smooth tile height (halo 1)
#define SF_NPF ((SF_RW * SF_RH + 255) / 256)

// Shared by both pipelines: V = float (akaze) and V = int (fastakaze, 16.16 fixed point: every pass of the separable
// Gaussian ends in >> 16 (akazed.cu:2922-2985) and the conductivity is stored as (int)(g * 65536 + 0.5f), akazed.cu:3444).
template <typename V>
__device__ __forceinline__ void sf_fetch(V (&pf)[SF_NPF], const V* __restrict__ s, int w, int h, int p,
                                         int x0, int y0, int tid)
{
#pragma unroll
    for (int i = 0; i < SF_NPF; i++) {
        const int idx = tid + 256 * i;
        if (idx < SF_RW * SF_RH) {
            const int r = idx / SF_RW, c = idx - r * SF_RW;
            pf[i] = s[(long)hak_refl(y0 - 3 + r, h) * p + hak_refl(x0 - 3 + c, w)];
        }
    }
}

template <typename V>
__global__ __launch_bounds__(256) void k_smooth_flow(const V* __restrict__ src, V* __restrict__ smooth,
                                                     V* __restrict__ flow, long stride, int w, int h, int p,
                                                     SfTaps<V> t, int type, const HakImgState* __restrict__ state,
                                                     int octave, float fixed_ikc, int tiles_per_block, int nbx, int nby, int nimg)
{
    __shared__ V raw[SF_RH * SF_RW];         // raw tile; reused for the smooth tile after the row pass
    __shared__ V rowp[SF_RH * SF_PW];
    int bx, by, img;
    if (!hak_xcd_decode(nbx, nby, nimg, bx, by, img)) return;
    const V* s = src + (long)img * stride;
    V* osm = smooth + (long)img * stride;
    V* og = flow + (long)img * stride;
    const float ikc = state ? state[img].ikc[octave] : fixed_ikc;
    const int tid = threadIdx.x;
    const int x0 = bx * SF_TX;
    const int ty0 = by * tiles_per_block;
    const int ty1 = min(ty0 + tiles_per_block, (h + SF_TY - 1) / SF_TY);
    V pf[SF_NPF];
    if (ty0 < ty1) sf_fetch<V>(pf, s, w, h, p, x0, ty0 * SF_TY, tid);
    for (int ty = ty0; ty < ty1; ty++) {
        const int y0 = ty * SF_TY;
        hak_lds_barrier();                                    // previous tile's readers are done
#pragma unroll
        for (int i = 0; i < SF_NPF; i++)
            if (tid + 256 * i < SF_RW * SF_RH) raw[tid + 256 * i] = pf[i];
        hak_lds_barrier();
        if (ty + 1 < ty1) sf_fetch<V>(pf, s, w, h, p, x0, y0 + SF_TY, tid);      // in flight during the compute below
        // ---- row pass (akazed.cu:227-239): rowp[r][c] <-> image column x0-1+c, raw column c+1+... (offset 2)
        for (int idx = tid; idx < SF_RH * SF_PW; idx += 256) {
            const int r = idx / SF_PW, c = idx - r * SF_PW;
            const V* q = raw + r * SF_RW + c + 2;           // raw column of image column x0-1+c is c+2
            rowp[idx] = sf_conv(q[0], q[-1], q[1], q[-2], q[2], t);
        }
        hak_lds_barrier();
        // ---- column pass (akazed.cu:283-288) -> smooth tile (halo 1) in LDS, centre -> HBM
        V* sm = raw;
        for (int idx = tid; idx < SF_SH * SF_PW; idx += 256) {
            const int r = idx / SF_PW, c = idx - r * SF_PW;
            const V* q = rowp + (r + 2) * SF_PW + c;        // rowp row of image row y0-1+r is r+2
            const V ws = sf_conv(q[0], q[-SF_PW], q[SF_PW], q[-2 * SF_PW], q[2 * SF_PW], t);
            sm[idx] = ws;
            const int x = x0 - 1 + c, y = y0 - 1 + r;
            if (c >= 1 && c <= SF_TX && r >= 1 && r <= SF_TY && x < w && y < h) osm[(long)y * p + x] = ws;
        }
        hak_lds_barrier();
        // ---- Scharr + conductivity (akazed.cu:1088-1106) on the output tile
        for (int idx = tid; idx < SF_TY * SF_TX; idx += 256) {
            const int r = idx >> 6, c = idx & 63;
            const int x = x0 + c, y = y0 + r;
            if (x >= w || y >= h) continue;
            const V* q = sm + (r + 1) * SF_PW + c + 1;
            const V ul = q[-SF_PW - 1], uc = q[-SF_PW], ur = q[-SF_PW + 1];
            const V cl = q[-1], cr = q[1];
            const V ll = q[SF_PW - 1], lc = q[SF_PW], lr = q[SF_PW + 1];
            const V dx = 10 * (cr - cl) + 3 * (ur + lr - ul - ll);
            const V dy = 10 * (lc - uc) + 3 * (ll + lr - ul - ur);
            const float dif2 = sf_dif2(dx, dy, ikc);
            float g;
            if (type == HAK_PM_G2) g = 1.f / (1.f + dif2);
            else if (type == HAK_PM_G1) g = hak_expf(-dif2);
            else if (type == HAK_WEICKERT) {
                float d2 = dif2 * dif2;
                g = 1.f - hak_expf(-3.315f / (d2 * d2));
            } else g = 1.f / sqrtf(1.f + dif2);
            sf_store_g(og + (long)y * p + x, g);
        }
    }
}

template <typename V>
static void launch_sf(hipStream_t st, const V* src, V* smooth, V* flow, long stride, int w, int h, int p, int nimg,
                      SfTaps<V> t, int diffusivity, const HakImgState* state, int octave, float fixed_ikc)
{
    const int ntx = (w + SF_TX - 1) / SF_TX, nty = (h + SF_TY - 1) / SF_TY;
    int tpb = 8;
    while (tpb > 1 && (long)ntx * ((nty + tpb - 1) / tpb) * nimg < 4096) tpb >>= 1;
    const int nby = (nty + tpb - 1) / tpb;
    k_smooth_flow<V><<<hak_xcd_grid(ntx, nby, nimg), 256, 0, st>>>(src, smooth, flow, stride, w, h, p, t, diffusivity, state, octave,
                                                                   fixed_ikc, tpb, ntx, nby, nimg);
}

void hak_launch_smooth_flow(hipStream_t st, const float* src, float* smooth, float* flow, long stride,
                            int w, int h, int p, int nimg, const float* taps, int diffusivity,
                            const HakImgState* state, int octave, float fixed_ikc)
{
    launch_sf<float>(st, src, smooth, flow, stride, w, h, p, nimg, SfTaps<float>{taps[0], taps[1], taps[2]}, diffusivity, state, octave, fixed_ikc);
}

// integer FAST path: hLowPass(int, var 1) + hFlow (akaze.cpp:664-680) in one pass; itaps = (int)(tap * 65536 + 0.5f)
void hakf_launch_smooth_flow(hipStream_t st, const int* src, int* smooth, int* flow, long stride,
                             int w, int h, int p, int nimg, const int* itaps, int diffusivity,
                             const HakImgState* state, int octave)
{
    launch_sf<int>(st, src, smooth, flow, stride, w, h, p, nimg, SfTaps<int>{itaps[0], itaps[1], itaps[2]}, diffusivity, state, octave, 0.f);
}
