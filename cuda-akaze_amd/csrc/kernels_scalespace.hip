// kernels_scalespace.hip -- nonlinear scale space for gfx950 (CDNA4, wave64).
//
// Stages (reference wrapper -> kernel here):
//   hLowPass/gConv2d<R>            akazed.cu:2336, 204   -> k_lowpass<R>
//   hDownWithSmooth                akazed.cu:2389, 449   -> k_down_smooth
//   hScharrContrast                akazed.cu:2410, 644, 827, 901 -> k_grad_max, k_grad_hist, k_kcontrast
//   hFlow/gFlowNaive               akazed.cu:2487, 1068  -> k_flow
//   hNldStep/gNldStepNaive         akazed.cu:2509, 1241  -> kernels_fed.hip (the FED hot loop)
//   hHessianDeterminant            akazed.cu:2531, 1267, 1299 -> k_derivate, k_hessian
//
// All kernels take the batch image in blockIdx.z.  Float evaluation order is
// the reference's source order (no contraction: built with -ffp-contract=off);
// the only fused op is the explicit fmaf of the FED step (akazed.cu:1263).
#include "hak_internal.h"

#define TILE_X 64
#define TILE_Y 16

struct HakTaps { float k[8]; };

// ------------------------------------------------------------------ lowpass
// Separable Gaussian, reflect-101.  One 64x16 output tile per 256-thread
// block: raw tile (+R halo) -> LDS, row pass -> LDS, column pass -> HBM.
template <int R>
__global__ __launch_bounds__(256) void k_lowpass(const float* __restrict__ src, long src_stride, int sp,
                                                 float* __restrict__ dst, long dst_stride,
                                                 int w, int h, int p, HakTaps taps)
{
    constexpr int RW = TILE_X + 2 * R, RH = TILE_Y + 2 * R;
    __shared__ float raw[RH][RW + 1];
    __shared__ float rowp[RH][TILE_X];
    const float* s = src + (long)blockIdx.z * src_stride;
    float* d = dst + (long)blockIdx.z * dst_stride;
    const int x0 = blockIdx.x * TILE_X, y0 = blockIdx.y * TILE_Y, tid = threadIdx.x;

    for (int i = tid; i < RH * RW; i += 256) {
        int r = i / RW, c = i - r * RW;
        raw[r][c] = s[(long)hak_refl(y0 - R + r, h) * sp + hak_refl(x0 - R + c, w)];
    }
    __syncthreads();
    for (int i = tid; i < RH * TILE_X; i += 256) {
        int r = i >> 6, c = i & 63;
        float ws = raw[r][c + R] * taps.k[0];
#pragma unroll
        for (int k = 1; k <= R; k++) ws += taps.k[k] * (raw[r][c + R - k] + raw[r][c + R + k]);
        rowp[r][c] = ws;
    }
    __syncthreads();
    const int c = tid & 63, x = x0 + c;
    if (x >= w) return;
    for (int rr = tid >> 6; rr < TILE_Y; rr += 4) {
        int y = y0 + rr;
        if (y >= h) break;
        float ws = rowp[rr + R][c] * taps.k[0];
#pragma unroll
        for (int k = 1; k <= R; k++) ws += taps.k[k] * (rowp[rr + R - k][c] + rowp[rr + R + k][c]);
        d[(long)y * p + x] = ws;
    }
}

void hak_launch_lowpass(hipStream_t st, const float* src, long src_stride, int src_pitch, float* dst, long dst_stride,
                        int w, int h, int p, int nimg, const float* taps, int R)
{
    HakTaps t;
    for (int i = 0; i < 8; i++) t.k[i] = i <= R ? taps[i] : 0.f;
    dim3 grid((w + TILE_X - 1) / TILE_X, (h + TILE_Y - 1) / TILE_Y, nimg);
    switch (R) {
    case 2: k_lowpass<2><<<grid, 256, 0, st>>>(src, src_stride, src_pitch, dst, dst_stride, w, h, p, t); break;
    case 3: k_lowpass<3><<<grid, 256, 0, st>>>(src, src_stride, src_pitch, dst, dst_stride, w, h, p, t); break;
    case 4: k_lowpass<4><<<grid, 256, 0, st>>>(src, src_stride, src_pitch, dst, dst_stride, w, h, p, t); break;
    default: k_lowpass<5><<<grid, 256, 0, st>>>(src, src_stride, src_pitch, dst, dst_stride, w, h, p, t); break;
    }
}

// ------------------------------------------------------------ down + smooth
// dst = src[2y][2x]; smooth = G(sigma=1, R=2) on the decimated lattice with the
// mirror taken on the SOURCE extents (akazed.cu:466, 477-494).
__device__ __forceinline__ int refl_src(int i, int m)
{
    i = i < 0 ? -i : i;
    i = i < m ? i : m + m - 2 - i;
    i = i < 0 ? 0 : i;
    return i < m ? i : m - 1;
}

__global__ __launch_bounds__(256) void k_down_smooth(const float* __restrict__ src, float* __restrict__ dst,
                                                     float* __restrict__ smooth, long stride,
                                                     HakOct so, HakOct dd, HakTaps taps)
{
    constexpr int RW = TILE_X + 4, RH = TILE_Y + 4;
    __shared__ float dec[RH][RW + 1];
    __shared__ float rowp[RH][TILE_X];
    const float* s = src + (long)blockIdx.z * stride;
    float* d = dst + (long)blockIdx.z * stride;
    float* sm = smooth + (long)blockIdx.z * stride;
    const int x0 = blockIdx.x * TILE_X, y0 = blockIdx.y * TILE_Y, tid = threadIdx.x;

    for (int i = tid; i < RH * RW; i += 256) {
        int r = i / RW, c = i - r * RW;
        int sy = refl_src(2 * (y0 - 2 + r), so.h), sx = refl_src(2 * (x0 - 2 + c), so.w);
        dec[r][c] = s[(long)sy * so.p + sx];
    }
    __syncthreads();
    for (int i = tid; i < RH * TILE_X; i += 256) {
        int r = i >> 6, c = i & 63;
        rowp[r][c] = taps.k[0] * dec[r][c + 2] + taps.k[1] * (dec[r][c + 1] + dec[r][c + 3]) +
                     taps.k[2] * (dec[r][c] + dec[r][c + 4]);
    }
    __syncthreads();
    const int c = tid & 63, x = x0 + c;
    if (x >= dd.w) return;
    for (int rr = tid >> 6; rr < TILE_Y; rr += 4) {
        int y = y0 + rr;
        if (y >= dd.h) break;
        long o = (long)y * dd.p + x;
        d[o] = dec[rr + 2][c + 2];
        sm[o] = taps.k[0] * rowp[rr + 2][c] + taps.k[1] * (rowp[rr + 1][c] + rowp[rr + 3][c]) +
                taps.k[2] * (rowp[rr][c] + rowp[rr + 4][c]);
    }
}

void hak_launch_down_smooth(hipStream_t st, const float* src, float* dst, float* smooth, long stride,
                            HakOct so, HakOct dd, int nimg, const float* taps)
{
    HakTaps t;
    for (int i = 0; i < 8; i++) t.k[i] = i <= 2 ? taps[i] : 0.f;
    dim3 grid((dd.w + TILE_X - 1) / TILE_X, (dd.h + TILE_Y - 1) / TILE_Y, nimg);
    k_down_smooth<<<grid, 256, 0, st>>>(src, dst, smooth, stride, so, dd, t);
}

// ------------------------------------------------------- Scharr + contrast
// un-normalised Scharr pair of akazed.cu:664-665 / 1088-1089
__device__ __forceinline__ void scharr_dxdy(const float* __restrict__ s, int x, int y, int w, int h, int p,
                                            float& dx, float& dy)
{
    int x0 = x - 1 < 0 ? 1 - x : x - 1;
    int x2 = x + 1 < w ? x + 1 : w + w - 3 - x;
    int y0 = y - 1 < 0 ? 1 - y : y - 1;
    int y2 = y + 1 < h ? y + 1 : h + h - 3 - y;
    const float* r0 = s + (long)y0 * p;
    const float* r1 = s + (long)y * p;
    const float* r2 = s + (long)y2 * p;
    float ul = r0[x0], uc = r0[x], ur = r0[x2];
    float cl = r1[x0], cr = r1[x2];
    float ll = r2[x0], lc = r2[x], lr = r2[x2];
    dx = 10 * (cr - cl) + 3 * (ur + lr - ul - ll);
    dy = 10 * (lc - uc) + 3 * (ll + lr - ul - ur);
}

__global__ __launch_bounds__(256) void k_reset_state(HakImgState* state)
{
    HakImgState* st = state + blockIdx.x;
    for (int i = threadIdx.x; i < HAK_NBINS; i += 256) st->hist[i] = 0;
    if (threadIdx.x == 0) {
        st->hmax_bits = __float_as_uint(0.03f);        // akazed.cu:2413
        st->ncand = 0;
        st->total_pts = 0;
        st->num_pts = 0;
        st->hist_done = 0;
    }
}

void hak_launch_reset_state(hipStream_t st, HakImgState* state, int nimg)
{
    k_reset_state<<<nimg, 256, 0, st>>>(state);
}

// pass 1: maximum gradient magnitude over the 16-px lattice (what gFindMaxContrastU4 delivers, hak_internal.h)
__global__ __launch_bounds__(256) void k_grad_max(const float* __restrict__ smooth, long stride, int w, int h, int p,
                                                  HakImgState* state)
{
    const float* s = smooth + (long)blockIdx.z * stride;
    const int x = blockIdx.x * TILE_X + (threadIdx.x & 63);
    const int y0 = blockIdx.y * TILE_Y + (threadIdx.x >> 6);
    float m = 0.f;
    if (x < w)
        for (int y = y0; y < blockIdx.y * TILE_Y + TILE_Y && y < h; y += 4) {
            if (!hak_on_lattice(x, y, w, h)) continue;
            float dx, dy;
            scharr_dxdy(s, x, y, w, h, p, dx, dy);
            m = fmaxf(m, sqrtf(dx * dx + dy * dy));
        }
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0 && m > 0.f) atomicMax(&state[blockIdx.z].hmax_bits, __float_as_uint(m));
}

// pass 2: 300-bin histogram, gradient recomputed (cheaper than storing it)
__global__ __launch_bounds__(256) void k_grad_hist(const float* __restrict__ smooth, long stride, int w, int h, int p,
                                                   HakImgState* state)
{
    __shared__ int shist[HAK_NBINS];
    for (int i = threadIdx.x; i < HAK_NBINS; i += 256) shist[i] = 0;
    __syncthreads();
    const float* s = smooth + (long)blockIdx.z * stride;
    const float hmax = __uint_as_float(state[blockIdx.z].hmax_bits);
    const float hfactor = HAK_NBINS / hmax;                       // akazed.cu:2450
    const int x = blockIdx.x * TILE_X + (threadIdx.x & 63);
    const int y0 = blockIdx.y * TILE_Y + (threadIdx.x >> 6);
    if (x < w)
        for (int y = y0; y < blockIdx.y * TILE_Y + TILE_Y && y < h; y += 4) {
            float dx, dy;
            scharr_dxdy(s, x, y, w, h, p, dx, dy);
            float g = sqrtf(dx * dx + dy * dy);
            // (int)__fmul_rz(g, factor): exact double product, truncated (akazed.cu:924)
            int hi = (int)((double)g * (double)hfactor);
            hi = hi >= HAK_NBINS ? HAK_NBINS - 1 : hi;
            atomicAdd(&shist[hi], 1);
        }
    __syncthreads();
    for (int i = threadIdx.x; i < HAK_NBINS; i += 256)
        if (shist[i]) atomicAdd(&state[blockIdx.z].hist[i], shist[i]);
}

// host half of hScharrContrast (akazed.cu:2467-2481) + the per-octave 0.75 decay
// (akaze.cpp:371) and ikc = 1/(k*k) (akazed.cu:2493), kept on the device.
__global__ void k_kcontrast(HakImgState* state, int npix, int extra0, float per, int noct)
{
    HakImgState* st = state + blockIdx.x;
    if (threadIdx.x != 0) return;
    const float hmax = __uint_as_float(st->hmax_bits);
    const float hfactor = HAK_NBINS / hmax;
    int thresh = (int)((npix - (st->hist[0] + extra0)) * per);
    int cumuv = 0, k = 1;
    while (k < HAK_NBINS) {
        if (cumuv >= thresh) break;
        cumuv += st->hist[k];
        k++;
    }
    float kc = k / hfactor;
    for (int o = 0; o < noct; o++) {
        if (o > 0) kc *= 0.75f;
        st->kcontrast[o] = kc;
        st->ikc[o] = 1.f / (kc * kc);
    }
}

void hak_launch_contrast(hipStream_t st, const float* smooth, long stride, int w, int h, int p, int nimg,
                         HakImgState* state, float per, int noct)
{
    dim3 grid((w + TILE_X - 1) / TILE_X, (h + TILE_Y - 1) / TILE_Y, nimg);
    k_grad_max<<<grid, 256, 0, st>>>(smooth, stride, w, h, p, state);
    k_grad_hist<<<grid, 256, 0, st>>>(smooth, stride, w, h, p, state);
    k_kcontrast<<<nimg, 64, 0, st>>>(state, w * h, hak_hist_extra0(w, h), per, noct);
}

// --------------------------------------------------------------------- flow
__global__ __launch_bounds__(256) void k_flow(const float* __restrict__ src, float* __restrict__ dst, long stride,
                                              int w, int h, int p, int type, const HakImgState* state, int octave,
                                              float fixed_ikc)
{
    const float* s = src + (long)blockIdx.z * stride;
    float* d = dst + (long)blockIdx.z * stride;
    const float ikc = state ? state[blockIdx.z].ikc[octave] : fixed_ikc;
    const int x = blockIdx.x * TILE_X + (threadIdx.x & 63);
    const int y0 = blockIdx.y * TILE_Y + (threadIdx.x >> 6);
    if (x >= w) return;
    for (int y = y0; y < blockIdx.y * TILE_Y + TILE_Y && y < h; y += 4) {
        float dx, dy;
        scharr_dxdy(s, x, y, w, h, p, dx, dy);
        float dif2 = ikc * (dx * dx + dy * dy);
        float g;
        if (type == HAK_PM_G2) g = 1.f / (1.f + dif2);
        else if (type == HAK_PM_G1) g = hak_expf(-dif2);
        else if (type == HAK_WEICKERT) {
            float d2 = dif2 * dif2;
            g = 1.f - hak_expf(-3.315f / (d2 * d2));
        } else g = 1.f / sqrtf(1.f + dif2);
        d[(long)y * p + x] = g;
    }
}

void hak_launch_flow(hipStream_t st, const float* src, float* dst, long stride, int w, int h, int p, int nimg,
                     int diffusivity, const HakImgState* state, int octave, float fixed_ikc)
{
    dim3 grid((w + TILE_X - 1) / TILE_X, (h + TILE_Y - 1) / TILE_Y, nimg);
    k_flow<<<grid, 256, 0, st>>>(src, dst, stride, w, h, p, diffusivity, state, octave, fixed_ikc);
}

// ------------------------------------------------- derivatives + determinant
// unfused fallback for dilations > 4 and the debug / test planes: dxy = interleaved {Lx, Ly} (HakLayout)
__global__ __launch_bounds__(256) void k_derivate(const float* __restrict__ src, float* __restrict__ dxy, long stride, int w, int h, int p,
                                                  int step, float fac1, float fac2)
{
    const float* s = src + (long)blockIdx.z * stride;
    float2* o = reinterpret_cast<float2*>(dxy + (long)blockIdx.z * stride);
    const int x = blockIdx.x * TILE_X + (threadIdx.x & 63);
    const int y0 = blockIdx.y * TILE_Y + (threadIdx.x >> 6);
    if (x >= w) return;
    const int x0 = hak_refl(x - step, w), x2 = hak_refl(x + step, w);
    for (int y = y0; y < blockIdx.y * TILE_Y + TILE_Y && y < h; y += 4) {
        const float* r0 = s + (long)hak_refl(y - step, h) * p;
        const float* r1 = s + (long)y * p;
        const float* r2 = s + (long)hak_refl(y + step, h) * p;
        float ul = r0[x0], uc = r0[x], ur = r0[x2];
        float cl = r1[x0], cr = r1[x2];
        float ll = r2[x0], lc = r2[x], lr = r2[x2];
        o[(long)y * p + x] = make_float2(fac1 * (ur + lr - ul - ll) + fac2 * (cr - cl),        // akazed.cu:1294
                                         fac1 * (lr + ll - ur - ul) + fac2 * (lc - uc));       // akazed.cu:1295
    }
}

__global__ __launch_bounds__(256) void k_hessian(const float* __restrict__ dxy, float* __restrict__ det, long stride, int w, int h, int p,
                                                 int step, float fac1, float fac2)
{
    const float* d = dxy + (long)blockIdx.z * stride;
    float* o = det + (long)blockIdx.z * stride;
    const int x = blockIdx.x * TILE_X + (threadIdx.x & 63);
    const int y0 = blockIdx.y * TILE_Y + (threadIdx.x >> 6);
    if (x >= w) return;
    for (int y = y0; y < blockIdx.y * TILE_Y + TILE_Y && y < h; y += 4)
        o[(long)y * p + x] = hak_det_at<float>(d, x, y, step, w, h, p, fac1, fac2);             // akazed.cu:1318-1330
}

void hak_deriv_factors(float* fac1, float* fac2)
{
    float wv = 10.f / 3.f;                                   // akazed.cu:2537-2539
    *fac1 = 1.f / (2.f * (wv + 2.f));
    *fac2 = wv * *fac1;
}

void hak_launch_derivate(hipStream_t st, const float* src, float* dxy, long stride,
                         int w, int h, int p, int nimg, int step)
{
    float f1, f2;
    hak_deriv_factors(&f1, &f2);
    dim3 grid((w + TILE_X - 1) / TILE_X, (h + TILE_Y - 1) / TILE_Y, nimg);
    k_derivate<<<grid, 256, 0, st>>>(src, dxy, stride, w, h, p, step, f1, f2);
}

void hak_launch_hessian(hipStream_t st, const float* dxy, float* det, long stride,
                        int w, int h, int p, int nimg, int step)
{
    float f1, f2;
    hak_deriv_factors(&f1, &f2);
    dim3 grid((w + TILE_X - 1) / TILE_X, (h + TILE_Y - 1) / TILE_Y, nimg);
    k_hessian<<<grid, 256, 0, st>>>(dxy, det, stride, w, h, p, step, f1, f2);
}

// interleaved {a, b} plane (pitch 2p) -> two dense planes (pitch p): stage tests and hak_debug_plane only
__global__ __launch_bounds__(256) void k_deinterleave(const float2* __restrict__ ab, float* __restrict__ a, float* __restrict__ b,
                                                      int w, int h, int p)
{
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= w) return;
    const float2 v = ab[(long)y * p + x];
    a[(long)y * p + x] = v.x;
    b[(long)y * p + x] = v.y;
}
void hak_launch_deinterleave(hipStream_t st, const float* ab, float* a, float* b, int w, int h, int p)
{
    k_deinterleave<<<dim3((w + 255) / 256, h), 256, 0, st>>>(reinterpret_cast<const float2*>(ab), a, b, w, h, p);
}

// ------------------------------------------------------------------ ingest
// uint8 -> float32 [0,1] as main.cpp:149: (float)(v * (1.0 / 255.0)).  One thread converts 16 pixels of a row (one 16-byte load,
// four 16-byte stores) when the row starts and pitches allow it, else pixel by pixel; the threads of an image are a flat
// index over (row, 16-px chunk), so a 1920-px row does not leave a partly idle block at its right end.
__global__ __launch_bounds__(256) void k_ingest_u8(const unsigned char* __restrict__ src, long src_stride, int sp,
                                                   float* __restrict__ dst, long dst_stride, int dp, int w, int h, int cpr, int vec)
{
    const unsigned char* s = src + (long)blockIdx.y * src_stride;
    float* d = dst + (long)blockIdx.y * dst_stride;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const int y = (int)(idx / cpr), x = (int)(idx - (long)y * cpr) * 16;
    if (y >= h) return;
    const unsigned char* q = s + (long)y * sp + x;
    float* o = d + (long)y * dp + x;
    if (vec && x + 15 < w) {
        const uint4 v = *reinterpret_cast<const uint4*>(q);
        const unsigned wds[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const unsigned u = wds[k];
            *reinterpret_cast<float4*>(o + 4 * k) =
                make_float4((float)((u & 255u) * (1.0 / 255.0)), (float)(((u >> 8) & 255u) * (1.0 / 255.0)),
                            (float)(((u >> 16) & 255u) * (1.0 / 255.0)), (float)((u >> 24) * (1.0 / 255.0)));
        }
    } else {
        for (int e = 0; e < 16 && x + e < w; e++) o[e] = (float)(q[e] * (1.0 / 255.0));
    }
}

void hak_launch_ingest_u8(hipStream_t st, const unsigned char* src, long src_stride, int sp, float* dst, long dst_stride,
                          int dp, int w, int h, int nimg)
{
    const int cpr = (w + 15) / 16;                                  // 16-px chunks per row
    const int vec = ((uintptr_t)src % 16 == 0) && ((uintptr_t)dst % 16 == 0) && sp % 16 == 0 && src_stride % 16 == 0 &&
                    dp % 4 == 0 && dst_stride % 4 == 0;
    dim3 grid((unsigned)(((long)cpr * h + 255) / 256), nimg);
    k_ingest_u8<<<grid, 256, 0, st>>>(src, src_stride, sp, dst, dst_stride, dp, w, h, cpr, vec);
}
