// kernels_fast.hip -- the reference's integer "FAST" pipeline (namespace fastakaze, akazed.cu:2781-4367):
// the float scale space / detector / descriptor in int32 with 16.16 fixed-point weights on uint8 input.
//
// Straightforward kernels (tile or direct-load), batch image in blockIdx.z; this path is a "next" scope
// row: bit-exact first, not yet tuned like the float path.  32-bit products wrap
// (two's complement) exactly as in the oracle (akaze_oracle_fast.c, F1).
#include "hak_internal.h"

#define FT_X 64
#define FT_Y 16
struct FkTaps { int k[8]; };

__device__ __forceinline__ int wmul(int a, int b) { return (int)((unsigned)a * (unsigned)b); }
__device__ __forceinline__ int wadd(int a, int b) { return (int)((unsigned)a + (unsigned)b); }

// ---- separable Gaussian, T = unsigned char (akazed.cu:2990 gConv2d<R>, 2786 gConv2dR2) or int (2922)
template <typename T, int R>
__global__ __launch_bounds__(256) void kf_conv(const T* __restrict__ src, long src_stride, int sp, int* __restrict__ dst,
                                               long dst_stride, int w, int h, int p, FkTaps t)
{
    constexpr int RW = FT_X + 2 * R, RH = FT_Y + 2 * R;
    __shared__ int raw[RH][RW + 1];
    __shared__ int rowp[RH][FT_X];
    const T* s = src + (long)blockIdx.z * src_stride;
    int* d = dst + (long)blockIdx.z * dst_stride;
    const int x0 = blockIdx.x * FT_X, y0 = blockIdx.y * FT_Y, tid = threadIdx.x;
    for (int i = tid; i < RH * RW; i += 256) {
        int r = i / RW, c = i - r * RW;
        raw[r][c] = (int)s[(long)hak_refl(y0 - R + r, h) * sp + hak_refl(x0 - R + c, w)];
    }
    __syncthreads();
    for (int i = tid; i < RH * FT_X; i += 256) {
        int r = i >> 6, c = i & 63;
        int ws = wmul(t.k[0], raw[r][c + R]);
#pragma unroll
        for (int k = 1; k <= R; k++) ws = wadd(ws, wmul(t.k[k], raw[r][c + R - k] + raw[r][c + R + k]));
        rowp[r][c] = ws >> 16;
    }
    __syncthreads();
    const int c = tid & 63, x = x0 + c;
    if (x >= w) return;
    for (int rr = tid >> 6; rr < FT_Y; rr += 4) {
        int y = y0 + rr;
        if (y >= h) break;
        int ws = wmul(t.k[0], rowp[rr + R][c]);
#pragma unroll
        for (int k = 1; k <= R; k++) ws = wadd(ws, wmul(t.k[k], rowp[rr + R - k][c] + rowp[rr + R + k][c]));
        d[(long)y * p + x] = ws >> 16;
    }
}

template <typename T>
static void launch_conv(hipStream_t st, const T* src, long src_stride, int sp, int* dst, long dst_stride, int w, int h, int p,
                        int nimg, const int* taps, int R)
{
    FkTaps t;
    for (int i = 0; i < 8; i++) t.k[i] = i <= R ? taps[i] : 0;
    dim3 grid((w + FT_X - 1) / FT_X, (h + FT_Y - 1) / FT_Y, nimg);
    switch (R) {
    case 2: kf_conv<T, 2><<<grid, 256, 0, st>>>(src, src_stride, sp, dst, dst_stride, w, h, p, t); break;
    case 3: kf_conv<T, 3><<<grid, 256, 0, st>>>(src, src_stride, sp, dst, dst_stride, w, h, p, t); break;
    case 4: kf_conv<T, 4><<<grid, 256, 0, st>>>(src, src_stride, sp, dst, dst_stride, w, h, p, t); break;
    default: kf_conv<T, 5><<<grid, 256, 0, st>>>(src, src_stride, sp, dst, dst_stride, w, h, p, t); break;
    }
}
void hakf_launch_conv_u8(hipStream_t st, const unsigned char* src, long src_stride, int sp, int* dst, long dst_stride,
                         int w, int h, int p, int nimg, const int* taps, int R)
{ launch_conv<unsigned char>(st, src, src_stride, sp, dst, dst_stride, w, h, p, nimg, taps, R); }
void hakf_launch_conv_int(hipStream_t st, const int* src, int* dst, long stride, int w, int h, int p, int nimg, const int* taps, int R)
{ launch_conv<int>(st, src, stride, p, dst, stride, w, h, p, nimg, taps, R); }

// ---- akazed.cu:3143 fastakaze::gDownWithSmooth
__global__ __launch_bounds__(256) void kf_down_smooth(const int* __restrict__ src, int* __restrict__ dst, int* __restrict__ smooth,
                                                      long stride, HakOct so, HakOct dd, FkTaps t)
{
    constexpr int RW = FT_X + 4, RH = FT_Y + 4;
    __shared__ int dec[RH][RW + 1];
    __shared__ int rowp[RH][FT_X];
    const int* s = src + (long)blockIdx.z * stride;
    int* d = dst + (long)blockIdx.z * stride;
    int* sm = smooth + (long)blockIdx.z * stride;
    const int x0 = blockIdx.x * FT_X, y0 = blockIdx.y * FT_Y, tid = threadIdx.x;
    for (int i = tid; i < RH * RW; i += 256) {
        int r = i / RW, c = i - r * RW;
        dec[r][c] = s[(long)hak_refl(2 * (y0 - 2 + r), so.h) * so.p + hak_refl(2 * (x0 - 2 + c), so.w)];
    }
    __syncthreads();
    for (int i = tid; i < RH * FT_X; i += 256) {
        int r = i >> 6, c = i & 63;
        rowp[r][c] = wadd(wadd(wmul(t.k[0], dec[r][c + 2]), wmul(t.k[1], dec[r][c + 1] + dec[r][c + 3])),
                          wmul(t.k[2], dec[r][c] + dec[r][c + 4])) >> 16;
    }
    __syncthreads();
    const int c = tid & 63, x = x0 + c;
    if (x >= dd.w) return;
    for (int rr = tid >> 6; rr < FT_Y; rr += 4) {
        int y = y0 + rr;
        if (y >= dd.h) break;
        long o = (long)y * dd.p + x;
        d[o] = dec[rr + 2][c + 2];
        sm[o] = wadd(wadd(wmul(t.k[0], rowp[rr + 2][c]), wmul(t.k[1], rowp[rr + 1][c] + rowp[rr + 3][c])),
                     wmul(t.k[2], rowp[rr][c] + rowp[rr + 4][c])) >> 16;
    }
}
void hakf_launch_down_smooth(hipStream_t st, const int* src, int* dst, int* smooth, long stride, HakOct so, HakOct dd, int nimg,
                             const int* taps)
{
    FkTaps t;
    for (int i = 0; i < 8; i++) t.k[i] = i <= 2 ? taps[i] : 0;
    dim3 grid((dd.w + FT_X - 1) / FT_X, (dd.h + FT_Y - 1) / FT_Y, nimg);
    kf_down_smooth<<<grid, 256, 0, st>>>(src, dst, smooth, stride, so, dd, t);
}

// ---- Scharr (akazed.cu:3208-3232, 3406-3428)
__device__ __forceinline__ void fscharr(const int* __restrict__ s, int x, int y, int w, int h, int p, int& dx, int& dy)
{
    const int x0 = x - 1 < 0 ? 1 - x : x - 1, x2 = x + 1 < w ? x + 1 : w + w - 3 - x;
    const int y0 = y - 1 < 0 ? 1 - y : y - 1, y2 = y + 1 < h ? y + 1 : h + h - 3 - y;
    const int* r0 = s + (long)y0 * p;
    const int* r1 = s + (long)y * p;
    const int* r2 = s + (long)y2 * p;
    dx = 10 * (r1[x2] - r1[x0]) + 3 * (r0[x2] + r2[x2] - r0[x0] - r2[x0]);
    dy = 10 * (r2[x] - r0[x]) + 3 * (r2[x0] + r2[x2] - r0[x0] - r0[x2]);
}
__device__ __forceinline__ int fgrad(int dx, int dy) { return (int)(sqrtf((float)wadd(wmul(dx, dx), wmul(dy, dy))) + 0.5f); }

__global__ __launch_bounds__(256) void kf_reset(HakImgState* state)
{
    HakImgState* st = state + blockIdx.x;
    for (int i = threadIdx.x; i < HAK_NBINS; i += 256) st->hist[i] = 0;
    if (threadIdx.x == 0) { st->ihmax = 1; st->ncand = 0; st->total_pts = 0; st->num_pts = 0; }    // akazed.cu:4101
}
__global__ __launch_bounds__(256) void kf_grad_max(const int* __restrict__ smooth, long stride, int w, int h, int p, HakImgState* state)
{
    const int* s = smooth + (long)blockIdx.z * stride;
    const int x = blockIdx.x * FT_X + (threadIdx.x & 63), y0 = blockIdx.y * FT_Y + (threadIdx.x >> 6);
    int m = 0;
    if (x < w)
        for (int y = y0; y < blockIdx.y * FT_Y + FT_Y && y < h; y += 4) {
            if (!hak_on_lattice(x, y, w, h)) continue;                                           // akazed.cu:3245-3296
            int dx, dy;
            fscharr(s, x, y, w, h, p, dx, dy);
            m = max(m, fgrad(dx, dy));
        }
    for (int o = 32; o > 0; o >>= 1) m = max(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0 && m > 1) atomicMax(&state[blockIdx.z].ihmax, m);
}
__global__ __launch_bounds__(256) void kf_grad_hist(const int* __restrict__ smooth, long stride, int w, int h, int p, HakImgState* state)
{
    __shared__ int shist[HAK_NBINS];
    for (int i = threadIdx.x; i < HAK_NBINS; i += 256) shist[i] = 0;
    __syncthreads();
    const int* s = smooth + (long)blockIdx.z * stride;
    const int hfactor = (int)(HAK_NBINS / (float)state[blockIdx.z].ihmax * 65536 + 0.5f);     // akazed.cu:4133
    const int x = blockIdx.x * FT_X + (threadIdx.x & 63), y0 = blockIdx.y * FT_Y + (threadIdx.x >> 6);
    if (x < w)
        for (int y = y0; y < blockIdx.y * FT_Y + FT_Y && y < h; y += 4) {
            int dx, dy;
            fscharr(s, x, y, w, h, p, dx, dy);
            int hi = wmul(fgrad(dx, dy), hfactor) >> 16;                                       // akazed.cu:3319
            hi = hi >= HAK_NBINS ? HAK_NBINS - 1 : (hi < 0 ? 0 : hi);
            atomicAdd(&shist[hi], 1);
        }
    __syncthreads();
    for (int i = threadIdx.x; i < HAK_NBINS; i += 256)
        if (shist[i]) atomicAdd(&state[blockIdx.z].hist[i], shist[i]);
}
__global__ void kf_kcontrast(HakImgState* state, int npix, int extra0, float per, int noct)
{
    HakImgState* st = state + blockIdx.x;
    if (threadIdx.x != 0) return;
    int thresh = (int)((npix - (st->hist[0] + extra0)) * per);                                  // akazed.cu:4146, 3305
    int cumuv = 0, k = 1;
    while (k < HAK_NBINS) {
        if (cumuv >= thresh) break;
        cumuv += st->hist[k];
        k++;
    }
    int kc = k * st->ihmax / HAK_NBINS;                                                          // akazed.cu:4162
    for (int o = 0; o < noct; o++) {
        if (o > 0) kc = (int)(kc * 0.75f + 0.5f);                                                // akaze.cpp:642
        st->ikcontrast[o] = kc;
        st->ikc[o] = 1.f / (kc * kc);                                                            // akazed.cu:4215
    }
}
void hakf_launch_contrast(hipStream_t st, const int* smooth, long stride, int w, int h, int p, int nimg, HakImgState* state,
                          float per, int noct)
{
    dim3 grid((w + FT_X - 1) / FT_X, (h + FT_Y - 1) / FT_Y, nimg);
    kf_grad_max<<<grid, 256, 0, st>>>(smooth, stride, w, h, p, state);
    kf_grad_hist<<<grid, 256, 0, st>>>(smooth, stride, w, h, p, state);
    kf_kcontrast<<<nimg, 64, 0, st>>>(state, w * h, hak_hist_extra0(w, h), per, noct);
}
// ---- fused octave-0 prologue of the FAST path (akaze.cpp:589-612): one pass over the uint8 image gives Lt(0,0) = the base
// Gaussian, the maximum Scharr magnitude of the sigma=1 image, and that magnitude as a plane for the histogram pass.  The
// sigma=1 plane itself is only an intermediate of the contrast factor and is never written.  (Unfused: four kernels,
// kf_conv<u8,2> + kf_grad_max + kf_grad_hist + kf_conv<u8,R>, each re-reading its input from HBM.)
#define FB_TX 64
#define FB_TY 32
template <int R>
__global__ __launch_bounds__(256) void kf_base(const unsigned char* __restrict__ img, long img_stride, int sp, int* __restrict__ lt,
                                               int* __restrict__ grad, long stride, int w, int h, int p, FkTaps t1, FkTaps tb,
                                               HakImgState* state, int nbx, int nby, int nimg)
{
    constexpr int H = R < 3 ? 3 : R;
    constexpr int RW = FB_TX + 2 * H, RH = FB_TY + 2 * H;         // raw tile
    constexpr int PW = FB_TX + 2, PH = FB_TY + 6, SH = FB_TY + 2;   // sigma=1 row-pass tile (halo 1 in x, 3 in y), smooth tile (halo 1)
    __shared__ int raw[RH * (RW + 1)];                              // reused for the sigma=1 smooth tile [SH][PW]
    __shared__ int rowb[RH * FB_TX];
    __shared__ int rowp[PH * PW];
    int bx, by, im;
    if (!hak_xcd_decode(nbx, nby, nimg, bx, by, im)) return;
    const unsigned char* s = img + (long)im * img_stride;
    int* o = lt + (long)im * stride;
    int* go = grad + (long)im * stride;
    const int x0 = bx * FB_TX, y0 = by * FB_TY, tid = threadIdx.x;
    for (int i = tid; i < RH * RW; i += 256) {
        const int r = i / RW, c = i - r * RW;
        raw[r * (RW + 1) + c] = (int)s[(long)hak_refl(y0 - H + r, h) * sp + hak_refl(x0 - H + c, w)];
    }
    __syncthreads();
    for (int i = tid; i < RH * FB_TX; i += 256) {                   // base row pass, output columns only
        const int r = i >> 6, c = i & 63;
        const int* q = raw + r * (RW + 1) + c + H;
        int ws = wmul(tb.k[0], q[0]);
#pragma unroll
        for (int k = 1; k <= R; k++) ws = wadd(ws, wmul(tb.k[k], q[-k] + q[k]));
        rowb[i] = ws >> 16;
    }
    for (int i = tid; i < PH * PW; i += 256) {                      // sigma=1 row pass: rows y0-3.., columns x0-1..
        const int r = i / PW, c = i - r * PW;
        const int* q = raw + (r + H - 3) * (RW + 1) + c + H - 1;
        rowp[i] = wadd(wadd(wmul(t1.k[0], q[0]), wmul(t1.k[1], q[-1] + q[1])), wmul(t1.k[2], q[-2] + q[2])) >> 16;
    }
    __syncthreads();
    for (int i = tid; i < FB_TY * FB_TX; i += 256) {                // base column pass -> Lt(0,0)
        const int r = i >> 6, c = i & 63;
        const int x = x0 + c, y = y0 + r;
        const int* q = rowb + (r + H) * FB_TX + c;
        int ws = wmul(tb.k[0], q[0]);
#pragma unroll
        for (int k = 1; k <= R; k++) ws = wadd(ws, wmul(tb.k[k], q[-k * FB_TX] + q[k * FB_TX]));
        if (x < w && y < h) o[(long)y * p + x] = ws >> 16;
    }
    int* sm = raw;                                                  // raw is dead: both row passes are done
    for (int i = tid; i < SH * PW; i += 256) {                      // sigma=1 column pass -> smooth tile (halo 1)
        const int r = i / PW, c = i - r * PW;
        const int* q = rowp + (r + 2) * PW + c;
        sm[i] = wadd(wadd(wmul(t1.k[0], q[0]), wmul(t1.k[1], q[-PW] + q[PW])), wmul(t1.k[2], q[-2 * PW] + q[2 * PW])) >> 16;
    }
    __syncthreads();
    int m = 0;
    for (int i = tid; i < FB_TY * FB_TX; i += 256) {                // Scharr magnitude (akazed.cu:3208-3232)
        const int r = i >> 6, c = i & 63;
        const int x = x0 + c, y = y0 + r;
        if (x < w && y < h) {
            const int* q = sm + (r + 1) * PW + c + 1;
            const int dx = 10 * (q[1] - q[-1]) + 3 * (q[-PW + 1] + q[PW + 1] - q[-PW - 1] - q[PW - 1]);
            const int dy = 10 * (q[PW] - q[-PW]) + 3 * (q[PW - 1] + q[PW + 1] - q[-PW - 1] - q[-PW + 1]);
            const int g = fgrad(dx, dy);
            go[(long)y * p + x] = g;
            if (hak_on_lattice(x, y, w, h)) m = max(m, g);                                        // akazed.cu:3245-3296
        }
    }
    for (int off = 32; off > 0; off >>= 1) m = max(m, __shfl_xor(m, off));
    if ((tid & 63) == 0 && m > 1) atomicMax(&state[im].ihmax, m);
}

// histogram of the gradient plane kf_base left behind (akazed.cu:3299-3330)
#define FHIST_COPIES 8         // private LDS histograms per block, selected by lane (LDS atomics on one address serialise)
__global__ __launch_bounds__(256) void kf_grad_hist_plane(const int* __restrict__ grad, long stride, int w, int h, int p,
                                                          HakImgState* state, int rows_per_block)
{
    __shared__ int shist[FHIST_COPIES * HAK_NBINS];
    const int im = blockIdx.y;
    const int* g0 = grad + (long)im * stride;
    for (int i = threadIdx.x; i < FHIST_COPIES * HAK_NBINS; i += 256) shist[i] = 0;
    const int hfactor = (int)(HAK_NBINS / (float)state[im].ihmax * 65536 + 0.5f);                // akazed.cu:4133
    __syncthreads();
    int* mine = shist + (threadIdx.x & (FHIST_COPIES - 1)) * HAK_NBINS;
    const int y0 = blockIdx.x * rows_per_block, y1 = min(y0 + rows_per_block, h);
    for (int y = y0; y < y1; y++) {
        const int* row = g0 + (long)y * p;
        for (int x = threadIdx.x; x < w; x += 256) {
            int hi = wmul(row[x], hfactor) >> 16;                                                 // akazed.cu:3319
            hi = hi >= HAK_NBINS ? HAK_NBINS - 1 : (hi < 0 ? 0 : hi);
            atomicAdd(&mine[hi], 1);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < HAK_NBINS; i += 256) {
        int sum = 0;
#pragma unroll
        for (int c = 0; c < FHIST_COPIES; c++) sum += shist[c * HAK_NBINS + i];
        if (sum) atomicAdd(&state[im].hist[i], sum);
    }
}

// img (uint8) -> Lt(0,0), contrast factors; `grad_scratch` = a free int32 plane of the arena.  false: R not covered
// (caller: hakf_launch_conv_u8 x2 + hakf_launch_contrast)
bool hakf_launch_base_level(hipStream_t st, const unsigned char* img, long img_stride, int sp, int* lt, int* grad_scratch, long stride,
                            int w, int h, int p, int nimg, const int* itaps1, const int* itaps_base, int R, HakImgState* state,
                            float per, int noct, const HakKnobs& knobs)
{
    if (R < 2 || R > 5) return false;
    FkTaps t1, tb;
    for (int i = 0; i < 8; i++) { t1.k[i] = i <= 2 ? itaps1[i] : 0; tb.k[i] = i <= R ? itaps_base[i] : 0; }
    const int nbx = (w + FB_TX - 1) / FB_TX, nby = (h + FB_TY - 1) / FB_TY;
    const unsigned grid = hak_xcd_grid(nbx, nby, nimg);
    // the streaming form with the histogram inside (round 5): contrast maximum first, from the lattice points alone, then ONE pass
    if (knobs.base_hist && hakf_launch_base_stream(st, img, img_stride, sp, lt, nullptr, stride, w, h, p, nimg, itaps1, itaps_base, R, state, knobs.base_stream)) {
        kf_kcontrast<<<nimg, 64, 0, st>>>(state, w * h, hak_hist_extra0(w, h), per, noct);
        return true;
    }
    if (hakf_launch_base_stream(st, img, img_stride, sp, lt, grad_scratch, stride, w, h, p, nimg, itaps1, itaps_base, R, state, knobs.base_stream)) {
        // pass A done by the streaming kernel
    } else
    switch (R) {
    case 2: kf_base<2><<<grid, 256, 0, st>>>(img, img_stride, sp, lt, grad_scratch, stride, w, h, p, t1, tb, state, nbx, nby, nimg); break;
    case 3: kf_base<3><<<grid, 256, 0, st>>>(img, img_stride, sp, lt, grad_scratch, stride, w, h, p, t1, tb, state, nbx, nby, nimg); break;
    case 4: kf_base<4><<<grid, 256, 0, st>>>(img, img_stride, sp, lt, grad_scratch, stride, w, h, p, t1, tb, state, nbx, nby, nimg); break;
    default: kf_base<5><<<grid, 256, 0, st>>>(img, img_stride, sp, lt, grad_scratch, stride, w, h, p, t1, tb, state, nbx, nby, nimg); break;
    }
    int rpb = 8;
    while (rpb > 1 && (long)((h + rpb - 1) / rpb) * nimg < 2048) rpb >>= 1;
    kf_grad_hist_plane<<<dim3((h + rpb - 1) / rpb, nimg), 256, 0, st>>>(grad_scratch, stride, w, h, p, state, rpb);
    kf_kcontrast<<<nimg, 64, 0, st>>>(state, w * h, hak_hist_extra0(w, h), per, noct);
    return true;
}

void hakf_launch_reset(hipStream_t st, HakImgState* state, int nimg) { kf_reset<<<nimg, 256, 0, st>>>(state); }

// ---- akazed.cu:3406 gFlowNaive (conductivity as 16.16 int)
__global__ __launch_bounds__(256) void kf_flow(const int* __restrict__ src, int* __restrict__ dst, long stride, int w, int h, int p,
                                               int type, const HakImgState* state, int octave)
{
    const int* s = src + (long)blockIdx.z * stride;
    int* d = dst + (long)blockIdx.z * stride;
    const float ikc = state[blockIdx.z].ikc[octave];
    const int x = blockIdx.x * FT_X + (threadIdx.x & 63), y0 = blockIdx.y * FT_Y + (threadIdx.x >> 6);
    if (x >= w) return;
    for (int y = y0; y < blockIdx.y * FT_Y + FT_Y && y < h; y += 4) {
        int dx, dy;
        fscharr(s, x, y, w, h, p, dx, dy);
        const float dif2 = wadd(wmul(dx, dx), wmul(dy, dy)) * ikc;
        float g;
        if (type == HAK_PM_G2) g = 1.f / (1.f + dif2);
        else if (type == HAK_PM_G1) g = hak_expf(-dif2);
        else if (type == HAK_WEICKERT) { float d2 = dif2 * dif2; g = 1.f - hak_expf(-3.315f / (d2 * d2)); }
        else g = 1.f / sqrtf(1.f + dif2);
        d[(long)y * p + x] = (int)(g * 65536 + 0.5f);
    }
}
void hakf_launch_flow(hipStream_t st, const int* src, int* dst, long stride, int w, int h, int p, int nimg, int type,
                      const HakImgState* state, int octave)
{
    dim3 grid((w + FT_X - 1) / FT_X, (h + FT_Y - 1) / FT_Y, nimg);
    kf_flow<<<grid, 256, 0, st>>>(src, dst, stride, w, h, p, type, state, octave);
}

// ---- akazed.cu:3448 gNldStepNaive
__global__ __launch_bounds__(256) void kf_nld_step(const int* __restrict__ src, const int* __restrict__ flow, int* __restrict__ dst,
                                                   long stride, int w, int h, int p, int stepfac)
{
    const int* s = src + (long)blockIdx.z * stride;
    const int* f = flow + (long)blockIdx.z * stride;
    int* d = dst + (long)blockIdx.z * stride;
    const int x = blockIdx.x * FT_X + (threadIdx.x & 63), y0 = blockIdx.y * FT_Y + (threadIdx.x >> 6);
    if (x >= w) return;
    const int x0 = x - 1 < 0 ? 1 - x : x - 1, x2 = x + 1 < w ? x + 1 : w + w - 3 - x;
    for (int y = y0; y < blockIdx.y * FT_Y + FT_Y && y < h; y += 4) {
        const long r0 = (long)(y - 1 < 0 ? 1 - y : y - 1) * p, r1 = (long)y * p, r2 = (long)(y + 1 < h ? y + 1 : h + h - 3 - y) * p;
        const int L = s[r1 + x], F = f[r1 + x];
        const int step = wadd(wadd(wadd(wmul(F + f[r1 + x2], s[r1 + x2] - L), wmul(F + f[r1 + x0], s[r1 + x0] - L)),
                                   wmul(F + f[r2 + x], s[r2 + x] - L)), wmul(F + f[r0 + x], s[r0 + x] - L)) >> 16;
        d[r1 + x] = wadd(wmul(stepfac, step) >> 16, L);
    }
}
void hakf_launch_nld_step(hipStream_t st, const int* src, const int* flow, int* dst, long stride, int w, int h, int p, int nimg, float tau)
{
    const int stepfac = (int)(0.5f * tau * 65536 + 0.5f);                                         // akazed.cu:4237
    dim3 grid((w + FT_X - 1) / FT_X, (h + FT_Y - 1) / FT_Y, nimg);
    kf_nld_step<<<grid, 256, 0, st>>>(src, flow, dst, stride, w, h, p, stepfac);
}

// ---- akazed.cu:3339 gDerivate, 3371 gHessianDeterminant (unfused fallback for dilations > 4 and debug planes);
// dxy = interleaved {Lx, Ly} (HakLayout)
__global__ __launch_bounds__(256) void kf_derivate(const int* __restrict__ src, int* __restrict__ dxy, long stride,
                                                   int w, int h, int p, int step, int fac1, int fac2)
{
    const int* s = src + (long)blockIdx.z * stride;
    int2* o = reinterpret_cast<int2*>(dxy + (long)blockIdx.z * stride);
    const int x = blockIdx.x * FT_X + (threadIdx.x & 63), y0 = blockIdx.y * FT_Y + (threadIdx.x >> 6);
    if (x >= w) return;
    const int x0 = hak_refl(x - step, w), x2 = hak_refl(x + step, w);
    for (int y = y0; y < blockIdx.y * FT_Y + FT_Y && y < h; y += 4) {
        const int* r0 = s + (long)hak_refl(y - step, h) * p;
        const int* r1 = s + (long)y * p;
        const int* r2 = s + (long)hak_refl(y + step, h) * p;
        const int ul = r0[x0], uc = r0[x], ur = r0[x2], cl = r1[x0], cr = r1[x2], ll = r2[x0], lc = r2[x], lr = r2[x2];
        o[(long)y * p + x] = make_int2(wadd(wmul(fac1, ur + lr - ul - ll), wmul(fac2, cr - cl)) >> 16,
                                       wadd(wmul(fac1, lr + ll - ur - ul), wmul(fac2, lc - uc)) >> 16);
    }
}
__global__ __launch_bounds__(256) void kf_hessian(const int* __restrict__ dxy, int* __restrict__ det, long stride,
                                                  int w, int h, int p, int step, int fac1, int fac2)
{
    const int* d = dxy + (long)blockIdx.z * stride;
    int* o = det + (long)blockIdx.z * stride;
    const int x = blockIdx.x * FT_X + (threadIdx.x & 63), y0 = blockIdx.y * FT_Y + (threadIdx.x >> 6);
    if (x >= w) return;
    for (int y = y0; y < blockIdx.y * FT_Y + FT_Y && y < h; y += 4)
        o[(long)y * p + x] = hak_det_at<int>(d, x, y, step, w, h, p, fac1, fac2);
}
static void ifactors(int& f1, int& f2)
{
    float fac1, fac2;                                                                             // akazed.cu:4177-4184
    hak_deriv_factors(&fac1, &fac2);
    f1 = (int)(fac1 * 65536 + 0.5f);
    f2 = (int)(fac2 * 65536 + 0.5f);
}
void hakf_launch_hessian(hipStream_t st, const int* src, int* dxy, int* det, long stride, int w, int h, int p, int nimg, int step)
{
    int f1, f2;
    ifactors(f1, f2);
    dim3 grid((w + FT_X - 1) / FT_X, (h + FT_Y - 1) / FT_Y, nimg);
    kf_derivate<<<grid, 256, 0, st>>>(src, dxy, stride, w, h, p, step, f1, f2);
    kf_hessian<<<grid, 256, 0, st>>>(dxy, det, stride, w, h, p, step, f1, f2);
}
void hakf_launch_det(hipStream_t st, const int* dxy, int* det, long stride, int w, int h, int p, int nimg, int step)
{
    int f1, f2;
    ifactors(f1, f2);
    dim3 grid((w + FT_X - 1) / FT_X, (h + FT_Y - 1) / FT_Y, nimg);
    kf_hessian<<<grid, 256, 0, st>>>(dxy, det, stride, w, h, p, step, f1, f2);
}

// ---- akazed.cu:3476 gCalcExtremaMap (int): key = response << 32 | ~layer, candidates appended
__global__ __launch_bounds__(256) void kf_extrema(const int* __restrict__ base, long stride, unsigned long long* maps, long map_stride,
                                                  unsigned long long* cand, long cand_cap, HakImgState* state, HakLayout L,
                                                  const HakTables* __restrict__ tab, int octave, int s, int threshold, long det_off)
{
    const int img = blockIdx.z;
    const HakOct oc = L.oct[octave];
    const int* det = base + (long)img * stride + det_off;
    const int layer = octave * L.ms + s;
    const float border = tab->borders[layer];
    const int psz = (int)tab->borders[octave * L.ms];
    const int lane = threadIdx.x & 63;
    const int x = blockIdx.x * 64 + lane, y0 = blockIdx.y * 16 + (threadIdx.x >> 6);
    const bool xok = x >= psz && x < oc.w && (int)(x - border + 0.5f) - 1 >= 0 && (int)(x + border + 0.5f) + 1 < oc.w;
    for (int y = y0; y < blockIdx.y * 16 + 16; y += 4) {
        bool hit = false;
        int v = 0;
        if (xok && y >= psz && y < oc.h && (int)(y - border + 0.5f) - 1 >= 0 && (int)(y + border + 0.5f) + 1 < oc.h) {
            const int* vp = det + (long)y * oc.p + x;
            v = *vp;
            hit = v > threshold && v > vp[-oc.p] && v > vp[oc.p] && v > vp[-1] && v > vp[1] && v > vp[-oc.p - 1] &&
                  v > vp[-oc.p + 1] && v > vp[oc.p - 1] && v > vp[oc.p + 1];
        }
        const unsigned long long m = __ballot(hit);
        if (m) {
            int cbase = 0;
            if (lane == 0) cbase = atomicAdd(&state[img].ncand, __popcll(m));
            cbase = __builtin_amdgcn_readfirstlane(cbase);
            if (hit) {
                const int fx = x << octave, fy = y << octave;
                const unsigned long long key = ((unsigned long long)(unsigned)v << 32) | (0xFFFFFFFFu - (unsigned)layer);
                atomicMax(&maps[(long)img * map_stride + (long)fy * L.oct[0].p + fx], key);
                const long slot = cbase + __popcll(m & ((1ull << lane) - 1ull));
                if (slot < cand_cap) cand[(long)img * cand_cap + slot] = ((unsigned long long)layer << 32) | ((unsigned)fy << 16) | (unsigned)fx;
            }
        }
    }
}
void hakf_launch_extrema(hipStream_t st, const HakBatch& b, const HakLayout& L, const HakTables* tab, int octave, int s, int threshold,
                         long det_off)
{
    const HakOct oc = L.oct[octave];
    dim3 grid((oc.w + 63) / 64, (oc.h + 15) / 16, b.nimg);
    kf_extrema<<<grid, 256, 0, st>>>(reinterpret_cast<const int*>(b.base), b.stride, b.maps, b.map_stride, b.cand, b.cand_cap, b.state, L,
                                     tab, octave, s, threshold, det_off);
}

// refine (akazed.cu:3600) + orientation (3649) + MLDB (3723): k_orient<int> / k_describe<int> in kernels_describe.hip
