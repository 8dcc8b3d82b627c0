// kernels_base.hip -- octave-0 prologue: base level + contrast factor (akaze.cpp:325-333).
//
//   hLowPass(img -> smooth, var 1, ksz 5)         akazed.cu:2336, 204   (gConv2d<2>)
//   hScharrContrast(smooth -> kcontrast)          akazed.cu:2410, 644, 827, 901
//   hLowPass(img -> Lt(0,0), var soffset^2)       akazed.cu:2336, 204   (gConv2d<R>, R = 4 for soffset 1.6)
//
// The sigma=1 plane is only an intermediate of the contrast factor, so it is never written:
//   pass A  reads the image once and produces Lt(0,0) AND the maximum Scharr magnitude of the
//           sigma=1 image (one atomicMax per block);
//   pass B  reads the image again, recomputes sigma=1 + gradient and fills the 300-bin histogram
//           (needs the global maximum of pass A).
// 12 B/px instead of the 24 B/px of four separate passes.  Both are persistent tile kernels with
// register prefetch of the next tile (see kernels_hessian.hip for why).  As in
// kernels_smoothflow.hip, tiles are loaded with reflect-101 indices and then indexed plainly.
#include "hak_internal.h"

#define BS_TX 64
#define BS_TY 32

struct BsTaps { float a[3]; float b[6]; };       // sigma=1 taps (R=2), base taps (R <= 5)

// un-normalised Scharr magnitude at the centre of a 3x3 neighbourhood in a tile of width W (akazed.cu:664-666)
template <int W>
__device__ __forceinline__ float scharr_mag(const float* q)
{
    const float ul = q[-W - 1], uc = q[-W], ur = q[-W + 1];
    const float cl = q[-1], cr = q[1];
    const float ll = q[W - 1], lc = q[W], lr = q[W + 1];
    const float dx = 10 * (cr - cl) + 3 * (ur + lr - ul - ll);
    const float dy = 10 * (lc - uc) + 3 * (ll + lr - ul - ur);
    return sqrtf(dx * dx + dy * dy);
}

template <int H>                                   // H = tile halo
struct BsGeo {
    static constexpr int RW = BS_TX + 2 * H, RH = BS_TY + 2 * H;    // raw tile
    static constexpr int NPF = (RW * RH + 255) / 256;
    static constexpr int PW = BS_TX + 2, PH = BS_TY + 6;            // sigma=1 row-pass tile (halo 1 in x, 3 in y)
    static constexpr int SH = BS_TY + 2;                            // sigma=1 smooth tile (halo 1)
};

template <int H>
__device__ __forceinline__ void bs_fetch(float (&pf)[BsGeo<H>::NPF], const float* __restrict__ s, int w, int h, int sp,
                                         int x0, int y0, int tid)
{
    using G = BsGeo<H>;
#pragma unroll
    for (int i = 0; i < G::NPF; i++) {
        const int idx = tid + 256 * i;
        if (idx < G::RW * G::RH) {
            const int r = idx / G::RW, c = idx - r * G::RW;
            pf[i] = s[(long)hak_refl(y0 - H + r, h) * sp + hak_refl(x0 - H + c, w)];
        }
    }
}

// sigma=1 row pass (rows y0-3 .. y0+TY+2, columns x0-1 .. x0+TX) from a raw tile with halo H >= 3
template <int H>
__device__ __forceinline__ void bs_rowpass1(const float* raw, float* rowp, const BsTaps& t, int tid)
{
    using G = BsGeo<H>;
    for (int idx = tid; idx < G::PH * G::PW; idx += 256) {
        const int r = idx / G::PW, c = idx - r * G::PW;
        const float* q = raw + (r + H - 3) * G::RW + c + H - 1;
        float ws = q[0] * t.a[0];
        ws += t.a[1] * (q[-1] + q[1]);
        ws += t.a[2] * (q[-2] + q[2]);
        rowp[idx] = ws;
    }
}

// sigma=1 column pass -> smooth tile (rows y0-1 .. y0+TY, columns x0-1 .. x0+TX)
template <int H>
__device__ __forceinline__ void bs_colpass1(const float* rowp, float* sm, const BsTaps& t, int tid)
{
    using G = BsGeo<H>;
    for (int idx = tid; idx < G::SH * G::PW; idx += 256) {
        const int r = idx / G::PW, c = idx - r * G::PW;
        const float* q = rowp + (r + 2) * G::PW + c;
        float ws = q[0] * t.a[0];
        ws += t.a[1] * (q[-G::PW] + q[G::PW]);
        ws += t.a[2] * (q[-2 * G::PW] + q[2 * G::PW]);
        sm[idx] = ws;
    }
}

// ---- pass A: Lt(0,0) = G(base) * img, and max |Scharr(G(1) * img)|
template <int R>
__global__ __launch_bounds__(256) void k_base_a(const float* __restrict__ img, long img_stride, int sp,
                                                float* __restrict__ lt, float* __restrict__ grad, long stride, int w, int h, int p,
                                                BsTaps t, HakImgState* state, int tiles_per_block, int nbx, int nby, int nimg)
{
    constexpr int H = R < 3 ? 3 : R;
    using G = BsGeo<H>;
    __shared__ float raw[G::RH * G::RW];              // reused for the sigma=1 smooth tile
    __shared__ float rowb[G::RH * BS_TX];             // base row pass (all tile rows, output columns)
    __shared__ float rowp[G::PH * G::PW];             // sigma=1 row pass
    __shared__ float wmax[4];
    int bx, by, im;
    if (!hak_xcd_decode(nbx, nby, nimg, bx, by, im)) return;
    const float* s = img + (long)im * img_stride;
    float* o = lt + (long)im * stride;
    float* go = grad ? grad + (long)im * stride : nullptr;
    const int tid = threadIdx.x;
    const int x0 = bx * BS_TX;
    const int ty0 = by * tiles_per_block;
    const int ty1 = min(ty0 + tiles_per_block, (h + BS_TY - 1) / BS_TY);
    float pf[G::NPF];
    float tmax = 0.f;
    if (ty0 < ty1) bs_fetch<H>(pf, s, w, h, sp, x0, ty0 * BS_TY, tid);
    for (int ty = ty0; ty < ty1; ty++) {
        const int y0 = ty * BS_TY;
        hak_lds_barrier();
#pragma unroll
        for (int i = 0; i < G::NPF; i++)
            if (tid + 256 * i < G::RW * G::RH) raw[tid + 256 * i] = pf[i];
        hak_lds_barrier();
        if (ty + 1 < ty1) bs_fetch<H>(pf, s, w, h, sp, x0, y0 + BS_TY, tid);
        // base row pass on every raw row, output columns only (akazed.cu:227-239)
        for (int idx = tid; idx < G::RH * BS_TX; idx += 256) {
            const int r = idx >> 6, c = idx & 63;
            const float* q = raw + r * G::RW + c + H;
            float ws = q[0] * t.b[0];
#pragma unroll
            for (int k = 1; k <= R; k++) ws += t.b[k] * (q[-k] + q[k]);
            rowb[idx] = ws;
        }
        bs_rowpass1<H>(raw, rowp, t, tid);
        hak_lds_barrier();
        // base column pass -> Lt(0,0) (akazed.cu:283-288)
        for (int idx = tid; idx < BS_TY * BS_TX; idx += 256) {
            const int r = idx >> 6, c = idx & 63;
            const int x = x0 + c, y = y0 + r;
            const float* q = rowb + (r + H) * BS_TX + c;
            float ws = q[0] * t.b[0];
#pragma unroll
            for (int k = 1; k <= R; k++) ws += t.b[k] * (q[-k * BS_TX] + q[k * BS_TX]);
            if (x < w && y < h) o[(long)y * p + x] = ws;
        }
        float* sm = raw;
        bs_colpass1<H>(rowp, sm, t, tid);
        hak_lds_barrier();
        for (int idx = tid; idx < BS_TY * BS_TX; idx += 256) {
            const int r = idx >> 6, c = idx & 63;
            if (x0 + c < w && y0 + r < h) {
                const float g = scharr_mag<G::PW>(sm + (r + 1) * G::PW + c + 1);
                if (hak_on_lattice(x0 + c, y0 + r, w, h)) tmax = fmaxf(tmax, g);
                if (go) go[(long)(y0 + r) * p + x0 + c] = g;      // kept for the histogram pass (k_grad_hist_plane)
            }
        }
    }
    for (int off = 32; off > 0; off >>= 1) tmax = fmaxf(tmax, __shfl_xor(tmax, off));
    if ((tid & 63) == 0) wmax[tid >> 6] = tmax;
    hak_lds_barrier();
    if (tid == 0) {
        const float m = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
        if (m > 0.f) atomicMax(&state[im].hmax_bits, __float_as_uint(m));   // lattice maximum (akazed.cu:827-877)
    }
}

// ---- pass B: 300-bin histogram of |Scharr(G(1) * img)|
__global__ __launch_bounds__(256) void k_base_b(const float* __restrict__ img, long img_stride, int sp,
                                                int w, int h, BsTaps t, HakImgState* state, int tiles_per_block, int nbx, int nby, int nimg)
{
    constexpr int H = 3;
    using G = BsGeo<H>;
    __shared__ float raw[G::RH * G::RW];
    __shared__ float rowp[G::PH * G::PW];
    __shared__ int shist[HAK_NBINS];
    int bx, by, im;
    if (!hak_xcd_decode(nbx, nby, nimg, bx, by, im)) return;
    const float* s = img + (long)im * img_stride;
    const int tid = threadIdx.x;
    const int x0 = bx * BS_TX;
    const int ty0 = by * tiles_per_block;
    const int ty1 = min(ty0 + tiles_per_block, (h + BS_TY - 1) / BS_TY);
    for (int i = tid; i < HAK_NBINS; i += 256) shist[i] = 0;
    const float hmax = __uint_as_float(state[im].hmax_bits);
    const float hfactor = HAK_NBINS / hmax;                         // akazed.cu:2450
    float pf[G::NPF];
    if (ty0 < ty1) bs_fetch<H>(pf, s, w, h, sp, x0, ty0 * BS_TY, tid);
    for (int ty = ty0; ty < ty1; ty++) {
        const int y0 = ty * BS_TY;
        hak_lds_barrier();
#pragma unroll
        for (int i = 0; i < G::NPF; i++)
            if (tid + 256 * i < G::RW * G::RH) raw[tid + 256 * i] = pf[i];
        hak_lds_barrier();
        if (ty + 1 < ty1) bs_fetch<H>(pf, s, w, h, sp, x0, y0 + BS_TY, tid);
        bs_rowpass1<H>(raw, rowp, t, tid);
        hak_lds_barrier();
        float* sm = raw;
        bs_colpass1<H>(rowp, sm, t, tid);
        hak_lds_barrier();
        for (int idx = tid; idx < BS_TY * BS_TX; idx += 256) {
            const int r = idx >> 6, c = idx & 63;
            if (x0 + c < w && y0 + r < h) {
                const float g = scharr_mag<G::PW>(sm + (r + 1) * G::PW + c + 1);
                // (int)__fmul_rz(g, factor): exact double product, truncated (akazed.cu:924)
                int hi = (int)((double)g * (double)hfactor);
                hi = hi >= HAK_NBINS ? HAK_NBINS - 1 : hi;
                atomicAdd(&shist[hi], 1);
            }
        }
    }
    hak_lds_barrier();
    for (int i = tid; i < HAK_NBINS; i += 256)
        if (shist[i]) atomicAdd(&state[im].hist[i], shist[i]);
}

// ---- pass B': the same histogram from the gradient plane pass A left behind (4 B/px read, no recomputation)
#define HIST_COPIES 8          // private LDS histograms per block, selected by lane: smooth images put most pixels into a few
                                // bins, and LDS atomics on one address serialise
// host half of hScharrContrast (akazed.cu:2467-2481) + the per-octave 0.75 decay (akaze.cpp:371) and ikc = 1/(k*k)
// (akazed.cu:2493), kept on the device.  hist_words: the image's 300 bins (global memory, or an LDS copy fetched coherently).
__device__ __forceinline__ void hak_finish_kcontrast(HakImgState* st, const int* hist_words, int npix, int extra0, float per, int noct)
{
    auto hist = [&](int k) { return hist_words[k]; };
    const float hmax = __uint_as_float(st->hmax_bits);
    const float hfactor = HAK_NBINS / hmax;
    int thresh = (int)((npix - (hist(0) + extra0)) * per);                  // akazed.cu:2468; extra0: hak_hist_extra0
    int cumuv = 0, k = 1;
    while (k < HAK_NBINS) {
        if (cumuv >= thresh) break;
        cumuv += hist(k);
        k++;
    }
    float kc = k / hfactor;
    for (int o = 0; o < noct; o++) {
        if (o > 0) kc *= 0.75f;
        st->kcontrast[o] = kc;
        st->ikc[o] = 1.f / (kc * kc);
    }
}

// noct > 0: the block that adds its bins last also finishes the contrast factor (one launch less on the critical chain of a
// single-image call); noct == 0: histogram only (k_kcontrast2 follows)
__global__ __launch_bounds__(256) void k_grad_hist_plane(const float* __restrict__ grad, long stride, int w, int h, int p,
                                                         HakImgState* state, int rows_per_block, float per, int noct)
{
    __shared__ int shist[HIST_COPIES * HAK_NBINS];
    __shared__ int last_block;
    const int im = blockIdx.y;
    const float* g0 = grad + (long)im * stride;
    const int tid = threadIdx.x;
    for (int i = tid; i < HIST_COPIES * HAK_NBINS; i += 256) shist[i] = 0;
    const float hmax = __uint_as_float(state[im].hmax_bits);
    const float hfactor = HAK_NBINS / hmax;                         // akazed.cu:2450
    hak_lds_barrier();
    const int y0 = blockIdx.x * rows_per_block, y1 = min(y0 + rows_per_block, h);
    int* mine = shist + (tid & (HIST_COPIES - 1)) * HAK_NBINS;
    auto bin = [&](float g) {
        // (int)__fmul_rz(g, factor): exact double product, truncated (akazed.cu:924)
        int hi = (int)((double)g * (double)hfactor);
        hi = hi >= HAK_NBINS ? HAK_NBINS - 1 : hi;
        atomicAdd(&mine[hi], 1);
    };
    const int w4 = w >> 2;                                          // rows are 16-byte aligned (pitch % 64 == 0)
    for (int y = y0; y < y1; y++) {
        const float* row = g0 + (long)y * p;
        for (int x = tid; x < w4; x += 256) {
            const float4 g = reinterpret_cast<const float4*>(row)[x];
            bin(g.x); bin(g.y); bin(g.z); bin(g.w);
        }
        if (tid < (w & 3)) bin(row[4 * w4 + tid]);
    }
    hak_lds_barrier();
    for (int i = tid; i < HAK_NBINS; i += 256) {
        int sum = 0;
#pragma unroll
        for (int c = 0; c < HIST_COPIES; c++) sum += shist[c * HAK_NBINS + i];
        if (sum) atomicAdd(&state[im].hist[i], sum);
    }
    if (noct > 0) {
        __threadfence();                                            // this block's bins are visible before its ticket is
        __syncthreads();
        if (tid == 0) last_block = atomicAdd(&state[im].hist_done, 1) == (int)gridDim.x - 1;
        __syncthreads();
        if (last_block) {
            // every other block's bins have landed in L2 (their fences precede their tickets); fetch the 300 words past this CU's
            // L1 -- which may hold a stale line of the state record -- with all lanes at once, then scan them from LDS
            __threadfence();
            for (int i = tid; i < HAK_NBINS; i += 256)
                shist[i] = __hip_atomic_load(&state[im].hist[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __syncthreads();
            if (tid == 0) hak_finish_kcontrast(state + im, shist, w * h, hak_hist_extra0(w, h), per, noct);
        }
    }
}

__global__ void k_kcontrast2(HakImgState* state, int npix, int extra0, float per, int noct)
{
    if (threadIdx.x != 0) return;
    hak_finish_kcontrast(state + blockIdx.x, state[blockIdx.x].hist, npix, extra0, per, noct);
}

// img -> Lt(0,0) and the per-image contrast factors.  Returns false when R is not supported here.
// `grad_scratch` (optional): a free plane of the arena (same stride / pitch as lt) that receives the gradient magnitude so
// that the histogram pass reads 4 B/px instead of recomputing sigma=1 + Scharr from the image.
bool hak_launch_base_level(hipStream_t st, const float* img, long img_stride, int sp, float* lt, float* grad_scratch, long stride,
                           int w, int h, int p, int nimg, const float* taps1, const float* taps_base, int R,
                           HakImgState* state, float per, int noct, const HakKnobs& knobs)
{
    if (R < 2 || R > 5) return false;
    BsTaps t;
    for (int i = 0; i < 3; i++) t.a[i] = taps1[i];
    for (int i = 0; i < 6; i++) t.b[i] = i <= R ? taps_base[i] : 0.f;
    const int ntx = (w + BS_TX - 1) / BS_TX, nty = (h + BS_TY - 1) / BS_TY;
    int tpb = 8;
    while (tpb > 1 && (long)ntx * ((nty + tpb - 1) / tpb) * nimg < 4096) tpb >>= 1;
    const int nby = (nty + tpb - 1) / tpb;
    const unsigned grid = hak_xcd_grid(ntx, nby, nimg);
    // the streaming form with the histogram inside (round 5): contrast maximum first, from the lattice points alone, then ONE pass
    if (knobs.base_hist && hak_launch_base_stream(st, img, img_stride, sp, lt, nullptr, stride, w, h, p, nimg, taps1, taps_base, R, state, knobs.base_stream)) {
        k_kcontrast2<<<nimg, 64, 0, st>>>(state, w * h, hak_hist_extra0(w, h), per, noct);
        return true;
    }
    if (grad_scratch && hak_launch_base_stream(st, img, img_stride, sp, lt, grad_scratch, stride, w, h, p, nimg, taps1, taps_base, R, state, knobs.base_stream)) {
        // pass A done by the streaming kernel
    } else
    switch (R) {
    case 2: k_base_a<2><<<grid, 256, 0, st>>>(img, img_stride, sp, lt, grad_scratch, stride, w, h, p, t, state, tpb, ntx, nby, nimg); break;
    case 3: k_base_a<3><<<grid, 256, 0, st>>>(img, img_stride, sp, lt, grad_scratch, stride, w, h, p, t, state, tpb, ntx, nby, nimg); break;
    case 4: k_base_a<4><<<grid, 256, 0, st>>>(img, img_stride, sp, lt, grad_scratch, stride, w, h, p, t, state, tpb, ntx, nby, nimg); break;
    default: k_base_a<5><<<grid, 256, 0, st>>>(img, img_stride, sp, lt, grad_scratch, stride, w, h, p, t, state, tpb, ntx, nby, nimg); break;
    }
    if (grad_scratch) {
        static const long hist_min_blocks = [] { const char* e = getenv("HAK_HIST_MIN_BLOCKS"); const long v = e ? atol(e) : 256; return v < 1 ? 1 : v; }();
        static const int hist_rpb_max = [] { const char* e = getenv("HAK_HIST_RPB_MAX"); const int v = e ? atoi(e) : 8; return v < 1 ? 1 : v; }();
        int rpb = hist_rpb_max;
        while (rpb > 1 && (long)((h + rpb - 1) / rpb) * nimg < hist_min_blocks) rpb >>= 1;
        // rows per block: 8 unless that leaves fewer than 256 blocks (round 5; 2048 before: a pair's two images then ran 2 160 one-row
        // blocks, each zeroing 2 400 LDS words and flushing its bins with global atomics for 1 920 pixels -- 27 us alone, on the
        // critical chain of the pair call; with 8 rows per block the call is 0.547 instead of 0.567 ms, batches are unchanged)
        // (noct = 0: the contrast factor stays a launch of its own.  Letting the last histogram block finish it was measured:
        // the fence every block then needs behind its ~300 bin atomics, which are otherwise fire-and-forget, took the kernel from
        // 22 to 73 us on a single 1080p image -- far more than the launch it saves)
        k_grad_hist_plane<<<dim3((h + rpb - 1) / rpb, nimg), 256, 0, st>>>(grad_scratch, stride, w, h, p, state, rpb, per, 0);
    } else {
        k_base_b<<<grid, 256, 0, st>>>(img, img_stride, sp, w, h, t, state, tpb, ntx, nby, nimg);
    }
    k_kcontrast2<<<nimg, 64, 0, st>>>(state, w * h, hak_hist_extra0(w, h), per, noct);
    return true;
}
