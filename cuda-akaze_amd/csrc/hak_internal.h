// hak_internal.h -- shared declarations of libhipakaze (gfx950 only).
#pragma once
#include <cstdlib>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/hipakaze.h"

#define HAK_NBINS 300        // akazed.cu:8
#define HAK_WAVE 64

// ---------------------------------------------------------------- geometry
struct HakOct {
    int w, h, p;             // width, height, pitch (elements) of one octave plane
    long plane;              // h * p
};

// Per-image arena (floats). Planes persist until the descriptors are done.
//   per (octave o, sublevel s): Lt (1 plane) and Dxy (2 planes' worth: the first derivatives INTERLEAVED, element
//                               (y, x) = {Lx, Ly} at dxy + 2 * (y * p + x))                      persistent
//   per octave: smooth, flow, tmp                                                                  scratch
// The determinant is never stored: its only consumers are the extrema test (fused into the Hessian kernels) and the
// sub-pixel refinement of the ~2 k keypoints, which re-evaluates the nine values it needs from Dxy (hak_det_at).
struct HakLayout {
    int noct, ms;
    HakOct oct[HAK_MAX_OCTAVES];
    long lvl_off[HAK_MAX_OCTAVES];              // start of the 3*ms persistent planes of octave o
    long smooth_off[HAK_MAX_OCTAVES], flow_off[HAK_MAX_OCTAVES], tmp_off[HAK_MAX_OCTAVES];
    long arena;                                 // floats per image

    __host__ __device__ long lt(int o, int s) const { return lvl_off[o] + (long)s * oct[o].plane; }
    __host__ __device__ long dxy(int o, int s) const { return lvl_off[o] + (long)(ms + 2 * s) * oct[o].plane; }
};

// Per-image device scalars.
struct HakImgState {
    unsigned int hmax_bits;                     // max Scharr gradient magnitude over the 16-px lattice (float bits), floored at 0.03f
    int hist[HAK_NBINS];
    float kcontrast[HAK_MAX_OCTAVES];           // per octave: k0, k0*0.75, ...
    float ikc[HAK_MAX_OCTAVES];                 // 1/(k*k)
    int ihmax;                                  // FAST path: max integer gradient magnitude over the 16-px lattice (floored at 1)
    int ikcontrast[HAK_MAX_OCTAVES];            // FAST path: integer contrast factor per octave
    int ncand;                                  // entries in the image's extrema candidate list
    int total_pts;                              // NMS survivors before clamping to max_pts
    int num_pts;                                // min(total, max_pts)
    int hist_done;                              // blocks of the histogram pass that have added their bins (the last one finishes the contrast factor)
};

// Read-only tables shared by all images of a context.
struct HakTables {
    float sizes[HAK_MAX_OCTAVES * HAK_MAX_SCALES];     // d_extrema_param sizes, per layer
    float borders[HAK_MAX_OCTAVES * HAK_MAX_SCALES];
    int sigma_size[HAK_MAX_OCTAVES * HAK_MAX_SCALES];
    float orient_w[36];                                // exp(-r2*0.08f)
    int orient_ij[128];                                // k_orient's 109 disc samples in the reference's thread order: i | j << 8 (bytes)
    float orient_gw[128];                              // ... and their weight orient_w[i*i + j*j]; slots 109.. unused (0)
    float fac1, fac2;                                  // dilated-Scharr factors (akazed.cu:2537-2539); FAST: ifac = (int)(fac * 65536 + 0.5f)
    int ifac1, ifac2;
    int comp1[488], comp2[488];                        // MLDB pair table (akazed.cu:65-159)
    unsigned char comp_packed[64 * 16];                // per descriptor byte: 8 x (idx1, idx2) as bytes
    // MLDB sample plan of the configured descriptor_pattern_size (hak_describe_plan, kernels_describe.hip): which sample a
    // lane takes in its n-th turn and where it goes depends on (lane, n) only, never on the keypoint.  [n * 64 + lane]:
    //   dsc_pos : bits 0..7 l = x - size2 (signed), 8..15 k = y - size2 (signed), bit 16 sample exists
    //   dsc_cell: byte g (2x2, 3x3, 4x4 grid) = accumulator row (0x7F none) | 0x80 when the lane's previous sample went to the
    //             same row; bit 24 + g: last sample of this lane in that row
    unsigned int dsc_pos[7 * 64];
    unsigned int dsc_cell[7 * 64];
    int dsc_plan_ok;                                   // 0: pattern too large or a lane revisits a row -> generic k_describe
};
void hak_describe_plan(HakTables* t, int patsize);

// ---------------------------------------------------------------- device helpers
// hScharrContrast as the reference's kernels actually compute it (akazed.cu:827-877, 901-938 / 3245-3336; derivation:
// DESIGN.md 2, D2 / D3):
//   the maximum -- gFindMaxContrastU4's "reduction" compares with absolute pixels of the image's top-left tile, not with the
//   block's other threads, and only thread 0 of each 32 x 32 block feeds atomicMax: what arrives deterministically is the maximum
//   over the pixels x % 16 == 0 && y % 16 == 0 that grid1 = ceil((n / 2) / 16) blocks of 32 cover (an odd n with
//   (n - 1) % 32 == 0 leaves its last column / row out);
//   the histogram -- the guard `ix >= width && iy >= height` returns only when BOTH are outside, so the 32 x 16 blocks' threads
//   right of and below the image count as well, zeros of the reused arena (akaze.cpp:142-149): hak_hist_extra0 entries in bin 0.
__host__ __device__ inline int hak_lattice_cov(int n) { const int c = 32 * ((n / 2 + 15) / 16); return c < n ? c : n; }
__host__ __device__ inline bool hak_on_lattice(int x, int y, int w, int h)
{
    return ((x | y) & 15) == 0 && x < hak_lattice_cov(w) && y < hak_lattice_cov(h);
}
__host__ __device__ inline int hak_hist_extra0(int w, int h) { return ((w + 31) / 32 * 32 - w) * h + ((h + 15) / 16 * 16 - h) * w; }

// reflect-101 as the reference: left/top abs(i), right/bottom borderAdd (akazed.cu:162-170)
__device__ __forceinline__ int hak_refl(int i, int m)
{
    i = i < 0 ? -i : i;
    i = i < m ? i : m + m - 2 - i;
    // tiles hanging over the image can index past one reflection; those values are never used
    i = i < 0 ? 0 : i;
    return i < m ? i : m - 1;
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() lowers to
// `s_waitcnt vmcnt(0) lgkmcnt(0); s_barrier`, i.e. it also drains every outstanding global load and
// store of the wave -- which serialises a register prefetch and exposes the store latency at each
// phase boundary of the persistent tile kernels.  Use this between phases that only exchange data
// through LDS; global results are complete at kernel end as usual.
// XCD-aware block order.  The hardware deals consecutive workgroup ids round-robin to the 8 XCDs (each has its own
// L2).  A 1-D grid of hak_xcd_grid() = nbx * nby * nimg blocks is decoded so that, within every full group of 8
// images, all blocks of one image carry ids congruent mod 8: an image's tiles (which share halos) then run on one
// XCD.  The nimg % 8 images left over are laid out plainly (no padding blocks: a single-image call launches exactly
// its own tiles).  Always returns true (kept as a predicate so callers read `if (!decode) return`).
__device__ __forceinline__ bool hak_xcd_decode(int nbx, int nby, int nimg, int& bx, int& by, int& img)
{
    const int nb = nbx * nby;
    const int full = nimg & ~7;                             // images covered by complete groups of 8
    const int bid = blockIdx.x;
    int t;
    if (bid < full * nb) {
        const int xcd = bid & 7, j = bid >> 3;
        const int g = j / nb;
        t = j - g * nb;
        img = g * 8 + xcd;
    } else {
        const int r = bid - full * nb;
        const int k = r / nb;
        t = r - k * nb;
        img = full + k;
    }
    by = t / nbx;
    bx = t - by * nbx;
    return img < nimg;
}
static inline unsigned hak_xcd_grid(int nbx, int nby, int nimg) { return (unsigned)nimg * (unsigned)nbx * (unsigned)nby; }

__device__ __forceinline__ void hak_lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// ---- dilated Scharr / determinant / key helpers of the Hessian kernels (kernels_hessian.hip, kernels_hessian_stream.hip,
// kernels_scalespace.hip) and of the keypoint refinement.  Shared by both pipelines: V = float (akaze) and V = int
// (fastakaze 16.16 fixed point, akazed.cu:3339-3403: every weighted sum is followed by >> 16; the determinant is not shifted).
__device__ __forceinline__ float hs_d(float f1, float f2, float a, float b) { return f1 * a + f2 * b; }
__device__ __forceinline__ int hs_d(int f1, int f2, int a, int b)
{
    return (int)((unsigned)f1 * (unsigned)a + (unsigned)f2 * (unsigned)b) >> 16;
}
__device__ __forceinline__ float hs_det(float dxx, float dyy, float dxy) { return dxx * dyy - dxy * dxy; }
__device__ __forceinline__ int hs_det(int dxx, int dyy, int dxy)
{
    return (int)((unsigned)dxx * (unsigned)dyy - (unsigned)dxy * (unsigned)dxy);
}
__device__ __forceinline__ unsigned hs_key_bits(float v) { return __float_as_uint(v); }     // positive floats order like their bits
__device__ __forceinline__ unsigned hs_key_bits(int v) { return (unsigned)v; }              // positive ints

// Hessian determinant of one level at (x, y), re-evaluated from the interleaved derivative plane with the expressions and
// the reflect-101 index rule of the Hessian kernels (akazed.cu:1318-1330): bit-identical to the value those kernels
// compared in their extrema test.  The determinant plane itself is never written (hak_internal.h HakLayout).
template <typename V>
__device__ __forceinline__ V hak_det_at(const V* __restrict__ dxy, int x, int y, int S, int w, int h, int p, V f1, V f2)
{
    const long x0 = hak_refl(x - S, w), x1 = x, x2 = hak_refl(x + S, w);
    const long o0 = (long)hak_refl(y - S, h) * p, o1 = (long)y * p, o2 = (long)hak_refl(y + S, h) * p;
    const V xul = dxy[2 * (o0 + x0)], xuc = dxy[2 * (o0 + x1)], xur = dxy[2 * (o0 + x2)];
    const V xcl = dxy[2 * (o1 + x0)], xcr = dxy[2 * (o1 + x2)];
    const V xll = dxy[2 * (o2 + x0)], xlc = dxy[2 * (o2 + x1)], xlr = dxy[2 * (o2 + x2)];
    const V yul = dxy[2 * (o0 + x0) + 1], yuc = dxy[2 * (o0 + x1) + 1], yur = dxy[2 * (o0 + x2) + 1];
    const V yll = dxy[2 * (o2 + x0) + 1], ylc = dxy[2 * (o2 + x1) + 1], ylr = dxy[2 * (o2 + x2) + 1];
    const V dxx = hs_d(f1, f2, xur + xlr - xul - xll, xcr - xcl);
    const V dxy_ = hs_d(f1, f2, xlr + xll - xur - xul, xlc - xuc);
    const V dyy = hs_d(f1, f2, ylr + yll - yur - yul, ylc - yuc);
    return hs_det(dxx, dyy, dxy_);
}

// ------------------------------------------------- deterministic float32 math
// Same operation sequence as the parity oracle's (oracle/okz_math.h): one IEEE
// binary32 op per step, explicit fmaf only.  The reference's libdevice /
// fast-math calls (akazed.cu:1697,1701,1887-1888) are not reproducible.
#define HAK_PI_F  3.14159274101257324f
#define HAK_HPI_F 1.57079637050628662f
#define HAK_PI_D  3.14159265358979323846

__host__ __device__ __forceinline__ void hak_sincosf(float a, float* s_out, float* c_out)
{
    float q = floorf(a * 0.636619747f + 0.5f);
    float r = fmaf(q, -1.57079601287841796875f, a);
    r = fmaf(q, -3.139164786504813217e-7f, r);
    r = fmaf(q, -5.390302529957764765e-15f, r);
    int n = ((int)q) & 3;
    float r2 = r * r;
    float sp = fmaf(r2, -1.9515295891e-4f, 8.3321608736e-3f);
    sp = fmaf(sp, r2, -1.6666654611e-1f);
    float sr = fmaf(r * r2, sp, r);
    float cp = fmaf(r2, 2.443315711809948e-5f, -1.388731625493765e-3f);
    cp = fmaf(cp, r2, 4.166664568298827e-2f);
    float cr = fmaf(r2 * r2, cp, fmaf(r2, -0.5f, 1.0f));
    float s, c;
    if (n == 0)      { s = sr;  c = cr;  }
    else if (n == 1) { s = cr;  c = -sr; }
    else if (n == 2) { s = -sr; c = -cr; }
    else             { s = -cr; c = sr;  }
    *s_out = s;
    *c_out = c;
}

__host__ __device__ __forceinline__ float hak_atan2f(float y, float x)
{
    float ax = fabsf(x), ay = fabsf(y);
    float mx = ax > ay ? ax : ay;
    float mn = ax > ay ? ay : ax;
    if (mx == 0.0f) return 0.0f;
    float a = mn / mx;
    float t, base;
    if (a > 0.4142135679721832275f) {
        t = (a - 1.0f) / (a + 1.0f);
        base = 0.785398185253143310546875f;
    } else {
        t = a;
        base = 0.0f;
    }
    float z = t * t;
    float p = fmaf(8.05374449538e-2f, z, -1.38776856032e-1f);
    p = fmaf(p, z, 1.99777106478e-1f);
    p = fmaf(p, z, -3.33329491539e-1f);
    float r = base + fmaf(p * z, t, t);
    if (ay > ax) r = HAK_HPI_F - r;
    if (x < 0.0f) r = HAK_PI_F - r;
    if (y < 0.0f) r = -r;
    return r;
}

__host__ __device__ __forceinline__ float hak_expf(float x)
{
    if (x < -87.0f) return 0.0f;
    if (x > 88.0f) x = 88.0f;
    float k = floorf(x * 1.44269502162933349609375f + 0.5f);
    float r = fmaf(k, -0.693359375f, x);
    r = fmaf(k, 2.12194440e-4f, r);
    float p = fmaf(1.9875691500e-4f, r, 1.3981999507e-3f);
    p = fmaf(p, r, 8.3334519073e-3f);
    p = fmaf(p, r, 4.1665795894e-2f);
    p = fmaf(p, r, 1.6666665459e-1f);
    p = fmaf(p, r, 5.0000001201e-1f);
    float e = fmaf(p, r * r, r) + 1.0f;
    return ldexpf(e, (int)k);
}

// ---------------------------------------------------------------- launchers
// (defined in the kernel files; all asynchronous on `st`; nimg = batch images,
//  base = arena of image 0, stride = floats between image arenas)
// Kernel-selection knobs of one context (hak_create reads them from the environment once; tests and A/B runs set them
// there).  They never change results, only which of two bit-identical kernels runs.
struct HakKnobs {
    int hess_stream = 1;          // HAK_HESS_STREAM: register-streaming Hessian kernel 0 never / 1 by the size rule / 2 always where it applies
    int base_stream = 1;          // HAK_BASE_STREAM: same for pass A of the octave-0 prologue
    int base_hist = 0;            // HAK_BASE_HIST=1: the streaming prologue finds the (lattice) contrast maximum first and bins the gradient on the
                                  // fly (no gradient plane, no histogram pass).  Off by default: half the bytes, 4 % SLOWER (the pass is bound by
                                  // vector issue; kernels_base_stream.hip)
    int hess_cbuf = 256;          // HAK_HESS_CBUF: staged candidates per block of the tile kernel (1..256; tests drive the overflow path)
    int desc_order = 4;           // HAK_DESC_ORDER: image group size of the describe kernels' block order
    int desc_plan = 1;            // HAK_DESC_PLAN: planned MLDB kernel (k_describe_runs) on / off
    int desc_sort = 1;            // HAK_DESC_SORT: the keypoint kernels visit an image's keypoints level by level (raster order within a
                                  // level) instead of in output order: 0 never / 1 in batches of 8 images and more / 2 always
    int hess_lp = 0;              // HAK_HESS_LP=1: the streaming Hessian low-passes Lt(o,s-1) itself and k_fed_sf stops storing `smooth`.
                                  // Off by default: measured 0.6 ms per 384 x 1080p SLOWER (FED -1.2 ms, Hessian +1.9 ms)
    int level_tile = 1;           // HAK_LEVEL_TILE: one launch per sublevel out of LDS tiles (k_level_tile) 0 never / 1 for launches of at
                                  // most HAK_LEVEL_TILE_MAX_PX pixels unless the streaming kernels are forced / 2 always
};
HakKnobs hak_knobs_from_env();    // defaults overridden by the HAK_* variables (hak_api.hip)

struct HakBatch {
    float* base;                  // arena of image 0
    long stride;                  // floats between consecutive image arenas
    int nimg;
    HakImgState* state;           // [nimg]
    unsigned long long* maps;     // [nimg][map_stride] packed response/layer keys, full resolution
    long map_stride;              // h0 * p0
    unsigned long long* bitmap;   // [nimg][h0][ceil(w0/64)] NMS survivor bits
    int* rowcount;                // [nimg][h0]
    unsigned long long* cand;     // [nimg][cand_cap] extrema candidates: layer<<32 | y<<16 | x (full-res)
    long cand_cap;
    const HakKnobs* knobs = nullptr;   // the owning context's; nullptr (stage operators): hak_knobs_from_env()
    int* perm = nullptr;          // [nimg][perm_cap] visiting order of the keypoint kernels (k_desc_perm), or nullptr: output order
    int perm_cap = 0;
    // per-image clamps of a PAIR call (hak_detect_and_compute_pair: the two AkazeData capacities, akaze.cpp:246, 451); 0: every
    // image is clamped at the launch's max_pts, which is always the record stride between images
    int cap0 = 0, cap1 = 0;
};

// scale space (kernels_scalespace.hip)
void hak_launch_lowpass(hipStream_t st, const float* src, long src_stride, int src_pitch, float* dst, long dst_stride,
                        int w, int h, int p, int nimg, const float* taps, int R);
void hak_launch_down_smooth(hipStream_t st, const float* src, float* dst, float* smooth, long stride,
                            HakOct so, HakOct dd, int nimg, const float* taps);
void hak_launch_contrast(hipStream_t st, const float* smooth, long stride, int w, int h, int p, int nimg,
                         HakImgState* state, float per, int noct);
void hak_launch_reset_state(hipStream_t st, HakImgState* state, int nimg);
void hak_launch_flow(hipStream_t st, const float* src, float* dst, long stride, int w, int h, int p, int nimg,
                     int diffusivity, const HakImgState* state, int octave, float fixed_ikc);
// derivate + determinant of one level, with the level's extrema search fused in when b != nullptr
// (kernels_hessian.hip); returns false when the caller still has to run hak_launch_extrema_level
// The register-streaming kernels (k_hessian_stream, k_fed_sf) need a few thousand waves of >= 32 rows to fill the chip;
// below that (small octaves, single-image calls) the LDS tile kernels are faster.  mode: 0 = never, 1 = by this size
// rule (default), 2 = always where the kernel applies (tests).  Measured on MI355X: 1080p octave 3 at 128 images and
// octave 0 at 1 image favour the tile kernels by 20-40 %, everything above ~2000 waves favours streaming by 6-35 %.
static inline bool hak_stream_pays(int mode, int w, int h, int nimg)
{
    if (mode != 1) return mode != 0;
    const long strips = (w + 239) / 240;
    return strips * ((h + 31) / 32) * nimg >= 2048;
}
// Rows per wave of the register-streaming kernels.  A block's four waves take four consecutive row segments of a strip, so the
// number of segments is a multiple of four (no block with idle waves) and the segments are equally tall (no short last one);
// about 256 rows each amortise the warm-up rows (NS + 4 .. 4S + 2 per segment); more, shorter segments while the grid cannot
// fill the chip.  1080 rows: 4 x 270, 2160: 8 x 270, 540: 4 x 135, 720: 4 x 180.  Measured on 256 x 1080p (FED / Hessian class,
// ms): 9 segments of 128 rows, the ninth 56 rows tall in a block of its own (the rule this replaces): 9.70 / 9.14; 8 x 135:
// 9.16 / 8.32; 4 x 270: 9.08 / 8.31; 12 x 90: 9.43 / 8.49; 16 x 68: 9.63 / 8.84.
// Segments are halved only while the launch has fewer than 2048 waves (round 3; 4096 before; env HAK_STREAM_MIN_WAVES): at 384
// images octave 2 then keeps 4 x 68 rows and octave 3 gets 8 x 17 instead of 16 x 9 -- a 9-row segment spends half its rows
// on warm-up.  FED class per 384 x 1080p images, A/B on one box: 8192: 13.68 ms, 4096: 13.18, 2048: 12.75-12.92, 1536: 12.78-12.85.
static inline int hak_stream_rows(int h, long strips_times_images, int min_rows)
{
    static const long want = [] { const char* e = getenv("HAK_STREAM_MIN_WAVES"); const long v = e ? atol(e) : 2048; return v < 1 ? 1 : v; }();
    int nseg = 4 * ((h + 512) / 1024 > 1 ? (h + 512) / 1024 : 1);
    while (strips_times_images * nseg < want && (h + 2 * nseg - 1) / (2 * nseg) >= min_rows) nseg *= 2;
    const int ry = (h + nseg - 1) / nseg;
    return ry > min_rows ? ry : min_rows;                    // (small images: fewer, not shorter, segments)
}
// dxy: interleaved {Lx, Ly} plane (2 * h * p elements).  det: where the determinant goes -- the fused kernels write it only
// when store_det is set (stage tests); the launch sequence passes a scratch plane that only the dilation > 4 fallback fills.
// lp_taps != nullptr: `src` is Lt(o,s-1) and the kernel applies the sigma=1 low-pass (taps k0, k1, k2) on the way in.
// hak_hessian_stream_covers: the cases the streaming kernel takes (the launch sequence asks before it lets k_fed_sf drop `smooth`)
static inline bool hak_hessian_stream_covers(int w, int h, int step, bool lp)
{
    return step >= 1 && step <= 4 && (w & 3) == 0 && w >= 16 && h >= 2 * step + 2 && (!lp || h >= 8);
}
bool hak_launch_hessian_stream(hipStream_t st, const float* src, float* dxy, float* det, bool store_det, long stride,
                               int w, int h, int p, int nimg, int step, float fac1, float fac2,
                               const HakBatch* b, const HakLayout* L, const HakTables* htab, int octave, int sub, float dthreshold,
                               const float* lp_taps = nullptr);
bool hakf_launch_hessian_stream(hipStream_t st, const int* src, int* dxy, int* det, bool store_det, long stride,
                                int w, int h, int p, int nimg, int step, int fac1, int fac2,
                                const HakBatch* b, const HakLayout* L, const HakTables* htab, int octave, int sub, int idthreshold);
bool hakf_launch_hessian_level(hipStream_t st, const int* src, int* dxy, int* det, bool store_det, long stride,
                               int w, int h, int p, int nimg, int step,
                               const HakBatch* b, const HakLayout* L, const HakTables* htab, int octave, int sub, int idthreshold);
bool hak_launch_hessian_level(hipStream_t st, const float* src, float* dxy, float* det, bool store_det, long stride,
                              int w, int h, int p, int nimg, int step,
                              const HakBatch* b, const HakLayout* L, const HakTables* htab, int octave, int sub, float dthreshold,
                              const float* lp_taps = nullptr);
// octave-0 prologue fused (kernels_base.hip): Lt(0,0) + contrast factors, sigma=1 plane never written
// register-streaming pass A of the octave-0 prologue (kernels_base_stream.hip); false: not covered / does not pay
bool hak_launch_base_stream(hipStream_t st, const float* img, long img_stride, int sp, float* lt, float* grad, long stride, int w, int h,
                            int p, int nimg, const float* taps1, const float* taps_base, int R, HakImgState* state, int mode);
bool hakf_launch_base_stream(hipStream_t st, const unsigned char* img, long img_stride, int sp, int* lt, int* grad, long stride, int w,
                             int h, int p, int nimg, const int* itaps1, const int* itaps_base, int R, HakImgState* state, int mode);
bool hak_launch_base_level(hipStream_t st, const float* img, long img_stride, int sp, float* lt, float* grad_scratch, long stride,
                           int w, int h, int p, int nimg, const float* taps1, const float* taps_base, int R,
                           HakImgState* state, float per, int noct, const HakKnobs& knobs);
// sigma=1 low-pass + conductivity fused (kernels_smoothflow.hip)
void hakf_launch_smooth_flow(hipStream_t st, const int* src, int* smooth, int* flow, long stride,
                             int w, int h, int p, int nimg, const int* itaps, int diffusivity,
                             const HakImgState* state, int octave);
void hak_launch_smooth_flow(hipStream_t st, const float* src, float* smooth, float* flow, long stride,
                            int w, int h, int p, int nimg, const float* taps, int diffusivity,
                            const HakImgState* state, int octave, float fixed_ikc);
// one whole sublevel (low-pass | decimation, conductivity, every FED step) per launch out of LDS tiles, for launches too small
// to fill the chip (kernels_level.hip).  Returns the number of launches (1 up to 36 steps).
#define HAK_LEVEL_TILE_MAX_PX (1920L * 1088L)      // by-size rule: at most one 1080p plane's worth of pixels per launch
// dxy / step / b / L / htab / sub / threshold: the level's Hessian inside the same launch where the cycle is long enough
// (first launch has >= 2 * step steps); *hess_done tells the caller whether it still has to launch the Hessian kernel
struct HakBatch;
int hak_launch_level_tile(hipStream_t st, const float* src, HakOct so, bool head, float* smooth, float* dst, float* tmp, long stride,
                          HakOct dd, int nimg, const float* taps, int diffusivity, const float* tau, int n,
                          const HakImgState* state, int octave, float fixed_ikc,
                          float* dxy = nullptr, int step = 0, const HakBatch* b = nullptr, const HakLayout* L = nullptr,
                          const HakTables* htab = nullptr, int sub = 0, float dthreshold = 0.f, bool* hess_done = nullptr);
int hakf_launch_level_tile(hipStream_t st, const int* src, HakOct so, bool head, int* smooth, int* dst, int* tmp, long stride,
                           HakOct dd, int nimg, const int* itaps, int diffusivity, const float* tau, int n,
                           const HakImgState* state, int octave,
                           int* dxy = nullptr, int step = 0, const HakBatch* b = nullptr, const HakLayout* L = nullptr,
                           const HakTables* htab = nullptr, int sub = 0, int idthreshold = 0, bool* hess_done = nullptr);
// fused FED groups (kernels_fed.hip)
#define HAK_FED_MAX_FUSE 4
bool hakf_launch_base_level(hipStream_t st, const unsigned char* img, long img_stride, int sp, int* lt, int* grad_scratch, long stride,
                            int w, int h, int p, int nimg, const int* itaps1, const int* itaps_base, int R, HakImgState* state,
                            float per, int noct, const HakKnobs& knobs);
int hak_launch_rcp_check(unsigned lo, unsigned hi, unsigned long long* d_bad);
bool hak_launch_fed_sf_head(hipStream_t st, const float* src, HakOct so, float* smooth, float* flow, float* dst, long stride,
                            HakOct dd, int nimg, const float* taps, int diffusivity, const float* tau, int ns,
                            const HakImgState* state, int octave, bool write_g);
bool hakf_launch_fed_sf_head(hipStream_t st, const int* src, HakOct so, int* smooth, int* flow, int* dst, long stride,
                             HakOct dd, int nimg, const int* itaps, int diffusivity, const float* tau, int ns,
                             const HakImgState* state, int octave, bool write_g);
bool hakf_launch_fed_sf(hipStream_t st, const int* src, int* smooth, int* flow, int* dst, long stride,
                        int w, int h, int p, int nimg, const int* itaps, int diffusivity, const float* tau, int ns,
                        const HakImgState* state, int octave, bool write_g);
// store_smooth = false: the low-pass is not written (its only reader, the level's Hessian, runs in the LP variant)
bool hak_launch_fed_sf(hipStream_t st, const float* src, float* smooth, float* flow, float* dst, long stride,
                       int w, int h, int p, int nimg, const float* taps, int diffusivity, const float* tau, int ns,
                       const HakImgState* state, int octave, float fixed_ikc, bool write_g, bool store_smooth = true);
void hakf_launch_fed_group(hipStream_t st, const int* src, const int* flow, int* dst, long stride,
                           int w, int h, int p, int nimg, const float* tau, int ns);
int hak_fed_groups(int n, int max_fuse, int w);
int hak_fed_group_size(int n, int G, int g);
void hak_launch_fed_group(hipStream_t st, const float* src, const float* flow, float* dst, long stride,
                          int w, int h, int p, int nimg, const float* tau, int ns);
void hak_launch_derivate(hipStream_t st, const float* src, float* dxy, long stride,
                         int w, int h, int p, int nimg, int step);
void hak_launch_hessian(hipStream_t st, const float* dxy, float* det, long stride,
                        int w, int h, int p, int nimg, int step);
void hak_deriv_factors(float* fac1, float* fac2);            // akazed.cu:2537-2539
// interleaved {a, b} plane (pitch 2p) -> two dense planes (pitch p); bytes are copied, so it serves both element types
void hak_launch_deinterleave(hipStream_t st, const float* ab, float* a, float* b, int w, int h, int p);

void hak_launch_ingest_u8(hipStream_t st, const unsigned char* src, long src_stride, int sp, float* dst, long dst_stride,
                          int dp, int w, int h, int nimg);

// detector tail (kernels_detect.hip)
void hak_launch_extrema_level(hipStream_t st, const HakBatch& b, const HakLayout& L, const HakTables* tab, int octave,
                              int s, float dthreshold, long det_off);
void hak_launch_download(hipStream_t st, const hak_point* d_points, const int* d_num, long max_pts, int nimg, hak_point* h_points,
                         int* h_num);
// destinations of a pair call's records: device arrays, pinned host arrays (or NULL), capacity of each in records
struct HakPairDst { hak_point* d[2]; hak_point* h[2]; int cap[2]; };
void hak_launch_download_pair(hipStream_t st, const hak_point* src, const int* d_num, long max_pts, const HakPairDst& dst, int* h_num);
void hak_launch_nms_emit(hipStream_t st, const HakBatch& b, const HakLayout& L, const HakTables* tab, int psz,
                         hak_point* points, int max_pts, int* num_out, int fast = 0, int refine = 1);
void hak_launch_clear_maps(hipStream_t st, const HakBatch& b, const HakLayout& L);
void hak_launch_seed_maps(hipStream_t st, const HakBatch& b, const HakLayout& L, const unsigned* d_resp_bits, const int* d_layer);

// integer FAST path (kernels_fast.hip); planes are int32 in the same arena layout
void hakf_launch_reset(hipStream_t st, HakImgState* state, int nimg);
void hakf_launch_conv_u8(hipStream_t st, const unsigned char* src, long src_stride, int sp, int* dst, long dst_stride,
                         int w, int h, int p, int nimg, const int* taps, int R);
void hakf_launch_conv_int(hipStream_t st, const int* src, int* dst, long stride, int w, int h, int p, int nimg, const int* taps, int R);
void hakf_launch_down_smooth(hipStream_t st, const int* src, int* dst, int* smooth, long stride, HakOct so, HakOct dd, int nimg,
                             const int* taps);
void hakf_launch_contrast(hipStream_t st, const int* smooth, long stride, int w, int h, int p, int nimg, HakImgState* state,
                          float per, int noct);
void hakf_launch_flow(hipStream_t st, const int* src, int* dst, long stride, int w, int h, int p, int nimg, int type,
                      const HakImgState* state, int octave);
void hakf_launch_nld_step(hipStream_t st, const int* src, const int* flow, int* dst, long stride, int w, int h, int p, int nimg, float tau);
void hakf_launch_hessian(hipStream_t st, const int* src, int* dxy, int* det, long stride, int w, int h, int p, int nimg, int step);
void hakf_launch_det(hipStream_t st, const int* dxy, int* det, long stride, int w, int h, int p, int nimg, int step);
void hakf_launch_extrema(hipStream_t st, const HakBatch& b, const HakLayout& L, const HakTables* tab, int octave, int s, int threshold,
                         long det_off);
void hakf_launch_describe(hipStream_t st, const HakBatch& b, const HakLayout& L, const HakTables* tab, hak_point* points, int max_pts,
                          int patsize, int upright, int desc, int planned);

// descriptors (kernels_describe.hip)
// orient = 0 (stage tests): skip k_orient and describe with the angles the records hold
void hak_launch_describe(hipStream_t st, const HakBatch& b, const HakLayout& L, const HakTables* tab,
                         hak_point* points, int max_pts, int patsize, int upright, int desc, int planned, int orient = 1);

// bandwidth probes (kernels_probe.hip)
#define HAK_COPY_SHAPES 26                  // 3 x 2 x 4 copy shapes + read-only + write-only (kernels_probe.hip)
int hak_launch_copy_probe(long bytes, int iters, double* ms_per_copy, double* shapes_ms = nullptr);
int hak_launch_gather_probe(long bytes, int blocks, int per_lane, int iters, double* ms_per_launch);
int hak_launch_stream_probe(int w, int h, int nimg, int nwrite, int warm, int iters, double* ms, double* bytes);
int hak_launch_hess_probe(int w, int h, int nimg, int step, int iters, double* ms, double* bytes);

// matcher (kernels_match.hip)
// Scratch of the SLICED searches (one big pair through hak_match / hak_match_knn2, e.g. 10k x 10k: the train set is cut into
// slices so that the query blocks x slices fill the chip).  Owned by a context or handed out by the per-device pool of
// hak_api.hip (ctx == NULL: cuMatch is a free function in the reference); lives on one device, used by one call at a time.
// Invariant between calls: every key is 0xFFFFFFFF and every ticket 0 -- the kernels restore that themselves (the block that
// draws a query block's last ticket reads the merged keys with exchanges and resets the ticket), so no call starts with a memset.
struct HakMatchScratch {
    unsigned* keys = nullptr; long keys_cap = 0;        // [queries][16] class minima, merged with atomicMin          (1-NN)
    int* ticket = nullptr; long ticket_cap = 0;         // [query blocks of 128]
    uint2* part = nullptr; long part_cap = 0;           // [slices][queries padded to 128] two smallest keys per slice (2-NN)
    int4* knn = nullptr; long knn_cap = 0;              // forward | reverse 2-NN results of hak_match_knn2
    int* blk = nullptr; long blk_cap = 0;               // accepted matches per 1024-query block (compaction of hak_match_knn2)
    int* d_cnt = nullptr; int* h_cnt = nullptr;         // accepted-match count: device word and its pinned host mirror
    int device = -1;
};
// grow-only; returns false (and leaves a usable smaller state) when the device is out of memory.  `st`: the stream earlier
// users of the buffers ran on (synchronised before a buffer is replaced).
bool hak_match_scratch_reserve(HakMatchScratch* sc, hipStream_t st, long keys, long tickets, long parts, long knn, long blks);
void hak_match_scratch_free(HakMatchScratch* sc);
void hak_launch_match(hipStream_t st, hak_point* pts1, const hak_point* pts2, const int* n1_dev, const int* n2_dev,
                      int n1_host, int n2_host, long pair_stride1, long pair_stride2, int npairs,
                      HakMatchScratch* scratch = nullptr);
void hak_launch_knn2(hipStream_t st, const hak_point* ptsA, const hak_point* ptsB, const int* nA_dev, const int* nB_dev,
                     int nA_host, int nB_host, long strideA, long strideB, int npairs, int4* out, long out_stride,
                     HakMatchScratch* scratch = nullptr);
void hak_launch_knn2_finish(hipStream_t st, hak_point* pts1, const hak_point* pts2, const int* n1_dev, int n1_host, long stride1,
                            long stride2, int npairs, const int4* fwd, const int4* rev, long knn_stride, int ratio_num,
                            int ratio_den, int cross, int max_dist, hak_match_pair* out, long out_stride, int* out_count,
                            HakMatchScratch* scratch = nullptr);

// A launcher that cannot do what it was asked (a precondition its caller should have checked) records the reason here instead
// of aborting; enqueue_detect turns it into the call's error (hak_api.hip).  Thread-local, like hak_last_error().
void hak_note_launch_error(const char* msg);
