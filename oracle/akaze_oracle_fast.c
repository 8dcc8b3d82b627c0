/*
 * akaze_oracle_fast.c -- CPU restatement of the reference's integer "FAST" path
 * (Akazer::fastDetectAndCompute, namespace fastakaze): the float pipeline in int32 with 16.16
 * fixed-point weights, uint8 input in [0,255].
 *
 * TEST INFRASTRUCTURE ONLY (see oracle/README.md).  The stage functions are exported so that the literal micro-fixture
 * tests (tests/test_reference_literal_cpu.py) can drive them one by one.  Each function cites the reference file:line it
 * follows.  Same deterministic choices D2-D10 as akaze_oracle.c; additionally:
 *   F1  32-bit products wrap (two's complement) exactly as the device's v_mul_lo_u32 does; the
 *       reference's `int * int` can overflow for large tau (stepfac * step, akazed.cu:3465).
 *   F2  the reference's __expf / __cosf / __sinf are replaced by the oracle's deterministic
 *       okz_expf / okz_sincosf (as in the float path); the per-sample angle uses dFastAtan2 as the
 *       source does (akazed.cu:3685), which is plain float arithmetic.
 *   F3  fastDetect (akaze.cpp:506-743) cannot even be compiled without OpenCV (it creates cv::Mat
 *       unconditionally); the algorithm below follows its device calls only.
 * Parity of this path is unpinned against a CUDA run for the same reasons as the float path.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "okz_math.h"

#define FK_FLEN 61
#define FK_NBINS 300
#define FK_MAX_OCT 8
#define FK_MAX_SCALE 5

typedef struct FkPoint {
    float x, y;
    int octave;
    float response, size, angle;
    unsigned char features[FK_FLEN];
    int match, distance;
    float match_x, match_y;
} FkPoint;

typedef struct FkParams {
    int noctaves, max_scale;
    float per, kcontrast, soffset;
    int reordering;
    float derivative_factor, dthreshold;
    int diffusivity, descriptor_pattern_size;
    int upright;
} FkParams;

int okz_fed_tau(float T, int M, float tau_max, int reordering, float* tau, int cap);
void okz_compare_indices(int* idx1, int* idx2);
extern int okz_reading_variant;   /* akaze_oracle.c: alternative readings for tools/ref_render_check.py; 0 everywhere else */
int okz_layout(int w, int h, int p, int noctaves, int max_scale, int* owhps, int* osizes, int* offsets);

static inline int fborder_add(int a, int b, int m) { int c = a + b; return c < m ? c : m + m - 2 - c; }
static inline int fiabs(int a) { return a < 0 ? -a : a; }
static inline int wmul(int a, int b) { return (int)((unsigned)a * (unsigned)b); }   /* F1 */
static inline int wadd(int a, int b) { return (int)((unsigned)a + (unsigned)b); }
static inline int wsub(int a, int b) { return (int)((unsigned)a - (unsigned)b); }
static inline int wneg(int a) { return (int)(0u - (unsigned)a); }
/* (every +, - and * on plane values goes through these: the device wraps, C leaves signed overflow undefined -- and the planes do
 * leave the int range once a long FED cycle has blown a coarse level up; `make asan` runs such a case under UBSan) */
static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
/* float -> int as the device does it (cvt.rzi.s32.f32 / v_cvt_i32_f32: toward zero, SATURATING, NaN -> 0).  A C cast of a NaN or of
 * a value outside the int range is undefined and x86 delivers INT_MIN for all of them.  It matters where the integer pipeline has
 * left its range: a long FED cycle at a coarse level (tau up to ~50 against 16-bit truncations) can blow the plane up, the wrapped
 * sum of squares of gFlowNaive (akazed.cu:3427) turns NEGATIVE, and the conductivity is then sqrt of a negative number (Charbonnier:
 * NaN) or exp of a huge one (PM_G1: inf) -- found by tests/fuzz_parity.py, round 5; likewise the rotated derivatives of gDescribe2
 * (akazed.cu:3779-3780). */
static inline int d2i_sat(double v)
{
    if (v != v) return 0;
    if (v >= 2147483648.0) return 2147483647;
    if (v <= -2147483648.0) return (int)0x80000000;
    return (int)v;
}
static inline int f2i_sat(float v)
{
    if (v != v) return 0;
    if (v >= 2147483648.0f) return 2147483647;
    if (v <= -2147483648.0f) return (int)0x80000000;
    return (int)v;
}

/* akazed.cu:3855-3900 fastakaze::createGaussKernel: ik[i] = (int)(k[i]*65536 + 0.5f) */
void fkz_gauss_taps(float var, int radius, int* ik)
{
    float k[8];
    float denom = 1.f / (2.f * var);
    float ksum = 0;
    for (int i = 0; i <= radius; i++) {
        k[i] = expf(-i * i * denom);
        ksum += (i == 0) ? k[i] : k[i] + k[i];
    }
    ksum = 1 / ksum;
    for (int i = 0; i <= radius; i++) {
        k[i] *= ksum;
        ik[i] = (int)(k[i] * 65536 + 0.5f);
    }
}

/* akazed.cu:2990-3075 gConv2d<R> (uchar src) and 2786-2850 / 2922-2985 gConv2dR2 (uchar / int src):
 * row pass `(k0*c + sum k_i*(l+r)) >> 16`, column pass the same on the row results */
static void conv_rows_cols(const int* in, int* dst, int w, int h, int p, const int* k, int R)
{
    int* rows = (int*)malloc(sizeof(int) * (size_t)w * h);
#pragma omp parallel for schedule(static)
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            const int* s = in + (size_t)y * p;
            int ws = wmul(k[0], s[x]);
            for (int i = 1; i <= R; i++) ws = wadd(ws, wmul(k[i], wadd(s[fiabs(x - i)], s[fborder_add(x, i, w)])));
            rows[(size_t)y * w + x] = ws >> 16;
        }
#pragma omp parallel for schedule(static)
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            int ws = wmul(k[0], rows[(size_t)y * w + x]);
            for (int i = 1; i <= R; i++)
                ws = wadd(ws, wmul(k[i], wadd(rows[(size_t)fiabs(y - i) * w + x], rows[(size_t)fborder_add(y, i, h) * w + x])));
            dst[(size_t)y * p + x] = ws >> 16;
        }
    free(rows);
}

void fkz_conv_u8(const unsigned char* src, int sp, int* dst, int w, int h, int p, const int* k, int R)
{
    int* tmp = (int*)malloc(sizeof(int) * (size_t)p * h);
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) tmp[(size_t)y * p + x] = src[(size_t)y * sp + x];
    conv_rows_cols(tmp, dst, w, h, p, k, R);
    free(tmp);
}

void fkz_conv_int(const int* src, int* dst, int w, int h, int p, const int* k, int R)
{
    conv_rows_cols(src, dst, w, h, p, k, R);
}

/* akazed.cu:3143-3205 fastakaze::gDownWithSmooth */
void fkz_down_smooth(const int* src, int* dst, int* smooth, int sw, int sh, int sp, int dw, int dh, int dp, const int* k)
{
    int* rows = (int*)malloc(sizeof(int) * (size_t)dw * sh);
    for (int sy = 0; sy < sh; sy++)
        for (int dx = 0; dx < dw; dx++) {
            const int* s = src + (size_t)sy * sp;
            int six = dx + dx;
            int x0 = fiabs(six - 4), x1 = fiabs(six - 2), x3 = fborder_add(six, 2, sw), x4 = fborder_add(six, 4, sw);
            rows[(size_t)sy * dw + dx] = wadd(wadd(wmul(k[0], s[six]), wmul(k[1], wadd(s[x1], s[x3]))), wmul(k[2], wadd(s[x0], s[x4]))) >> 16;
        }
    for (int dy = 0; dy < dh; dy++) {
        int siy = dy + dy;
        int y0 = fiabs(siy - 4), y1 = fiabs(siy - 2), y3 = fborder_add(siy, 2, sh), y4 = fborder_add(siy, 4, sh);
        for (int dx = 0; dx < dw; dx++) {
            dst[(size_t)dy * dp + dx] = src[(size_t)siy * sp + dx + dx];
            smooth[(size_t)dy * dp + dx] =
                wadd(wadd(wmul(k[0], rows[(size_t)siy * dw + dx]),
                          wmul(k[1], wadd(rows[(size_t)y1 * dw + dx], rows[(size_t)y3 * dw + dx]))),
                     wmul(k[2], wadd(rows[(size_t)y0 * dw + dx], rows[(size_t)y4 * dw + dx]))) >> 16;
        }
    }
    free(rows);
}

static inline void fscharr(const int* src, int x, int y, int w, int h, int p, int* dx, int* dy)
{
    int x0 = fiabs(x - 1), x2 = fborder_add(x, 1, w), y0 = fiabs(y - 1), y2 = fborder_add(y, 1, h);
    const int* r0 = src + (size_t)y0 * p;
    const int* r1 = src + (size_t)y * p;
    const int* r2 = src + (size_t)y2 * p;
    *dx = wadd(wmul(10, wsub(r1[x2], r1[x0])), wmul(3, wsub(wsub(wadd(r0[x2], r2[x2]), r0[x0]), r2[x0])));
    *dy = wadd(wmul(10, wsub(r2[x], r0[x])), wmul(3, wsub(wsub(wadd(r2[x0], r2[x2]), r0[x0]), r0[x2])));
}

/* akazed.cu:3208-3232 gScharrContrastNaive + 4098-4165 hScharrContrast + 3245-3296 gFindMaxContrastU4 + 3299-3336
 * gConstrastHistShared: the same two kernels as the float path on int32 (akaze_oracle.c, okz_kcontrast, has the derivation).
 * D2: hmax = max(1, maximum over the lattice x % 16 == 0 && y % 16 == 0 inside grid1's coverage), the deterministic core;
 * D3: the threads beyond the right / bottom edge count zeros of the zeroed arena into bin 0.  A wrapped-negative bin index
 * (grad * hfactor overflows int32 when grad > ~109 hmax: possible now that hmax is a subsample) is an out-of-bounds shared
 * atomicAdd in the reference (:3321-3326 checks only the upper end); counted in bin 0 here. */
int fkz_kcontrast(const int* smooth, int w, int h, int p, float per, int* hmax_out, int* hist_out)
{
    int* grad = (int*)malloc(sizeof(int) * (size_t)w * h);
    int hmax = 1;                                                       /* :4101 */
    const int full = okz_reading_variant & 2;
    const int lat = full ? 1 : 16;
    const int wcov = full ? w : (32 * ((w / 2 + 15) / 16) < w ? 32 * ((w / 2 + 15) / 16) : w);   /* grid1 :4122 */
    const int hcov = full ? h : (32 * ((h / 2 + 15) / 16) < h ? 32 * ((h / 2 + 15) / 16) : h);
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            int dx, dy;
            fscharr(smooth, x, y, w, h, p, &dx, &dy);
            int g = f2i_sat(sqrtf((float)wadd(wmul(dx, dx), wmul(dy, dy))) + 0.5f);   /* :3231 */
            grad[(size_t)y * w + x] = g;
            if (g > hmax && x % lat == 0 && y % lat == 0 && x < wcov && y < hcov) hmax = g;   /* :3245-3296 */
        }
    int hist[FK_NBINS];
    memset(hist, 0, sizeof(hist));
    int hfactor = (int)(FK_NBINS / (float)hmax * 65536 + 0.5f);          /* :4133 */
    for (size_t i = 0; i < (size_t)w * h; i++) {
        int hi = wmul(grad[i], hfactor) >> 16;                          /* :3319 */
        if (hi >= FK_NBINS) hi = FK_NBINS - 1;
        if (hi < 0) hi = 0;
        hist[hi]++;
    }
    free(grad);
    if (!(okz_reading_variant & 4))                                     /* :3305 with grid2 :4142 */
        hist[0] += ((w + 31) / 32 * 32 - w) * h + ((h + 15) / 16 * 16 - h) * w;
    int thresh = (int)((w * h - hist[0]) * per);
    int cumuv = 0, k = 1;
    while (k < FK_NBINS) {
        if (cumuv >= thresh) break;
        cumuv += hist[k];
        k++;
    }
    if ((okz_reading_variant & 8) && k < FK_NBINS) k++;                 /* power table only */
    if (hmax_out) *hmax_out = hmax;
    if (hist_out) memcpy(hist_out, hist, sizeof(hist));
    return k * hmax / FK_NBINS;                                         /* :4162 */
}

/* akazed.cu:3406-3445 gFlowNaive + 4209-4216 hFlow: conductivity as 16.16 int */
void fkz_flow(const int* src, int* dst, int type, int kcontrast, int w, int h, int p)
{
    float ikc = 1.f / (kcontrast * kcontrast);
#pragma omp parallel for schedule(static)
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            int dx, dy;
            fscharr(src, x, y, w, h, p, &dx, &dy);
            float dif2 = wadd(wmul(dx, dx), wmul(dy, dy)) * ikc;
            float g;
            if (type == 0) g = okz_expf(-dif2);
            else if (type == 1) g = 1.f / (1.f + dif2);
            else if (type == 2) { float d2 = dif2 * dif2; g = 1.f - okz_expf(-3.315f / (d2 * d2)); }
            else g = 1.f / sqrtf(1.f + dif2);
            dst[(size_t)y * p + x] = f2i_sat(g * 65536 + 0.5f);         /* :3431-3443: a device cast (NaN -> 0, saturating): dif2 < 0 once the sum of squares has wrapped */
        }
}

/* akazed.cu:3448-3470 gNldStepNaive + 4231-4238 hNldStep */
void fkz_nld_step(const int* src, const int* flow, int* dst, float tau, int w, int h, int p)
{
    int stepfac = (int)(0.5f * tau * 65536 + 0.5f);
#pragma omp parallel for schedule(static)
    for (int y = 0; y < h; y++) {
        int y0 = fiabs(y - 1), y2 = fborder_add(y, 1, h);
        const int *s0 = src + (size_t)y0 * p, *s1 = src + (size_t)y * p, *s2 = src + (size_t)y2 * p;
        const int *f0 = flow + (size_t)y0 * p, *f1 = flow + (size_t)y * p, *f2 = flow + (size_t)y2 * p;
        for (int x = 0; x < w; x++) {
            int x0 = fiabs(x - 1), x2 = fborder_add(x, 1, w);
            int step = wadd(wadd(wadd(wmul(wadd(f1[x], f1[x2]), wsub(s1[x2], s1[x])), wmul(wadd(f1[x], f1[x0]), wsub(s1[x0], s1[x]))),
                                 wmul(wadd(f1[x], f2[x]), wsub(s2[x], s1[x]))), wmul(wadd(f1[x], f0[x]), wsub(s0[x], s1[x]))) >> 16;
            dst[(size_t)y * p + x] = wadd(wmul(stepfac, step) >> 16, s1[x]);
        }
    }
}

/* akazed.cu:4175-4195 hHessianDeterminant factors; 3339-3403 gDerivate / gHessianDeterminant */
void fkz_deriv_factors(int* f1, int* f2)
{
    float w = 10.f / 3.f;
    float fac1 = 1.f / (2.f * (w + 2.f));
    float fac2 = w * fac1;
    *f1 = (int)(fac1 * 65536 + 0.5f);
    *f2 = (int)(fac2 * 65536 + 0.5f);
}

void fkz_hessian(const int* src, int* dxo, int* dyo, int* det, int step, int w, int h, int p)
{
    int fac1, fac2;
    fkz_deriv_factors(&fac1, &fac2);
#pragma omp parallel for schedule(static)
    for (int y = 0; y < h; y++) {
        const int* r0 = src + (size_t)fiabs(y - step) * p;
        const int* r1 = src + (size_t)y * p;
        const int* r2 = src + (size_t)fborder_add(y, step, h) * p;
        for (int x = 0; x < w; x++) {
            int x0 = fiabs(x - step), x2 = fborder_add(x, step, w);
            int ul = r0[x0], uc = r0[x], ur = r0[x2], cl = r1[x0], cr = r1[x2], ll = r2[x0], lc = r2[x], lr = r2[x2];
            dxo[(size_t)y * p + x] = wadd(wmul(fac1, wsub(wsub(wadd(ur, lr), ul), ll)), wmul(fac2, wsub(cr, cl))) >> 16;
            dyo[(size_t)y * p + x] = wadd(wmul(fac1, wsub(wsub(wadd(lr, ll), ur), ul)), wmul(fac2, wsub(lc, uc))) >> 16;
        }
    }
#pragma omp parallel for schedule(static)
    for (int y = 0; y < h; y++) {
        size_t o0 = (size_t)fiabs(y - step) * p, o1 = (size_t)y * p, o2 = (size_t)fborder_add(y, step, h) * p;
        for (int x = 0; x < w; x++) {
            int x0 = fiabs(x - step), x2 = fborder_add(x, step, w);
            const int* dx = dxo; const int* dy = dyo;
            int dxx = wadd(wmul(fac1, wsub(wsub(wadd(dx[o0 + x2], dx[o2 + x2]), dx[o0 + x0]), dx[o2 + x0])), wmul(fac2, wsub(dx[o1 + x2], dx[o1 + x0]))) >> 16;
            int dxy = wadd(wmul(fac1, wsub(wsub(wadd(dx[o2 + x2], dx[o2 + x0]), dx[o0 + x2]), dx[o0 + x0])), wmul(fac2, wsub(dx[o2 + x], dx[o0 + x]))) >> 16;
            int dyy = wadd(wmul(fac1, wsub(wsub(wadd(dy[o2 + x2], dy[o2 + x0]), dy[o0 + x2]), dy[o0 + x0])), wmul(fac2, wsub(dy[o2 + x], dy[o0 + x]))) >> 16;
            det[o1 + x] = wsub(wmul(dxx, dyy), wmul(dxy, dxy));
        }
    }
}

/* akazed.cu:3476-3515 gCalcExtremaMap (int), sublevels ascending (D5) */
void fkz_extrema(const int* dets, int* rmap, float* smap, int* lmap, const float* params, int octave,
                        int ms, int threshold, int w, int h, int p, int opitch)
{
    int psz = (int)params[0];
    for (int s = 0; s < ms; s++) {
        float border = params[s], size = params[ms + s];
        const int* det = dets + (size_t)s * h * p;
        for (int iy = psz; iy < h; iy++) {
            if ((int)(iy - border + 0.5f) - 1 < 0 || (int)(iy + border + 0.5f) + 1 >= h) continue;
            for (int ix = psz; ix < w; ix++) {
                if ((int)(ix - border + 0.5f) - 1 < 0 || (int)(ix + border + 0.5f) + 1 >= w) continue;
                const int* vp = det + (size_t)iy * p + ix;
                int v = *vp;
                if (v > threshold && v > vp[-p] && v > vp[p] && v > vp[-1] && v > vp[1] && v > vp[-p - 1] &&
                    v > vp[-p + 1] && v > vp[p - 1] && v > vp[p + 1]) {
                    size_t o = (size_t)(iy << octave) * opitch + (size_t)(ix << octave);
                    if (rmap[o] < v) { rmap[o] = v; smap[o] = size; lmap[o] = octave * ms + s; }
                }
            }
        }
    }
}

/* akazed.cu:3538-3598 gNmsRNaive (int), raster order (D6), response filled (D8) */
int fkz_nms(FkPoint* pts, int max_pts, const int* rmap, const float* smap, const int* lmap, int psz, int w, int h, int p)
{
    int n = 0;
    for (int iy = psz; iy + psz < h; iy++)
        for (int ix = psz; ix + psz < w; ix++) {
            size_t idx = (size_t)iy * p + ix;
            if (lmap[idx] < 0) continue;
            float fsz = smap[idx];
            int isz = (int)(fsz + 0.5f), sqsz = (int)(fsz * fsz), to_nms = 0;
            for (int i = -isz; i <= isz && !to_nms; i++) {
                /* :3562-3577 same cursor as the float kernel: `continue` at the centre skips `new_idx++`, so on
                 * row i == 0 every j > 0 reads column ix + j - 1 (Q1) */
                int col = ix - isz;
                for (int j = -isz; j <= isz; j++) {
                    if (i == 0 && j == 0) { col += okz_reading_variant & 1; continue; }        /* :3565 */
                    int rn = rmap[(size_t)(iy + i) * p + col];
                    if (i * i + j * j < sqsz && rn > -1000000 && (rn > rmap[idx] || (rn == rmap[idx] && i <= 0 && j <= 0))) to_nms = 1;
                    col++;                                                                     /* :3577 */
                }
            }
            if (!to_nms) {
                if (n < max_pts) {
                    FkPoint* pt = pts + n;
                    pt->x = (float)ix; pt->y = (float)iy; pt->octave = lmap[idx]; pt->size = smap[idx];
                    pt->response = (float)rmap[idx];
                }
                n++;
            }
        }
    return n;
}

/* akazed.cu:3600-3646 gRefine (int det, float offsets) */
void fkz_refine(FkPoint* pt, const int* det, int o, int p)
{
    int y = (int)pt->y >> o, x = (int)pt->x >> o;
    size_t idx = (size_t)y * p + x;
    int v2 = wadd(det[idx], det[idx]);
    int dx = wsub(det[idx + 1], det[idx - 1]) >> 1;
    int dy = wsub(det[idx + p], det[idx - p]) >> 1;
    int dxx = wsub(wadd(det[idx + 1], det[idx - 1]), v2);
    int dyy = wsub(wadd(det[idx + p], det[idx - p]), v2);
    int dxy = wsub(wsub(wadd(det[idx + p + 1], det[idx - p - 1]), det[idx - p + 1]), det[idx + p - 1]) >> 2;
    int dd = wsub(wmul(dxx, dyy), wmul(dxy, dxy));
    float idd = dd != 0 ? (1.f / dd) : 0.f;
    float dst0 = idd * wsub(wmul(dxy, dy), wmul(dyy, dx));
    float dst1 = idd * wsub(wmul(dxy, dx), wmul(dxx, dy));
    if (dst0 < -1.f || dst0 > 1.f || dst1 < -1.f || dst1 > 1.f) return;
    int ratio = 1 << o;
    pt->y = ratio * (y + dst1);
    pt->x = ratio * (x + dst0);
}

static inline float fast_atan2(float y, float x)                       /* akazed.cu:173-185 (0/0 -> 0) */
{
    float absx = fabsf(x), absy = fabsf(y);
    float mn = absx < absy ? absx : absy, mx = absx < absy ? absy : absx;
    float a = mx > 0.f ? mn / mx : 0.f;
    float s = a * a;
    float r = fmaf(fmaf(fmaf(-0.0464964749f, s, 0.15931422f), s, -0.327622764f), s * a, a);
    r = (absy > absx ? OKZ_HPI_F - r : r);
    r = (x < 0 ? (float)(OKZ_PI_D - r) : r);
    r = (y < 0 ? -r : r);
    return r;
}

/* akazed.cu:3649-3718 gCalcOrient (int planes; per-sample angle by dFastAtan2) */
void fkz_orient(FkPoint* pt, const int* dxd, const int* dyd, int o, int w, int h, int p, const float* wtab)
{
    float resx[42], resy[42], re8x[42], re8y[42];
    for (int t = 0; t < 42; t++) { resx[t] = 0.f; resy[t] = 0.f; }
    int step = f2i_sat(pt->size + 0.5f);
    int x = f2i_sat(pt->x + 0.5f) >> o, y = f2i_sat(pt->y + 0.5f) >> o;
    for (int tix = 0; tix < 13 * 16; tix++) {
        int i = (tix & 15) - 6, j = (tix / 16) - 6, r2 = i * i + j * j;
        if (r2 >= 36) continue;
        size_t pos = (size_t)clampi(y + step * j, 0, h - 1) * p + clampi(x + step * i, 0, w - 1);
        float dx = wtab[r2] * dxd[pos], dy = wtab[r2] * dyd[pos];
        float angle = fast_atan2(dy, dx);
        int a = d2i_sat(angle * (21 / OKZ_PI_D)) + 21;                 /* (NaN for a sample without gradient: 0 on the device) */
        a = a > 41 ? 41 : a; a = a < 0 ? 0 : a;
        resx[a] += dx; resy[a] += dy;
    }
    for (int t = 0; t < 42; t++) {
        re8x[t] = resx[t]; re8y[t] = resy[t];
        for (int k = t + 1; k < t + 7; k++) { re8x[t] += resx[k < 42 ? k : k - 42]; re8y[t] += resy[k < 42 ? k : k - 42]; }
    }
    float maxr = 0.0f; int maxk = 0;
    for (int k = 0; k < 42; k++) { float r = re8x[k] * re8x[k] + re8y[k] * re8y[k]; if (r > maxr) { maxr = r; maxk = k; } }
    float r = fast_atan2(re8y[maxk], re8x[maxk]);
    pt->angle = (r < 0.0f ? (float)(r + 2.0f * OKZ_PI_D) : r);
}

/* akazed.cu:3723-3850 gDescribe2 (int accumulators: sums are exact and order-free) */
void fkz_describe(FkPoint* pt, const int* imd, const int* dxd, const int* dyd, int o, int w, int h, int p,
                         int patsize, const int* idx1, const int* idx2)
{
    int acc[90];
    memset(acc, 0, sizeof(acc));
    int size2 = patsize, size3 = (int)ceilf(2.0f * patsize / 3.0f), size4 = (int)ceilf(0.5f * patsize);
    float iratio = 1.f / (1 << o);
    int scale = f2i_sat(pt->size + 0.5f);
    float xf = pt->x * iratio, yf = pt->y * iratio, co, si;
    okz_sincosf(pt->angle, &si, &co);
    int winsize = 3 * size3 > 4 * size4 ? 3 * size3 : 4 * size4;
    for (int i = 0; i < winsize * winsize; i++) {
        int y = i / winsize, x = i - winsize * y, m = x > y ? x : y;
        int l = x - size2, k = y - size2;
        int xp = clampi(f2i_sat(xf + scale * (k * co - l * si) + 0.5f), 0, w - 1);
        int yp = clampi(f2i_sat(yf + scale * (k * si + l * co) + 0.5f), 0, h - 1);
        size_t pos = (size_t)yp * p + xp;
        int im = imd[pos], dx = dxd[pos], dy = dyd[pos];
        int rx = f2i_sat(wneg(dx) * si + dy * co);                           /* akazed.cu:3777 */
        int ry = f2i_sat(dx * co + dy * si);
        if (m < 2 * size2) { int c = 3 * ((y < size2 ? 0 : 2) + (x < size2 ? 0 : 1)); acc[c] = wadd(acc[c], im); acc[c + 1] = wadd(acc[c + 1], rx); acc[c + 2] = wadd(acc[c + 2], ry); }
        if (m < 3 * size3) {
            int x3 = (x < size3 ? 0 : (x < 2 * size3 ? 1 : 2)), y3 = (y < size3 ? 0 : (y < 2 * size3 ? 1 : 2));
            int c = 3 * (4 + y3 * 3 + x3); acc[c] = wadd(acc[c], im); acc[c + 1] = wadd(acc[c + 1], rx); acc[c + 2] = wadd(acc[c + 2], ry);
        }
        if (m < 4 * size4) {
            int x4 = (x < 2 * size4 ? (x < size4 ? 0 : 1) : (x < 3 * size4 ? 2 : 3)), y4 = (y < 2 * size4 ? (y < size4 ? 0 : 1) : (y < 3 * size4 ? 2 : 3));
            int c = 3 * (13 + y4 * 4 + x4); acc[c] = wadd(acc[c], im); acc[c + 1] = wadd(acc[c + 1], rx); acc[c + 2] = wadd(acc[c + 2], ry);
        }
    }
    for (int b = 0; b < FK_FLEN; b++) {
        unsigned char d = 0;
        for (int i = 0; i < (b == 60 ? 6 : 8); ++i) d |= (unsigned char)((acc[idx1[b * 8 + i]] > acc[idx2[b * 8 + i]] ? 1 : 0) << i);
        pt->features[b] = d;
    }
}

long fkz_arena_ints(int w, int h, int p, int noctaves, int max_scale)
{
    int owhps[3 * FK_MAX_OCT], osizes[FK_MAX_OCT], offsets[FK_MAX_OCT + 1];
    int n = okz_layout(w, h, p, noctaves, max_scale, owhps, osizes, offsets);
    return offsets[n];
}

/* Akazer::fastDetectAndCompute akaze.cpp:153-201 + fastDetect 506-743.  image: uint8, pitch sp bytes.
 * arena: fkz_arena_ints() ints (layout SURVEY 9.1 with int planes). Returns num_pts. */
int fkz_detect_and_compute(const unsigned char* image, int w, int h, int sp, int p, const FkParams* prm, FkPoint* pts,
                           int max_pts, int desc, int* tmem, int* kcontrast_out)
{
    int ms = prm->max_scale;
    int owhps[3 * FK_MAX_OCT], osizes[FK_MAX_OCT], offsets[FK_MAX_OCT + 1];
    int noct = okz_layout(w, h, p, prm->noctaves, ms, owhps, osizes, offsets);
    int* rmap = tmem;
    float* smap = (float*)(tmem + osizes[0]);
    int* lmap = tmem + 2 * osizes[0];
    for (int i = 0; i < osizes[0]; i++) { rmap[i] = (int)0xC0C0C0C0; smap[i] = -1e6f; lmap[i] = -1; }   /* akaze.cpp:520-525 */
    int k1[3], kbase[8];
    fkz_gauss_taps(1.f, 2, k1);
    float tmax = 0.25f, esigma = prm->soffset;
    float last_etime = (float)(0.5 * prm->soffset * prm->soffset), curr_etime = 0, ttime = 0;
    int naux = 0, oratio = 1, sigma_size = 0, mstep = 0;
    float smax = (float)(10.0 * sqrtf(2.0f));
    float params[2 * FK_MAX_SCALE + 2];
    float* borders = params; float* sizes = params + ms;
    float psz = 10000;
    float tau[4096];
    int ikcontrast = 1, idthreshold = 65;                                /* akaze.cpp:558-559 */
    for (int i = 0; i < noct; i++) {
        int ow = owhps[3 * i], oh = owhps[3 * i + 1], op = owhps[3 * i + 2], msz = osizes[i], ms_msz = msz * ms;
        int* nldimg = tmem + offsets[i];
        int* smooth = nldimg + ms_msz; int* flow = smooth + ms_msz; int* temp = flow + ms_msz;
        int* dx = flow; int* dy = temp;
        for (int j = 0; j < ms; j++) {
            if (j == 0 && i == 0) {                                     /* akaze.cpp:589-623 */
                float var = prm->soffset * prm->soffset;
                int ksz = (int)(2 * ceilf((prm->soffset - 0.8f) / 0.3f) + 3);
                int R = ksz <= 5 ? 2 : ksz <= 7 ? 3 : ksz <= 9 ? 4 : 5;
                fkz_conv_u8(image, sp, smooth, ow, oh, op, k1, 2);
                ikcontrast = fkz_kcontrast(smooth, ow, oh, op, prm->per, NULL, NULL);
                if (kcontrast_out) *kcontrast_out = ikcontrast;
                fkz_gauss_taps(var, R, kbase);
                fkz_conv_u8(image, sp, nldimg, ow, oh, op, kbase, R);
                memcpy(smooth, nldimg, sizeof(int) * (size_t)msz);
                sizes[j] = esigma * prm->derivative_factor;
                sigma_size = (int)(esigma * prm->derivative_factor + 0.5f);
                borders[j] = smax * sigma_size;
                fkz_hessian(smooth, dx, dy, smooth, sigma_size, ow, oh, op);
                continue;
            }
            esigma = prm->soffset * powf(2, (float)j / ms + i);
            curr_etime = 0.5f * esigma * esigma;
            ttime = curr_etime - last_etime;
            naux = okz_fed_tau(ttime, 1, tmax, prm->reordering, tau, 4096);
            sizes[j] = esigma * prm->derivative_factor / oratio;
            sigma_size = (int)(sizes[j] + 0.5f);
            borders[j] = smax * sigma_size;
            if (j == 0) {                                               /* akaze.cpp:640-662 */
                ikcontrast = (int)(ikcontrast * 0.75f + 0.5f);
                int* oldnld = nldimg - mstep;
                fkz_down_smooth(oldnld, nldimg, smooth, owhps[3 * (i - 1)], owhps[3 * (i - 1) + 1], owhps[3 * (i - 1) + 2], ow, oh, op, k1);
                fkz_flow(smooth, flow, prm->diffusivity, ikcontrast, ow, oh, op);
                for (int k = 0; k < naux; k++) {
                    fkz_nld_step(nldimg, flow, temp, tau[k], ow, oh, op);
                    memcpy(nldimg, temp, sizeof(int) * (size_t)msz);
                }
            } else {                                                    /* akaze.cpp:664-695 */
                int* oldnld = nldimg;
                nldimg += msz; smooth += msz; flow += msz; temp += msz; dx = flow; dy = temp;
                fkz_conv_int(oldnld, smooth, ow, oh, op, k1, 2);
                fkz_flow(smooth, flow, prm->diffusivity, ikcontrast, ow, oh, op);
                fkz_nld_step(oldnld, flow, nldimg, tau[0], ow, oh, op);
                for (int k = 1; k < naux; k++) {
                    fkz_nld_step(nldimg, flow, temp, tau[k], ow, oh, op);
                    memcpy(nldimg, temp, sizeof(int) * (size_t)msz);
                }
            }
            /* fkz_hessian writes det last, so src == det (the smooth slot) is safe: dx/dy are complete first */
            {
                int* tmpdet = (int*)malloc(sizeof(int) * (size_t)msz);
                fkz_hessian(smooth, dx, dy, tmpdet, sigma_size, ow, oh, op);
                memcpy(smooth, tmpdet, sizeof(int) * (size_t)msz);
                free(tmpdet);
            }
            last_etime = curr_etime;
        }
        fkz_extrema(tmem + offsets[i] + ms_msz, rmap, smap, lmap, params, i, ms, idthreshold, ow, oh, op, owhps[2]);
        psz = psz < borders[0] * oratio ? psz : borders[0] * oratio;
        mstep = ms_msz * 4;
        oratio *= 2;
    }
    int total = fkz_nms(pts, max_pts, rmap, smap, lmap, (int)psz, owhps[0], owhps[1], owhps[2]);
    int num = total < max_pts ? total : max_pts;
    float wtab[36];
    int idx1[488], idx2[488];
    for (int r2 = 0; r2 < 36; r2++) wtab[r2] = okz_expf(-r2 * 0.08f);
    okz_compare_indices(idx1, idx2);
#pragma omp parallel for schedule(dynamic, 16)
    for (int n = 0; n < num; n++) {
        FkPoint* pt = pts + n;
        int o = pt->octave / ms, s = pt->octave % ms;
        int ow = owhps[3 * o], oh = owhps[3 * o + 1], op = owhps[3 * o + 2];
        int* lt = tmem + offsets[o] + (size_t)s * osizes[o];
        int* det = tmem + offsets[o] + (size_t)(ms + s) * osizes[o];
        int* lx = tmem + offsets[o] + (size_t)(2 * ms + s) * osizes[o];
        int* ly = tmem + offsets[o] + (size_t)(3 * ms + s) * osizes[o];
        fkz_refine(pt, det, o, op);
        pt->angle = 0.f;
        memset(pt->features, 0, FK_FLEN);
        if (desc) {
            if (!prm->upright) fkz_orient(pt, lx, ly, o, ow, oh, op, wtab);
            fkz_describe(pt, lt, lx, ly, o, ow, oh, op, prm->descriptor_pattern_size, idx1, idx2);
        }
        pt->match = -1; pt->distance = -1; pt->match_x = -1; pt->match_y = -1;
    }
    return num;
}
