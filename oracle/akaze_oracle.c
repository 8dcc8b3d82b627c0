/*
 * akaze_oracle.c -- CPU restatement of the CUDA-AKAZE float hot path
 * (Akazer::detectAndCompute + cuMatch).
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the parity oracle: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.  The
 * shipped HIP library never links, imports or calls anything in oracle/.
 *
 * Each function cites the reference file:line it follows (paths relative to
 * the reference checkout).  Float evaluation order follows the source text;
 * no contraction except the two explicit __fmaf_rn sites (akazed.cu:179,
 * 1263).  Compile with -ffp-contract=off.
 *
 * Deterministic choices where the reference is racy or undefined
 * (SURVEY.md 2.3):
 *   D2  hmax = max(0.03f, maximum of grad over the 16-px LATTICE x % 16 == 0 && y % 16 == 0): the deterministic core of
 *       gFindMaxContrastU4 (akazed.cu:827-877), followed literally since round 5 -- see okz_kcontrast.  The kernel's racy
 *       remainder (block-local maxima swapped against absolute pixels of the image's top-left tile) can only raise hmax, and
 *       by no more than the maximum of the top-left 32 x 32 region; the oracle takes none of it.
 *   D3  the histogram guard `ix >= width && iy >= height` (akazed.cu:909) lets the threads beyond the right and bottom edge
 *       count too; in the steady state of a reused Akazer (arena zeroed at the end of every call, akaze.cpp:142-149) they
 *       read zeros: (ceil32(w) - w) * h + (ceil16(h) - h) * w extra entries in bin 0.  Followed since round 5.
 *   D4  clean reflect-101 separable Gaussian (no partial-block defect)
 *   D5  sublevels scatter into the maps in ascending order, strict '<'
 *   D6  keypoints are emitted in raster order of the full-resolution map
 *   D7  orientation histogram accumulates in ascending sample-thread order
 *   D8  AkazePoint::response = response_map value of the winning level
 *   D9  Hamming distance over exactly 61 bytes
 *   D10 n2 < 16 handled (lanes without a candidate do not take part)
 *
 * PARITY PINNING (oracle/README.md has the table): the reference has no tests and no golden vectors; its CUDA path cannot be
 * built here.  The only compilable piece (fed.cpp) is built by oracle/Makefile into oracle/_ref/ and pins okz_fed_tau() bit for
 * bit.  Constants are pinned by the KATs of SURVEY.md 4, every control-flow-heavy stage by hand-derived literal fixtures
 * (tests/literal_fixtures.py).  The whole path is pinned STATISTICALLY against the reference's own CUDA run through the result
 * pictures in its data/ directory (tools/ref_render_check.py, tests/test_ref_render_cpu.py): keypoint counts
 * within -1.0 .. -3.6 % of the printed ones, 86-89 % of the keypoints at the drawn pixel +-1 with the drawn radius class (as
 * many as the oracle scores against its own drawing), the NMS cursor lag confirmed.  That pin cannot see the contrast factor to
 * better than +-1 bin, response values or the descriptor's bit layout (measured: its power table): those are reading-only.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "okz_math.h"

#define OKZ_FLEN 61
#define OKZ_NBINS 300           /* akazed.cu:8 */
#define OKZ_MAX_DIST 96         /* akazed.cu:11 */
#define OKZ_MAX_OCT 8           /* akazed.cu:10 */
#define OKZ_MAX_SCALE 5         /* akazed.cu:9 */

/* akaze_structures.h:19-40 -- 104-byte POD */
typedef struct OkzPoint {
    float x, y;
    int octave;
    float response, size, angle;
    unsigned char features[OKZ_FLEN];
    int match, distance;
    float match_x, match_y;
} OkzPoint;

/* the 11 Akazer::init arguments (akaze.h:25-26) + build-side extension */
typedef struct OkzParams {
    int noctaves, max_scale;
    float per, kcontrast, soffset;
    int reordering;
    float derivative_factor, dthreshold;
    int diffusivity, descriptor_pattern_size;
    int upright;                /* 1: skip orientation, angle = 0 */
} OkzParams;

/* Alternative READINGS of the reference, selectable for tools/ref_render_check.py only (it scores each reading against the
 * keypoint circles the reference's own CUDA run drew into data/akaze_show*.jpg).  0 = the reading every test, golden and
 * the HIP library follow.  bit 0: gNmsRNaive's read cursor also advances at the skipped centre (a "clean disc": what the
 * kernel would do if `new_idx++` stood in front of the `continue`, akazed.cu:1581-1593).  bit 1: hmax is the TRUE maximum of
 * the gradient over w x h (what gFindMaxContrastU4 was meant to compute; the reading of rounds 1-4).  bit 2: the histogram
 * counts exactly the w x h valid pixels (the guard of akazed.cu:909 read as `||`; rounds 1-4).  bits 3-5 exist for the power
 * table of the statistical pin only (deliberately WRONG readings whose visibility in the reference's pictures is measured):
 * bit 3: k + 1 histogram bins; bit 4: in the scatter of one octave's sublevels into the maps the LAST writer wins whatever the
 * responses (the extreme outcome of the unsynchronised compare-then-write of akazed.cu:1368-1373, D5); bit 5: main orientation
 * + 5 degrees (float path). */
int okz_reading_variant = 0;
void okz_set_reading_variant(int v) { okz_reading_variant = v; }

/* akazed.cu:162-170 */
static inline int border_add(int a, int b, int m)
{
    int c = a + b;
    return c < m ? c : m + m - 2 - c;
}

static inline int iabs(int a) { return a < 0 ? -a : a; }

int okz_sizeof_point(void) { return (int)sizeof(OkzPoint); }

/* The team size of the oracle's parallel loops, set explicitly: OMP_NUM_THREADS is read once, when libgomp initialises -- in a
 * process that has imported torch that happened long before the oracle's first call, and the loops then ran with one thread per
 * logical CPU of the box (256 on the GPU boxes: 10 x slower than 16, which is what bench.py's cpu_baseline reported until round 5). */
#ifdef _OPENMP
#include <omp.h>
void okz_set_num_threads(int n) { if (n > 0) omp_set_num_threads(n); }
int okz_get_max_threads(void) { return omp_get_max_threads(); }
#else
void okz_set_num_threads(int n) { (void)n; }
int okz_get_max_threads(void) { return 1; }
#endif

/* ---------------------------------------------------------------- FED tau */

/* fed.cpp:128-148 */
static int fed_is_prime(int number)
{
    if (number <= 1) return 0;
    if (number == 2 || number == 3 || number == 5 || number == 7) return 1;
    if ((number % 2) == 0 || (number % 3) == 0 || (number % 5) == 0 || (number % 7) == 0) return 0;
    int upper = (int)sqrt(number + 1.0);
    for (int d = 11; d <= upper; d += 2)
        if (number % d == 0) return 0;
    return 1;
}

/* fed.cpp:41-119: fed_tau_by_process_time -> _by_cycle_time -> _internal.
 * tau must hold at least the returned n entries (n <= 4096 checked). */
int okz_fed_tau(float T, int M, float tau_max, int reordering, float* tau, int cap)
{
    float t = T / (float)M;                                             /* fed.cpp:44 */
    int n = (int)(ceil(sqrt(3.0 * t / tau_max + 0.25f) - 0.5f - 1.0e-8f) + 0.5f);   /* :55 */
    float scale = (float)(3.0 * t / (tau_max * (float)(n * (n + 1)))); /* :56 */
    if (n <= 0) return 0;
    if (n > cap) return -n;
    float c = 1.0f / (4.0f * (float)n + 2.0f);                          /* :79 */
    float d = scale * tau_max / 2.0f;                                   /* :80 */
    float* tauh = (float*)malloc(sizeof(float) * (size_t)n);
    for (int k = 0; k < n; ++k) {
        float h = (float)cos(OKZ_PI_D * (2.0f * (float)k + 1.0f) * c);  /* :84 */
        tauh[k] = d / (h * h);
    }
    if (!reordering) {
        memcpy(tau, tauh, sizeof(float) * (size_t)n);
    } else {
        int kappa = n / 2;                                              /* :98 */
        int prime = n + 1;
        while (!fed_is_prime(prime)) prime++;
        for (int k = 0, l = 0; l < n; ++k, ++l) {                       /* :108-115 */
            int index;
            while ((index = ((k + 1) * kappa) % prime - 1) >= n) k++;
            tau[l] = tauh[index];
        }
    }
    free(tauh);
    return n;
}

/* ------------------------------------------------------------- Gaussians */

/* akazed.cu:2298-2333 createGaussKernel: taps k[0..radius] */
void okz_gauss_taps(float var, int radius, float* k)
{
    float denom = 1.f / (2.f * var);
    float ksum = 0;
    for (int i = 0; i <= radius; i++) {
        k[i] = expf(-i * i * denom);
        if (i == 0) ksum += k[i];
        else ksum += k[i] + k[i];
    }
    ksum = 1 / ksum;
    for (int i = 0; i <= radius; i++) k[i] *= ksum;
}

/* akazed.cu:204-290 gConv2d<R> (row pass then column pass, reflect-101) */
void okz_lowpass(const float* src, float* dst, int w, int h, int p, const float* k, int R)
{
    float* rows = (float*)malloc(sizeof(float) * (size_t)w * (size_t)h);
#pragma omp parallel for schedule(static)
    for (int y = 0; y < h; y++) {
        const float* s = src + (size_t)y * p;
        float* r = rows + (size_t)y * w;
        for (int x = 0; x < w; x++) {
            float wsum = s[x] * k[0];                                   /* :227 */
            for (int i = 1; i <= R; i++)
                wsum += k[i] * (s[iabs(x - i)] + s[border_add(x, i, w)]);   /* :237 */
            r[x] = wsum;
        }
    }
#pragma omp parallel for schedule(static)
    for (int y = 0; y < h; y++) {
        for (int x = 0; x < w; x++) {
            float wsum = rows[(size_t)y * w + x] * k[0];                /* :283 */
            for (int i = 1; i <= R; i++)
                wsum += k[i] * (rows[(size_t)iabs(y - i) * w + x] + rows[(size_t)border_add(y, i, h) * w + x]);   /* :286 */
            dst[(size_t)y * p + x] = wsum;
        }
    }
    free(rows);
}

/* akazed.cu:449-511 gDownWithSmooth: dst = src[2y][2x]; smooth = G(k, R=2)
 * evaluated on the decimated lattice, mirror taken on the SOURCE extents. */
void okz_down_smooth(const float* src, float* dst, float* smooth,
                     int sw, int sh, int sp, int dw, int dh, int dp, const float* k)
{
    float* rows = (float*)malloc(sizeof(float) * (size_t)dw * (size_t)sh);
    /* row pass for every even-offset source row that can be referenced */
#pragma omp parallel for schedule(static)
    for (int sy = 0; sy < sh; sy++) {
        const float* s = src + (size_t)sy * sp;
        float* r = rows + (size_t)sy * dw;
        for (int dx = 0; dx < dw; dx++) {
            int six = dx + dx;
            int x0 = iabs(six - 4), x1 = iabs(six - 2), x3 = border_add(six, 2, sw), x4 = border_add(six, 4, sw);
            r[dx] = k[0] * s[six] + k[1] * (s[x1] + s[x3]) + k[2] * (s[x0] + s[x4]);   /* :469-471 */
        }
    }
#pragma omp parallel for schedule(static)
    for (int dy = 0; dy < dh; dy++) {
        int siy = dy + dy;
        int y0 = iabs(siy - 4), y1 = iabs(siy - 2), y3 = border_add(siy, 2, sh), y4 = border_add(siy, 4, sh);
        for (int dx = 0; dx < dw; dx++) {
            dst[(size_t)dy * dp + dx] = src[(size_t)siy * sp + dx + dx];           /* :506 */
            smooth[(size_t)dy * dp + dx] =
                k[0] * rows[(size_t)siy * dw + dx] +
                k[1] * (rows[(size_t)y1 * dw + dx] + rows[(size_t)y3 * dw + dx]) +
                k[2] * (rows[(size_t)y0 * dw + dx] + rows[(size_t)y4 * dw + dx]);   /* :507-509 */
        }
    }
    free(rows);
}

/* ------------------------------------------------------ contrast factor */

/* un-normalised Scharr pair shared by akazed.cu:664-665 and 1088-1089 */
static inline void scharr_dxdy(const float* src, int x, int y, int w, int h, int p, float* dx, float* dy)
{
    int x0 = iabs(x - 1), x2 = border_add(x, 1, w);
    int y0 = iabs(y - 1), y2 = border_add(y, 1, h);
    const float* r0 = src + (size_t)y0 * p;
    const float* r1 = src + (size_t)y * p;
    const float* r2 = src + (size_t)y2 * p;
    *dx = 10 * (r1[x2] - r1[x0]) + 3 * (r0[x2] + r2[x2] - r0[x0] - r2[x0]);
    *dy = 10 * (r2[x] - r0[x]) + 3 * (r2[x0] + r2[x2] - r0[x0] - r0[x2]);
}

/* akazed.cu:644-667 gScharrContrastNaive */
void okz_scharr_grad(const float* src, float* grad, int w, int h, int p)
{
#pragma omp parallel for schedule(static)
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            float dx, dy;
            scharr_dxdy(src, x, y, w, h, p, &dx, &dy);
            grad[(size_t)y * p + x] = sqrtf(dx * dx + dy * dy);
        }
}


/* double / float -> int as the device does it (cvt.rzi.s32: toward zero, SATURATING, NaN -> 0); a C cast of a NaN or of a value
 * outside the int range is undefined (x86: INT_MIN).  Needed where the reference's arithmetic can leave the finite range: a constant
 * image has hmax == 0, hfactor == inf and 0 * inf == NaN in every histogram bin index (akazed.cu:924), and a sample without gradient has
 * atan2(0, 0) -- NaN through the polynomial -- as its angle (akazed.cu:1702). */
static inline int okz_d2i(double v)
{
    if (v != v) return 0;
    if (v >= 2147483648.0) return 2147483647;
    if (v <= -2147483648.0) return (int)0x80000000;
    return (int)v;
}

/* akazed.cu:2410-2484 hScharrContrast host half + 827-877 gFindMaxContrastU4 + 901-938 gConstrastHistShared.
 * hist (300 ints) and hmax are optional outputs.
 *
 * hmax (D2).  gFindMaxContrastU4 runs 16 x 16 threads per 32 x 32 pixel block (grid1 :2435).  Thread (tix, tiy) swaps the
 * largest of its four pixels (ix0 + {0,16}, iy0 + {0,16}) into x0y0 (:842-857, in place).  The "reduction" (:860-873) then
 * sorts x0y0 against nidx = (nid / 16) * pitch + nid % 16 -- an ABSOLUTE pixel of the image's top-left 16 x 16 tile, not
 * another thread's value -- and only thread 0 of each block feeds atomicMax (:874-877).  Thread 0's four pixels are
 * (32 bx + {0,16}, 32 by + {0,16}): over all blocks, exactly the pixels with x % 16 == 0 && y % 16 == 0 inside the image.
 * No other thread ever writes them (every thread's four pixels are its own; the tile pixels it writes have nid >= 1, and
 * block (0,0)'s thread 0 owns pixel 0), so the value block b contributes is at least the maximum of its lattice pixels, and
 * exactly that unless one of the eight tile pixels nid = 128, 64, .., 1 holds something larger at the moment thread 0 reads
 * it.  Those pixels only ever receive the smaller side of a swap from blocks other than (0,0), and from block (0,0) values
 * of the image's top-left 32 x 32 region: the racy excess is bounded by max(grad[0..31][0..31]) and vanishes whenever that
 * is below the lattice maximum.  The oracle takes the deterministic core.
 *
 * histogram (D3).  gConstrastHistShared (32 x 16 threads, grid2 :2454) returns only when BOTH ix >= width and iy >= height
 * (:909), so columns [w, ceil32(w)) of rows < h and rows [h, ceil16(h)) of columns < w are counted as well.  grad is the
 * octave's `temp` plane 0 (akaze.cpp:319, 326): the columns are pitch padding nobody writes, the rows are the head of temp
 * plane 1 -- all zero once a reused Akazer has finished one call (cudaMemset(omem, 0, ..), akaze.cpp:142-149; the very
 * first call, or a call with another image size, reads a fresh cudaMalloc).  They land in bin 0 and lower `thresh` (:2468).
 * With a caller pitch below ceil32(w) -- main.cpp:174 aligns to 128 -- (ceil32(w) - pitch) * (h - 1) of the column reads
 * fall on real pixels of the swap-permuted plane instead; their values depend on the race above, the oracle keeps zeros.
 * The swaps leave the multiset of gradient values unchanged when they do not collide, so the histogram itself is taken over
 * the un-permuted plane. */
float okz_kcontrast(const float* grad, int w, int h, int p, float per, float* hmax_out, int* hist_out)
{
    float hmax = 0.03f;                                                 /* :2413 */
    /* grid1 (:2435) has ceil((w / 2) / 16) blocks of 32 columns: an odd w with (w - 1) % 32 == 0 leaves its last column --
     * a lattice column -- to no block at all; rows alike */
    const int lat = (okz_reading_variant & 2) ? 1 : 16;
    const int wcov = (okz_reading_variant & 2) ? w : (32 * ((w / 2 + 15) / 16) < w ? 32 * ((w / 2 + 15) / 16) : w);
    const int hcov = (okz_reading_variant & 2) ? h : (32 * ((h / 2 + 15) / 16) < h ? 32 * ((h / 2 + 15) / 16) : h);
    for (int y = 0; y < hcov; y += lat)
        for (int x = 0; x < wcov; x += lat) {
            float g = grad[(size_t)y * p + x];
            if (g > hmax) hmax = g;
        }
    int hist[OKZ_NBINS];
    memset(hist, 0, sizeof(hist));
    float hfactor = OKZ_NBINS / hmax;                                   /* :2450 */
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            /* __fmul_rz(grad, factor) then float->int truncation (:924):
             * the double product of two floats is exact, so truncating it
             * equals truncating the RZ-rounded float product. */
            int hi = okz_d2i((double)grad[(size_t)y * p + x] * (double)hfactor);
            if (hi >= OKZ_NBINS) hi = OKZ_NBINS - 1;
            hist[hi]++;
        }
    if (!(okz_reading_variant & 4))                                     /* :909 with grid2 :2454 */
        hist[0] += ((w + 31) / 32 * 32 - w) * h + ((h + 15) / 16 * 16 - h) * w;
    int thresh = (int)((w * h - hist[0]) * per);                        /* :2468 */
    int cumuv = 0, k = 1;
    while (k < OKZ_NBINS) {                                             /* :2472-2480 */
        if (cumuv >= thresh) break;
        cumuv += hist[k];
        k++;
    }
    if ((okz_reading_variant & 8) && k < OKZ_NBINS) k++;                /* power table only */
    if (hmax_out) *hmax_out = hmax;
    if (hist_out) memcpy(hist_out, hist, sizeof(hist));
    return k / hfactor;                                                 /* :2481 */
}

/* --------------------------------------------------- conductivity + step */

/* akazed.cu:1068-1107 gFlowNaive + 2493 (ikc).  PM_G1 / WEICKERT use the
 * oracle's own exp (the reference's __expf/__powf are unpinned). */
void okz_flow(const float* src, float* dst, int type, float kcontrast, int w, int h, int p)
{
    float ikc = 1.f / (kcontrast * kcontrast);
#pragma omp parallel for schedule(static)
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            float dx, dy;
            scharr_dxdy(src, x, y, w, h, p, &dx, &dy);
            float dif2 = ikc * (dx * dx + dy * dy);
            float g;
            if (type == 0) g = okz_expf(-dif2);
            else if (type == 1) g = 1.f / (1.f + dif2);
            else if (type == 2) {
                float d2 = dif2 * dif2;
                g = 1.f - okz_expf(-3.315f / (d2 * d2));
            } else g = 1.f / sqrtf(1.f + dif2);
            dst[(size_t)y * p + x] = g;
        }
}

/* akazed.cu:1241-1264 gNldStepNaive + 2515 (stepfac = 0.5*tau). dst != src. */
void okz_nld_step(const float* src, const float* flow, float* dst, float tau, int w, int h, int p)
{
    float stepfac = 0.5f * tau;
#pragma omp parallel for schedule(static)
    for (int y = 0; y < h; y++) {
        int y0 = iabs(y - 1), y2 = border_add(y, 1, h);
        const float* s0 = src + (size_t)y0 * p;
        const float* s1 = src + (size_t)y * p;
        const float* s2 = src + (size_t)y2 * p;
        const float* f0 = flow + (size_t)y0 * p;
        const float* f1 = flow + (size_t)y * p;
        const float* f2 = flow + (size_t)y2 * p;
        for (int x = 0; x < w; x++) {
            int x0 = iabs(x - 1), x2 = border_add(x, 1, w);
            float step = (f1[x] + f1[x2]) * (s1[x2] - s1[x]) +
                         (f1[x] + f1[x0]) * (s1[x0] - s1[x]) +
                         (f1[x] + f2[x]) * (s2[x] - s1[x]) +
                         (f1[x] + f0[x]) * (s0[x] - s1[x]);             /* :1259-1262 */
            dst[(size_t)y * p + x] = fmaf(stepfac, step, s1[x]);        /* :1263 */
        }
    }
}

/* ---------------------------------------------------- Hessian determinant */

/* akazed.cu:2537-2539 */
void okz_deriv_factors(float* fac1, float* fac2)
{
    float w = 10.f / 3.f;
    *fac1 = 1.f / (2.f * (w + 2.f));
    *fac2 = w * *fac1;
}

/* akazed.cu:1267-1296 gDerivate */
void okz_derivate(const float* src, float* dxo, float* dyo, int step, int w, int h, int p)
{
    float fac1, fac2;
    okz_deriv_factors(&fac1, &fac2);
#pragma omp parallel for schedule(static)
    for (int y = 0; y < h; y++) {
        const float* r0 = src + (size_t)iabs(y - step) * p;
        const float* r1 = src + (size_t)y * p;
        const float* r2 = src + (size_t)border_add(y, step, h) * p;
        for (int x = 0; x < w; x++) {
            int x0 = iabs(x - step), x2 = border_add(x, step, w);
            float ul = r0[x0], uc = r0[x], ur = r0[x2];
            float cl = r1[x0], cr = r1[x2];
            float ll = r2[x0], lc = r2[x], lr = r2[x2];
            dxo[(size_t)y * p + x] = fac1 * (ur + lr - ul - ll) + fac2 * (cr - cl);   /* :1294 */
            dyo[(size_t)y * p + x] = fac1 * (lr + ll - ur - ul) + fac2 * (lc - uc);   /* :1295 */
        }
    }
}

/* akazed.cu:1299-1331 gHessianDeterminant */
void okz_hessian(const float* dx, const float* dy, float* det, int step, int w, int h, int p)
{
    float fac1, fac2;
    okz_deriv_factors(&fac1, &fac2);
#pragma omp parallel for schedule(static)
    for (int y = 0; y < h; y++) {
        size_t o0 = (size_t)iabs(y - step) * p, o1 = (size_t)y * p, o2 = (size_t)border_add(y, step, h) * p;
        for (int x = 0; x < w; x++) {
            int x0 = iabs(x - step), x2 = border_add(x, step, w);
            size_t iul = o0 + x0, iuc = o0 + x, iur = o0 + x2, icl = o1 + x0, icr = o1 + x2,
                   ill = o2 + x0, ilc = o2 + x, ilr = o2 + x2;
            float dxx = fac1 * (dx[iur] + dx[ilr] - dx[iul] - dx[ill]) + fac2 * (dx[icr] - dx[icl]);
            float dxy = fac1 * (dx[ilr] + dx[ill] - dx[iur] - dx[iul]) + fac2 * (dx[ilc] - dx[iuc]);
            float dyy = fac1 * (dy[ilr] + dy[ill] - dy[iur] - dy[iul]) + fac2 * (dy[ilc] - dy[iuc]);
            det[o1 + x] = dxx * dyy - dxy * dxy;                        /* :1330 */
        }
    }
}

/* ------------------------------------------------------- detector tail */

/* akazed.cu:1334-1393 gCalcExtremaMap + 2563-2587, sublevels ascending (D5).
 * params = [borders[0..ms) | sizes[0..ms)] as in d_extrema_param. */
void okz_extrema_map(const float* dets, float* response_map, float* size_map, int* layer_map,
                     const float* params, int octave, int max_scale, float threshold,
                     int w, int h, int p, int opitch)
{
    int psz = (int)params[0];
    for (int s = 0; s < max_scale; s++) {
        float border = params[s];
        float size = params[max_scale + s];
        const float* det = dets + (size_t)s * h * p;
        for (int iy = psz; iy < h; iy++) {
            int up_y = (int)(iy - border + 0.5f) - 1;
            int down_y = (int)(iy + border + 0.5f) + 1;
            if (up_y < 0 || down_y >= h) continue;
            for (int ix = psz; ix < w; ix++) {
                int left_x = (int)(ix - border + 0.5f) - 1;
                int right_x = (int)(ix + border + 0.5f) + 1;
                if (left_x < 0 || right_x >= w) continue;
                const float* vp = det + (size_t)iy * p + ix;
                const float* vp0 = vp - p;
                const float* vp2 = vp + p;
                float v = *vp;
                if (v > threshold && v > *vp0 && v > *vp2 && v > vp[-1] && v > vp[1] &&
                    v > vp0[-1] && v > vp0[1] && v > vp2[-1] && v > vp2[1]) {
                    size_t oidx = (size_t)(iy << octave) * opitch + (size_t)(ix << octave);
                    /* (variant bit 4, power table only: a later sublevel of the SAME octave overwrites unconditionally) */
                    if (response_map[oidx] < v ||                        /* :1368 */
                        ((okz_reading_variant & 16) && layer_map[oidx] >= octave * max_scale)) {
                        response_map[oidx] = v;
                        size_map[oidx] = size;
                        layer_map[oidx] = octave * max_scale + s;
                    }
                }
            }
        }
    }
}

/* akazed.cu:1554-1613 gNmsRNaive, emitted in raster order (D6); response
 * filled from the map (D8).  Returns the number of survivors (may exceed
 * max_pts; only the first max_pts are written). */
int okz_nms(OkzPoint* points, int max_pts, const float* response_map, const float* size_map,
            const int* layer_map, int psz, int w, int h, int p)
{
    int n = 0;
    for (int iy = psz; iy + psz < h; iy++)
        for (int ix = psz; ix + psz < w; ix++) {
            size_t idx = (size_t)iy * p + ix;
            if (layer_map[idx] < 0) continue;
            float fsz = size_map[idx];
            int isz = (int)(fsz + 0.5f);
            int sqsz = (int)(fsz * fsz);
            int to_nms = 0;
            for (int i = -isz; i <= isz && !to_nms; i++) {
                /* :1578 the read cursor restarts at column ix - isz on every row and advances at the END of the
                 * j loop body (:1593) -- which the `continue` of the centre (:1581-1584) skips.  So on row i == 0
                 * every j > 0 reads column ix + j - 1 (j == 1 re-reads the centre itself), while the disc test
                 * and the tie rule still use j.  Followed literally (deviation table: Q1). */
                int col = ix - isz;
                for (int j = -isz; j <= isz; j++) {
                    if (i == 0 && j == 0) { col += okz_reading_variant & 1; continue; }        /* :1581 */
                    float rn = response_map[(size_t)(iy + i) * p + col];
                    if (i * i + j * j < sqsz && (rn > -1e6f &&                                 /* :1585-1586 */
                        (rn > response_map[idx] || (rn == response_map[idx] && i <= 0 && j <= 0))))
                        to_nms = 1;
                    col++;                                                                     /* :1593 */
                }
            }
            if (!to_nms) {
                if (n < max_pts) {
                    OkzPoint* pt = points + n;
                    pt->x = (float)ix;
                    pt->y = (float)iy;
                    pt->octave = layer_map[idx];
                    pt->size = size_map[idx];
                    pt->response = response_map[idx];
                }
                n++;
            }
        }
    return n;
}

/* akazed.cu:1615-1662 gRefine on one point; det = det plane of its level */
void okz_refine_point(OkzPoint* pt, const float* det, int o, int p)
{
    int y = (int)pt->y >> o;
    int x = (int)pt->x >> o;
    size_t idx = (size_t)y * p + x;
    float v2 = det[idx] + det[idx];
    float dx = 0.5f * (det[idx + 1] - det[idx - 1]);
    float dy = 0.5f * (det[idx + p] - det[idx - p]);
    float dxx = det[idx + 1] + det[idx - 1] - v2;
    float dyy = det[idx + p] + det[idx - p] - v2;
    float dxy = 0.25f * (det[idx + p + 1] + det[idx - p - 1] - det[idx - p + 1] - det[idx + p - 1]);
    float dd = dxx * dyy - dxy * dxy;
    float idd = dd != 0.f ? 1.f / dd : 0.f;
    float dst0 = idd * (dxy * dy - dyy * dx);
    float dst1 = idd * (dxy * dx - dxx * dy);
    int weak = dst0 < -1.f || dst0 > 1.f || dst1 < -1.f || dst1 > 1.f;
    if (weak) return;
    int ratio = 1 << o;
    pt->y = ratio * (y + dst1);
    pt->x = ratio * (x + dst0);
}

/* exp(-r2*0.08f) table for r2 in [0,36) (akazed.cu:1697) */
void okz_orient_weights(float* tab)
{
    for (int r2 = 0; r2 < 36; r2++) tab[r2] = okz_expf(-r2 * 0.08f);
}

static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* akazed.cu:1665-1736 gCalcOrient on one point (D7: ascending thread order).
 * Sample coordinates are clamped to the plane (never triggers for points the
 * detector accepts; guards out-of-range reads for hand-made inputs). */
void okz_orient_point(OkzPoint* pt, const float* dxd, const float* dyd, int o, int w, int h, int p,
                      const float* wtab)
{
    float resx[42], resy[42], re8x[42], re8y[42];
    for (int t = 0; t < 42; t++) { resx[t] = 0.f; resy[t] = 0.f; }
    /* (device casts: gRefine can leave NaN coordinates behind -- a determinant of inf - inf at a coarse level of a deep pyramid -- and the
     * device turns them into 0 where a C cast is undefined; found by tests/fuzz_parity.py --huge) */
    int step = okz_d2i(pt->size + 0.5f);
    int x = okz_d2i(pt->x + 0.5f) >> o;
    int y = okz_d2i(pt->y + 0.5f) >> o;
    for (int tix = 0; tix < 13 * 16; tix++) {
        int i = (tix & 15) - 6;
        int j = (tix / 16) - 6;
        int r2 = i * i + j * j;
        if (r2 >= 36) continue;
        float gweight = wtab[r2];
        int yy = clampi(y + step * j, 0, h - 1), xx = clampi(x + step * i, 0, w - 1);
        size_t pos = (size_t)yy * p + xx;
        float dx = gweight * dxd[pos];
        float dy = gweight * dyd[pos];
        float angle = okz_atan2f(dy, dx);
        int a = okz_d2i(angle * (21 / OKZ_PI_D)) + 21;                  /* :1702 (double) */
        a = a > 41 ? 41 : a;
        a = a < 0 ? 0 : a;
        resx[a] += dx;
        resy[a] += dy;
    }
    for (int t = 0; t < 42; t++) {                                      /* :1708-1717 */
        re8x[t] = resx[t];
        re8y[t] = resy[t];
        for (int k = t + 1; k < t + 7; k++) {
            re8x[t] += resx[k < 42 ? k : k - 42];
            re8y[t] += resy[k < 42 ? k : k - 42];
        }
    }
    float maxr = 0.0f;
    int maxk = 0;
    for (int k = 0; k < 42; k++) {
        float r = re8x[k] * re8x[k] + re8y[k] * re8y[k];
        if (r > maxr) { maxr = r; maxk = k; }
    }
    /* dFastAtan2 akazed.cu:173-185 */
    float yv = re8y[maxk], xv = re8x[maxk];
    float absx = fabsf(xv), absy = fabsf(yv);
    float mn = absx < absy ? absx : absy, mx = absx < absy ? absy : absx;
    float a = mx > 0.f ? mn / mx : 0.f; /* reference: 0/0 -> NaN; oracle defines angle 0 (never hit by detected points) */
    float s = a * a;
    float r = fmaf(fmaf(fmaf(-0.0464964749f, s, 0.15931422f), s, -0.327622764f), s * a, a);
    r = (absy > absx ? OKZ_HPI_F - r : r);
    r = (xv < 0 ? (float)(OKZ_PI_D - r) : r);
    r = (yv < 0 ? -r : r);
    pt->angle = (r < 0.0f ? (float)(r + 2.0f * OKZ_PI_D) : r);          /* :1734 */
    if (okz_reading_variant & 32) {                                     /* power table only */
        pt->angle += (float)(5.0 * OKZ_PI_D / 180.0);
        if (pt->angle >= (float)(2.0 * OKZ_PI_D)) pt->angle -= (float)(2.0 * OKZ_PI_D);
    }
}

/* akazed.cu:65-159 setCompareIndices: 486 pairs into idx1/idx2 (>= 488 ints) */
void okz_compare_indices(int* idx1, int* idx2)
{
    static const int lo[3] = {0, 4, 13}, hi[3] = {4, 13, 29};
    int cntr = 0;
    for (int g = 0; g < 3; g++)
        for (int ch = 0; ch < 3; ch++)
            for (int j = lo[g]; j < hi[g] - 1; ++j)
                for (int i = j + 1; i < hi[g]; ++i) {
                    idx1[cntr] = 3 * j + ch;
                    idx2[cntr] = 3 * i + ch;
                    cntr++;
                }
    for (; cntr < 488; cntr++) { idx1[cntr] = 0; idx2[cntr] = 0; }
}

/* akazed.cu:1869-2001 gDescribe2 on one point (64-thread accumulation order,
 * t/t+32 pairing, shfl-down tree 1,2,4,8,16).  imd/dxd/dyd: Lt, Lx, Ly planes
 * of the point's level. */
void okz_describe_point(OkzPoint* pt, const float* imd, const float* dxd, const float* dyd,
                        int o, int w, int h, int p, int patsize, const int* idx1, const int* idx2)
{
    enum { S = 64 };
    static _Thread_local float acc[3 * 30 * S];
    int size2 = patsize;
    int size3 = (int)ceilf(2.0f * patsize / 3.0f);                      /* :2682 */
    int size4 = (int)ceilf(0.5f * patsize);                             /* :2683 */
    float iratio = 1.f / (1 << o);
    int scale = okz_d2i(pt->size + 0.5f);
    float xf = pt->x * iratio;
    float yf = pt->y * iratio;
    float co, si;
    okz_sincosf(pt->angle, &si, &co);
    int winsize = 3 * size3 > 4 * size4 ? 3 * size3 : 4 * size4;
    memset(acc, 0, sizeof(acc));
    for (int tix = 0; tix < S; tix++) {
        float* a = acc + 3 * 30 * tix;
        for (int i = tix; i < winsize * winsize; i += S) {
            int y = i / winsize;
            int x = i - winsize * y;
            int m = x > y ? x : y;
            if (m >= winsize) continue;
            int l = x - size2;
            int k = y - size2;
            int xp = okz_d2i(xf + scale * (k * co - l * si) + 0.5f);    /* :1921 */
            int yp = okz_d2i(yf + scale * (k * si + l * co) + 0.5f);    /* :1922 */
            xp = clampi(xp, 0, w - 1);
            yp = clampi(yp, 0, h - 1);
            size_t pos = (size_t)yp * p + xp;
            float im = imd[pos];
            float dx = dxd[pos];
            float dy = dyd[pos];
            float rx = -dx * si + dy * co;
            float ry = dx * co + dy * si;
            if (m < 2 * size2) {
                int x2 = (x < size2 ? 0 : 1);
                int y2 = (y < size2 ? 0 : 1);
                int c = 3 * (y2 * 2 + x2);
                a[c] += im; a[c + 1] += rx; a[c + 2] += ry;
            }
            if (m < 3 * size3) {
                int x3 = (x < size3 ? 0 : (x < 2 * size3 ? 1 : 2));
                int y3 = (y < size3 ? 0 : (y < 2 * size3 ? 1 : 2));
                int c = 3 * (4 + y3 * 3 + x3);
                a[c] += im; a[c + 1] += rx; a[c + 2] += ry;
            }
            if (m < 4 * size4) {
                int x4 = (x < 2 * size4 ? (x < size4 ? 0 : 1) : (x < 3 * size4 ? 2 : 3));
                int y4 = (y < 2 * size4 ? (y < size4 ? 0 : 1) : (y < 3 * size4 ? 2 : 3));
                int c = 3 * (4 + 9 + y4 * 4 + x4);
                a[c] += im; a[c + 1] += rx; a[c + 2] += ry;
            }
        }
    }
    /* reduce: b_t = a_t + a_{t+32}; then v_l += v_{l+1}, +2, +4, +8, +16 (:1957-1981) */
    float vals[90];
    for (int c = 0; c < 90; c++) {
        float v[32];
        for (int t = 0; t < 32; t++) v[t] = acc[3 * 30 * t + c] + acc[3 * 30 * (t + 32) + c];
        for (int d = 1; d < 32; d <<= 1)
            for (int t = 0; t + d < 32; t += 2 * d) v[t] = v[t] + v[t + d];
        vals[c] = v[0];
    }
    for (int b = 0; b < OKZ_FLEN; b++) {                                /* :1987-1999 */
        unsigned char desc_r = 0;
        for (int i = 0; i < (b == 60 ? 6 : 8); ++i)
            desc_r |= (unsigned char)((vals[idx1[b * 8 + i]] > vals[idx2[b * 8 + i]] ? 1 : 0) << i);
        pt->features[b] = desc_r;
    }
}

/* akazed.cu:2125-2241 gHammingMatch + dHammingDistance2 (D9, D10) */
void okz_match(OkzPoint* pts1, int n1, const OkzPoint* pts2, int n2)
{
#pragma omp parallel for schedule(static)
    for (int q = 0; q < n1; q++) {
        int distance[16], indice[16];
        for (int t = 0; t < 16; t++) { distance[t] = 1 << 30; indice[t] = -1; }
        for (int j = 0; j < n2; j++) {
            int dist = 0;
            for (int b = 0; b < OKZ_FLEN; b++)
                dist += __builtin_popcount((unsigned)(pts1[q].features[b] ^ pts2[j].features[b]));
            int t = j & 15;
            if (dist < distance[t]) { distance[t] = dist; indice[t] = j; }   /* :2180 strict */
        }
        int best = 0;
        for (int t = 1; t < 16; t++)
            if (distance[t] < distance[best]) best = t;
        int nflag = 0;
        for (int t = 0; t < 16; t++) nflag += distance[best] < distance[t] ? 1 : 0;   /* :2206 */
        OkzPoint* p1 = pts1 + q;
        if (indice[best] >= 0 && nflag == 15 && distance[best] < OKZ_MAX_DIST) {    /* :2223 */
            p1->match = indice[best];
            p1->distance = distance[best];
            p1->match_x = pts2[indice[best]].x;
            p1->match_y = pts2[indice[best]].y;
        } else {
            p1->match = -1;
            p1->distance = -1;
            p1->match_x = -1;
            p1->match_y = -1;
        }
    }
}

/* Match post-processing: the 2-NN rule of the reference's unused gMatch (akazed.cu:2028-2122:
 * keep best and second-best score, accept iff best < second && best < MAX_DIST; its LDS reduction has no
 * barriers and it writes pt2->match_y, so only the stated rule is restated here), generalised to a ratio
 * num/den and an optional symmetric cross-check, with the accepted matches listed in query order.
 * Parity for this row is pinned by numpy brute force (tests), not by reference output: the kernel is dead code. */
typedef struct { int query, train, distance, second; float x1, y1, x2, y2; } OkzMatchPair;

static void okz_nn2(const OkzPoint* a, const OkzPoint* B, int nB, int* j1, int* d1, int* d2)
{
    int best = 512, second = 512, bi = -1;
    for (int j = 0; j < nB; j++) {
        int dist = 0;
        for (int b = 0; b < OKZ_FLEN; b++) dist += __builtin_popcount((unsigned)(a->features[b] ^ B[j].features[b]));
        if (dist < best) { second = best; best = dist; bi = j; }
        else if (dist < second) second = dist;
    }
    *j1 = bi; *d1 = best; *d2 = second;
}

int okz_match_knn2(OkzPoint* pts1, int n1, const OkzPoint* pts2, int n2, int ratio_num, int ratio_den, int cross,
                   int max_dist, OkzMatchPair* out)
{
    int* acc = (int*)calloc((size_t)(n1 > 0 ? n1 : 1), 3 * sizeof(int));
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n1; i++) {
        int j1, d1, d2;
        okz_nn2(pts1 + i, pts2, n2, &j1, &d1, &d2);
        int ok = j1 >= 0 && d1 < max_dist && (long)d1 * ratio_den < (long)d2 * ratio_num;
        if (ok && cross) {
            int i1, e1, e2;
            okz_nn2(pts2 + j1, pts1, n1, &i1, &e1, &e2);
            ok = i1 == i;
        }
        acc[3 * i] = ok ? j1 : -1; acc[3 * i + 1] = d1; acc[3 * i + 2] = d2;
    }
    int cnt = 0;
    for (int i = 0; i < n1; i++) {
        OkzPoint* p1 = pts1 + i;
        int j = acc[3 * i];
        if (j >= 0) {
            p1->match = j; p1->distance = acc[3 * i + 1]; p1->match_x = pts2[j].x; p1->match_y = pts2[j].y;
            if (out) {
                OkzMatchPair r = {i, j, acc[3 * i + 1], acc[3 * i + 2], p1->x, p1->y, pts2[j].x, pts2[j].y};
                out[cnt] = r;
            }
            cnt++;
        } else {
            p1->match = -1; p1->distance = -1; p1->match_x = -1; p1->match_y = -1;
        }
    }
    free(acc);
    return cnt;
}

/* --------------------------------------------------------- whole pipeline */

/* akaze.cpp:204-237 allocMemory (sized by the effective octave count, D15).
 * owhps: 3 ints per octave (w,h,p); osizes[noct]; offsets[noct+1].
 * Returns the effective number of octaves. */
int okz_layout(int w, int h, int p, int noctaves, int max_scale, int* owhps, int* osizes, int* offsets)
{
    owhps[0] = w; owhps[1] = h; owhps[2] = p;
    osizes[0] = h * p;
    offsets[0] = 3 * osizes[0];
    offsets[1] = offsets[0] + osizes[0] * max_scale * 4;
    int n = noctaves;
    for (int i = 0, j = 1, k = 2; j < noctaves; i++, j++, k++) {
        int ww = owhps[3 * i] >> 1, hh = owhps[3 * i + 1] >> 1;
        if (ww < 80 || hh < 80) { n = j; break; }
        owhps[3 * j] = ww;
        owhps[3 * j + 1] = hh;
        owhps[3 * j + 2] = (ww % 128 != 0) ? (ww - ww % 128 + 128) : ww;   /* iAlignUp cuda_utils.h:160 */
        osizes[j] = hh * owhps[3 * j + 2];
        offsets[k] = offsets[j] + osizes[j] * max_scale * 4;
    }
    return n;
}

/* number of floats okz_detect_and_compute needs in `arena` */
long okz_arena_floats(int w, int h, int p, int noctaves, int max_scale)
{
    int owhps[3 * OKZ_MAX_OCT], osizes[OKZ_MAX_OCT], offsets[OKZ_MAX_OCT + 1];
    int n = okz_layout(w, h, p, noctaves, max_scale, owhps, osizes, offsets);
    return offsets[n];
}

/* Akazer::detectAndCompute akaze.cpp:101-150 + detect 240-503.
 * image: float32 [0,1], pitch p elements.  arena: okz_arena_floats() floats
 * (layout SURVEY.md 9.1; left in its post-call state so tests can read any
 * plane).  kcontrast_out (nullable) receives the octave-0 contrast factor.
 * Returns num_pts (<= max_pts). */
int okz_detect_and_compute(const float* image, int w, int h, int p, const OkzParams* prm,
                           OkzPoint* pts, int max_pts, int desc, float* tmem, float* kcontrast_out)
{
    int noct = prm->noctaves, ms = prm->max_scale;
    int owhps[3 * OKZ_MAX_OCT], osizes[OKZ_MAX_OCT], offsets[OKZ_MAX_OCT + 1];
    noct = okz_layout(w, h, p, noct, ms, owhps, osizes, offsets);

    float* response_map = tmem;
    float* size_map = tmem + osizes[0];
    int* layer_map = (int*)(size_map + osizes[0]);
    for (int i = 0; i < osizes[0]; i++) {                               /* akaze.cpp:252-258 (D1) */
        response_map[i] = -0.0926474631f;
        size_map[i] = -0.0926474631f;
        layer_map[i] = -1;
    }

    float k1[3], kbase[8];
    okz_gauss_taps(1.f, 2, k1);

    float kcontrast = prm->kcontrast;
    float tmax = 0.25f;
    float esigma = prm->soffset;
    float last_etime = (float)(0.5 * prm->soffset * prm->soffset);     /* akaze.cpp:270 */
    float curr_etime = 0, ttime = 0;
    int naux = 0, oratio = 1, sigma_size = 0;
    float smax = (float)(10.0 * sqrtf(2.0f));                           /* akaze.cpp:279 */
    float params[2 * OKZ_MAX_SCALE + 2];
    float* borders = params;
    float* sizes = params + ms;
    float psz = 10000;
    float tau[4096];
    int mstep = 0;

    for (int i = 0; i < noct; i++) {
        int ow = owhps[3 * i], oh = owhps[3 * i + 1], op = owhps[3 * i + 2];
        int msz = osizes[i];
        int ms_msz = msz * ms;
        float* nldimg = tmem + offsets[i];
        float* smooth = nldimg + ms_msz;
        float* flow = smooth + ms_msz;
        float* temp = flow + ms_msz;
        float* dx = flow;
        float* dy = temp;
        for (int j = 0; j < ms; j++) {
            if (j == 0 && i == 0) {                                     /* akaze.cpp:325-354 */
                float var = prm->soffset * prm->soffset;
                int ksz = (int)(2 * ceilf((prm->soffset - 0.8f) / 0.3f) + 3);
                int R = ksz <= 5 ? 2 : ksz <= 7 ? 3 : ksz <= 9 ? 4 : 5;   /* akazed.cu:2345-2377 */
                okz_lowpass(image, smooth, ow, oh, op, k1, 2);
                okz_scharr_grad(smooth, temp, ow, oh, op);
                kcontrast = okz_kcontrast(temp, ow, oh, op, prm->per, NULL, NULL);
                if (kcontrast_out) *kcontrast_out = kcontrast;
                okz_gauss_taps(var, R, kbase);
                okz_lowpass(image, nldimg, ow, oh, op, kbase, R);
                memcpy(smooth, nldimg, sizeof(float) * (size_t)msz);
                sizes[j] = esigma * prm->derivative_factor;
                sigma_size = (int)(esigma * prm->derivative_factor + 0.5f);
                borders[j] = smax * sigma_size;
                okz_derivate(smooth, dx, dy, sigma_size, ow, oh, op);
                okz_hessian(dx, dy, smooth, sigma_size, ow, oh, op);
                continue;
            }
            esigma = prm->soffset * powf(2, (float)j / ms + i);         /* akaze.cpp:357 */
            curr_etime = 0.5f * esigma * esigma;
            ttime = curr_etime - last_etime;
            naux = okz_fed_tau(ttime, 1, tmax, prm->reordering, tau, 4096);
            sizes[j] = esigma * prm->derivative_factor / oratio;
            sigma_size = (int)(sizes[j] + 0.5f);
            borders[j] = smax * sigma_size;
            if (j == 0) {                                               /* akaze.cpp:369-392 */
                kcontrast *= 0.75f;
                float* oldnld = nldimg - mstep;
                okz_down_smooth(oldnld, nldimg, smooth, owhps[3 * (i - 1)], owhps[3 * (i - 1) + 1],
                                owhps[3 * (i - 1) + 2], ow, oh, op, k1);
                okz_flow(smooth, flow, prm->diffusivity, kcontrast, ow, oh, op);
                for (int k = 0; k < naux; k++) {
                    okz_nld_step(nldimg, flow, temp, tau[k], ow, oh, op);
                    memcpy(nldimg, temp, sizeof(float) * (size_t)msz);
                }
            } else {                                                    /* akaze.cpp:393-421 */
                float* oldnld = nldimg;
                nldimg += msz; smooth += msz; flow += msz; temp += msz;
                dx = flow; dy = temp;
                okz_lowpass(oldnld, smooth, ow, oh, op, k1, 2);
                okz_flow(smooth, flow, prm->diffusivity, kcontrast, ow, oh, op);
                okz_nld_step(oldnld, flow, nldimg, tau[0], ow, oh, op);
                for (int k = 1; k < naux; k++) {
                    okz_nld_step(nldimg, flow, temp, tau[k], ow, oh, op);
                    memcpy(nldimg, temp, sizeof(float) * (size_t)msz);
                }
            }
            okz_derivate(smooth, dx, dy, sigma_size, ow, oh, op);       /* akaze.cpp:423 */
            okz_hessian(dx, dy, smooth, sigma_size, ow, oh, op);
            last_etime = curr_etime;
        }
        float* dets = tmem + offsets[i] + ms_msz;                       /* akaze.cpp:431-433 */
        okz_extrema_map(dets, response_map, size_map, layer_map, params, i, ms, prm->dthreshold,
                        ow, oh, op, owhps[2]);
        psz = psz < borders[0] * oratio ? psz : borders[0] * oratio;
        mstep = ms_msz * 4;
        oratio *= 2;
    }

    int total = okz_nms(pts, max_pts, response_map, size_map, layer_map, (int)psz, owhps[0], owhps[1], owhps[2]);
    int num = total < max_pts ? total : max_pts;                        /* akaze.cpp:451 */

    float wtab[36];
    int idx1[488], idx2[488];
    okz_orient_weights(wtab);
    okz_compare_indices(idx1, idx2);
#pragma omp parallel for schedule(dynamic, 16)
    for (int n = 0; n < num; n++) {
        OkzPoint* pt = pts + n;
        int o = pt->octave / ms, s = pt->octave % ms;
        int ow = owhps[3 * o], oh = owhps[3 * o + 1], op = owhps[3 * o + 2];
        float* lt = tmem + offsets[o] + (size_t)s * osizes[o];
        float* det = tmem + offsets[o] + (size_t)(ms + s) * osizes[o];
        float* lx = tmem + offsets[o] + (size_t)(2 * ms + s) * osizes[o];
        float* ly = tmem + offsets[o] + (size_t)(3 * ms + s) * osizes[o];
        okz_refine_point(pt, det, o, op);
        pt->angle = 0.f;
        memset(pt->features, 0, OKZ_FLEN);
        if (desc) {
            if (!prm->upright) okz_orient_point(pt, lx, ly, o, ow, oh, op, wtab);
            okz_describe_point(pt, lt, lx, ly, o, ow, oh, op, prm->descriptor_pattern_size, idx1, idx2);
        }
        pt->match = -1; pt->distance = -1; pt->match_x = -1; pt->match_y = -1;
    }
    return num;
}
