/*
 * asan_main.c -- runs the CPU oracle (float path, integer FAST path, both matchers) under AddressSanitizer + UBSan on a few
 * seeded scenes, including odd sizes, a clamp below the keypoint count, no-descriptor and upright runs (SURVEY.md 5:
 * "sanitizers on the CPU build").  TEST INFRASTRUCTURE ONLY, like everything in oracle/: `make -C oracle asan`.
 * Signed wrap-around in the FAST path is spelled out with unsigned arithmetic there, so UBSan's signed-overflow check stays on.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef struct { float x, y; int octave; float response, size, angle; unsigned char features[61]; int match, distance; float match_x, match_y; } Pt;
typedef struct { int noctaves, max_scale; float per, kcontrast, soffset; int reordering; float derivative_factor, dthreshold;
                 int diffusivity, descriptor_pattern_size, upright; } Prm;
typedef struct { int query, train, distance, second; float x1, y1, x2, y2; } MatchPair;

long okz_arena_floats(int w, int h, int p, int noctaves, int max_scale);
long fkz_arena_ints(int w, int h, int p, int noctaves, int max_scale);
int okz_detect_and_compute(const float* image, int w, int h, int p, const Prm* prm, Pt* pts, int max_pts, int desc, float* tmem, float* kc);
int fkz_detect_and_compute(const unsigned char* image, int w, int h, int sp, int p, const Prm* prm, Pt* pts, int max_pts, int desc, int* tmem, int* kc);
void okz_match(Pt* pts1, int n1, const Pt* pts2, int n2);
int okz_match_knn2(Pt* pts1, int n1, const Pt* pts2, int n2, int ratio_num, int ratio_den, int cross, int max_dist, MatchPair* out);
int okz_sizeof_point(void);

static unsigned rng_state;
static unsigned rnd(void) { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 17; rng_state ^= rng_state << 5; return rng_state; }

/* gradient + rectangles + discs + a little noise: enough structure for a few hundred keypoints */
static void scene(unsigned char* u8, int w, int h, unsigned seed, int shift)
{
    rng_state = seed * 2654435761u + 1;
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) u8[(size_t)y * w + x] = (unsigned char)(60 + (x + shift) * 80 / w + y * 40 / h);
    int nshapes = 20 + w * h / 4000;
    for (int s = 0; s < nshapes; s++) {
        int cx = (int)(rnd() % (unsigned)w) + shift, cy = (int)(rnd() % (unsigned)h), r = 3 + (int)(rnd() % 14), v = (int)(rnd() % 256), disc = rnd() & 1;
        for (int y = cy - r; y <= cy + r; y++)
            for (int x = cx - r; x <= cx + r; x++)
                if (x >= 0 && x < w && y >= 0 && y < h && (!disc || (x - cx) * (x - cx) + (y - cy) * (y - cy) <= r * r))
                    u8[(size_t)y * w + x] = (unsigned char)v;
    }
    for (size_t i = 0; i < (size_t)w * h; i++) { int v = u8[i] + (int)(rnd() % 5) - 2; u8[i] = (unsigned char)(v < 0 ? 0 : v > 255 ? 255 : v); }
}

static int run_case(int w, int h, const Prm* prm, int max_pts, int desc)
{
    int p = (w + 127) / 128 * 128, n[2], nf[2];
    unsigned char* u8 = malloc((size_t)w * h);
    float* img = malloc(sizeof(float) * (size_t)h * p);
    float* arena = malloc(sizeof(float) * (size_t)okz_arena_floats(w, h, p, prm->noctaves, prm->max_scale));
    int* iarena = malloc(sizeof(int) * (size_t)fkz_arena_ints(w, h, p, prm->noctaves, prm->max_scale));
    Pt* pts[2], *fpts[2];
    for (int k = 0; k < 2; k++) {
        pts[k] = malloc(sizeof(Pt) * (size_t)max_pts);              /* exactly max_pts records: an overrun is an ASan error */
        fpts[k] = malloc(sizeof(Pt) * (size_t)max_pts);
        scene(u8, w, h, 7u + (unsigned)(w * 31 + h), k * 3);
        for (int y = 0; y < h; y++)
            for (int x = 0; x < p; x++) img[(size_t)y * p + x] = x < w ? u8[(size_t)y * w + x] * (float)(1.0 / 255.0) : 0.f;
        float kc; int ikc;
        n[k] = okz_detect_and_compute(img, w, h, p, prm, pts[k], max_pts, desc, arena, &kc);
        nf[k] = fkz_detect_and_compute(u8, w, h, w, p, prm, fpts[k], max_pts, desc, iarena, &ikc);
    }
    okz_match(pts[0], n[0], pts[1], n[1]);
    okz_match(fpts[0], nf[0], fpts[1], nf[1]);
    okz_match(pts[0], n[0], pts[1], n[1] < 5 ? n[1] : 5);           /* n2 < 16 (D10) */
    okz_match(pts[0], n[0], pts[1], 0);
    MatchPair* out = malloc(sizeof(MatchPair) * (size_t)(n[0] > 0 ? n[0] : 1));
    int acc = okz_match_knn2(pts[0], n[0], pts[1], n[1], 4, 5, 1, 96, out);
    printf("  %4d x %-4d oct %d ms %d diff %d pat %2d upright %d max_pts %5d desc %d: float %d / %d, FAST %d / %d, knn2 %d\n", w, h, prm->noctaves,
           prm->max_scale, prm->diffusivity, prm->descriptor_pattern_size, prm->upright, max_pts, desc, n[0], n[1], nf[0], nf[1], acc);
    free(out);
    for (int k = 0; k < 2; k++) { free(pts[k]); free(fpts[k]); }
    free(u8); free(img); free(arena); free(iarena);
    return n[0] + nf[0];
}

int main(void)
{
    if (okz_sizeof_point() != (int)sizeof(Pt) || sizeof(Pt) != 104) { fprintf(stderr, "record layout drifted\n"); return 1; }
    Prm d = {4, 4, 0.7f, 0.03f, 1.6f, 1, 1.5f, 0.001f, 1, 10, 0};     /* main.cpp:156-166 */
    int total = 0;
    total += run_case(320, 240, &d, 10000, 1);
    total += run_case(211, 173, &d, 10000, 1);                         /* odd sizes, two octaves survive the 80 px rule */
    total += run_case(400, 300, &d, 25, 1);                            /* clamp below the keypoint count */
    total += run_case(324, 200, &d, 10000, 0);                         /* no descriptors */
    Prm u = d; u.upright = 1; u.noctaves = 3; u.max_scale = 3;
    total += run_case(360, 280, &u, 10000, 1);
    Prm c = d; c.diffusivity = 3; c.descriptor_pattern_size = 12; c.soffset = 1.2f;
    total += run_case(300, 220, &c, 10000, 1);
    Prm g = d; g.diffusivity = 0; g.descriptor_pattern_size = 6;
    total += run_case(256, 256, &g, 10000, 1);
    Prm wk = d; wk.diffusivity = 2; wk.reordering = 0;
    total += run_case(288, 200, &wk, 10000, 1);
    /* the integer pipeline far outside its range: two sublevels per octave make FED cycles of ~31 steps at octave 3 (tau up to 50), the
     * 16-bit truncations blow the plane up, sums of squares wrap negative and the conductivity casts see NaN / inf (f2i_sat); found by
     * tests/fuzz_parity.py (seed 5, case 1504) */
    Prm b = d; b.max_scale = 2; b.diffusivity = 3; b.derivative_factor = 1.0f; b.dthreshold = 0.0005f; b.descriptor_pattern_size = 6;
    total += run_case(754, 869, &b, 3000, 1);
    b.diffusivity = 0;
    total += run_case(720, 700, &b, 3000, 1);
    if (total < 200) { fprintf(stderr, "asan_main: the scenes hold too few keypoints (%d) to exercise the tail\n", total); return 1; }
    printf("asan_main: oracle clean\n");
    return 0;
}
