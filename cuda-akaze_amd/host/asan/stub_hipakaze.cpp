// stub_hipakaze.cpp -- a HOST-MEMORY stand-in for the entry points of libhipakaze.so (and the two HIP runtime calls) that the
// C++ layer host/akaze.cpp uses, so that the layer's own logic -- pinned-buffer registry, context re-creation on a size change,
// per-call clamp, error style -- can run under AddressSanitizer / UBSan on the CPU (GPU sanitizers are not available on this
// pool).  Test scaffolding for `make -C cuda-akaze_amd/host asan` only: it computes nothing of AKAZE and is never shipped.
#include "hipakaze.h"
#include <hip/hip_runtime_api.h>
#include <cstdlib>
#include <cstring>
#include <string>

struct hak_ctx { hak_config cfg; int w, h; };
static std::string g_err;
int g_live_ctx = 0, g_live_dev = 0, g_live_host = 0;          // leak accounting checked by the driver

extern "C" {
const char* hak_last_error(void) { return g_err.c_str(); }
void hak_default_config(hak_config* c)
{
    memset(c, 0, sizeof(*c));
    c->noctaves = 4; c->max_scale = 4; c->per = 0.7f; c->kcontrast = 0.03f; c->soffset = 1.6f; c->reordering = 1;
    c->derivative_factor = 1.5f; c->dthreshold = 0.001f; c->diffusivity = HAK_PM_G2; c->descriptor_pattern_size = 10;
    c->max_pts = 10000; c->batch = 1;
}
int hak_create(const hak_config* cfg, int w, int h, hak_ctx** out)
{
    if (w < 80 || h < 80) { g_err = "image smaller than 80 px"; return 1; }
    *out = new hak_ctx{*cfg, w, h};
    g_live_ctx++;
    return 0;
}
void hak_destroy(hak_ctx* c) { if (c) { g_live_ctx--; delete c; } }
int hak_host_alloc(void** p, long bytes) { *p = malloc((size_t)bytes); g_live_host += *p != nullptr; return *p ? 0 : 1; }
int hak_host_free(void* p) { g_live_host--; free(p); return 0; }
int hak_points_alloc(hak_point** d, int count) { *d = (hak_point*)malloc(sizeof(hak_point) * (size_t)count); g_live_dev++; return *d ? 0 : 1; }
int hak_points_free(hak_point* d) { g_live_dev--; free(d); return 0; }
// "detects" w*h/4096 points (a number that depends on the image size), clamped to the CALL's max_pts; writes every byte of
// every record it reports, so an undersized caller buffer is an ASan error
static int fake_detect(hak_ctx* c, int pitch, hak_point* d_points, int max_pts, int* num_pts, hak_point* h_points)
{
    if (!c || !d_points || !num_pts) { g_err = "null argument"; return 1; }
    if (pitch < c->w) { g_err = "pitch smaller than width"; return 1; }
    int n = c->w * c->h / 4096;
    if (n > max_pts) n = max_pts;
    for (int i = 0; i < n; i++) {
        memset(&d_points[i], 0, sizeof(hak_point));
        d_points[i].x = (float)(i % c->w); d_points[i].y = (float)(i / c->w); d_points[i].octave = i % 16;
        d_points[i].features[i % HAK_FLEN] = (unsigned char)i;
        d_points[i].match = -1;
    }
    *num_pts = n;
    if (h_points && n) memcpy(h_points, d_points, sizeof(hak_point) * (size_t)n);
    return 0;
}
int hak_detect_and_compute(hak_ctx* c, const float* img, int pitch, hak_point* d, int max_pts, int* n, hak_point* h, int)
{ return img ? fake_detect(c, pitch, d, max_pts, n, h) : (g_err = "null image", 1); }
int hak_fast_detect_and_compute(hak_ctx* c, const unsigned char* img, int pitch, hak_point* d, int max_pts, int* n, hak_point* h, int)
{ return img ? fake_detect(c, pitch, d, max_pts, n, h) : (g_err = "null image", 1); }
int hak_match(hak_ctx*, hak_point* p1, int n1, const hak_point* p2, int n2, hak_point* h1);
int hak_detect_and_compute_pair(hak_ctx* c, const float* i1, const float* i2, int pitch, hak_point* d1, hak_point* d2, int m1, int m2,
                                int* n1, int* n2, hak_point* h1, hak_point* h2, int, int match)
{
    if (!c || c->cfg.batch < 2) { g_err = "pair call needs batch >= 2"; return 1; }
    const int clamp = m1 < m2 ? m1 : m2;
    if (!i1 || !i2 || fake_detect(c, pitch, d1, clamp, n1, nullptr) || fake_detect(c, pitch, d2, clamp, n2, nullptr)) return 1;
    if (match) hak_match(nullptr, d1, *n1, d2, *n2, nullptr);
    if (h1 && *n1) memcpy(h1, d1, sizeof(hak_point) * (size_t)*n1);
    if (h2 && *n2) memcpy(h2, d2, sizeof(hak_point) * (size_t)*n2);
    return 0;
}
int hak_match(hak_ctx*, hak_point* p1, int n1, const hak_point* p2, int n2, hak_point* h1)
{
    for (int i = 0; i < n1; i++) {
        p1[i].match = n2 ? i % n2 : -1; p1[i].distance = n2 ? 7 : -1;
        p1[i].match_x = n2 ? p2[i % n2].x : -1; p1[i].match_y = n2 ? p2[i % n2].y : -1;
        if (h1) { h1[i].match = p1[i].match; h1[i].distance = p1[i].distance; h1[i].match_x = p1[i].match_x; h1[i].match_y = p1[i].match_y; }
    }
    return 0;
}
int hak_match_knn2(hak_ctx*, hak_point* p1, int n1, const hak_point* p2, int n2, int, int, int, int, hak_point* h1,
                   hak_match_pair* d_out, int* count, hak_match_pair* h_out)
{
    hak_match(nullptr, p1, n1, p2, n2, h1);
    int c = 0;
    for (int i = 0; i < n1 && n2; i += 2, c++)
        if (d_out) { d_out[c] = hak_match_pair{i, i % n2, 7, 9, p1[i].x, p1[i].y, p2[i % n2].x, p2[i % n2].y}; if (h_out) h_out[c] = d_out[c]; }
    *count = c;
    return 0;
}
// the two HIP runtime calls of cuMatchKnn
hipError_t hipMalloc(void** p, size_t n) { *p = malloc(n); return *p ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipFree(void* p) { free(p); return hipSuccess; }
}
