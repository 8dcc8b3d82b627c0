// akaze.cpp -- akaze::Akazer / initAkazeData / freeAkazeData / cuMatch over the C ABI.
// Behavioural mirror of the reference's akaze.cpp:24-150 (same call contract and error style);
// all device work happens behind hipakaze.h.
#include "akaze.h"
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <set>

namespace
{
    void die(const char* what)
    {
        fprintf(stderr, "hip-akaze: %s failed: %s\n", what, hak_last_error());
        exit(-1);                                                     // cuda_utils.h:23
    }
    // host buffers handed out pinned (device-visible), so that freeAkazeData knows how to release them
    std::mutex g_pinned_mu;
    std::set<void*> g_pinned;
}

namespace akaze
{
    void initAkazeData(AkazeData& data, const int max_pts, const bool host, const bool dev)       // akaze.cpp:26-40
    {
        data.num_pts = 0;
        data.max_pts = max_pts;
        data.h_data = NULL;
        if (host) {
            // pinned host memory: detectAndCompute's launch sequence writes the records (and the count) straight into it, so a
            // call ends with ONE synchronisation instead of a count copy, a sync and a record copy (akaze.cpp:134-139 does the
            // latter on pageable memory).  Without a usable device the plain allocation of the reference remains.
            void* p = NULL;
            if (hak_host_alloc(&p, (long)(sizeof(AkazePoint) * (size_t)max_pts)) == 0 && p) {
                std::lock_guard<std::mutex> lock(g_pinned_mu);
                g_pinned.insert(p);
            } else p = malloc(sizeof(AkazePoint) * (size_t)max_pts);
            data.h_data = (AkazePoint*)p;
        }
        data.d_data = NULL;
        if (dev && hak_points_alloc(&data.d_data, max_pts)) die("initAkazeData");
    }

    void freeAkazeData(AkazeData& data)                                                          // akaze.cpp:43-52
    {
        if (data.d_data != NULL && hak_points_free(data.d_data)) die("freeAkazeData");
        if (data.h_data != NULL) {
            bool pinned;
            { std::lock_guard<std::mutex> lock(g_pinned_mu); pinned = g_pinned.erase(data.h_data) > 0; }
            if (pinned) { if (hak_host_free(data.h_data)) die("freeAkazeData"); }
            else free(data.h_data);
        }
        data.d_data = NULL;
        data.h_data = NULL;
        data.num_pts = 0;
        data.max_pts = 0;
    }

    void cuMatch(AkazeData& result1, AkazeData& result2)                                         // akaze.cpp:55-64
    {
        if (hak_match(NULL, result1.d_data, result1.num_pts, result2.d_data, result2.num_pts, result1.h_data))
            die("cuMatch");
    }

    int cuMatchKnn(AkazeData& result1, AkazeData& result2, hak_match_pair* matches, int ratio_num, int ratio_den, bool cross_check)
    {
        int count = 0;
        hak_match_pair* d_out = nullptr;
        const int cap = result1.num_pts > 0 ? result1.num_pts : 1;
        if (matches && hipMalloc((void**)&d_out, sizeof(hak_match_pair) * (size_t)cap) != hipSuccess) die("cuMatchKnn alloc");
        if (hak_match_knn2(NULL, result1.d_data, result1.num_pts, result2.d_data, result2.num_pts, ratio_num, ratio_den,
                           cross_check ? 1 : 0, 0, result1.h_data, d_out, &count, matches))
            die("cuMatchKnn");
        if (d_out) (void)hipFree(d_out);
        return count;
    }

    Akazer::Akazer() { hak_default_config(&cfg); }

    Akazer::~Akazer() { hak_destroy(ctx); }                              // akaze.cpp:74-77

    void Akazer::setMaxPoints(int max_pts) { cfg.max_pts = max_pts; hak_destroy(ctx); ctx = nullptr; }
    void Akazer::setUpright(bool upright) { cfg.upright = upright ? 1 : 0; hak_destroy(ctx); ctx = nullptr; }

    void Akazer::ensureContext(int w, int h)
    {
        if (ctx && ctx_w == w && ctx_h == h) return;
        hak_destroy(ctx);
        ctx = nullptr;
        if (hak_create(&cfg, w, h, &ctx)) die("Akazer: hak_create");
        ctx_w = w;
        ctx_h = h;
    }

    void Akazer::init(int3 whp0, int _noctaves, int _max_scale, float _per, float _kcontrast, float _soffset, bool _reordering,
                      float _derivative_factor, float _dthreshold, int _diffusivity, int _descriptor_pattern_size)
    {
        whp = whp0;                                                                             // akaze.cpp:83-95
        cfg.noctaves = _noctaves;
        cfg.max_scale = _max_scale;
        cfg.per = _per;
        cfg.kcontrast = _kcontrast;
        cfg.soffset = _soffset;
        cfg.reordering = _reordering ? 1 : 0;
        cfg.derivative_factor = _derivative_factor;
        cfg.dthreshold = _dthreshold;
        cfg.diffusivity = _diffusivity;
        cfg.descriptor_pattern_size = _descriptor_pattern_size;
        // ONE context serves detectAndCompute (one image of it) and detectAndComputePair (both): a second context would double the
        // streams of the process, and launch chains that share a hardware queue run one after the other
        cfg.batch = 2;
        hak_destroy(ctx);
        ctx = nullptr;
        ensureContext(whp.x, whp.y);                  // arena allocated once for the init size ("reused", akaze.cpp:109-113)
    }

    void Akazer::detectAndCompute(float* image, AkazeData& result, int3 whp0, const bool desc)   // akaze.cpp:101-150
    {
        ensureContext(whp0.x, whp0.y);                // other size than init(): new arena (akaze.cpp:114-117)
        // the clamp of THIS call is result.max_pts, as the reference's setMaxNumPoints(result.max_pts) (akaze.cpp:246, 451);
        // nothing in the context is sized by it, so neither a smaller nor a larger AkazeData rebuilds anything
        if (hak_detect_and_compute(ctx, image, whp0.z, result.d_data, result.max_pts, &result.num_pts, result.h_data, desc ? 1 : 0))
            die("detectAndCompute");
    }

    void Akazer::detectAndComputePair(float* image1, float* image2, AkazeData& result1, AkazeData& result2, int3 whp0, const bool desc,
                                      const bool match)
    {
        ensureContext(whp0.x, whp0.y);
        if (hak_detect_and_compute_pair(ctx, image1, image2, whp0.z, result1.d_data, result2.d_data, result1.max_pts, result2.max_pts,
                                        &result1.num_pts, &result2.num_pts, result1.h_data, result2.h_data, desc ? 1 : 0, match ? 1 : 0))
            die("detectAndComputePair");
    }

    void Akazer::fastDetectAndCompute(unsigned char* image, AkazeData& result, int3 whp0, const bool desc)   // akaze.cpp:153-201
    {
        ensureContext(whp0.x, whp0.y);
        if (hak_fast_detect_and_compute(ctx, image, whp0.z, result.d_data, result.max_pts, &result.num_pts, result.h_data, desc ? 1 : 0))
            die("fastDetectAndCompute");
    }
}
