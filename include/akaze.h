// akaze.h -- the CUDA-AKAZE public API (akaze.h:10-30 of the reference) on top of libhipakaze's C ABI.
//
// Drop-in for main.cpp-style callers: same namespace, names, argument order and types.  `int3` is
// HIP's vector type (layout-identical to CUDA's).  Images are DEVICE pointers to float32 in [0,1]
// with pitch whp0.z elements (main.cpp:174); AkazeData is caller-owned and filled in place.  Errors
// print to stderr and exit(-1) like the reference's CHECK (cuda_utils.h:18-37).
#pragma once
#include "akaze_structures.h"
#include "hip_utils.h"

namespace akaze
{
    void initAkazeData(AkazeData& data, const int max_pts, const bool host, const bool dev);     // akaze.h:10
    void freeAkazeData(AkazeData& data);                                                        // akaze.h:12
    void cuMatch(AkazeData& result1, AkazeData& result2);                                       // akaze.h:14

    // build-side addition: 2-NN ratio test (the reference's unused gMatch, akazed.cu:2028-2122) +
    // symmetric cross-check; fills result1 like cuMatch and returns the accepted matches in query order
    // (host array `matches`, capacity >= result1.num_pts; may be NULL to only count).
    int cuMatchKnn(AkazeData& result1, AkazeData& result2, hak_match_pair* matches, int ratio_num = 1, int ratio_den = 1,
                   bool cross_check = true);

    class Akazer
    {
    public:
        Akazer();
        ~Akazer();

        // akaze.h:25-26
        void init(int3 whp0, int _noctaves, int _max_scale, float _per, float _kcontrast, float _soffset, bool _reordering,
                  float _derivative_factor, float _dthreshold, int _diffusivity, int _descriptor_pattern_size);

        // akaze.h:29-30
        void detectAndCompute(float* image, AkazeData& result, int3 whp0, const bool desc = true);
        void fastDetectAndCompute(unsigned char* image, AkazeData& result, int3 whp0, const bool desc = true);

        // build-side additions (no reference counterpart)
        // both images of a pair + cuMatch(result1, result2) as ONE launch sequence and one synchronisation (hak_detect_and_compute_pair):
        // same results in result1 / result2 as the two detectAndCompute calls followed by cuMatch of main.cpp:201-209
        void detectAndComputePair(float* image1, float* image2, AkazeData& result1, AkazeData& result2, int3 whp0,
                                  const bool desc = true, const bool match = true);
        void setMaxPoints(int max_pts);      // capacity the context is built for (default 10000, main.cpp:155)
        void setUpright(bool upright);       // MLDB-upright extension
        hak_ctx* context() { return ctx; }

    private:
        hak_config cfg;
        int3 whp{0, 0, 0};
        hak_ctx* ctx = nullptr;      // owns the arena (the reference's omem; room for the two images of a pair call), freed in the destructor
        int ctx_w = 0, ctx_h = 0;
        void ensureContext(int w, int h);
    };
}
