// akaze_structures.h -- data types of the CUDA-AKAZE public API, re-declared over the C ABI.
//
// Same names / layout as the reference's header of the same name (akaze_structures.h:19-59) so
// existing callers compile unchanged; the POD itself is the C struct hak_point of hipakaze.h.
#pragma once
#include <cstddef>
#include "hipakaze.h"

// descriptor selector kept for source compatibility (akaze_structures.h:7-15): only MLDB (5) has kernels
#define FEATURE_TYPE 5
#define FLEN HAK_FLEN

namespace akaze
{
    typedef ::hak_point AkazePoint;                       // 104-byte POD, see hipakaze.h

    static_assert(sizeof(AkazePoint) == 104, "AkazePoint must stay 104 bytes");
    static_assert(offsetof(AkazePoint, features) == 24 && offsetof(AkazePoint, match) == 88 &&
                  offsetof(AkazePoint, match_y) == 100, "AkazePoint layout drifted");

    // akaze_structures.h:44-50 -- caller-owned handle filled in place by detectAndCompute / cuMatch
    struct AkazeData
    {
        int num_pts;            // valid points
        int max_pts;            // allocated points
        AkazePoint* h_data;     // host copy (malloc) or NULL
        AkazePoint* d_data;     // device array (hipMalloc) or NULL
    };

    enum DiffusivityType { PM_G1 = HAK_PM_G1, PM_G2 = HAK_PM_G2, WEICKERT = HAK_WEICKERT, CHARBONNIER = HAK_CHARBONNIER };
}
