// fed_ref_shim.cpp -- C entry point around the REFERENCE's own fed.cpp.
//
// TEST INFRASTRUCTURE ONLY.  Built by oracle/Makefile together with
// /root/reference/fed.cpp (compiled where it lies, never copied) into
// oracle/_ref/libfedref.so.  Used to pin oracle/akaze_oracle.c:okz_fed_tau and
// to generate tests/golden/fed_tau.json.  Not present on the GPU box unless the
// prebuilt .so travelled with the snapshot; tests skip when it is missing.
#include "fed.h"   // -I/root/reference
#include <vector>

extern "C" int fedref_tau_by_process_time(float T, int M, float tau_max, int reordering, float* out, int cap)
{
    std::vector<float> tau;
    int n = fed_tau_by_process_time(T, M, tau_max, reordering != 0, tau);
    for (int i = 0; i < n && i < cap; i++) out[i] = tau[i];
    return n;
}
