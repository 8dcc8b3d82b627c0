// asan_main.cpp -- drives host/akaze.cpp (the C++ drop-in layer) against stub_hipakaze.cpp under ASan + UBSan:
// the call patterns of the reference demo (main.cpp:190-232) plus the layer's own edge cases.
#include "akaze.h"
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <vector>

extern int g_live_ctx, g_live_dev, g_live_host;
#define REQUIRE(c) do { if (!(c)) { fprintf(stderr, "asan_main: %s:%d: %s\n", __FILE__, __LINE__, #c); exit(1); } } while (0)

int main()
{
    using namespace akaze;
    std::vector<float> img(1920 * 1080, 0.5f);
    std::vector<unsigned char> img8(1920 * 1080, 128);
    {
        AkazeData d1, d2, small, hostless, devless;
        initAkazeData(d1, 10000, true, true);                       // main.cpp:193-194
        initAkazeData(d2, 10000, true, true);
        initAkazeData(small, 7, true, true);                        // a clamp far below what the image holds
        initAkazeData(hostless, 10000, false, true);
        initAkazeData(devless, 100, true, false);
        REQUIRE(hostless.h_data == NULL && devless.d_data == NULL && g_live_host == 4 && g_live_dev == 4);
        std::unique_ptr<Akazer> det(new Akazer);
        int3 whp{1920, 1080, 1920};
        det->init(whp, 4, 4, 0.7f, 0.03f, 1.6f, true, 1.5f, 0.001f, 1, 10);
        REQUIRE(g_live_ctx == 1);
        for (int i = 0; i < 3; i++) {                               // main.cpp:199-205
            det->detectAndCompute(img.data(), d1, whp, true);
            det->detectAndCompute(img.data(), d2, whp, true);
        }
        REQUIRE(d1.num_pts == 1920 * 1080 / 4096 && d1.h_data[d1.num_pts - 1].match == -1);
        cuMatch(d1, d2);                                            // main.cpp:209
        REQUIRE(d1.h_data[5].match == 5 && d1.h_data[5].distance == 7);
        det->detectAndCompute(img.data(), small, whp, true);        // per-call clamp = the caller's max_pts (akaze.cpp:246, 451)
        REQUIRE(small.num_pts == 7);
        det->detectAndCompute(img.data(), hostless, whp, false);    // no host copy requested
        REQUIRE(hostless.num_pts == d1.num_pts);
        int3 whp2{1280, 720, 1280};                                 // another size than init(): a new arena (akaze.cpp:114-117)
        det->detectAndCompute(img.data(), d2, whp2, true);
        REQUIRE(d2.num_pts == 1280 * 720 / 4096 && g_live_ctx == 1);
        det->fastDetectAndCompute(img8.data(), d2, whp, true);
        REQUIRE(d2.num_pts == d1.num_pts && g_live_ctx == 1);
        std::vector<hak_match_pair> pairs(d1.num_pts);
        REQUIRE(cuMatchKnn(d1, d2, pairs.data()) == (d1.num_pts + 1) / 2 && pairs[1].query == 2);
        REQUIRE(cuMatchKnn(d1, d2, NULL) == (d1.num_pts + 1) / 2);
        cuMatch(d1, hostless);                                      // train side without a host buffer
        d2.num_pts = 0;
        cuMatch(d1, d2);                                            // empty train set (D10)
        REQUIRE(d1.h_data[0].match == -1);
        // the pair call on the same (two-image) context; results as the three calls give them; the smaller capacity is the clamp
        det->detectAndComputePair(img.data(), img.data(), d1, d2, whp, true, true);
        REQUIRE(d1.num_pts == 1920 * 1080 / 4096 && d2.num_pts == d1.num_pts && d1.h_data[7].match == 7 && g_live_ctx == 1);
        det->detectAndComputePair(img.data(), img.data(), d1, small, whp, true, false);
        REQUIRE(d1.num_pts == 7 && small.num_pts == 7);
        det->detectAndComputePair(img.data(), img.data(), d1, d2, whp2, false, true);   // another size: the context is rebuilt
        REQUIRE(d1.num_pts == 1280 * 720 / 4096 && g_live_ctx == 1);
        det->setMaxPoints(500);                                     // drops the context; the next call rebuilds it
        det->setUpright(true);
        REQUIRE(g_live_ctx == 0);
        det->detectAndCompute(img.data(), d1, whp, true);
        det->init(whp2, 3, 3, 0.7f, 0.03f, 1.2f, false, 1.5f, 0.001f, 3, 8);    // re-init replaces the context
        REQUIRE(g_live_ctx == 1);
        freeAkazeData(d1); freeAkazeData(d2); freeAkazeData(small); freeAkazeData(hostless); freeAkazeData(devless);
        freeAkazeData(d1);                                          // freeing twice is harmless (pointers are reset)
        REQUIRE(d1.h_data == NULL && d1.max_pts == 0);
    }
    REQUIRE(g_live_ctx == 0 && g_live_dev == 0 && g_live_host == 0);
    printf("asan_main: host layer clean\n");
    return 0;
}
