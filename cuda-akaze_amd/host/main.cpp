// main.cpp -- counterpart of the reference demo cudaAkazeDemo2 (main.cpp:128-233) without OpenCV:
// reads two binary PGMs (or synthesises a pair), uploads them, runs detectAndCompute on both images
// `nrepeats` times and cuMatch once, and prints the same five result lines.
//
//   hipakaze_demo [device] [left.pgm right.pgm] [nrepeats] [--dump file] [--api-checks] [--pair]
//
// --dump file   writes the host-side results as raw 104-byte AkazePoint records:
//               int32 n1, n2, then n1 + n2 records of the float path (image 1 after cuMatch),
//               then int32 f1, f2 and f1 + f2 records of the FAST path (image 1 after cuMatch).
// --pair        the loop calls Akazer::detectAndComputePair (both images + cuMatch in ONE launch sequence, akaze.h) instead of
//               detectAndCompute x 2 (+ cuMatch after the loop); the printed counts and the dumped records are the same
// --api-checks  additionally drives Akazer through the call patterns of akaze.cpp:101-150 that the demo loop does not:
//               an AkazeData smaller and larger than the default capacity, and an image size other than init()'s.
#include "akaze.h"
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <memory>
#include <string>
#include <vector>

static bool readPgm(const std::string& path, std::vector<unsigned char>& px, int& w, int& h)
{
    std::ifstream f(path, std::ios::binary);
    if (!f) return false;
    std::string magic;
    f >> magic;
    if (magic != "P5") return false;
    auto next = [&]() {
        std::string t;
        while (f >> t) {
            if (t[0] == '#') { std::getline(f, t); continue; }
            return std::stoi(t);
        }
        return -1;
    };
    w = next(); h = next();
    int maxv = next();
    if (w <= 0 || h <= 0 || maxv != 255) return false;
    f.get();
    px.resize((size_t)w * h);
    f.read((char*)px.data(), px.size());
    return (bool)f;
}

static void synthPair(std::vector<unsigned char>& a, std::vector<unsigned char>& b, int w, int h)
{
    a.resize((size_t)w * h); b.resize((size_t)w * h);
    unsigned s = 1;
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 17; s ^= s << 5; return s; };      // xorshift32, seed 1
    std::vector<float> img((size_t)w * h);
    for (int y = 0; y < h; y++) for (int x = 0; x < w; x++) img[(size_t)y * w + x] = 0.35f + 0.25f * x / w + 0.15f * y / h;
    for (int n = 0; n < 180; n++) {
        int cx = rnd() % w, cy = rnd() % h, rw = 8 + rnd() % (w / 10), rh = 8 + rnd() % (h / 10);
        float amp = (0.08f + (rnd() % 1000) * 0.00037f) * ((rnd() & 1) ? 1.f : -1.f);
        for (int y = std::max(0, cy - rh / 2); y < std::min(h, cy + rh / 2); y++)
            for (int x = std::max(0, cx - rw / 2); x < std::min(w, cx + rw / 2); x++) img[(size_t)y * w + x] += amp;
    }
    for (size_t i = 0; i < img.size(); i++) a[i] = (unsigned char)std::min(255.f, std::max(0.f, std::round(img[i] * 255.f)));
    for (int y = 0; y < h; y++) for (int x = 0; x < w; x++)          // second view: 20 px shift
        b[(size_t)y * w + x] = a[(size_t)std::min(h - 1, y + 20) * w + std::min(w - 1, x + 20)];
}

static void dumpPoints(std::ofstream& f, const akaze::AkazeData& a, const akaze::AkazeData& b)
{
    const int n[2] = {a.num_pts, b.num_pts};
    f.write((const char*)n, sizeof(n));
    f.write((const char*)a.h_data, sizeof(akaze::AkazePoint) * (size_t)a.num_pts);
    f.write((const char*)b.h_data, sizeof(akaze::AkazePoint) * (size_t)b.num_pts);
}

int main(int argc, char** argv)
{
    // One image per call keeps four launch chains in flight: ask the HIP runtime for more than its default of four
    // hardware queues -- before its first call, and only if the user has not chosen a value ("Hardware queues")
    // (--pair: one launch sequence per call, replayed as a graph -- the runtime's default of four queues is the better one there)
    {
        bool pair = false;
        for (int i = 1; i < argc; i++) pair |= !strcmp(argv[i], "--pair");
        if (!pair) setenv("GPU_MAX_HW_QUEUES", "8", 0);
    }
    std::cout << "===== Registration by HIP-AKAZE (MI355X) =====" << std::endl;
    std::string dumpPath;
    bool apiChecks = false, pairCalls = false;
    {   // strip the options; what is left are the reference demo's positional arguments (main.cpp:131-135)
        int n = 1;
        for (int i = 1; i < argc; i++) {
            if (!strcmp(argv[i], "--dump") && i + 1 < argc) dumpPath = argv[++i];
            else if (!strcmp(argv[i], "--api-checks")) apiChecks = true;
            else if (!strcmp(argv[i], "--pair")) pairCalls = true;
            else argv[n++] = argv[i];
        }
        argc = n;
    }
    int devNum = argc > 1 ? std::atoi(argv[1]) : 0;
    int nrepeats = argc > 4 ? std::atoi(argv[4]) : 100;
    std::ofstream dump;
    if (!dumpPath.empty()) {
        dump.open(dumpPath, std::ios::binary);
        if (!dump) { std::cerr << "cannot open " << dumpPath << std::endl; return 1; }
    }
    std::vector<unsigned char> l8, r8;
    int w = 1920, h = 1080, w2 = 0, h2 = 0;
    if (argc > 3) {
        if (!readPgm(argv[2], l8, w, h) || !readPgm(argv[3], r8, w2, h2) || w != w2 || h != h2) {
            std::cerr << "cannot read the PGM pair" << std::endl;
            return 1;
        }
    } else synthPair(l8, r8, w, h);
    std::vector<float> limg(l8.size()), rimg(r8.size());
    for (size_t i = 0; i < l8.size(); i++) { limg[i] = (float)(l8[i] * (1.0 / 255.0)); rimg[i] = (float)(r8[i] * (1.0 / 255.0)); }   // main.cpp:149
    std::cout << "Image size = (" << w << "," << h << ")" << std::endl;

    // configuration: main.cpp:155-166
    int max_npts = 10000, noctaves = 4, max_scale = 4;
    float per = 0.7f, kcontrast = 0.03f, soffset = 1.6f, derivative_factor = 1.5f, dthreshold = 0.001f;
    bool reordering = true;
    int diffusivity = 1, descriptor_pattern_size = 10;

    std::cout << "Initializing data..." << std::endl;
    initDevice(devNum);
    GpuTimer timer(0);
    int3 whp1, whp2;
    whp1.x = w; whp1.y = h; whp1.z = iAlignUp(w, 128);
    whp2 = whp1;
    float *img1 = NULL, *img2 = NULL;
    CHECK(hipMalloc((void**)&img1, sizeof(float) * (size_t)whp1.y * whp1.z));
    CHECK(hipMalloc((void**)&img2, sizeof(float) * (size_t)whp2.y * whp2.z));
    CHECK(hipMemcpy2D(img1, sizeof(float) * whp1.z, limg.data(), sizeof(float) * w, sizeof(float) * w, h, hipMemcpyHostToDevice));
    CHECK(hipMemcpy2D(img2, sizeof(float) * whp2.z, rimg.data(), sizeof(float) * w, sizeof(float) * w, h, hipMemcpyHostToDevice));
    float t0 = timer.read();

    akaze::AkazeData akaze_data1, akaze_data2;
    akaze::initAkazeData(akaze_data1, max_npts, true, true);
    akaze::initAkazeData(akaze_data2, max_npts, true, true);
    std::unique_ptr<akaze::Akazer> detector(new akaze::Akazer);
    detector->init(whp1, noctaves, max_scale, per, kcontrast, soffset, reordering, derivative_factor, dthreshold, diffusivity,
                   descriptor_pattern_size);

    float t1 = timer.read();
    for (int i = 0; i < nrepeats; i++) {
        if (pairCalls) detector->detectAndComputePair(img1, img2, akaze_data1, akaze_data2, whp1, true, true);
        else {
            detector->detectAndCompute(img1, akaze_data1, whp1, true);
            detector->detectAndCompute(img2, akaze_data2, whp2, true);
        }
    }
    float t2 = timer.read();
    if (!pairCalls) akaze::cuMatch(akaze_data1, akaze_data2);
    float t3 = timer.read();

    int nmatch = 0;
    for (int i = 0; i < akaze_data1.num_pts; i++) nmatch += akaze_data1.h_data[i].match >= 0;
    std::cout << "Number of features1: " << akaze_data1.num_pts << std::endl
              << "Number of features2: " << akaze_data2.num_pts << std::endl
              << "Number of accepted matches: " << nmatch << std::endl;
    std::cout << "Time for allocating image memory:  " << t0 << std::endl
              << "Time for allocating point memory:  " << t1 - t0 << std::endl
              << "Time of detection and computation: " << (t2 - t1) / nrepeats << std::endl
              << "Time of matching AKAZE keypoints:   " << (t3 - t2) << std::endl;

    if (dump.is_open()) dumpPoints(dump, akaze_data1, akaze_data2);

    // match post-processing (build-side addition): ratio 4/5 + cross-check, compacted on the device
    std::vector<hak_match_pair> good(akaze_data1.num_pts > 0 ? akaze_data1.num_pts : 1);
    float t4 = timer.read();
    int ngood = akaze::cuMatchKnn(akaze_data1, akaze_data2, good.data(), 4, 5, true);
    float t5 = timer.read();
    std::cout << "2-NN ratio 0.8 + cross-check matches: " << ngood << "  (" << t5 - t4 << " ms)" << std::endl;

    // ---- the reference's second demo (main.cpp:227-300): the integer FAST path on the uint8 images
    std::cout << "===== FAST (16.16 fixed-point) path =====" << std::endl;
    unsigned char *fimg1 = NULL, *fimg2 = NULL;
    CHECK(hipMalloc((void**)&fimg1, (size_t)whp1.y * whp1.z));
    CHECK(hipMalloc((void**)&fimg2, (size_t)whp2.y * whp2.z));
    CHECK(hipMemcpy2D(fimg1, whp1.z, l8.data(), w, w, h, hipMemcpyHostToDevice));
    CHECK(hipMemcpy2D(fimg2, whp2.z, r8.data(), w, w, h, hipMemcpyHostToDevice));
    float f1 = timer.read();
    for (int i = 0; i < nrepeats; i++) {
        detector->fastDetectAndCompute(fimg1, akaze_data1, whp1, true);
        detector->fastDetectAndCompute(fimg2, akaze_data2, whp2, true);
    }
    float f2 = timer.read();
    akaze::cuMatch(akaze_data1, akaze_data2);
    float f3 = timer.read();
    nmatch = 0;
    for (int i = 0; i < akaze_data1.num_pts; i++) nmatch += akaze_data1.h_data[i].match >= 0;
    std::cout << "Number of features1: " << akaze_data1.num_pts << std::endl
              << "Number of features2: " << akaze_data2.num_pts << std::endl
              << "Number of accepted matches: " << nmatch << std::endl
              << "Time of detection and computation: " << (f2 - f1) / nrepeats << std::endl
              << "Time of matching AKAZE keypoints:   " << (f3 - f2) << std::endl;
    if (dump.is_open()) dumpPoints(dump, akaze_data1, akaze_data2);
    CHECK(hipFree(fimg1));
    CHECK(hipFree(fimg2));

    if (apiChecks) {
        // (a) an AkazeData smaller than the default capacity clamps THIS call (akaze.cpp:246 setMaxNumPoints(result.max_pts)) ...
        std::cout << "===== API checks =====" << std::endl;
        akaze::AkazeData small, large;
        akaze::initAkazeData(small, 500, true, true);
        akaze::initAkazeData(large, 20000, true, true);
        detector->detectAndCompute(img1, small, whp1, true);
        std::cout << "small AkazeData (500): " << small.num_pts << std::endl;
        // ... and does not shrink later calls; a larger one is not clamped to the default 10000 either
        detector->detectAndCompute(img1, large, whp1, true);
        std::cout << "large AkazeData (20000): " << large.num_pts << std::endl;
        detector->detectAndCompute(img1, akaze_data1, whp1, true);
        std::cout << "default AkazeData again: " << akaze_data1.num_pts << std::endl;
        int same = small.num_pts <= large.num_pts;
        for (int i = 0; i < small.num_pts && same; i++)                       // the clamp keeps the raster-order prefix
            same = memcmp(&small.h_data[i], &large.h_data[i], 85) == 0;
        std::cout << "small is a prefix of large: " << (same ? "yes" : "NO") << std::endl;
        // (b) a size other than init()'s: new arena (akaze.cpp:109-117), then back
        int3 whpc; whpc.x = w / 2 / 4 * 4; whpc.y = h / 2; whpc.z = iAlignUp(whpc.x, 128);
        float* crop = NULL;
        CHECK(hipMalloc((void**)&crop, sizeof(float) * (size_t)whpc.y * whpc.z));
        CHECK(hipMemcpy2D(crop, sizeof(float) * whpc.z, limg.data(), sizeof(float) * w, sizeof(float) * whpc.x, whpc.y, hipMemcpyHostToDevice));
        detector->detectAndCompute(crop, large, whpc, true);
        std::cout << "top-left quarter (" << whpc.x << "x" << whpc.y << "): " << large.num_pts << std::endl;
        if (dump.is_open()) dumpPoints(dump, small, large);
        detector->detectAndCompute(img1, large, whp1, true);
        std::cout << "full size again: " << large.num_pts << std::endl;
        CHECK(hipFree(crop));
        akaze::freeAkazeData(small);
        akaze::freeAkazeData(large);
    }

    akaze::freeAkazeData(akaze_data1);
    akaze::freeAkazeData(akaze_data2);
    CHECK(hipFree(img1));
    CHECK(hipFree(img2));
    return 0;
}
