// hak_api.hip -- C ABI of libhipakaze: context, FED schedule, launch sequence.
//
// Host orchestration restates Akazer::detectAndCompute / detect
// (akaze.cpp:101-150, 240-503) with every per-image scalar kept on the device
// (kcontrast, point counts), one launch sequence per BATCH of images
// (blockIdx.z = image) and no host synchronisation inside the sequence.
#include "hak_internal.h"
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "hak_ctx.h"

// ------------------------------------------------------------------ errors
static thread_local std::string g_err;
int hak_fail(const std::string& m) { g_err = m; return 1; }
static thread_local const char* g_launch_err = nullptr;
void hak_note_launch_error(const char* msg) { if (!g_launch_err) g_launch_err = msg; }

extern "C" const char* hak_last_error(void) { return g_err.c_str(); }

extern "C" int hak_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

extern "C" int hak_set_device(int dev)
{
    int n = hak_device_count();
    if (n == 0) return fail("no HIP device: libhipakaze has no CPU fallback");
    dev = dev < 0 ? 0 : (dev >= n ? n - 1 : dev);                   // cuda_utils.h:50
    HIP_TRY(hipSetDevice(dev));
    return 0;
}

extern "C" void hak_default_config(hak_config* c)
{
    c->noctaves = 4; c->max_scale = 4; c->per = 0.7f; c->kcontrast = 0.03f; c->soffset = 1.6f;
    c->reordering = 1; c->derivative_factor = 1.5f; c->dthreshold = 0.001f; c->diffusivity = HAK_PM_G2;
    c->descriptor_pattern_size = 10; c->max_pts = 10000; c->upright = 0; c->batch = 1;
}

// --------------------------------------------------- host-side schedule math
static bool fed_is_prime(int number)                                // fed.cpp:128-148
{
    if (number <= 1) return false;
    if (number == 2 || number == 3 || number == 5 || number == 7) return true;
    if (number % 2 == 0 || number % 3 == 0 || number % 5 == 0 || number % 7 == 0) return false;
    int upper = (int)std::sqrt(number + 1.0);
    for (int d = 11; d <= upper; d += 2)
        if (number % d == 0) return false;
    return true;
}

extern "C" int hak_fed_tau(float T, int M, float tau_max, int reordering, float* tau, int cap)
{
    // fed.cpp:41-119; mixed float/double arithmetic as in the source
    const float t = T / (float)M;
    const int n = (int)(std::ceil(std::sqrt(3.0 * t / tau_max + 0.25f) - 0.5f - 1.0e-8f) + 0.5f);
    if (n <= 0) return 0;
    if (n > cap) return -n;
    const float scale = (float)(3.0 * t / (tau_max * (float)(n * (n + 1))));
    const float c = 1.0f / (4.0f * (float)n + 2.0f);
    const float d = scale * tau_max / 2.0f;
    std::vector<float> tauh(n);
    for (int k = 0; k < n; ++k) {
        float hh = (float)std::cos(HAK_PI_D * (2.0f * (float)k + 1.0f) * c);
        tauh[k] = d / (hh * hh);
    }
    if (!reordering) {
        for (int k = 0; k < n; k++) tau[k] = tauh[k];
        return n;
    }
    const int kappa = n / 2;
    int prime = n + 1;
    while (!fed_is_prime(prime)) prime++;
    for (int k = 0, l = 0; l < n; ++k, ++l) {
        int index;
        while ((index = ((k + 1) * kappa) % prime - 1) >= n) k++;
        tau[l] = tauh[index];
    }
    return n;
}

extern "C" void hak_gauss_taps(float var, int radius, float* taps)
{
    // akazed.cu:2298-2333
    const float denom = 1.f / (2.f * var);
    float ksum = 0;
    for (int i = 0; i <= radius; i++) {
        taps[i] = expf(-i * i * denom);
        ksum += (i == 0) ? taps[i] : taps[i] + taps[i];
    }
    ksum = 1 / ksum;
    for (int i = 0; i <= radius; i++) taps[i] *= ksum;
}

extern "C" int hak_describe_plan_query(int pattern_size, unsigned int* pos, unsigned int* cell)
{
    static HakTables t;                                     // (too large for the stack of a small thread; host-only scratch)
    static std::mutex mu;
    std::lock_guard<std::mutex> lock(mu);
    if (pattern_size < 1) return 0;
    memset(&t, 0, sizeof(t));
    hak_describe_plan(&t, pattern_size);
    if (pos) memcpy(pos, t.dsc_pos, sizeof(t.dsc_pos));
    if (cell) memcpy(cell, t.dsc_cell, sizeof(t.dsc_cell));
    return t.dsc_plan_ok;
}

extern "C" void hak_compare_indices(int* idx1, int* idx2)
{
    // akazed.cu:65-159: per grid (2x2 cells 0-3, 3x3 cells 4-12, 4x4 cells 13-28), channel-major, pairs j<i
    static const int lo[3] = {0, 4, 13}, hi[3] = {4, 13, 29};
    int n = 0;
    for (int g = 0; g < 3; g++)
        for (int ch = 0; ch < 3; ch++)
            for (int j = lo[g]; j < hi[g] - 1; ++j)
                for (int i = j + 1; i < hi[g]; ++i) {
                    idx1[n] = 3 * j + ch;
                    idx2[n] = 3 * i + ch;
                    n++;
                }
    for (; n < 488; n++) idx1[n] = idx2[n] = 0;
}

static inline int align_up(int a, int b) { return (a + b - 1) / b * b; }

HakKnobs hak_knobs_from_env()
{
    HakKnobs k;
    if (const char* e = getenv("HAK_HESS_STREAM")) k.hess_stream = atoi(e);
    if (const char* e = getenv("HAK_BASE_STREAM")) k.base_stream = atoi(e);
    if (const char* e = getenv("HAK_BASE_HIST")) k.base_hist = atoi(e);
    if (const char* e = getenv("HAK_HESS_CBUF")) { const int v = atoi(e); k.hess_cbuf = v < 1 ? 1 : (v > 256 ? 256 : v); }
    if (const char* e = getenv("HAK_DESC_ORDER")) { const int v = atoi(e); k.desc_order = v < 0 ? 0 : (v > 255 ? 255 : v); }
    if (const char* e = getenv("HAK_DESC_PLAN")) k.desc_plan = atoi(e);
    if (const char* e = getenv("HAK_LEVEL_TILE")) k.level_tile = atoi(e);
    if (const char* e = getenv("HAK_HESS_LP")) k.hess_lp = atoi(e);
    if (const char* e = getenv("HAK_DESC_SORT")) k.desc_sort = atoi(e);
    return k;
}

static int build_plan(hak_ctx* c, int w, int h)
{
    const hak_config& cfg = c->cfg;
    if (cfg.noctaves < 1 || cfg.noctaves > HAK_MAX_OCTAVES) return fail("noctaves out of range");
    if (cfg.max_scale < 1 || cfg.max_scale > HAK_MAX_SCALES) return fail("max_scale out of range");
    if (w < 80 || h < 80) return fail("image smaller than 80 px");
    // candidate entries pack full-resolution coordinates as (y << 16) | x (kernels_detect.hip, kernels_hessian*.hip)
    if (w > 65535 || h > 65535) return fail("image larger than 65535 px in one dimension");
    HakLayout& L = c->L;
    memset(&L, 0, sizeof(L));
    L.ms = cfg.max_scale;
    // akaze.cpp:204-237 allocMemory; octave count fixed up front
    int noct = 1;
    L.oct[0] = {w, h, align_up(w, 64), 0};
    for (int j = 1; j < cfg.noctaves; j++) {
        int ww = L.oct[j - 1].w >> 1, hh = L.oct[j - 1].h >> 1;
        if (ww < 80 || hh < 80) break;
        L.oct[j] = {ww, hh, align_up(ww, 64), 0};
        noct = j + 1;
    }
    L.noct = noct;
    long off = 0;
    for (int o = 0; o < noct; o++) {
        L.oct[o].plane = (long)L.oct[o].h * L.oct[o].p;
        L.lvl_off[o] = off;       off += 3L * L.ms * L.oct[o].plane;      // Lt[ms] + interleaved {Lx, Ly}[ms] (2 planes each)
        L.smooth_off[o] = off;    off += L.oct[o].plane;
        L.flow_off[o] = off;      off += L.oct[o].plane;
        L.tmp_off[o] = off;       off += L.oct[o].plane;
    }
    L.arena = off;

    // akaze.cpp:268-439 schedule
    c->plan.assign((size_t)noct * L.ms, LevelPlan());
    const float tmax = 0.25f;
    float esigma = cfg.soffset;
    float last_etime = (float)(0.5 * cfg.soffset * cfg.soffset);                 // akaze.cpp:270
    const float smax = (float)(10.0 * sqrtf(2.0f));                               // akaze.cpp:279 (MLDB)
    int oratio = 1;
    float psz = 10000;
    float tau[4096];
    for (int i = 0; i < noct; i++) {
        for (int j = 0; j < L.ms; j++) {
            LevelPlan& lp = c->plan[(size_t)i * L.ms + j];
            if (i == 0 && j == 0) {
                lp.size = esigma * cfg.derivative_factor;                         // akaze.cpp:336-338
                lp.sigma_size = (int)(esigma * cfg.derivative_factor + 0.5f);
                lp.border = smax * lp.sigma_size;
                continue;
            }
            esigma = cfg.soffset * powf(2, (float)j / L.ms + i);                  // akaze.cpp:357
            float curr_etime = 0.5f * esigma * esigma;
            float ttime = curr_etime - last_etime;
            int n = hak_fed_tau(ttime, 1, tmax, cfg.reordering, tau, 4096);
            if (n < 0) return fail("FED cycle longer than 4096 steps");
            if (n == 0) return fail("FED cycle with zero steps (non-increasing scale schedule)");
            lp.nsteps = n;
            lp.tau.assign(tau, tau + n);
            lp.size = esigma * cfg.derivative_factor / oratio;
            lp.sigma_size = (int)(lp.size + 0.5f);
            lp.border = smax * lp.sigma_size;
            last_etime = curr_etime;
        }
        float b0 = c->plan[(size_t)i * L.ms].border * oratio;
        psz = psz < b0 ? psz : b0;                                                // akaze.cpp:434
        oratio *= 2;
    }
    c->psz = (int)psz;
    memset(&c->htab, 0, sizeof(c->htab));
    for (int l = 0; l < noct * L.ms; l++) {
        c->htab.sizes[l] = c->plan[l].size;
        c->htab.borders[l] = c->plan[l].border;
        c->htab.sigma_size[l] = c->plan[l].sigma_size;
    }
    for (int r2 = 0; r2 < 36; r2++) c->htab.orient_w[r2] = hak_expf(-r2 * 0.08f);  // akazed.cu:1697
    hak_deriv_factors(&c->htab.fac1, &c->htab.fac2);
    c->htab.ifac1 = (int)(c->htab.fac1 * 65536 + 0.5f);                           // akazed.cu:4183-4184
    c->htab.ifac2 = (int)(c->htab.fac2 * 65536 + 0.5f);
    hak_compare_indices(c->htab.comp1, c->htab.comp2);
    for (int b = 0; b < 61; b++)
        for (int i = 0; i < 8; i++) {
            c->htab.comp_packed[b * 16 + 2 * i] = (unsigned char)c->htab.comp1[b * 8 + i];
            c->htab.comp_packed[b * 16 + 2 * i + 1] = (unsigned char)c->htab.comp2[b * 8 + i];
        }
    hak_describe_plan(&c->htab, cfg.descriptor_pattern_size);
    hak_gauss_taps(1.f, 2, c->taps1);
    int ksz = (int)(2 * ceilf((cfg.soffset - 0.8f) / 0.3f) + 3);                 // akaze.cpp:328
    c->base_R = ksz <= 5 ? 2 : ksz <= 7 ? 3 : ksz <= 9 ? 4 : 5;                   // akazed.cu:2345-2377
    if (ksz > 11) return fail("Kernels larger than 11 not implemented");
    hak_gauss_taps(cfg.soffset * cfg.soffset, c->base_R, c->taps_base);
    for (int i = 0; i < 8; i++) {
        c->itaps1[i] = i <= 2 ? (int)(c->taps1[i] * 65536 + 0.5f) : 0;
        c->itaps_base[i] = i <= c->base_R ? (int)(c->taps_base[i] * 65536 + 0.5f) : 0;
    }
    return 0;
}

extern "C" int hak_create(const hak_config* cfg, int w, int h, hak_ctx** out)
{
    if (!cfg || !out) return fail("null argument");
    if (hak_device_count() == 0) return fail("no HIP device: libhipakaze has no CPU fallback");
    hak_ctx* c = new hak_ctx();
    c->cfg = *cfg;
    if (c->cfg.batch < 1) c->cfg.batch = 1;
    if (c->cfg.max_pts < 1) c->cfg.max_pts = 1;
    if (const char* e = getenv("HAK_FUSE_SF")) c->fuse_sf = atoi(e);
    if (const char* e = getenv("HAK_FUSE_HEAD")) c->fuse_head = atoi(e);
    if (const char* e = getenv("HAK_LEVEL_MIN_STEPS")) c->level_min_steps = atoi(e);
    c->knobs = hak_knobs_from_env();
    if (const char* e = getenv("HAK_FED_MAX_FUSE")) {
        int v = atoi(e);
        c->max_fuse = v < 1 ? 1 : (v > HAK_FED_MAX_FUSE ? HAK_FED_MAX_FUSE : v);
    }
    if (build_plan(c, w, h)) { delete c; return 1; }
    const int B = c->cfg.batch;
    const HakLayout& L = c->L;
    const int words = (L.oct[0].w + 63) / 64;
    hipError_t e = hipSuccess;
    auto A = [&](void** p, size_t bytes) { if (e == hipSuccess) e = hipMalloc(p, bytes); };
    A((void**)&c->arena, sizeof(float) * (size_t)L.arena * B);
    A((void**)&c->maps, sizeof(unsigned long long) * (size_t)L.oct[0].plane * B);
    A((void**)&c->bitmap, sizeof(unsigned long long) * (size_t)L.oct[0].h * words * B);
    A((void**)&c->rowcount, sizeof(int) * (size_t)L.oct[0].h * B);
    // a 3x3 strict maximum occurs at most once per 2x2 block: the list can never overflow
    c->cand_cap = 0;
    for (int o = 0; o < L.noct; o++) c->cand_cap += (long)L.ms * ((L.oct[o].w + 1) / 2) * ((L.oct[o].h + 1) / 2);
    A((void**)&c->cand, sizeof(unsigned long long) * (size_t)c->cand_cap * B);
    A((void**)&c->perm, sizeof(int) * (size_t)c->cfg.max_pts * B);
    A((void**)&c->state, sizeof(HakImgState) * (size_t)B);
    A((void**)&c->d_num, sizeof(int) * (size_t)B);
    A((void**)&c->dtab, sizeof(HakTables));
    if (e == hipSuccess) e = hipHostMalloc((void**)&c->h_num, sizeof(int) * (size_t)B);
    if (e == hipSuccess) e = hipMemcpy(c->dtab, &c->htab, sizeof(HakTables), hipMemcpyHostToDevice);
    // the key map must be all zero at the start of every call; calls restore that themselves (k_clear_cand_maps)
    if (e == hipSuccess) e = hipMemset(c->maps, 0, sizeof(unsigned long long) * (size_t)L.oct[0].plane * B);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_last, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_tail_fork, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_phase, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_null, hipEventDisableTiming);
    if (const char* s = getenv("HAK_NULL_ORDER")) c->null_order = atoi(s) != 0;
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_tail_join, hipEventDisableTiming);
    // ... and so must the survivor bitmap and the row counts (hak_launch_clear_maps restores all three)
    if (e == hipSuccess) e = hipMemset(c->bitmap, 0, sizeof(unsigned long long) * (size_t)L.oct[0].h * words * B);
    if (e == hipSuccess) e = hipMemset(c->rowcount, 0, sizeof(int) * (size_t)L.oct[0].h * B);
    // (hipMemset fills on the NULL stream and may return before the fill has run; the context's streams are non-blocking and would
    // not wait for it)
    if (e == hipSuccess) e = hipStreamSynchronize(nullptr);
    if (const char* s = getenv("HAK_HESS_SIDE")) c->hess_side = atoi(s);
    // (only when asked for, and behind the octave streams: the runtime deals a process's streams to its four hardware queues in
    // creation order, so one more stream per context moves every later stream to another queue -- creating it unconditionally put
    // the two pipeline contexts of the bench on ONE queue: 45.0 instead of 41-42 ms per step, single-image calls 1.12 instead of 0.94 ms)
    for (int o = 0; o < L.noct && e == hipSuccess; o++) {
        if (o > 0) e = hipStreamCreateWithFlags(&c->oct_stream[o], hipStreamNonBlocking);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_ready[o], hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_done[o], hipEventDisableTiming);
    }
    if (c->hess_side && e == hipSuccess) {
        e = hipStreamCreateWithFlags(&c->hess_stream, hipStreamNonBlocking);
        for (int s = 0; s < HAK_MAX_SCALES && e == hipSuccess; s++) {
            e = hipEventCreateWithFlags(&c->ev_hs[s], hipEventDisableTiming);
            if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_hd[s], hipEventDisableTiming);
        }
    }
    if (const char* s = getenv("HAK_SERIAL")) c->concurrent = atoi(s) == 0;
    if (const char* s = getenv("HAK_GRAPH")) { c->graph_mode = atoi(s); c->use_graph = c->graph_mode != 0; }
    if (e != hipSuccess) {
        fail(std::string("hak_create: ") + hipGetErrorString(e));
        hak_destroy(c);
        return 1;
    }
    c->stream = c->own_stream;
    *out = c;
    return 0;
}

extern "C" void hak_destroy(hak_ctx* c)
{
    if (!c) return;
    // work may still be queued on a caller-provided stream (hak_set_stream), which may itself be gone by now: wait for the
    // event recorded after the context's last enqueue instead of touching that stream
    if (c->ev_last) { (void)hipEventSynchronize(c->ev_last); (void)hipEventDestroy(c->ev_last); }
    if (c->ev_tail_fork) (void)hipEventDestroy(c->ev_tail_fork);
    if (c->ev_phase) (void)hipEventDestroy(c->ev_phase);
    if (c->ev_null) (void)hipEventDestroy(c->ev_null);
    if (c->ev_tail_join) (void)hipEventDestroy(c->ev_tail_join);
    if (c->own_stream) (void)hipStreamSynchronize(c->own_stream);
    for (auto& g : c->graph_exec) if (g) (void)hipGraphExecDestroy(g);
    for (int o = 0; o < HAK_MAX_OCTAVES; o++) {
        if (c->oct_stream[o]) { (void)hipStreamSynchronize(c->oct_stream[o]); (void)hipStreamDestroy(c->oct_stream[o]); }
        if (c->ev_ready[o]) (void)hipEventDestroy(c->ev_ready[o]);
        if (c->ev_done[o]) (void)hipEventDestroy(c->ev_done[o]);
    }
    if (c->hess_stream) { (void)hipStreamSynchronize(c->hess_stream); (void)hipStreamDestroy(c->hess_stream); }
    for (int s = 0; s < HAK_MAX_SCALES; s++) {
        if (c->ev_hs[s]) (void)hipEventDestroy(c->ev_hs[s]);
        if (c->ev_hd[s]) (void)hipEventDestroy(c->ev_hd[s]);
    }
    for (auto& p : c->prof)
        for (auto ev : p.ev) (void)hipEventDestroy(ev);
    hak_match_scratch_free(&c->msc);
    void* bufs[] = {c->arena, c->maps, c->bitmap, c->rowcount, c->cand, c->state, c->d_num, c->dtab, c->knn, c->d_cnt, c->perm, c->pair_pts};
    for (void* b : bufs) (void)hipFree(b);
    if (c->h_num) (void)hipHostFree(c->h_num);
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    delete c;
}

extern "C" int hak_set_stream(hak_ctx* c, void* s)
{
    if (!c) return fail("null context");
    c->stream = s ? (hipStream_t)s : c->own_stream;
    return 0;
}

extern "C" int hak_set_concurrency(hak_ctx* c, int on)
{
    if (!c) return fail("null context");
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->concurrent = on != 0;
    return 0;
}

extern "C" int hak_set_null_order(hak_ctx* c, int on)
{
    if (!c) return fail("null context");
    c->null_order = on != 0;
    return 0;
}

// The reference issues everything on stream 0 (akaze.cpp:101-150, 55-64): whatever its caller enqueued on the default stream before a call
// -- a hipMemset of an output array, an asynchronous upload, a kernel of its own -- is finished when the call's first kernel starts.  A
// context's streams are non-blocking; this makes the call's stream wait for the NULL stream's work enqueued so far (device-side: an event
// record there, a wait here; nothing when the context runs on the NULL stream itself).  Never inside a stream capture: every caller sits
// in front of run_detect_inner's capture.
static void order_after_null_stream(hak_ctx* c, hipStream_t st)
{
    if (!c || !c->null_order || !c->ev_null || st == nullptr) return;
    // (an idle NULL stream -- the reference's own call pattern with its blocking copies -- costs one query; the cross-queue dependency
    // itself was measured at ~15 us per call: 0.54 -> 0.55 ms for the pair call when taken unconditionally)
    const hipError_t q = hipStreamQuery(nullptr);
    if (q == hipSuccess) return;
    (void)hipGetLastError();
    if (hipEventRecord(c->ev_null, nullptr) != hipSuccess || hipStreamWaitEvent(st, c->ev_null, 0) != hipSuccess) (void)hipGetLastError();
}

extern "C" int hak_phase_event(hak_ctx* c, void** ev)
{
    if (!c || !ev) return fail("null argument");
    *ev = (void*)c->ev_phase;
    return 0;
}

extern "C" int hak_wait_event(hak_ctx* c, void* ev)
{
    if (!c || !ev) return fail("null argument");
    HIP_TRY(hipStreamWaitEvent(c->stream, (hipEvent_t)ev, 0));
    return 0;
}

extern "C" int hak_sync(hak_ctx* c)
{
    if (!c) return fail("null context");
    HIP_TRY(hipStreamSynchronize(c->stream));
    return 0;
}

// ------------------------------------------------------- the launch sequence
// The key map must be all zero when a launch sequence starts; every sequence restores that itself (k_clear_cand_maps).  If a
// call fails between writing the map and cleaning it up, the flag stays set and the next call clears the map in full -- eagerly
// on the context's stream and never inside a stream capture, so a replayed graph cannot miss (or needlessly carry) the clear.
void maps_guard_begin(hak_ctx* c)
{
    if (c->maps_dirty) {
        const size_t h = c->L.oct[0].h, words = (c->L.oct[0].w + 63) / 64, B = c->cfg.batch;
        (void)hipMemsetAsync(c->maps, 0, sizeof(unsigned long long) * (size_t)c->L.oct[0].plane * B, c->stream);
        (void)hipMemsetAsync(c->bitmap, 0, sizeof(unsigned long long) * h * words * B, c->stream);
        (void)hipMemsetAsync(c->rowcount, 0, sizeof(int) * h * B, c->stream);
    }
    c->maps_dirty = true;
}
int maps_guard_end(hak_ctx* c, int rc)
{
    if (!rc) c->maps_dirty = false;
    // (on the stream the sequence ended on: a marker on the caller's idle stream would make the next call's idle test fail)
    if (c->ev_last) (void)hipEventRecord(c->ev_last, c->sync_stream ? c->sync_stream : c->stream);
    return rc;
}

// kernels_level.hip's one-launch sublevel: by the size rule unless a test forces the streaming kernels (fuse_sf == 2)
static bool level_tile_pays(const hak_ctx* c, const HakOct& oc, int nimg)
{
    const int mode = c->knobs.level_tile;
    if (mode != 1) return mode != 0;
    // (round 4, measured: taking the tile kernel for the latency-bound small octaves of a LARGE batch as well -- octave 3, or octaves
    // 2-3, of 512 images -- cuts the FED launches from 46 to 26 / 18 and costs 8-17 ms per sequence: its halo work, (T + 2n)^2 / T^2
    // of the useful work, is only worth paying where launches, not bytes or arithmetic, are the cost)
    return c->fuse_sf != 2 && (long)oc.w * oc.h * nimg <= HAK_LEVEL_TILE_MAX_PX;
}

// Launch-bound sequences -- a single image: octave 0 small enough for k_level_tile -- are issued in SPINE order (enqueue_detect) and
// eagerly instead of as a replayed graph: their time is the longest dependency chain, not bytes.  Round 4 tried the same for a PAIR
// (two to four images; HAK_SPINE_MAX_PX widens the rule): with four chains of two-image kernels in flight the kernels slow each other
// down and the call got slower, 0.60 -> 0.635 ms (8 hardware queues) / 0.675 (4) -- a pair is bound by the GPU time of its small
// kernels, the replayed per-octave order is the better one for it (profiles/r04_pair_timeline.txt).
static bool spine_pays(const hak_ctx* c, int nimg)
{
    static const long max_px = [] { const char* e = getenv("HAK_SPINE_MAX_PX"); return e ? atol(e) : 0L; }();
    if (max_px > 0 && c->knobs.level_tile == 1) return c->fuse_sf != 2 && (long)c->L.oct[0].w * c->L.oct[0].h * nimg <= max_px;
    return level_tile_pays(c, c->L.oct[0], nimg);
}

static int enqueue_detect(hak_ctx* c, const float* d_images, long image_stride, int pitch, int nimg,
                          hak_point* d_points, int* d_num_pts, int desc, int max_pts, hak_point* h_points = nullptr, int cap0 = 0, int cap1 = 0)
{
    const hak_config& cfg = c->cfg;
    const HakLayout& L = c->L;
    // Octave o+1 depends only on Lt(o, 0) (the reference decimates from sublevel 0, akaze.cpp:371-375).
    const bool spine = c->concurrent && L.noct > 1 && spine_pays(c, nimg);
    const hipStream_t main_st = c->stream;
    c->sync_stream = c->stream;
    float* A = c->arena;
    const long S = L.arena;
    HakBatch b{A, S, nimg, c->state, c->maps, L.oct[0].plane, c->bitmap, c->rowcount, c->cand, c->cand_cap, &c->knobs, c->perm, cfg.max_pts};
    b.cap0 = cap0; b.cap1 = cap1;
    c->last_fast = false;
    c->fed_launches = 0;
    c->fed_fused_bytes = 0;

    bool hess_fused[HAK_MAX_OCTAVES * HAK_MAX_SCALES] = {};
    bool hess_lp[HAK_MAX_OCTAVES * HAK_MAX_SCALES] = {};     // the level's Hessian low-passes Lt(o,s-1) itself: `smooth` was not written
    static const bool level_hess_on = [] { const char* e = getenv("HAK_LEVEL_HESS"); return !e || atoi(e) != 0; }();
    // ---- part A of level (o, s): build Lt(o, s) and the sigma=1 low-pass `smooth` the level's Hessian reads (akaze.cpp:325-421)
    // smooth_alt != nullptr: the level's sigma=1 low-pass goes there instead of the octave's `smooth` plane (side-stream Hessians below)
    auto build_level = [&](int o, int s, hipStream_t st, float* smooth_alt = nullptr) {
        const HakOct oc = L.oct[o];
        float* smooth = smooth_alt ? smooth_alt : A + L.smooth_off[o];
        float* flow = A + L.flow_off[o];
        float* tmp = A + L.tmp_off[o];
        const LevelPlan& lp = c->plan[(size_t)o * L.ms + s];
        float* Lt = A + L.lt(o, s);
        if (o == 0 && s == 0) {                                                   // akaze.cpp:325-332 in two passes over img
            ProfScope ps(c, HAK_PROF_CONTRAST, st);
            hak_launch_base_level(st, d_images, image_stride, pitch, Lt, tmp /* free until the FED cycle of (0,1) */, S, oc.w, oc.h, oc.p, nimg, c->taps1,
                                  c->taps_base, c->base_R, c->state, cfg.per, L.noct, c->knobs);
            return;
        }
        const int n = lp.nsteps;
        // small launches (single images, small octaves of small batches): the whole sublevel in one launch out of LDS tiles --
        // octave heads (one launch instead of decimation + conductivity + FED groups) and every cycle long enough that the tile
        // kernel's halo work costs less than the launches it saves (by the size rule: n >= 8, i.e. octaves 2 and up of the demo
        // schedule; shorter cycles keep k_smooth_flow + k_fed_multi, which spend less GPU time per pixel)
        if (level_tile_pays(c, oc, nimg) && (s == 0 || n >= c->level_min_steps || c->knobs.level_tile == 2)) {
            ProfScope ps(c, HAK_PROF_FED, st);
            // (the level's Hessian rides along when the cycle is long enough: hess_fused tells hessian_level below)
            const int nl = hak_launch_level_tile(st, s == 0 ? A + L.lt(o - 1, 0) : A + L.lt(o, s - 1), s == 0 ? L.oct[o - 1] : oc, s == 0, smooth, Lt, tmp, S,
                                                 oc, nimg, c->taps1, cfg.diffusivity, lp.tau.data(), n, c->state, o, 0.f,
                                                 level_hess_on ? A + L.dxy(o, s) : nullptr, lp.sigma_size, &b, &L, &c->htab, s, cfg.dthreshold,
                                                 &hess_fused[o * HAK_MAX_SCALES + s]);
            c->fed_launches += nl;
            c->fed_fused_bytes += (s == 0 ? 1.0 * L.oct[o - 1].w * L.oct[o - 1].h : 4.0 * oc.w * oc.h) + 8.0 * oc.w * oc.h + (nl - 1) * 12.0 * oc.w * oc.h;
            return;
        }
        const int G = hak_fed_groups(n, c->max_fuse, oc.w);     // launches of this FED cycle
        const float* fsrc;          // input of the first FED launch
        bool fused_first = false;
        if (s == 0) {                                                             // akaze.cpp:369-392
            // octave head: decimation + low-pass + conductivity + the first FED group in one streaming pass when covered
            if (c->fuse_head && hak_stream_pays(c->fuse_sf, oc.w, oc.h, nimg)) {
                ProfScope ps(c, HAK_PROF_FED, st);
                fused_first = hak_launch_fed_sf_head(st, A + L.lt(o - 1, 0), L.oct[o - 1], smooth, flow, (G % 2 == 1) ? Lt : tmp, S, oc,
                                                     nimg, c->taps1, cfg.diffusivity, lp.tau.data(), hak_fed_group_size(n, G, 0),
                                                     c->state, o, G > 1);
                if (fused_first) {
                    c->fed_launches++;           // reads the even rows of Lt(o-1,0), writes smooth, L' (+ g for later launches)
                    c->fed_fused_bytes += 2.0 * L.oct[o - 1].w * L.oct[o - 1].h + (G > 1 ? 12.0 : 8.0) * oc.w * oc.h;
                }
            }
            // otherwise decimate Lt(o-1,0) so that the last of G ping-pong launches lands in Lt(o,0)
            float* first = (G % 2 == 0) ? Lt : tmp;
            if (!fused_first) {
                ProfScope ps(c, HAK_PROF_DOWN, st);
                hak_launch_down_smooth(st, A + L.lt(o - 1, 0), first, smooth, S, L.oct[o - 1], oc, nimg, c->taps1);
            }
            fsrc = first;
        } else {                                                                  // akaze.cpp:393-421
            fsrc = A + L.lt(o, s - 1);
        }
        // sublevels > 0: low-pass + conductivity + the first FED group in one streaming pass when the case is covered
        // (PM_G2, 16-byte rows); the conductivity plane is written only if later groups of the cycle need it
        if (s == 0) {
            if (!fused_first) {
                ProfScope ps(c, HAK_PROF_FLOW, st);
                hak_launch_flow(st, smooth, flow, S, oc.w, oc.h, oc.p, nimg, cfg.diffusivity, c->state, o, 0.f);
            }
        } else if (hak_stream_pays(c->fuse_sf, oc.w, oc.h, nimg) && cfg.diffusivity == HAK_PM_G2 && (oc.w & 3) == 0 && oc.w >= 16 &&
                   oc.h >= 8) {
            const int ns0 = hak_fed_group_size(n, G, 0);
            float* dst0 = (G % 2 == 1) ? Lt : tmp;
            // the low-pass has one reader, the level's Hessian: when that runs as the streaming kernel it low-passes Lt(o,s-1)
            // itself (LP variant) and the plane is not written at all
            const bool lp_hess = c->knobs.hess_lp != 0 && hak_stream_pays(c->knobs.hess_stream, oc.w, oc.h, nimg) &&
                                 hak_hessian_stream_covers(oc.w, oc.h, lp.sigma_size, true);
            ProfScope ps(c, HAK_PROF_FED, st);
            fused_first = hak_launch_fed_sf(st, fsrc, smooth, flow, dst0, S, oc.w, oc.h, oc.p, nimg, c->taps1, cfg.diffusivity,
                                            lp.tau.data(), ns0, c->state, o, 0.f, G > 1, !lp_hess);
            if (fused_first) {
                hess_lp[o * HAK_MAX_SCALES + s] = lp_hess;
                c->fed_launches++;               // reads L, writes L' (+ smooth unless the Hessian is LP, + g for later launches)
                c->fed_fused_bytes += ((G > 1 ? 16.0 : 12.0) - (lp_hess ? 4.0 : 0.0)) * oc.w * oc.h;
            }
        }
        if (s != 0 && !fused_first) {                                             // akaze.cpp:403-404 in one pass
            ProfScope ps(c, HAK_PROF_LOWPASS, st);
            hak_launch_smooth_flow(st, fsrc, smooth, flow, S, oc.w, oc.h, oc.p, nimg, c->taps1, cfg.diffusivity,
                                   c->state, o, 0.f);
        }
        // the n explicit steps of the cycle in G fused launches, ping-pong Lt <-> tmp, ending in Lt
        const float* src = fsrc;
        int done = 0;
        for (int g = 0; g < G; g++) {
            const int ns = hak_fed_group_size(n, G, g);
            float* dst = ((G - g) % 2 == 1) ? Lt : tmp;
            if (!(g == 0 && fused_first)) {
                ProfScope ps(c, HAK_PROF_FED, st);
                hak_launch_fed_group(st, src, flow, dst, S, oc.w, oc.h, oc.p, nimg, lp.tau.data() + done, ns);
                c->fed_launches++;
                c->fed_fused_bytes += 12.0 * oc.w * oc.h;       // reads L and g, writes L'
            }
            done += ns;
            src = dst;
        }
    };
    // ---- part B of level (o, s): derivatives + determinant + extrema (akaze.cpp:354, 423, 431-433).  Level (0, 0) differentiates
    // Lt itself, every other level the low-pass of its predecessor (D13).  (The determinant goes to HBM only in the dilation > 4
    // fallback: `flow` is free at every call.)
    auto hessian_level = [&](int o, int s, hipStream_t st, const float* smooth_alt = nullptr) {
        if (hess_fused[o * HAK_MAX_SCALES + s]) return;              // done inside k_level_tile
        const HakOct oc = L.oct[o];
        const LevelPlan& lp = c->plan[(size_t)o * L.ms + s];
        const bool lph = hess_lp[o * HAK_MAX_SCALES + s];
        const float* hsrc = (o == 0 && s == 0) ? A + L.lt(0, 0) : lph ? A + L.lt(o, s - 1) : smooth_alt ? smooth_alt : A + L.smooth_off[o];
        ProfScope ps(c, HAK_PROF_HESSIAN, st);
        if (!hak_launch_hessian_level(st, hsrc, A + L.dxy(o, s), A + L.flow_off[o], false, S, oc.w, oc.h, oc.p, nimg,
                                      lp.sigma_size, &b, &L, &c->htab, o, s, cfg.dthreshold, lph ? c->taps1 : nullptr))
            hak_launch_extrema_level(st, b, L, c->dtab, o, s, cfg.dthreshold, L.flow_off[o]);
    };

    hak_launch_reset_state(main_st, c->state, nimg);   // (the key map is all zero here: hak_create / k_clear_cand_maps / maps_guard)

    if (spine) {
        // Launch-bound calls (a single image): the dependency chain base -> head(1) -> head(2) -> ... -> every sublevel of the
        // last octave is the critical path, so it runs on ONE stream without cross-queue waits (each costs 15-35 us in a replayed
        // graph, profiles/r03_single_*); what hangs off it -- the remaining sublevels and all Hessians of octaves 0 .. noct-2 --
        // goes to side streams, one per octave.
        const int last = L.noct - 1;
        hipGraphNode_t head_node[HAK_MAX_OCTAVES] = {};
        hipGraph_t cap_graph = nullptr;
        for (int o = 0; o <= last; o++) {
            build_level(o, 0, main_st);
            if (o < last) {
                (void)hipEventRecord(c->ev_ready[o], main_st);                    // Lt(o,0) + its low-pass ready: side stream o may start
                // while capturing: remember the head's graph node (see below)
                hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
                const hipGraphNode_t* deps = nullptr;
                size_t ndeps = 0;
                if (hipStreamGetCaptureInfo_v2(main_st, &cs, nullptr, &cap_graph, &deps, &ndeps) == hipSuccess &&
                    cs == hipStreamCaptureStatusActive && ndeps == 1) head_node[o] = deps[0];
                else (void)hipGetLastError();
            }
        }
        // The graph executor deals a fork's branches to its queues by position: the first outgoing edge of a node stays on the
        // node's queue, the k-th goes k-1 queues further (of four).  Every side stream forks from the spine as some head's SECOND
        // edge, so all three would share one queue and run one after the other (measured: 440 us of side work in a row).  Empty
        // nodes in front of a fork move its side branch further along.  The replay submits queue by queue -- the spine's first, then
        // the others from the last to the first -- so the longest side chain (octave 0's) gets the last queue, the shortest the
        // first.  Pure placement: results and ordering are unaffected, and a runtime that places nodes differently merely ignores
        // the hint (HAK_GRAPH_PADS=0 switches it off).
        static const bool pads_on = [] { const char* e = getenv("HAK_GRAPH_PADS"); return !e || atoi(e) != 0; }();
        for (int o = 0; o < last && pads_on; o++)
            for (int k = 0; k < last - 1 - o && head_node[o] && cap_graph; k++) {
                hipGraphNode_t pad = nullptr;
                if (hipGraphAddEmptyNode(&pad, cap_graph, &head_node[o], 1) != hipSuccess) (void)hipGetLastError();
            }
        // Side streams: one per remaining octave by default.  The chain + noct-1 side streams want noct hardware queues besides the
        // null stream's; the runtime gives a process four (GPU_MAX_HW_QUEUES), so at four octaves two side chains share a queue and
        // run one after the other (C++ demo: 1.08 instead of 0.91 ms per 1080p pair).  A process that makes single-image calls
        // should start with GPU_MAX_HW_QUEUES=8 (the demo does) -- the library does not set it itself, because a
        // process that runs BATCHES loses 2 % (1080p) to 13 % (720p) with eight queues.  HAK_SIDE_STREAMS = n < noct-1 makes
        // octaves n-1 .. noct-2 share the last side stream by design (same time as the shared queue).
        static const int nside_env = [] { const char* e = getenv("HAK_SIDE_STREAMS"); const int v = e ? atoi(e) : HAK_MAX_OCTAVES; return v < 1 ? 1 : v; }();
        const int nside = nside_env < last ? nside_env : (last > 0 ? last : 1);
        auto side_of = [&](int o) { return c->oct_stream[1 + (o < nside ? o : nside - 1)]; };
        // the remaining work, one level per octave in turn, each octave's first node behind the wait for its head
        for (int s = 0; s < L.ms; s++)
            for (int k = 0; k <= last; k++) {
                const int o = k == 0 ? last : k - 1;                              // the spine's own octave first
                const hipStream_t st = o == last ? main_st : side_of(o);
                if (s == 0 && o != last && hipStreamWaitEvent(st, c->ev_ready[o], 0) != hipSuccess) return fail("stream wait");
                if (s > 0) build_level(o, s, st);
                hessian_level(o, s, st);
            }
        for (int i = 0; i < nside && last > 0; i++) {
            (void)hipEventRecord(c->ev_done[i + 1], c->oct_stream[i + 1]);
            if (hipStreamWaitEvent(main_st, c->ev_done[i + 1], 0) != hipSuccess) return fail("stream join");
        }
    } else {
        // each octave on its own stream, chained by events: the small octaves' launches are latency chains of a few hundred waves
        // and hide under octave 0's chip-filling kernels
        // Small launches in the tile-kernel regime (a pair, a handful of images): octave 0 is the longest chain, and a third of it
        // are its four Hessians, which nothing in the scale space waits for.  They move to a stream of their own.  The only
        // hazard is the `smooth` plane (level s's Hessian reads it, level s+1's low-pass overwrites it): the levels alternate
        // between `smooth` and `tmp`, which is free in an octave whose FED cycles are single launches (G = 1: the cycle lands in
        // Lt directly), so level s+1's low-pass only waits for the Hessian of level s-1.  profiles/r05_pair_serial_timeline.txt:
        // 292 us of chain (alone) become 183 + the last Hessian.
        // MEASURED, OFF BY DEFAULT (HAK_HESS_SIDE=1): the pair call gets SLOWER, 0.567-0.572 -> 0.620-0.623 ms (4 or 5 hardware
        // queues alike; 6: 0.82): as with the spine order of round 4, a fifth chain of two-image kernels stretches the other four by
        // more than the critical chain shrinks -- the call is bound by the chip's throughput on these small kernels, not by the
        // order they are issued in.
        bool side0 = c->hess_side != 0 && c->concurrent && L.noct > 1 && c->hess_stream && L.ms <= HAK_MAX_SCALES &&
                     !hak_stream_pays(c->knobs.hess_stream, L.oct[0].w, L.oct[0].h, nimg) && !level_tile_pays(c, L.oct[0], nimg) &&
                     !hak_stream_pays(c->fuse_sf, L.oct[0].w, L.oct[0].h, nimg);
        for (int s = 1; s < L.ms && side0; s++)
            side0 = hak_fed_groups(c->plan[s].nsteps, c->max_fuse, L.oct[0].w) == 1 && c->plan[s].sigma_size <= 4;
        hipStream_t st = main_st;
        for (int o = 0; o < L.noct; o++) {
            if (c->concurrent && o > 0) {                       // this octave's chain waits only for Lt(o-1,0)
                st = c->oct_stream[o];
                if (hipStreamWaitEvent(st, c->ev_ready[o - 1], 0) != hipSuccess) return fail("stream wait");
            }
            for (int s = 0; s < L.ms; s++) {
                if (o == 0 && side0) {
                    float* sm = (s & 1) ? A + L.tmp_off[0] : A + L.smooth_off[0];
                    // (level s's low-pass target was last read by the Hessian of level s - 2)
                    if (s >= 2 && hipStreamWaitEvent(st, c->ev_hd[s - 2], 0) != hipSuccess) return fail("stream wait");
                    build_level(0, s, st, sm);
                    if (s == 0) (void)hipEventRecord(c->ev_ready[0], st);
                    (void)hipEventRecord(c->ev_hs[s], st);
                    if (hipStreamWaitEvent(c->hess_stream, c->ev_hs[s], 0) != hipSuccess) return fail("stream wait");
                    hessian_level(0, s, c->hess_stream, sm);
                    (void)hipEventRecord(c->ev_hd[s], c->hess_stream);
                    continue;
                }
                build_level(o, s, st);
                if (c->concurrent && s == 0) (void)hipEventRecord(c->ev_ready[o], st);   // Lt(o,0) final: octave o+1 may start
                hessian_level(o, s, st);
            }
            if (c->concurrent && o > 0) (void)hipEventRecord(c->ev_done[o], st);
        }
        if (c->concurrent)
            for (int o = 1; o < L.noct; o++)
                if (hipStreamWaitEvent(main_st, c->ev_done[o], 0) != hipSuccess) return fail("stream join");
        if (side0)                                              // (the stream runs in order: its last event covers all four)
            if (hipStreamWaitEvent(main_st, c->ev_hd[L.ms - 1], 0) != hipSuccess) return fail("stream join");
    }
    // the scale space (bound by HBM stores) is done, the keypoint stages (bound by gathers and integer work) begin: a caller that
    // runs two contexts lets the other one start its scale space here (hak_phase_event)
    {
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        hipGraph_t g = nullptr;
        const hipGraphNode_t* deps = nullptr;
        size_t ndeps = 0;
        if (hipStreamGetCaptureInfo_v2(main_st, &cs, nullptr, &g, &deps, &ndeps) != hipSuccess) { (void)hipGetLastError(); cs = hipStreamCaptureStatusNone; }
        hipError_t pe;
        if (cs == hipStreamCaptureStatusActive) {
            // inside a capture a plain record would be a capture-internal dependency: the record becomes an event-record NODE behind
            // the stream's current frontier, and the frontier moves to it
            hipGraphNode_t node = nullptr;
            pe = hipGraphAddEventRecordNode(&node, g, deps, ndeps, c->ev_phase);
            if (pe == hipSuccess) pe = hipStreamUpdateCaptureDependencies(main_st, &node, 1, hipStreamSetCaptureDependencies);
        } else pe = hipEventRecord(c->ev_phase, main_st);
        if (pe != hipSuccess) { fprintf(stderr, "hipakaze: phase event record: %s\n", hipGetErrorString(pe)); (void)hipGetLastError(); }
    }
    bool tail_fork = false;
    { ProfScope ps(c, HAK_PROF_NMS);                                              // akaze.cpp:449-455
      hak_launch_nms_emit(main_st, b, L, c->dtab, c->psz, d_points, max_pts, d_num_pts);
      // the clean-up for the next sequence needs only the candidate list: beside the descriptor kernels, not in front of them
      static const bool tail_fork_on = [] { const char* e = getenv("HAK_TAIL_FORK"); return !e || atoi(e) != 0; }();
      // (batches: no gain beside 5 ms of descriptor kernels, A/B 5 640 vs 5 710 pairs/s; the pair call: the fork's two cross-stream
      // waits in the replayed graph cost more than the 5 us kernel they move aside, 0.571 vs 0.544 ms, round 5)
      tail_fork = spine && tail_fork_on;
      if (tail_fork) {
          (void)hipEventRecord(c->ev_tail_fork, main_st);
          if (hipStreamWaitEvent(c->oct_stream[1], c->ev_tail_fork, 0) != hipSuccess) return fail("stream wait");
          hak_launch_clear_maps(c->oct_stream[1], b, L);
          (void)hipEventRecord(c->ev_tail_join, c->oct_stream[1]);
      } else hak_launch_clear_maps(main_st, b, L); }
    { ProfScope ps(c, HAK_PROF_DESCRIBE);                                         // akaze.cpp:124-131
      hak_launch_describe(main_st, b, L, c->dtab, d_points, max_pts, cfg.descriptor_pattern_size, cfg.upright, desc, c->htab.dsc_plan_ok); }
    if (h_points)                                                                 // pinned destination: records and count go out in the same sequence
        hak_launch_download(main_st, d_points, d_num_pts, max_pts, nimg, h_points, c->h_num);
    if (tail_fork && hipStreamWaitEvent(main_st, c->ev_tail_join, 0) != hipSuccess) return fail("stream join");
    if (hipGetLastError() != hipSuccess) return fail("kernel launch failed");
    if (g_launch_err) { const char* m = g_launch_err; g_launch_err = nullptr; return fail(m); }
    return 0;
}

// ------------------------------------------------------- integer FAST path
// Akazer::fastDetectAndCompute / fastDetect (akaze.cpp:153-201, 506-743): same orchestration on int32 planes.
static int enqueue_fast_detect(hak_ctx* c, const unsigned char* d_images, long image_stride, int pitch, int nimg,
                               hak_point* d_points, int* d_num_pts, int desc, int max_pts)
{
    const hak_config& cfg = c->cfg;
    const HakLayout& L = c->L;
    hipStream_t st = c->stream;
    int* A = reinterpret_cast<int*>(c->arena);
    const long S = L.arena;
    HakBatch b{c->arena, S, nimg, c->state, c->maps, L.oct[0].plane, c->bitmap, c->rowcount, c->cand, c->cand_cap, &c->knobs, c->perm, cfg.max_pts};
    const int idthreshold = 65;                                                   // akaze.cpp:559
    c->last_fast = true;
    c->sync_stream = c->stream;
    hakf_launch_reset(st, c->state, nimg);              // (the key map is all zero here: hak_create / k_clear_cand_maps / maps_guard)
    for (int o = 0; o < L.noct; o++) {
        const HakOct oc = L.oct[o];
        int* smooth = A + L.smooth_off[o];
        int* flow = A + L.flow_off[o];
        int* tmp = A + L.tmp_off[o];
        for (int s = 0; s < L.ms; s++) {
            const LevelPlan& lp = c->plan[(size_t)o * L.ms + s];
            int* Lt = A + L.lt(o, s);
            if (o == 0 && s == 0) {                                               // akaze.cpp:589-623
                // one fused pass + a histogram pass over the gradient plane it leaves in det(0,0) (free until the Hessian below)
                if (!hakf_launch_base_level(st, d_images, image_stride, pitch, Lt, tmp, S, oc.w, oc.h, oc.p, nimg, c->itaps1,
                                            c->itaps_base, c->base_R, c->state, cfg.per, L.noct, c->knobs)) {
                    hakf_launch_conv_u8(st, d_images, image_stride, pitch, smooth, S, oc.w, oc.h, oc.p, nimg, c->itaps1, 2);
                    hakf_launch_contrast(st, smooth, S, oc.w, oc.h, oc.p, nimg, c->state, cfg.per, L.noct);
                    hakf_launch_conv_u8(st, d_images, image_stride, pitch, Lt, S, oc.w, oc.h, oc.p, nimg, c->itaps_base, c->base_R);
                }
                if (!hakf_launch_hessian_level(st, Lt, A + L.dxy(o, s), flow, false, S, oc.w, oc.h, oc.p, nimg,
                                               lp.sigma_size, &b, &L, &c->htab, o, s, idthreshold)) {
                    hakf_launch_hessian(st, Lt, A + L.dxy(o, s), flow, S, oc.w, oc.h, oc.p, nimg, lp.sigma_size);
                    hakf_launch_extrema(st, b, L, c->dtab, o, s, idthreshold, L.flow_off[o]);
                }
                continue;
            }
            const int n = lp.nsteps;
            if (level_tile_pays(c, oc, nimg)) {
                bool hess_done = false;
                static const bool level_hess_on = [] { const char* e = getenv("HAK_LEVEL_HESS"); return !e || atoi(e) != 0; }();
                hakf_launch_level_tile(st, s == 0 ? A + L.lt(o - 1, 0) : A + L.lt(o, s - 1), s == 0 ? L.oct[o - 1] : oc, s == 0, smooth, Lt, tmp, S, oc,
                                       nimg, c->itaps1, cfg.diffusivity, lp.tau.data(), n, c->state, o,
                                       level_hess_on ? A + L.dxy(o, s) : nullptr, lp.sigma_size, &b, &L, &c->htab, s, idthreshold, &hess_done);
                if (!hess_done && !hakf_launch_hessian_level(st, smooth, A + L.dxy(o, s), flow, false, S, oc.w, oc.h, oc.p, nimg,
                                               lp.sigma_size, &b, &L, &c->htab, o, s, idthreshold)) {
                    hakf_launch_hessian(st, smooth, A + L.dxy(o, s), flow, S, oc.w, oc.h, oc.p, nimg, lp.sigma_size);
                    hakf_launch_extrema(st, b, L, c->dtab, o, s, idthreshold, L.flow_off[o]);
                }
                continue;
            }
            // FED cycle in G fused launches (the float path's streaming kernel instantiated for int32) when the width
            // allows 16-byte rows, else one step per launch; ping-pong Lt <-> tmp so that the last launch lands in Lt
            const bool fused = (oc.w % 4) == 0;
            const int G = fused ? hak_fed_groups(n, c->max_fuse, oc.w) : n;
            const int* src;
            bool fused_first = false;
            if (s == 0) {                                                         // akaze.cpp:640-662
                int* first = (G % 2 == 0) ? Lt : tmp;
                if (fused && c->fuse_head && hak_stream_pays(c->fuse_sf, oc.w, oc.h, nimg))
                    fused_first = hakf_launch_fed_sf_head(st, A + L.lt(o - 1, 0), L.oct[o - 1], smooth, flow, (G % 2 == 1) ? Lt : tmp, S, oc,
                                                          nimg, c->itaps1, cfg.diffusivity, lp.tau.data(), hak_fed_group_size(n, G, 0),
                                                          c->state, o, G > 1);
                if (!fused_first) {
                    hakf_launch_down_smooth(st, A + L.lt(o - 1, 0), first, smooth, S, L.oct[o - 1], oc, nimg, c->itaps1);
                    hakf_launch_flow(st, smooth, flow, S, oc.w, oc.h, oc.p, nimg, cfg.diffusivity, c->state, o);
                }
                src = first;
            } else {                                                              // akaze.cpp:664-695
                src = A + L.lt(o, s - 1);
                // low-pass + conductivity + first FED group in one streaming pass when covered, else low-pass + flow in one tile pass
                if (fused && hak_stream_pays(c->fuse_sf, oc.w, oc.h, nimg))
                    fused_first = hakf_launch_fed_sf(st, src, smooth, flow, (G % 2 == 1) ? Lt : tmp, S, oc.w, oc.h, oc.p, nimg, c->itaps1,
                                                     cfg.diffusivity, lp.tau.data(), hak_fed_group_size(n, G, 0), c->state, o, G > 1);
                if (!fused_first)
                    hakf_launch_smooth_flow(st, src, smooth, flow, S, oc.w, oc.h, oc.p, nimg, c->itaps1, cfg.diffusivity, c->state, o);
            }
            int done = 0;
            for (int g = 0; g < G; g++) {
                const int ns = fused ? hak_fed_group_size(n, G, g) : 1;
                int* dst = ((G - g) % 2 == 1) ? Lt : tmp;
                if (g == 0 && fused_first) { done += ns; src = dst; continue; }
                if (fused) hakf_launch_fed_group(st, src, flow, dst, S, oc.w, oc.h, oc.p, nimg, lp.tau.data() + done, ns);
                else hakf_launch_nld_step(st, src, flow, dst, S, oc.w, oc.h, oc.p, nimg, lp.tau[done]);
                done += ns;
                src = dst;
            }
            if (!hakf_launch_hessian_level(st, smooth, A + L.dxy(o, s), flow, false, S, oc.w, oc.h, oc.p, nimg,
                                           lp.sigma_size, &b, &L, &c->htab, o, s, idthreshold)) {
                hakf_launch_hessian(st, smooth, A + L.dxy(o, s), flow, S, oc.w, oc.h, oc.p, nimg, lp.sigma_size);
                hakf_launch_extrema(st, b, L, c->dtab, o, s, idthreshold, L.flow_off[o]);
            }
        }
    }
    hak_launch_nms_emit(st, b, L, c->dtab, c->psz, d_points, max_pts, d_num_pts, 1);
    hak_launch_clear_maps(st, b, L);
    hakf_launch_describe(st, b, L, c->dtab, d_points, max_pts, cfg.descriptor_pattern_size, cfg.upright, desc, c->htab.dsc_plan_ok);
    if (hipGetLastError() != hipSuccess) return fail("kernel launch failed");
    return 0;
}

extern "C" int hak_fast_detect_and_compute_batch(hak_ctx* c, const unsigned char* d_images, long image_stride, int pitch,
                                                 int nimg, hak_point* d_points, int* d_num_pts, int desc)
{
    if (!c || !d_images || !d_points || !d_num_pts) return fail("null argument");
    if (nimg < 1 || nimg > c->cfg.batch) return fail("nimg exceeds the context's batch capacity");
    if (pitch < c->L.oct[0].w) return fail("pitch smaller than width");
    order_after_null_stream(c, c->stream);
    maps_guard_begin(c);
    return maps_guard_end(c, enqueue_fast_detect(c, d_images, image_stride, pitch, nimg, d_points, d_num_pts, desc, c->cfg.max_pts));
}

extern "C" int hak_fast_detect_and_compute(hak_ctx* c, const unsigned char* d_image, int pitch, hak_point* d_points, int max_pts,
                                           int* num_pts, hak_point* h_points, int desc)
{
    if (!c || !d_image || !d_points || !num_pts) return fail("null argument");
    if (max_pts < 1) return fail("max_pts < 1");
    if (pitch < c->L.oct[0].w) return fail("pitch smaller than width");
    order_after_null_stream(c, c->stream);
    maps_guard_begin(c);
    if (maps_guard_end(c, enqueue_fast_detect(c, d_image, 0, pitch, 1, d_points, c->d_num, desc, max_pts))) return 1;
    HIP_TRY(hipMemcpyAsync(c->h_num, c->d_num, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    *num_pts = c->h_num[0];
    if (h_points && *num_pts > 0)
        HIP_TRY(hipMemcpy(h_points, d_points, sizeof(hak_point) * (size_t)*num_pts, hipMemcpyDeviceToHost));
    return 0;
}

// enqueue one detect+describe sequence: replay the captured graph when the arguments repeat, else capture it
static int run_detect_inner(hak_ctx* c, const float* d_images, long image_stride, int pitch, int nimg,
                            hak_point* d_points, int* d_num_pts, int desc, int max_pts, hak_point* h_pinned, int cap0, int cap1);
// h_pinned: device-visible host destination of the records (and c->h_num of the counts) written by the sequence itself, or NULL
// max_pts: the record stride between images and their clamp; cap0 / cap1 > 0 (two images): smaller clamps per image
static int run_detect(hak_ctx* c, const float* d_images, long image_stride, int pitch, int nimg,
                      hak_point* d_points, int* d_num_pts, int desc, int max_pts, hak_point* h_pinned = nullptr, int cap0 = 0, int cap1 = 0)
{
    order_after_null_stream(c, c->stream);
    maps_guard_begin(c);
    return maps_guard_end(c, run_detect_inner(c, d_images, image_stride, pitch, nimg, d_points, d_num_pts, desc, max_pts, h_pinned, cap0, cap1));
}
static int run_detect_inner(hak_ctx* c, const float* d_images, long image_stride, int pitch, int nimg,
                            hak_point* d_points, int* d_num_pts, int desc, int max_pts, hak_point* h_pinned, int cap0, int cap1)
{
    // A launch-bound sequence (single images: the spine order of enqueue_detect) is issued eagerly: with ~50 launches on four
    // streams the host keeps ahead of the GPU, and the graph replay of ROCm 7.2 submits queue by queue in an order of its own
    // (measured on the C++ demo, ms per 1080p pair: eager 1.18, replay 1.31; HAK_GRAPH=2 forces the replay).
    const bool launch_bound = c->concurrent && c->L.noct > 1 && spine_pays(c, nimg);
    if (!c->use_graph || c->prof_on || (launch_bound && c->graph_mode != 2))
        return enqueue_detect(c, d_images, image_stride, pitch, nimg, d_points, d_num_pts, desc, max_pts, h_pinned, cap0, cap1);
    hak_ctx::GraphKey key;
    memset(&key, 0, sizeof(key));
    key.img = d_images; key.stride = image_stride; key.pitch = pitch; key.nimg = nimg; key.pts = d_points;
    key.num = d_num_pts; key.desc = desc; key.max_pts = max_pts; key.conc = c->concurrent ? 1 : 0; key.st = c->stream; key.hpts = h_pinned; key.cap0 = cap0; key.cap1 = cap1;
    int slot = -1, victim = 0;
    for (int i = 0; i < hak_ctx::NGRAPH; i++) {
        if (c->graph_exec[i] && memcmp(&key, &c->gkey[i], sizeof(key)) == 0) slot = i;
        if (c->graph_age[i] < c->graph_age[victim]) victim = i;
    }
    if (slot >= 0) {
        c->graph_age[slot] = ++c->graph_clock;
        if (hipGraphLaunch(c->graph_exec[slot], c->stream) != hipSuccess) return fail("hipGraphLaunch");
        return 0;
    }
    slot = victim;                                              // least recently used (or empty) slot
    if (c->graph_exec[slot]) { (void)hipGraphExecDestroy(c->graph_exec[slot]); c->graph_exec[slot] = nullptr; }
    hipGraph_t graph = nullptr;
    if (hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal) != hipSuccess) {
        (void)hipGetLastError();
        c->use_graph = false;                                   // e.g. legacy default stream: fall back to eager launches
        return enqueue_detect(c, d_images, image_stride, pitch, nimg, d_points, d_num_pts, desc, max_pts, h_pinned, cap0, cap1);
    }
    const int rc = enqueue_detect(c, d_images, image_stride, pitch, nimg, d_points, d_num_pts, desc, max_pts, h_pinned, cap0, cap1);
    const hipError_t e = hipStreamEndCapture(c->stream, &graph);
    if (rc) { if (graph) (void)hipGraphDestroy(graph); return rc; }
    if (e != hipSuccess || !graph) return fail(std::string("hipStreamEndCapture: ") + hipGetErrorString(e));
    const hipError_t ei = hipGraphInstantiate(&c->graph_exec[slot], graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    if (ei != hipSuccess) { c->graph_exec[slot] = nullptr; return fail(std::string("hipGraphInstantiate: ") + hipGetErrorString(ei)); }
    c->gkey[slot] = key;
    c->graph_age[slot] = ++c->graph_clock;
    if (hipGraphLaunch(c->graph_exec[slot], c->stream) != hipSuccess) return fail("hipGraphLaunch");
    return 0;
}

extern "C" int hak_detect_and_compute_batch(hak_ctx* c, const float* d_images, long image_stride, int pitch,
                                            int nimg, hak_point* d_points, int* d_num_pts, int desc)
{
    if (!c || !d_images || !d_points || !d_num_pts) return fail("null argument");
    if (nimg < 1 || nimg > c->cfg.batch) return fail("nimg exceeds the context's batch capacity");
    if (pitch < c->L.oct[0].w) return fail("pitch smaller than width");
    return run_detect(c, d_images, image_stride, pitch, nimg, d_points, d_num_pts, desc, c->cfg.max_pts);
}

// is p device-visible (pinned) host memory?  A pageable pointer makes the query fail: not an error here.
// ... and may the download kernel write it directly?  k_download stores 8-byte words through the pointer itself, so it must be
// 8-byte aligned and mapped into this device at the same address (hipHostMalloc memory is; an offset into a pinned buffer or
// hipHostRegister'd memory need not be) -- anything else takes the count + hipMemcpy route.
static bool host_pinned(const void* p)
{
    hipPointerAttribute_t a{};
    bool pinned = p && ((uintptr_t)p & 7) == 0 && hipPointerGetAttributes(&a, p) == hipSuccess && a.type == hipMemoryTypeHost;
    if (pinned) {
        void* dp = nullptr;
        pinned = hipHostGetDevicePointer(&dp, const_cast<void*>(p), 0) == hipSuccess && dp == p;
    }
    (void)hipGetLastError();
    return pinned;
}

extern "C" int hak_detect_and_compute(hak_ctx* c, const float* d_image, int pitch, hak_point* d_points, int max_pts,
                                      int* num_pts, hak_point* h_points, int desc)
{
    if (!c || !d_image || !d_points || !num_pts) return fail("null argument");
    if (max_pts < 1) return fail("max_pts < 1");
    if (pitch < c->L.oct[0].w) return fail("pitch smaller than width");
    // A pinned h_points (hak_host_alloc: what initAkazeData of the C++ layer hands out) is filled by the launch sequence itself,
    // count included: one synchronisation and the results are there.  A pageable one takes the reference's route
    // (akaze.cpp:134-139): count first, then a copy of the valid records.
    // HAK_TIMING=1: host-side split of the call (submission vs waiting), printed every 100 calls -- diagnosis only
    static const bool timing = [] { const char* e = getenv("HAK_TIMING"); return e && atoi(e) != 0; }();
    static double t_sub = 0, t_wait = 0; static int t_n = 0;
    const auto t0 = std::chrono::steady_clock::now();
    hak_point* h_pinned = host_pinned(h_points) ? h_points : nullptr;
    if (run_detect(c, d_image, 0, pitch, 1, d_points, c->d_num, desc, max_pts, h_pinned)) return 1;
    if (!h_pinned) HIP_TRY(hipMemcpyAsync(c->h_num, c->d_num, sizeof(int), hipMemcpyDeviceToHost, c->sync_stream));
    const auto t1 = std::chrono::steady_clock::now();
    HIP_TRY(hipStreamSynchronize(c->sync_stream));
    if (timing) {
        const auto t2 = std::chrono::steady_clock::now();
        t_sub += std::chrono::duration<double, std::micro>(t1 - t0).count();
        t_wait += std::chrono::duration<double, std::micro>(t2 - t1).count();
        if (++t_n % 100 == 0) { fprintf(stderr, "hak timing: submit %.1f us, wait %.1f us per call\n", t_sub / 100, t_wait / 100); t_sub = t_wait = 0; }
    }
    *num_pts = c->h_num[0];
    if (h_points && !h_pinned && *num_pts > 0)                                    // akaze.cpp:134-139
        HIP_TRY(hipMemcpy(h_points, d_points, sizeof(hak_point) * (size_t)*num_pts, hipMemcpyDeviceToHost));
    return 0;
}

// Scratch for callers without a context (cuMatch is a free function in the reference, so hak_match / hak_match_knn2 accept
// ctx == NULL): a pool per device, guarded by a mutex.  A call takes a scratch of the CURRENT device for its duration -- both
// entry points synchronise before they return -- and puts it back; concurrent callers get different ones.
namespace {
std::mutex g_pool_mu;
std::vector<HakMatchScratch*> g_pool;
HakMatchScratch* pool_acquire()
{
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lock(g_pool_mu);
    for (size_t i = 0; i < g_pool.size(); i++)
        if (g_pool[i]->device == dev) { HakMatchScratch* sc = g_pool[i]; g_pool.erase(g_pool.begin() + (long)i); return sc; }
    HakMatchScratch* sc = new HakMatchScratch();
    sc->device = dev;
    return sc;
}
void pool_release(HakMatchScratch* sc, bool ok)
{
    // a call that failed may have left keys / tickets behind: such a scratch is not handed out again
    if (!ok) { hak_match_scratch_free(sc); delete sc; return; }
    std::lock_guard<std::mutex> lock(g_pool_mu);
    g_pool.push_back(sc);
}
}

// One PAIR per call (include/hipakaze.h): both images through ONE launch sequence (the batch path with two images: every launch
// covers both), the match appended, the records scattered to the caller's arrays by one more kernel, ONE synchronisation --
// instead of the three synchronous calls of main.cpp:201-209 (43 + 43 + 1 launches, three waits).
extern "C" int hak_detect_and_compute_pair(hak_ctx* c, const float* d_image1, const float* d_image2, int pitch,
                                           hak_point* d_points1, hak_point* d_points2, int max_pts1, int max_pts2,
                                           int* num_pts1, int* num_pts2, hak_point* h_points1, hak_point* h_points2, int desc, int match)
{
    if (!c || !d_image1 || !d_image2 || !d_points1 || !d_points2 || !num_pts1 || !num_pts2) return fail("null argument");
    if (c->cfg.batch < 2) return fail("hak_detect_and_compute_pair needs a context created with batch >= 2");
    if (max_pts1 < 1 || max_pts2 < 1) return fail("max_pts < 1");
    if (pitch < c->L.oct[0].w) return fail("pitch smaller than width");
    const long mp = c->cfg.max_pts;
    if (mp >= (1 << 20)) return fail("max_pts must stay below 2^20 for the matcher");
    if (!c->pair_pts) HIP_TRY(hipMalloc((void**)&c->pair_pts, sizeof(hak_point) * 2 * (size_t)mp));
    // each image keeps its own clamp, as in the three calls (setMaxNumPoints(result.max_pts), akaze.cpp:246, 451), bounded by the
    // context's max_pts -- the record stride of the pair buffer, which every kernel of the sequence and the matcher index with
    const int cap0 = max_pts1 < mp ? max_pts1 : (int)mp, cap1 = max_pts2 < mp ? max_pts2 : (int)mp;
    if (run_detect(c, d_image1, (long)(d_image2 - d_image1), pitch, 2, c->pair_pts, c->d_num, desc, (int)mp, nullptr, cap0, cap1)) return 1;
    if (match) {
        ProfScope ps(c, HAK_PROF_MATCH);
        hak_launch_match(c->sync_stream, c->pair_pts, c->pair_pts + mp, c->d_num, c->d_num + 1, (int)mp, (int)mp, 2 * mp, 2 * mp, 1, &c->msc);
    }
    HakPairDst dst{{d_points1, d_points2}, {host_pinned(h_points1) ? h_points1 : nullptr, host_pinned(h_points2) ? h_points2 : nullptr},
                   {max_pts1, max_pts2}};
    hak_launch_download_pair(c->sync_stream, c->pair_pts, c->d_num, mp, dst, c->h_num);
    if (hipGetLastError() != hipSuccess) return fail("pair launch failed");
    HIP_TRY(hipStreamSynchronize(c->sync_stream));
    *num_pts1 = c->h_num[0];
    *num_pts2 = c->h_num[1];
    // pageable host arrays take the reference's route (akaze.cpp:134-139): a copy of the valid records
    if (h_points1 && !dst.h[0] && *num_pts1 > 0)
        HIP_TRY(hipMemcpy(h_points1, d_points1, sizeof(hak_point) * (size_t)*num_pts1, hipMemcpyDeviceToHost));
    if (h_points2 && !dst.h[1] && *num_pts2 > 0)
        HIP_TRY(hipMemcpy(h_points2, d_points2, sizeof(hak_point) * (size_t)*num_pts2, hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int hak_match(hak_ctx* c, hak_point* d_pts1, int n1, const hak_point* d_pts2, int n2, hak_point* h_pts1)
{
    // ctx may be NULL (cuMatch is a free function in the reference): default stream, no profiling
    if (!d_pts1 || (!d_pts2 && n2 > 0)) return fail("null argument");
    if (n1 <= 0) return 0;
    if (n2 >= (1 << 20)) return fail("more than 2^20 - 1 train points");         // k_match packs distance << 20 | index
    // one big pair takes the sliced search (kernels_match.hip), whose scratch belongs to the context (its device) or, without
    // one, comes from the per-device pool above for the duration of the call -- no process-wide buffer
    hipStream_t st = c ? c->stream : nullptr;
    HakMatchScratch* sc = c ? &c->msc : pool_acquire();
    order_after_null_stream(c, st);
    {
        ProfScope ps(c, HAK_PROF_MATCH);
        hak_launch_match(st, d_pts1, d_pts2, nullptr, nullptr, n1, n2, 0, 0, 1, sc);
    }
    int rc = 0;
    if (hipGetLastError() != hipSuccess) rc = fail("match launch failed");
    if (hipStreamSynchronize(st) != hipSuccess) rc = fail("hipStreamSynchronize(match)");
    if (!c) pool_release(sc, rc == 0);
    else if (rc) hak_match_scratch_free(sc);
    if (rc) return rc;
    if (h_pts1)                                                                   // akaze.cpp:58-63
        HIP_TRY(hipMemcpy2D(&h_pts1[0].match, sizeof(hak_point), &d_pts1[0].match, sizeof(hak_point), 16, n1,
                            hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int hak_match_batch(hak_ctx* c, hak_point* d_points, const int* d_num_pts, int npairs)
{
    if (!c || !d_points || !d_num_pts || npairs < 1) return fail("bad argument");
    const long mp = c->cfg.max_pts;
    if (mp >= (1 << 20)) return fail("max_pts must stay below 2^20 for the matcher");   // k_match packs distance << 20 | index
    order_after_null_stream(c, c->stream);
    { ProfScope ps(c, HAK_PROF_MATCH);
      hak_launch_match(c->stream, d_points, d_points + mp, d_num_pts, d_num_pts + 1, (int)mp, (int)mp, 2 * mp, 2 * mp, npairs, &c->msc); }
    if (hipGetLastError() != hipSuccess) return fail("match launch failed");
    return 0;
}

// ----------------------------------------------------------- match post-processing
static int knn_scratch(hak_ctx* c)
{
    if (c->knn) return 0;
    const size_t npair = (size_t)(c->cfg.batch + 1) / 2;
    HIP_TRY(hipMalloc((void**)&c->knn, sizeof(int4) * 2 * npair * (size_t)c->cfg.max_pts));
    HIP_TRY(hipMalloc((void**)&c->d_cnt, sizeof(int) * npair));
    return 0;
}

extern "C" int hak_match_knn2(hak_ctx* c, hak_point* d_pts1, int n1, const hak_point* d_pts2, int n2, int ratio_num,
                              int ratio_den, int cross_check, int max_dist, hak_point* h_pts1, hak_match_pair* d_out,
                              int* count, hak_match_pair* h_out)
{
    if (!d_pts1 || (!d_pts2 && n2 > 0) || !count) return fail("null argument");
    if (ratio_num <= 0 || ratio_den <= 0) return fail("ratio must be a positive fraction");
    if (h_out && !d_out) return fail("h_out needs d_out");
    *count = 0;
    if (n1 <= 0) return 0;
    if (max_dist <= 0) max_dist = HAK_MAX_DIST;
    hipStream_t st = c ? c->stream : nullptr;
    HakMatchScratch* sc = c ? &c->msc : pool_acquire();
    order_after_null_stream(c, st);
    const int nb = (n1 + 1023) / 1024;
    if (!hak_match_scratch_reserve(sc, st, 0, 0, 0, (long)n1 + (long)(n2 > 0 ? n2 : 1), nb)) {
        if (!c) pool_release(sc, false);
        return fail("hak_match_knn2: out of device memory for the 2-NN scratch");
    }
    int4* fwd = sc->knn;
    int4* rev = sc->knn + n1;
    *sc->h_cnt = -1;
    {
        hak_launch_knn2(st, d_pts1, d_pts2, nullptr, nullptr, n1, n2, 0, 0, 1, fwd, 0, sc);
        if (cross_check && n2 > 0) hak_launch_knn2(st, d_pts2, d_pts1, nullptr, nullptr, n2, n1, 0, 0, 1, rev, 0, sc);
        hak_launch_knn2_finish(st, d_pts1, d_pts2, nullptr, n1, 0, 0, 1, fwd, cross_check ? rev : nullptr, 0, ratio_num, ratio_den,
                               cross_check ? 1 : 0, max_dist, d_out, 0, sc->d_cnt, sc);
    }
    int rc = 0;
    if (hipGetLastError() != hipSuccess) rc = fail("knn2 launch failed");
    if (!rc && hipStreamSynchronize(st) != hipSuccess) rc = fail("sync");
    // the multi-block finish leaves the count in the scratch's pinned word; the one-block finish only in device memory
    if (!rc && *sc->h_cnt < 0 && hipMemcpy(sc->h_cnt, sc->d_cnt, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) rc = fail("count download");
    if (!rc) *count = *sc->h_cnt;
    if (!c) pool_release(sc, rc == 0);
    else if (rc) hak_match_scratch_free(sc);
    if (!rc && h_out && *count > 0 &&
        hipMemcpy(h_out, d_out, sizeof(hak_match_pair) * (size_t)*count, hipMemcpyDeviceToHost) != hipSuccess)
        rc = fail("match list download");
    if (!rc && h_pts1 &&                                                          // akaze.cpp:58-63
        hipMemcpy2D(&h_pts1[0].match, sizeof(hak_point), &d_pts1[0].match, sizeof(hak_point), 16, n1, hipMemcpyDeviceToHost) != hipSuccess)
        rc = fail("match field download");
    return rc;
}

extern "C" int hak_match_knn2_batch(hak_ctx* c, hak_point* d_points, const int* d_num_pts, int npairs, int ratio_num,
                                    int ratio_den, int cross_check, int max_dist, hak_match_pair* d_out, int* d_counts)
{
    if (!c || !d_points || !d_num_pts || !d_counts || npairs < 1) return fail("bad argument");
    if (2 * npairs > c->cfg.batch + 1) return fail("npairs exceeds the context's batch capacity");
    if (ratio_num <= 0 || ratio_den <= 0) return fail("ratio must be a positive fraction");
    if (max_dist <= 0) max_dist = HAK_MAX_DIST;
    if (knn_scratch(c)) return 1;
    order_after_null_stream(c, c->stream);
    const long mp = c->cfg.max_pts;
    int4* fwd = c->knn;
    int4* rev = c->knn + (size_t)((c->cfg.batch + 1) / 2) * mp;
    { ProfScope ps(c, HAK_PROF_MATCH);
      hak_launch_knn2(c->stream, d_points, d_points + mp, d_num_pts, d_num_pts + 1, (int)mp, (int)mp, 2 * mp, 2 * mp, npairs, fwd, mp);
      if (cross_check)
          hak_launch_knn2(c->stream, d_points + mp, d_points, d_num_pts + 1, d_num_pts, (int)mp, (int)mp, 2 * mp, 2 * mp, npairs, rev, mp);
      hak_launch_knn2_finish(c->stream, d_points, d_points + mp, d_num_pts, 0, 2 * mp, 2 * mp, npairs, fwd, cross_check ? rev : nullptr,
                             mp, ratio_num, ratio_den, cross_check ? 1 : 0, max_dist, d_out, mp, d_counts); }
    if (hipGetLastError() != hipSuccess) return fail("knn2 launch failed");
    return 0;
}

// ----------------------------------------------------------- memory helpers
extern "C" int hak_points_alloc(hak_point** d, int count)
{
    HIP_TRY(hipMalloc((void**)d, sizeof(hak_point) * (size_t)count));
    HIP_TRY(hipMemset(*d, 0, sizeof(hak_point) * (size_t)count));
    HIP_TRY(hipStreamSynchronize(nullptr));                     // (the fill runs on the NULL stream; the caller's streams need not wait for that one)
    return 0;
}
extern "C" int hak_points_free(hak_point* d) { HIP_TRY(hipFree(d)); return 0; }

extern "C" int hak_image_alloc(float** d, int w, int h, int* pitch)
{
    int p = (w % 128 != 0) ? (w - w % 128 + 128) : w;                             // cuda_utils.h:160, main.cpp:174
    HIP_TRY(hipMalloc((void**)d, sizeof(float) * (size_t)p * h));
    if (pitch) *pitch = p;
    return 0;
}
extern "C" int hak_image_upload(float* d, int pitch, const float* hsrc, int w, int h)
{
    HIP_TRY(hipMemcpy2D(d, sizeof(float) * pitch, hsrc, sizeof(float) * w, sizeof(float) * w, h, hipMemcpyHostToDevice));
    return 0;
}
extern "C" int hak_image_free(float* d) { HIP_TRY(hipFree(d)); return 0; }

extern "C" int hak_ingest_u8(hak_ctx* c, const unsigned char* d_src, long src_stride, int src_pitch,
                             float* d_dst, long dst_stride, int dst_pitch, int w, int h, int nimg)
{
    if (!d_src || !d_dst || w < 1 || h < 1 || nimg < 1 || src_pitch < w || dst_pitch < w) return fail("bad ingest argument");
    if (hak_device_count() == 0) return fail("no HIP device: libhipakaze has no CPU fallback");
    order_after_null_stream(c, c ? c->stream : nullptr);
    hak_launch_ingest_u8(c ? c->stream : nullptr, d_src, src_stride, src_pitch, d_dst, dst_stride, dst_pitch, w, h, nimg);
    if (hipGetLastError() != hipSuccess) return fail("ingest launch failed");
    return 0;
}
extern "C" int hak_host_alloc(void** p, long bytes) { HIP_TRY(hipHostMalloc(p, (size_t)bytes)); return 0; }
extern "C" int hak_host_free(void* p) { HIP_TRY(hipHostFree(p)); return 0; }

extern "C" int hak_download_batch(hak_ctx* c, const hak_point* d_points, const int* d_num_pts, int nimg,
                                  hak_point* h_points, int* h_num_pts)
{
    if (!c || !d_points || !d_num_pts || !h_points || !h_num_pts) return fail("null argument");
    // pinned (device-visible) destination buffers -- hak_host_alloc -- take one kernel that stores counts and records over PCIe
    {
        hipPointerAttribute_t ap{}, an{};
        const bool pinned = hipPointerGetAttributes(&ap, h_points) == hipSuccess && ap.type == hipMemoryTypeHost &&
                            hipPointerGetAttributes(&an, h_num_pts) == hipSuccess && an.type == hipMemoryTypeHost;
        (void)hipGetLastError();                                    // a pageable pointer makes the query fail: not an error here
        if (pinned) {
            hak_launch_download(c->stream, d_points, d_num_pts, c->cfg.max_pts, nimg, h_points, h_num_pts);
            if (hipGetLastError() != hipSuccess) return fail("download launch failed");
            HIP_TRY(hipStreamSynchronize(c->stream));
            return 0;
        }
    }
    HIP_TRY(hipMemcpyAsync(h_num_pts, d_num_pts, sizeof(int) * (size_t)nimg, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    const long mp = c->cfg.max_pts;
    for (int i = 0; i < nimg; i++)
        if (h_num_pts[i] > 0)
            HIP_TRY(hipMemcpyAsync(h_points + i * mp, d_points + i * mp, sizeof(hak_point) * (size_t)h_num_pts[i],
                                   hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return 0;
}

extern "C" int hak_memcpy_d2h(void* dst, const void* src, long bytes)
{
    HIP_TRY(hipMemcpy(dst, src, (size_t)bytes, hipMemcpyDeviceToHost));
    return 0;
}
extern "C" int hak_memcpy_h2d(void* dst, const void* src, long bytes)
{
    HIP_TRY(hipMemcpy(dst, src, (size_t)bytes, hipMemcpyHostToDevice));
    return 0;
}

// ------------------------------------------------------------ introspection
extern "C" int hak_query_schedule(const hak_ctx* c, int* nsteps, int* sigma_size, float* sizes, float* borders)
{
    if (!c) return -1;
    for (size_t l = 0; l < c->plan.size(); l++) {
        if (nsteps) nsteps[l] = c->plan[l].nsteps;
        if (sigma_size) sigma_size[l] = c->plan[l].sigma_size;
        if (sizes) sizes[l] = c->plan[l].size;
        if (borders) borders[l] = c->plan[l].border;
    }
    return c->L.noct;
}

extern "C" int hak_query_geometry(const hak_ctx* c, int* whp)
{
    if (!c) return -1;
    for (int o = 0; o < c->L.noct; o++) {
        whp[3 * o] = c->L.oct[o].w; whp[3 * o + 1] = c->L.oct[o].h; whp[3 * o + 2] = c->L.oct[o].p;
    }
    return c->L.noct;
}

extern "C" int hak_query_traffic(const hak_ctx* c, int npts_hint, hak_traffic* out)
{
    if (!c || !out) return fail("null argument");
    const HakLayout& L = c->L;
    double pxsteps = 0, all = 0, folded = 0;
    int launches = 0;
    for (int o = 0; o < L.noct; o++) {
        const double N = (double)L.oct[o].w * L.oct[o].h;
        for (int s = 0; s < L.ms; s++) {
            const LevelPlan& lp = c->plan[(size_t)o * L.ms + s];
            pxsteps += N * lp.nsteps;
            launches += lp.nsteps ? hak_fed_groups(lp.nsteps, c->max_fuse, L.oct[o].w) : 0;
            // sublevels whose low-pass (8 B/px) + conductivity (8 B/px) run inside the first FED launch (k_fed_sf), and octave
            // heads whose decimation + low-pass (4 N_{o-1} + 8 N_o) + conductivity (8 N_o) do
            const bool covered = lp.nsteps && hak_stream_pays(c->fuse_sf, L.oct[o].w, L.oct[o].h, c->cfg.batch) &&
                                 c->cfg.diffusivity == HAK_PM_G2 && (L.oct[o].w & 3) == 0 && L.oct[o].w >= 16 && L.oct[o].h >= 8;
            if (covered && s > 0) folded += 16.0 * N;
            if (covered && c->fuse_head && s == 0 && o > 0 && !(L.oct[o - 1].w & 1) && !(L.oct[o - 1].h & 1))
                folded += 4.0 * L.oct[o - 1].w * L.oct[o - 1].h + 16.0 * N;
            if (o == 0 && s == 0) all += 56.0 * N;                                // o0 prologue
            else if (s == 0) all += 4.0 * L.oct[o - 1].w * L.oct[o - 1].h + 8.0 * N + 8.0 * N + 24.0 * N + 4.0 * N;
            else all += 44.0 * N;
        }
    }
    all += 16.0 * L.oct[0].w * L.oct[0].h;                                        // maps init + NMS scan
    all += 12.0 * pxsteps;
    all += (872.0 + 5292.0 + 104.0) * npts_hint;
    out->fed_px_steps = pxsteps;
    out->fed_bytes = 12.0 * pxsteps + folded;
    out->all_stage_bytes = all;
    out->fed_launches = launches;
    // per-class compulsory bytes of the launches AS BUILT (fused): what each class must move per image even with perfect
    // reuse inside a launch.  The FED figure is accumulated by the launch sequence itself (valid after the first detect call).
    out->fed_fused_bytes = c->fed_fused_bytes;
    double lvl_px = 0;
    for (int o = 0; o < L.noct; o++) lvl_px += (double)L.ms * L.oct[o].w * L.oct[o].h;
    out->hessian_bytes = 12.0 * lvl_px;                                           // read smooth, write the interleaved {Lx, Ly} plane
    out->prologue_bytes = 16.0 * L.oct[0].w * L.oct[0].h;                         // read img, write Lt + gradient; re-read gradient (histogram)
    out->describe_bytes = (872.0 + 5292.0) * npts_hint;                           // sampled bytes per keypoint (orientation + MLDB)
    out->nms_bytes = 104.0 * npts_hint;
    return 0;
}

extern "C" int hak_prof_enable(hak_ctx* c, int on) { if (!c) return 1; c->prof_on = on != 0; return 0; }
extern "C" int hak_prof_reset(hak_ctx* c)
{
    if (!c) return 1;
    for (auto& p : c->prof) { p.used = 0; p.acc_ms = 0; p.launches = 0; }
    return 0;
}
extern "C" int hak_prof_read(hak_ctx* c, int k, double* total_ms, int* launches)
{
    if (!c || k < 0 || k >= HAK_PROF_COUNT) return fail("bad profile class");
    HIP_TRY(hipStreamSynchronize(c->stream));
    ProfClass& p = c->prof[k];
    for (size_t i = 0; i + 1 < p.used; i += 2) {
        float ms = 0;
        HIP_TRY(hipEventElapsedTime(&ms, p.ev[i], p.ev[i + 1]));
        p.acc_ms += ms;
    }
    p.used = 0;
    if (total_ms) *total_ms = p.acc_ms;
    if (launches) *launches = p.launches;
    return 0;
}
