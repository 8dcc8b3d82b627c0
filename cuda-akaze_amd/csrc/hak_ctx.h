// hak_ctx.h -- the context behind the C ABI (struct hak_ctx) and its host-side helpers, shared by hak_api.hip (the product entry
// points, libhipakaze.so) and hak_test_api.hip (stage operators / probes of the test ABI, libhipakaze_test.so).  Internal.
#pragma once
#include "hak_internal.h"
#include <string>
#include <vector>

// ------------------------------------------------------------------ errors
int hak_fail(const std::string& m);          // hak_api.hip: sets the calling thread's hak_last_error() text, returns 1
static inline int fail(const std::string& m) { return hak_fail(m); }
#define HIP_TRY(call)                                                                           \
    do {                                                                                        \
        hipError_t e__ = (call);                                                                \
        if (e__ != hipSuccess)                                                                  \
            return fail(std::string(#call) + ": " + hipGetErrorString(e__));                    \
    } while (0)

// ----------------------------------------------------------------- context
struct LevelPlan {
    int nsteps = 0;
    std::vector<float> tau;
    int sigma_size = 0;
    float size = 0, border = 0;
};

struct ProfClass {
    std::vector<hipEvent_t> ev;     // pairs
    size_t used = 0;
    double acc_ms = 0;
    int launches = 0;
};

struct hak_ctx {
    hak_config cfg;
    HakLayout L;
    HakTables htab;
    HakTables* dtab = nullptr;
    std::vector<LevelPlan> plan;    // [noct*ms]
    float taps1[8], taps_base[8];
    int itaps1[8], itaps_base[8];       // FAST path: (int)(tap * 65536 + 0.5f)   akazed.cu:3896
    int base_R = 4;
    int psz = 28;
    float* arena = nullptr;
    unsigned long long* maps = nullptr;
    unsigned long long* bitmap = nullptr;
    int* rowcount = nullptr;
    unsigned long long* cand = nullptr;
    int* perm = nullptr;            // [batch][cfg.max_pts] visiting order of the keypoint kernels (HakBatch::perm)
    long cand_cap = 0;
    HakImgState* state = nullptr;
    int* d_num = nullptr;           // [batch] counts for the synchronous entry points
    int* h_num = nullptr;           // pinned
    hipStream_t own_stream = nullptr, stream = nullptr;
    // octave o+1 depends only on Lt(o,0) (akaze.cpp:371-375), so each octave runs on its own stream and the
    // latency-bound small-octave launches overlap octave 0's heavy kernels
    hipStream_t oct_stream[HAK_MAX_OCTAVES] = {};      // [0] unused (= stream)
    hipEvent_t ev_ready[HAK_MAX_OCTAVES] = {}, ev_done[HAK_MAX_OCTAVES] = {};
    // small launches (a pair, a handful of images): octave 0's Hessians leave its FED chain for a stream of their own (enqueue_detect)
    hipStream_t hess_stream = nullptr;
    hipEvent_t ev_hs[HAK_MAX_SCALES] = {}, ev_hd[HAK_MAX_SCALES] = {};   // level s's Hessian input ready / its Hessian done
    int hess_side = 0;              // HAK_HESS_SIDE=1: on for launches in the tile-kernel regime.  Off by default: measured SLOWER (pair call
                                    // 0.62 vs 0.57 ms) -- a fifth concurrent chain stretches the other four more than the shorter chain gains
    bool concurrent = true;
    // the launch sequence has no host-side data dependence, so it is captured once per argument set and replayed
    bool use_graph = true;          // env HAK_GRAPH=0 disables; profiling (event pairs) always runs eagerly
    int graph_mode = 1;             // HAK_GRAPH: 0 never, 1 replay except for launch-bound single-image sequences, 2 always
    struct GraphKey { const float* img; long stride; int pitch, nimg; hak_point* pts; int* num; int desc; int max_pts; int conc; hipStream_t st; hak_point* hpts; int cap0, cap1; };
    static constexpr int NGRAPH = 4;                    // e.g. the two images of a pair, alternating (main.cpp:201-205)
    hipGraphExec_t graph_exec[NGRAPH] = {};
    GraphKey gkey[NGRAPH] = {};
    unsigned long graph_age[NGRAPH] = {}, graph_clock = 0;
    bool prof_on = false;
    ProfClass prof[HAK_PROF_COUNT];
    int fed_launches = 0;
    double fed_fused_bytes = 0;     // compulsory HBM bytes per image of the FED launches as enqueued (read L [+ g], write L' [+ smooth, g])
    int max_fuse = 4;               // FED steps fused per launch (env HAK_FED_MAX_FUSE, 1..6)
    int fuse_head = 1;              // octave heads through the decimating k_fed_sf variant (env HAK_FUSE_HEAD=0 disables)
    int level_min_steps = 8;        // shortest FED cycle that goes through k_level_tile under the size rule (env HAK_LEVEL_MIN_STEPS)
    int fuse_sf = 1;                // low-pass + conductivity fused into the first FED launch of a sublevel: 0 never, 1 by size
                                    // (hak_stream_pays), 2 always where covered (env HAK_FUSE_SF)
    int4* knn = nullptr;            // 2-NN scratch: fwd[batch/2][max_pts] | rev[batch/2][max_pts], allocated on first use
    int* d_cnt = nullptr;
    hak_point* pair_pts = nullptr;  // [2][cfg.max_pts]: the contiguous pair layout hak_detect_and_compute_pair detects into and matches on
    HakMatchScratch msc;            // sliced searches of one big pair (hak_match / hak_match_knn2): grows on demand, on this context's device
    HakKnobs knobs;                 // kernel-selection knobs of THIS context (two contexts of a process may differ)
    hipEvent_t ev_last = nullptr;   // recorded after the last enqueue on c->stream: hak_destroy waits for it (external streams)
    hipEvent_t ev_tail_fork = nullptr, ev_tail_join = nullptr;   // the map clean-up runs beside the descriptor kernels
    hipEvent_t ev_null = nullptr;   // recorded on the NULL stream at the start of a call: the context's stream waits for it (hak_set_null_order)
    bool null_order = true;
    hipEvent_t ev_phase = nullptr;  // recorded in every detect sequence between the scale space and the keypoint stages (hak_phase_event)
    hipStream_t sync_stream = nullptr;                            // where the last detect sequence ends (c->stream unless it was left on the chain)
    bool last_fast = false;         // the arena holds the integer path's planes (hak_debug_plane)
    bool maps_dirty = false;        // a call failed between writing the key map and cleaning it up: clear it in full next time
};

struct ProfScope {
    hak_ctx* c; int k; hipEvent_t stop = nullptr; hipStream_t s;
    ProfScope(hak_ctx* ctx, int klass, hipStream_t st = nullptr) : c(ctx), k(klass), s(st ? st : (ctx ? ctx->stream : nullptr))
    {
        if (!c || !c->prof_on) return;      // (no context: hak_match(NULL, ...))
        ProfClass& p = c->prof[k];
        if (p.used + 2 > p.ev.size()) {
            // timing-only events: without the system-scope fence a default event carries (hip_runtime_api.h, hipEventDisableSystemFence:
            // "can improve the accuracy of timing measurements by avoiding the cost of cache writeback and invalidation, and the
            // performance impact of those actions on the execution of following work").  Measured on one box in alternating runs the
            // difference is 0.3-0.4 % of the FED class (DESIGN.md 4 lesson 34: the leg's first sequence is what read 4 % high).
            // hak_prof_read synchronises the stream before it reads the events.
            static const unsigned evflags = [] { const char* e = getenv("HAK_PROF_FENCE"); return e && atoi(e) ? 0u : (unsigned)hipEventDisableSystemFence; }();   // (=1: default events, for A/B)
            hipEvent_t a, b;
            (void)hipEventCreateWithFlags(&a, evflags); (void)hipEventCreateWithFlags(&b, evflags);
            p.ev.push_back(a); p.ev.push_back(b);
        }
        (void)hipEventRecord(p.ev[p.used], s);
        stop = p.ev[p.used + 1];
        p.used += 2;
        p.launches++;
    }
    ~ProfScope() { if (stop) (void)hipEventRecord(stop, s); }
};

// hak_api.hip: the key map must be all zero when a launch sequence starts; a failed call leaves the flag set (see there)
void maps_guard_begin(hak_ctx* c);
int maps_guard_end(hak_ctx* c, int rc);
