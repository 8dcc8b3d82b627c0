/*
 * hipakaze.h -- C ABI of libhipakaze.so, the MI355X-native AKAZE hot path
 * (detect + describe + match) behind the CUDA-AKAZE interface.
 *
 * The reference has no C layer: its boundary is the C++ class in akaze.h.  The
 * entry points below are what a binding for that class needs; each one cites
 * the reference interface it replaces (file:line in the reference checkout).
 * include/akaze.h re-creates akaze::Akazer / akaze::cuMatch on top of this
 * ABI, so main.cpp-style callers drop in.
 *
 * Conventions (as the reference): images are DEVICE pointers to
 * float32 in [0,1], row pitch in elements; point arrays are caller-owned
 * device arrays of 104-byte hak_point; every function returns 0 on success and
 * a non-zero status otherwise, with the message in hak_last_error().  There is
 * no CPU fallback: without a usable HIP device every compute call fails.
 */
#ifndef HIPAKAZE_H
#define HIPAKAZE_H

#ifdef __cplusplus
extern "C" {
#endif

#define HAK_FLEN 61          /* akaze_structures.h:29  (FEATURE_TYPE 5, MLDB) */
#define HAK_MAX_OCTAVES 8    /* akazed.cu:10 MAX_OCTAVE */
#define HAK_MAX_SCALES 5     /* akazed.cu:9  MAX_SCALE  */
#define HAK_MAX_DIST 96      /* akazed.cu:11 MAX_DIST   */

/* akaze_structures.h:19-40 AkazePoint: 104 bytes, align 4.
 * offsets: x 0, y 4, octave 8 (= octave*max_scale + sublevel), response 12,
 * size 16, angle 20, features 24..84, pad 85..87, match 88, distance 92,
 * match_x 96, match_y 100. */
typedef struct hak_point {
    float x;
    float y;
    int octave;
    float response;
    float size;
    float angle;
    unsigned char features[HAK_FLEN];
    int match;
    int distance;
    float match_x;
    float match_y;
} hak_point;

/* akaze_structures.h:53-59 DiffusivityType */
enum { HAK_PM_G1 = 0, HAK_PM_G2 = 1, HAK_WEICKERT = 2, HAK_CHARBONNIER = 3 };

/* The 11 arguments of Akazer::init (akaze.h:25-26, defaults akaze.h:35-54,
 * demo values main.cpp:156-166) plus what the reference keeps as macros or
 * call arguments. */
typedef struct hak_config {
    int noctaves;                 /* 4 */
    int max_scale;                /* 4  sublevels per octave */
    float per;                    /* 0.7  contrast percentile */
    float kcontrast;              /* 0.03 (overwritten per image, akazed.cu:2481) */
    float soffset;                /* 1.6 */
    int reordering;               /* 1 */
    float derivative_factor;      /* 1.5 */
    float dthreshold;             /* 0.001 */
    int diffusivity;              /* HAK_PM_G2 */
    int descriptor_pattern_size;  /* 10 */
    int max_pts;                  /* capacity of every per-image point array (main.cpp:155: 10000) */
    int upright;                  /* 0; 1 = MLDB-upright extension: angle = 0 (no reference behaviour) */
    int batch;                    /* images one launch sequence processes (>= 1) */
} hak_config;

typedef struct hak_ctx hak_ctx;

/* ---- device / errors: cuda_utils.h:41-67 initDevice, :18-37 CHECK/CheckMsg */
int hak_device_count(void);
int hak_set_device(int dev);
const char* hak_last_error(void);
void hak_default_config(hak_config* cfg);

/* ---- detector object: Akazer::Akazer/init/allocMemory/~Akazer (akaze.cpp:67-98, 204-237).
 * Geometry is fixed at creation (w x h pixels); the arena for `batch` images,
 * the FED schedule and all tables live in the context. */
int hak_create(const hak_config* cfg, int w, int h, hak_ctx** out);
/* Waits for the context's last launch sequence (an event recorded behind it, so a caller-provided stream that has been destroyed
 * meanwhile is never touched), then releases everything.  Kernel-selection knobs are read from the environment by
 * hak_create INTO the context: two contexts of one process may differ. */
void hak_destroy(hak_ctx* ctx);
/* Run on a caller-provided hipStream_t (e.g. torch's current stream); NULL = the context's own.  The stream must stay valid while
 * calls are made with it; switching streams does not synchronise -- order work across the two yourself (hak_sync before switching
 * is enough).  Launch-bound single-image sequences and batched ones both start and end on this stream. */
int hak_set_stream(hak_ctx* ctx, void* hip_stream);
int hak_sync(hak_ctx* ctx);
/* make the context's stream wait for a hipEvent_t recorded elsewhere (e.g. the end of an upload on a copy stream) without
 * blocking the host: everything enqueued on the context afterwards runs behind the event */
int hak_wait_event(hak_ctx* ctx, void* hip_event);
/* the context's phase event (a hipEvent_t owned by the context): every float detect sequence records it where its scale space
 * (FED / Hessian kernels: bound by HBM stores) ends and its keypoint stages (NMS, orientation, MLDB, then the caller's match:
 * bound by gathers and integer work) begin.  A caller that keeps two contexts busy makes each one's next sequence wait for the
 * OTHER one's phase event (hak_wait_event): the two kinds of work then run beside each other instead of in lockstep. */
int hak_phase_event(hak_ctx* ctx, void** hip_event);
/* 1 (default): octaves run on their own HIP streams (octave o+1 depends only on Lt(o,0), akaze.cpp:371-375);
 * 0: one stream, strictly serial launches (used for per-kernel timing). Env HAK_SERIAL=1 presets 0. */
int hak_set_concurrency(hak_ctx* ctx, int on);
/* 1 (default): every call that enqueues work on the context first makes the context's stream wait for what the caller has enqueued on
 * the NULL stream so far (an event recorded there; no host wait).  The reference runs on the default stream, so its callers' own
 * hipMemset / hipMemcpyAsync / kernels on that stream are ordered in front of detectAndCompute and cuMatch by themselves
 * (akaze.cpp:101-150 issues everything on stream 0); a context's streams are non-blocking and would not wait (hipMemset returns
 * before its fill has run: tools/probes/memset_order_probe.hip).  0: no such dependency (a caller that drives the context from its
 * own stream, hak_set_stream, or captures graphs while calling).  Env HAK_NULL_ORDER=0 presets 0. */
int hak_set_null_order(hak_ctx* ctx, int on);

/* ---- Akazer::detectAndCompute (akaze.h:29, akaze.cpp:101-150), one image,
 * synchronous.  d_image: device float32, pitch elements per row.  d_points:
 * device array of max_pts points; max_pts is also this call's clamp, as the
 * reference's setMaxNumPoints(result.max_pts) (akaze.cpp:246, 451) -- it may be
 * smaller or larger than cfg.max_pts (which sizes the batch entry points).
 * *num_pts (host) receives the count; when h_points != NULL the points are
 * copied to it (whole 104-byte records). */
int hak_detect_and_compute(hak_ctx* ctx, const float* d_image, int pitch,
                           hak_point* d_points, int max_pts, int* num_pts,
                           hak_point* h_points, int desc);

/* ---- batched, asynchronous form used by the frame-sharded driver.
 * nimg <= cfg.batch images at d_images + i*image_stride (elements); points of
 * image i at d_points + i*max_pts; counts written to d_num_pts[i] (device).
 * Returns after enqueueing; hak_sync() or stream sync to wait. */
int hak_detect_and_compute_batch(hak_ctx* ctx, const float* d_images, long image_stride, int pitch,
                                 int nimg, hak_point* d_points, int* d_num_pts, int desc);

/* ---- Akazer::fastDetectAndCompute (akaze.h:30, akaze.cpp:153-201, 506-743): the integer "FAST" path --
 * uint8 image in [0,255] (pitch in bytes), the whole pipeline in int32 with 16.16 fixed-point weights
 * (namespace fastakaze, akazed.cu:2781-4367), detector threshold fixed at 65 (akaze.cpp:559).  Same
 * output contract as hak_detect_and_compute; `response` holds the integer determinant as a float. */
int hak_fast_detect_and_compute(hak_ctx* ctx, const unsigned char* d_image, int pitch,
                                hak_point* d_points, int max_pts, int* num_pts,
                                hak_point* h_points, int desc);
int hak_fast_detect_and_compute_batch(hak_ctx* ctx, const unsigned char* d_images, long image_stride, int pitch,
                                      int nimg, hak_point* d_points, int* d_num_pts, int desc);

/* ---- one PAIR per call: detectAndCompute on both images + cuMatch of the pair (main.cpp:201-209's three synchronous calls) as ONE
 * launch sequence and ONE synchronisation.  Build-side addition for callers that are bound by launches, not by bytes (a single
 * 1080p pair: 0.94 ms through the three calls).  The context must have been created with batch >= 2.  d_points1 / d_points2
 * (device, max_pts1 / max_pts2 records), h_points1 / h_points2 (host or NULL; pinned host arrays are written by the launch
 * sequence itself) and the counts are filled exactly as two hak_detect_and_compute calls followed by hak_match(ctx, 1, 2) would;
 * each image keeps its own clamp min(max_pts_i, the context's max_pts) (setMaxNumPoints(result.max_pts), akaze.cpp:246, 451), and
 * the matcher sees the clamped sets.  match = 0 skips the matcher. */
int hak_detect_and_compute_pair(hak_ctx* ctx, const float* d_image1, const float* d_image2, int pitch,
                                hak_point* d_points1, hak_point* d_points2, int max_pts1, int max_pts2,
                                int* num_pts1, int* num_pts2, hak_point* h_points1, hak_point* h_points2, int desc, int match);

/* ---- cuMatch (akaze.h:14; ctx may be NULL = default stream, akaze.cpp:55-64, akazed.cu:2144-2241): 1-NN
 * Hamming, accepted iff dist < 96 and the minimum is attained in exactly one
 * of the 16 residue classes j mod 16.  Fills match/distance/match_x/match_y
 * of d_pts1; copies those 16 bytes per point into h_pts1 when not NULL. */
int hak_match(hak_ctx* ctx, hak_point* d_pts1, int n1, const hak_point* d_pts2, int n2,
              hak_point* h_pts1);
/* batched: pair k matches (d_pts + (2k)*max_pts) against (d_pts + (2k+1)*max_pts),
 * counts read from d_num_pts[2k], d_num_pts[2k+1] on the device. Asynchronous. */
int hak_match_batch(hak_ctx* ctx, hak_point* d_points, const int* d_num_pts, int npairs);

/* ---- match post-processing.  The reference ships an unused 2-NN matcher (gMatch,
 * akazed.cu:2028-2122: best and second-best score, accept iff best < second && best < MAX_DIST); this is
 * that rule made well-defined, plus the usual symmetric cross-check and a device-side compaction:
 *   j1(i) = nearest train point of query i (smallest index among ties), d1 its distance,
 *   d2    = distance to the nearest OTHER train point (512 when n2 < 2, gMatch's initial score);
 *   accept iff d1 < max_dist  and  d1 * ratio_den < d2 * ratio_num  (1/1 = gMatch's rule)
 *          and (cross_check == 0 or the nearest query of train point j1(i) is i, ties to the smallest index).
 * Writes match/distance/match_x/match_y of pts1 like cuMatch (rejected: -1; copied to h_pts1 when given) and appends the accepted
 * matches, in ascending query order, to d_out (capacity >= n1; may be NULL); *count receives their number.
 * ctx may be NULL (scratch is then allocated per call).  max_dist <= 0 selects 96 (akazed.cu:6). */
typedef struct hak_match_pair {
    int   query, train;      /* indices into pts1 / pts2 */
    int   distance, second;  /* d1, d2 */
    float x1, y1, x2, y2;    /* refined coordinates of both ends */
} hak_match_pair;
int hak_match_knn2(hak_ctx* ctx, hak_point* d_pts1, int n1, const hak_point* d_pts2, int n2,
                   int ratio_num, int ratio_den, int cross_check, int max_dist, hak_point* h_pts1,
                   hak_match_pair* d_out, int* count, hak_match_pair* h_out);
/* batched over the pairs of a detect batch (layout as hak_match_batch): pair k's accepted matches go to
 * d_out + k*max_pts, their number to d_counts[k].  Asynchronous on the context's stream. */
int hak_match_knn2_batch(hak_ctx* ctx, hak_point* d_points, const int* d_num_pts, int npairs,
                         int ratio_num, int ratio_den, int cross_check, int max_dist,
                         hak_match_pair* d_out, int* d_counts);

/* ---- memory helpers: initAkazeData/freeAkazeData (akaze.cpp:26-52) and the
 * image upload of main.cpp:172-188 */
int hak_points_alloc(hak_point** d_points, int count);
int hak_points_free(hak_point* d_points);
int hak_image_alloc(float** d_image, int w, int h, int* pitch);   /* pitch = iAlignUp(w,128), cuda_utils.h:160 */
int hak_image_upload(float* d_image, int pitch, const float* h_image, int w, int h);
int hak_image_free(float* d_image);
/* uint8 -> float32 in [0,1] on the device, exactly main.cpp:149 (convertTo(CV_32FC1, 1.0/255.0)):
 * dst = (float)(src * (1.0 / 255.0)).  nimg images, strides in elements; runs on the context's stream
 * (ctx may be NULL = default stream).  Quarters the H2D traffic of the float upload at main.cpp:187-188. */
int hak_ingest_u8(hak_ctx* ctx, const unsigned char* d_src, long src_stride, int src_pitch,
                  float* d_dst, long dst_stride, int dst_pitch, int w, int h, int nimg);
/* pinned host memory + the batched counterpart of the D2H copies at akaze.cpp:134-139 / 60-62:
 * counts first (one sync), then the first h_num_pts[i] records of every image, asynchronously,
 * then a final sync.  h_points is [nimg][max_pts]. */
int hak_host_alloc(void** p, long bytes);
int hak_host_free(void* p);
int hak_download_batch(hak_ctx* ctx, const hak_point* d_points, const int* d_num_pts, int nimg,
                       hak_point* h_points, int* h_num_pts);
int hak_memcpy_d2h(void* dst, const void* src, long bytes);
int hak_memcpy_h2d(void* dst, const void* src, long bytes);

/* ---- host-side schedule (pure CPU, usable without a GPU).
 * fed.cpp:41-119 fed_tau_by_process_time; returns n, writes tau[0..n). */
int hak_fed_tau(float T, int M, float tau_max, int reordering, float* tau, int cap);
/* akazed.cu:2298-2333 createGaussKernel */
void hak_gauss_taps(float var, int radius, float* taps);
/* akazed.cu:65-159 setCompareIndices (486 pairs, arrays of >= 488 ints) */
void hak_compare_indices(int* idx1, int* idx2);
/* The MLDB kernel's per-lane sample plan for one descriptor_pattern_size (akazed.cu:1905-1955 restated per (lane, turn): sample
 * i = lane + 64 * turn of the (winsize x winsize) window, its offset from the window centre and its accumulator row in the 2x2 /
 * 3x3 / 4x4 grid).  pos / cell: 7 * 64 words each, [turn * 64 + lane], or NULL.  pos: bits 0..7 x - size2 (signed), 8..15
 * y - size2 (signed), bit 16 the sample exists.  cell: byte g = row of grid g (0x7F none) | 0x80 when the lane's previous sample
 * went to the same row; bit 24 + g: the lane's last sample in that row.  Returns 1 when the planned kernel serves this size (at
 * most 7 samples per lane and no lane returning to a row it has left), 0 when the generic kernel does. */
int hak_describe_plan_query(int pattern_size, unsigned int* pos, unsigned int* cell);
/* schedule the context was built with: per (octave, sublevel) the number of FED
 * steps, sigma_size, size, border; returns effective number of octaves. */
int hak_query_schedule(const hak_ctx* ctx, int* nsteps, int* sigma_size, float* sizes, float* borders);
int hak_query_geometry(const hak_ctx* ctx, int* whp /* 3 ints per octave */);

/* ---- instrumentation (not on the hot path): byte accounting and per-class timing of the launch sequence.
 * The stage operators, plane introspection and bandwidth probes the tests use are a separate ABI:
 * include/hipakaze_test.h -> libhipakaze_test.so. */
/* algorithmic-byte accounting of the last detect call on this context */
typedef struct hak_traffic {
    double fed_px_steps;     /* sum over FED steps of pixels updated, per image */
    double fed_bytes;        /* 12 B x fed_px_steps, + 16 B/px (low-pass 8 + conductivity 8) for every
                                sublevel whose low-pass and conductivity run inside its first FED launch (k_fed_sf), + the decimation /
                                low-pass / conductivity bytes of every octave head that does the same */
    double all_stage_bytes;  /* all-stage compulsory bytes per image, keypoint part for npts_hint points */
    int fed_launches;        /* FED kernel launches per batch */
    /* compulsory HBM bytes per image of each kernel class AS BUILT (after fusion): what its launches must move even with
       perfect reuse inside a launch -- the numerator of the per-class roofline fractions in bench.py */
    double fed_fused_bytes;  /* FED launches as enqueued by the last detect call: read L (+ g), write L' (+ smooth, + g) */
    double hessian_bytes;    /* 12 B/px per level: read smooth, write the interleaved {Lx, Ly} plane (the determinant is not stored) */
    double prologue_bytes;   /* 16 B/px of octave 0: read image, write Lt(0,0) + gradient plane, re-read it for the histogram */
    double describe_bytes;   /* (872 + 5292) B sampled per keypoint (orientation + MLDB) x npts_hint */
    double nms_bytes;        /* 104 B record per keypoint x npts_hint */
} hak_traffic;
int hak_query_traffic(const hak_ctx* ctx, int npts_hint, hak_traffic* out);

/* ---- per-kernel-class timing with HIP events on the context's stream.
 * When enabled, every launch of the class is bracketed by an event pair; read
 * back accumulated milliseconds and launch count after hak_sync(). */
enum { HAK_PROF_FED = 0, HAK_PROF_LOWPASS, HAK_PROF_FLOW, HAK_PROF_HESSIAN, HAK_PROF_CONTRAST,
       HAK_PROF_DOWN, HAK_PROF_EXTREMA, HAK_PROF_NMS, HAK_PROF_DESCRIBE, HAK_PROF_MATCH, HAK_PROF_COUNT };
int hak_prof_enable(hak_ctx* ctx, int on);
int hak_prof_read(hak_ctx* ctx, int klass, double* total_ms, int* launches);
int hak_prof_reset(hak_ctx* ctx);

#ifdef __cplusplus
}
#endif
#endif /* HIPAKAZE_H */
