/*
 * hipakaze_test.h -- the TEST ABI of the HIP AKAZE path: libhipakaze_test.so.
 *
 * NOT part of the product: a maintainer who binds the reference's interface needs include/hipakaze.h (libhipakaze.so) or
 * include/akaze.h only.  This library links against libhipakaze.so and drives the SAME launchers and kernels the launch
 * sequence uses, one stage at a time, so that tests/ can compare every stage with the oracle and with hand-derived fixtures:
 *   hak_debug_*   planes / contrast factor of the last call on a context, writing a plane
 *   hak_op_*      single-stage operators (one h*() wrapper of akazed.cu each), the detector tail and the descriptor stages on
 *                 hand-made inputs, the conductivity's reciprocal self-check, bandwidth probes of the box
 */
#ifndef HIPAKAZE_TEST_H
#define HIPAKAZE_TEST_H
#include "hipakaze.h"
#ifdef __cplusplus
extern "C" {
#endif

/* ---- introspection of the last call on a context */
enum { HAK_PLANE_LT = 0, HAK_PLANE_DET = 1, HAK_PLANE_LX = 2, HAK_PLANE_LY = 3 };
/* copy plane (kind, octave, sublevel) of batch image `img`, densely packed w x h, to host */
int hak_debug_plane(hak_ctx* ctx, int img, int kind, int octave, int sublevel, float* h_dst);
int hak_debug_kcontrast(hak_ctx* ctx, int img, float* kcontrast);

/* ---- single-stage operators on caller-provided device planes (pitch p
 * elements, dense row-major), used by the per-kernel parity tests.  Each is
 * the HIP counterpart of one h*() wrapper of akazed.cu. Synchronous. */
int hak_op_lowpass(const float* d_src, float* d_dst, int w, int h, int p, float var, int radius);         /* hLowPass 2336 */
int hak_op_down_smooth(const float* d_src, float* d_dst, float* d_smooth, int sw, int sh, int sp,
                       int dw, int dh, int dp);                                                          /* hDownWithSmooth 2389 */
int hak_op_kcontrast(const float* d_smooth, int w, int h, int p, float per, float* kcontrast,
                     float* hmax, int* hist300);                                                         /* hScharrContrast 2410 */
int hak_op_flow(const float* d_src, float* d_dst, int w, int h, int p, int diffusivity, float kcontrast); /* hFlow 2487 */
int hak_op_nld_steps(const float* d_src, const float* d_flow, float* d_dst, float* d_tmp,
                     int w, int h, int p, const float* tau, int nsteps);                                  /* hNldStep 2509, n steps */
/* self-check of the conductivity's fast reciprocal (csrc/fed_common.h hak_rcp_newton): counts the floats with bit patterns
 * in [lo_bits, hi_bits) whose 3-instruction reciprocal differs from the IEEE quotient 1.0f / d.  Must be 0 on [1, 2^64). */
int hak_op_rcp_check(unsigned lo_bits, unsigned hi_bits, unsigned long long* mismatches);
int hak_op_smooth_flow(const float* d_src, float* d_smooth, float* d_flow, int w, int h, int p,
                       int diffusivity, float kcontrast);                                               /* hLowPass(var 1) + hFlow, akaze.cpp:403-404 */
int hak_op_hessian(const float* d_src, float* d_lx, float* d_ly, float* d_det, int w, int h, int p, int step); /* hHessianDeterminant 2531 */

/* ---- detector-tail and descriptor stages on hand-made inputs (tests/test_gpu_literal.py: micro-fixtures whose expected
 * output is derived by hand from the cited reference statements).  They drive the SAME launchers as the launch sequence, on
 * image 0 of the context, and synchronise before returning.  A sequence is begin -> {level | det_level | seed}* -> finish.
 *   hak_debug_set_plane   write plane (HAK_PLANE_LT / LX / LY) of (octave, sublevel) from a dense w x h host array
 *   hak_op_tail_begin     reset the per-image state (candidate list, counters); the key map is already clear
 *   hak_op_tail_level     hHessianDeterminant 2531 + gCalcExtremaMap 1334 of one level from a dense host L-plane, through the
 *                         fused kernel the context would pick (knobs as in hak_create), threshold = cfg.dthreshold
 *   hak_op_tail_det_level gCalcExtremaMap 1334 alone (the stand-alone kernel of the dilation > 4 fallback) on a dense host
 *                         determinant plane
 *   hak_op_tail_seed      hand-made full-resolution maps: response words (float or int bits) and layer ids (< 0: empty)
 *   hak_op_tail_finish    gNmsRNaive 1554 (+ gRefine 1615 when `refine`) -> d_points in raster order; *num_pts = survivors
 *   hak_op_orient_describe gCalcOrient 1665 + gDescribe2 1869 on the first n records of d_points, reading the planes the
 *                         arena holds now (desc: 0 none, 1 both, 2 descriptor only with the records' own angles) */
int hak_debug_set_plane(hak_ctx* ctx, int img, int kind, int octave, int sublevel, const float* h_src);
int hak_op_tail_begin(hak_ctx* ctx);
int hak_op_tail_level(hak_ctx* ctx, int octave, int sublevel, const float* h_src);
int hak_op_tail_det_level(hak_ctx* ctx, int octave, int sublevel, const float* h_det);
int hak_op_tail_seed(hak_ctx* ctx, const unsigned int* h_response_bits, const int* h_layer);
int hak_op_tail_finish(hak_ctx* ctx, hak_point* d_points, int max_pts, int refine, int fast, int* num_pts);
int hak_op_orient_describe(hak_ctx* ctx, hak_point* d_points, int n, int desc);

/* ---- bandwidth ceilings of the box (SURVEY 8d "copy-kernel ceiling"; not on the hot path).
 * hak_op_copy_probe: float4 copy of `bytes` with the streaming kernels' access shape (16 B/lane), `iters` times per launch
 * shape (24 shapes: loads in flight, nt / plain stores, grid size); *gbytes_per_s = (read + write bytes) / average kernel
 * time of the best shape.
 * hak_op_gather_probe: `blocks` x 256 lanes each gather `per_lane` dwords from pseudo-random 128-byte lines of a
 * `bytes`-sized buffer (the descriptor's access shape), `iters` times; *ms_per_launch = average kernel time.  Used to
 * calibrate the FETCH_SIZE counter for 4-byte gathers. */
int hak_op_copy_probe(long bytes, int iters, double* gbytes_per_s);
/* every launch shape of the probe: entry (g * 6 + s * 3 + l) = grid g {8, 16, 32 blocks per CU, one pass} x stores s {nt, plain} x
 * loads in flight per lane l {2, 4, 8}; then the read-only and the write-only stream.  Returns the number of entries (26). */
int hak_op_copy_probe_shapes(long bytes, int iters, double* gbytes_per_s, int n);
int hak_op_gather_probe(long bytes, int blocks, int per_lane, int iters, double* ms_per_launch);
/* hak_op_stream_probe: the FED family's access shape with the arithmetic taken out -- `nimg` planes of w x h read once (16 B per lane
 * and row, 256-lane strips with 8-column halos, row segments as the streaming kernels cut them, `warm_rows` warm-up rows per
 * segment) and `nwrite` (1..3) planes written with nt buffer stores; *gbytes_per_s = compulsory bytes (1 + nwrite) x 4 x w x h x nimg /
 * average kernel time: the data-movement floor of k_fed_sf / k_fed_multi at that launch geometry. */
int hak_op_stream_probe(int w, int h, int nimg, int nwrite, int warm_rows, int iters, double* ms_per_launch, double* gbytes_per_s);

/* hak_op_hess_probe: the streaming Hessian's access shape with the arithmetic taken out (k_hessian_stream, dilation `step` 1..4): its strips,
 * row segments and warm-up rows, 4 B/px read, the interleaved {Lx, Ly} plane written through the same per-wave LDS turn and the same
 * two dense 16-byte nt buffer stores per lane and row, at the kernel's LDS footprint and occupancy target; *gbytes_per_s = 12 B/px
 * compulsory / average kernel time: the data-movement floor of the Hessian class at that launch geometry. */
int hak_op_hess_probe(int w, int h, int nimg, int step, int iters, double* ms_per_launch, double* gbytes_per_s);
/* hak_debug_fill_match_scratch: fills the context's per-slice match summaries (kernels_match.hip: `part`) with `byte`, on the context's
 * stream -- the hand-off stress run makes a stale summary visible with it (tests/stress_handoff.py) */
int hak_debug_fill_match_scratch(hak_ctx* ctx, int byte);

#ifdef __cplusplus
}
#endif
#endif /* HIPAKAZE_TEST_H */
