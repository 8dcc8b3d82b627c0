// hak_test_api.hip -- the TEST ABI (include/hipakaze_test.h, libhipakaze_test.so): plane / kcontrast introspection, single-stage
// operators, detector-tail and descriptor stages on hand-made inputs, bandwidth probes.  Everything here drives the launchers
// of libhipakaze.so (the kernels of the launch sequence); nothing of it is part of the product ABI.
#include "hak_ctx.h"
#include "../../include/hipakaze_test.h"
#include <cmath>
#include <cstdio>
#include <cstring>

extern "C" int hak_debug_plane(hak_ctx* c, int img, int kind, int o, int s, float* h_dst)
{
    if (!c || img < 0 || img >= c->cfg.batch || o < 0 || o >= c->L.noct || s < 0 || s >= c->L.ms) return fail("bad plane");
    const HakLayout& L = c->L;
    const HakOct oc = L.oct[o];
    HIP_TRY(hipStreamSynchronize(c->stream));
    const float* arena = c->arena + (long)img * L.arena;
    if (kind == HAK_PLANE_LT) {
        HIP_TRY(hipMemcpy2D(h_dst, sizeof(float) * oc.w, arena + L.lt(o, s), sizeof(float) * oc.p, sizeof(float) * oc.w, oc.h, hipMemcpyDeviceToHost));
        return 0;
    }
    // Lx / Ly live interleaved; the determinant is not stored at all (HakLayout): both are produced here, for the tests, by
    // the unfused kernels from the stored derivative plane
    float* tmp = nullptr;
    HIP_TRY(hipMalloc((void**)&tmp, sizeof(float) * 2 * (size_t)oc.plane));
    int rc = 0;
    const float* src = tmp;
    if (kind == HAK_PLANE_DET) {
        const int step = c->plan[(size_t)o * L.ms + s].sigma_size;
        if (c->last_fast) hakf_launch_det(nullptr, reinterpret_cast<const int*>(arena + L.dxy(o, s)), reinterpret_cast<int*>(tmp), 0, oc.w, oc.h, oc.p, 1, step);
        else hak_launch_hessian(nullptr, arena + L.dxy(o, s), tmp, 0, oc.w, oc.h, oc.p, 1, step);
    } else {
        hak_launch_deinterleave(nullptr, arena + L.dxy(o, s), tmp, tmp + oc.plane, oc.w, oc.h, oc.p);
        if (kind == HAK_PLANE_LY) src = tmp + oc.plane;
    }
    if (hipDeviceSynchronize() != hipSuccess) rc = fail("debug plane kernel");
    if (!rc && hipMemcpy2D(h_dst, sizeof(float) * oc.w, src, sizeof(float) * oc.p, sizeof(float) * oc.w, oc.h, hipMemcpyDeviceToHost) != hipSuccess)
        rc = fail("debug plane copy");
    (void)hipFree(tmp);
    return rc;
}

extern "C" int hak_debug_kcontrast(hak_ctx* c, int img, float* kc)
{
    if (!c || img < 0 || img >= c->cfg.batch) return fail("bad image index");
    HakImgState s;
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipMemcpy(&s, c->state + img, sizeof(s), hipMemcpyDeviceToHost));
    *kc = s.kcontrast[0];
    return 0;
}

// ------------------------------------------------ single-stage test operators
extern "C" int hak_op_lowpass(const float* s, float* d, int w, int h, int p, float var, int radius)
{
    if (radius < 1 || radius > 5) return fail("radius must be 1..5");
    float taps[8];
    hak_gauss_taps(var, radius, taps);
    hak_launch_lowpass(nullptr, s, 0, p, d, 0, w, h, p, 1, taps, radius);
    HIP_TRY(hipDeviceSynchronize());
    return 0;
}

extern "C" int hak_op_down_smooth(const float* s, float* d, float* sm, int sw, int sh, int sp, int dw, int dh, int dp)
{
    float taps[8];
    hak_gauss_taps(1.f, 2, taps);
    HakOct so{sw, sh, sp, (long)sh * sp}, dd{dw, dh, dp, (long)dh * dp};
    hak_launch_down_smooth(nullptr, s, d, sm, 0, so, dd, 1, taps);
    HIP_TRY(hipDeviceSynchronize());
    return 0;
}

extern "C" int hak_op_kcontrast(const float* smooth, int w, int h, int p, float per, float* kc, float* hmax, int* hist)
{
    HakImgState* st = nullptr;
    HIP_TRY(hipMalloc((void**)&st, sizeof(HakImgState)));
    hak_launch_reset_state(nullptr, st, 1);
    hak_launch_contrast(nullptr, smooth, 0, w, h, p, 1, st, per, 1);
    HakImgState hs;
    HIP_TRY(hipMemcpy(&hs, st, sizeof(hs), hipMemcpyDeviceToHost));
    HIP_TRY(hipFree(st));
    if (kc) *kc = hs.kcontrast[0];
    if (hmax) memcpy(hmax, &hs.hmax_bits, 4);
    // the reference's h_hist: the w x h pixels plus what the threads beside / below the image add to bin 0 (hak_hist_extra0; the
    // kernels carry that constant into `thresh` instead of adding it to the device histogram)
    if (hist) { memcpy(hist, hs.hist, sizeof(hs.hist)); hist[0] += hak_hist_extra0(w, h); }
    return 0;
}

extern "C" int hak_op_flow(const float* s, float* d, int w, int h, int p, int diffusivity, float kcontrast)
{
    float ikc = 1.f / (kcontrast * kcontrast);                                    // akazed.cu:2493
    hak_launch_flow(nullptr, s, d, 0, w, h, p, 1, diffusivity, nullptr, 0, ikc);
    HIP_TRY(hipDeviceSynchronize());
    return 0;
}

extern "C" int hak_op_rcp_check(unsigned lo_bits, unsigned hi_bits, unsigned long long* mismatches)
{
    if (!mismatches) return fail("null argument");
    unsigned long long* d = nullptr;
    HIP_TRY(hipMalloc((void**)&d, sizeof(unsigned long long)));
    HIP_TRY(hipMemset(d, 0, sizeof(unsigned long long)));
    int rc = hak_launch_rcp_check(lo_bits, hi_bits, d);
    if (!rc && hipDeviceSynchronize() != hipSuccess) rc = fail("rcp check kernel");
    if (!rc && hipMemcpy(mismatches, d, sizeof(unsigned long long), hipMemcpyDeviceToHost) != hipSuccess) rc = fail("rcp check download");
    (void)hipFree(d);
    return rc;
}

extern "C" int hak_op_smooth_flow(const float* s, float* sm, float* fl, int w, int h, int p, int diffusivity, float kcontrast)
{
    float taps[8];
    hak_gauss_taps(1.f, 2, taps);
    const float ikc = 1.f / (kcontrast * kcontrast);                              // akazed.cu:2493
    hak_launch_smooth_flow(nullptr, s, sm, fl, 0, w, h, p, 1, taps, diffusivity, nullptr, 0, ikc);
    HIP_TRY(hipDeviceSynchronize());
    return 0;
}

extern "C" int hak_op_nld_steps(const float* src, const float* flow, float* dst, float* tmp, int w, int h, int p,
                                const float* tau, int nsteps)
{
    if (nsteps < 1) return fail("nsteps < 1");
    if (p % 4) return fail("pitch must be a multiple of 4");
    int fuse = 4;
    if (const char* e = getenv("HAK_FED_MAX_FUSE")) fuse = atoi(e);
    const int G = hak_fed_groups(nsteps, fuse, w);
    const float* s = src;
    int done = 0;
    for (int g = 0; g < G; g++) {
        const int ns = hak_fed_group_size(nsteps, G, g);
        float* d = ((G - g) % 2 == 1) ? dst : tmp;
        hak_launch_fed_group(nullptr, s, flow, d, 0, w, h, p, 1, tau + done, ns);
        done += ns;
        s = d;
    }
    HIP_TRY(hipDeviceSynchronize());
    return 0;
}

extern "C" int hak_op_hessian(const float* s, float* lx, float* ly, float* det, int w, int h, int p, int step)
{
    // the kernels write the derivatives interleaved (HakLayout); the test interface keeps the reference's three planes
    float* dxy = nullptr;
    HIP_TRY(hipMalloc((void**)&dxy, sizeof(float) * 2 * (size_t)h * p));
    hak_launch_hessian_level(nullptr, s, dxy, det, true, 0, w, h, p, 1, step, nullptr, nullptr, nullptr, 0, 0, 0.f);
    hak_launch_deinterleave(nullptr, dxy, lx, ly, w, h, p);
    const hipError_t e = hipDeviceSynchronize();
    (void)hipFree(dxy);
    if (e != hipSuccess) return fail(std::string("hak_op_hessian: ") + hipGetErrorString(e));
    return 0;
}

// ---- detector tail / descriptors on hand-made inputs (include/hipakaze.h; tests/test_gpu_literal.py)
static HakBatch tail_batch(hak_ctx* c)
{
    return HakBatch{c->arena, c->L.arena, 1, c->state, c->maps, c->L.oct[0].plane, c->bitmap, c->rowcount, c->cand, c->cand_cap, &c->knobs};
}
static int tail_level_ok(hak_ctx* c, int o, int s, const void* h)
{
    if (!c || !h) return fail("null argument");
    if (o < 0 || o >= c->L.noct || s < 0 || s >= c->L.ms) return fail("bad level");
    return 0;
}

extern "C" int hak_debug_set_plane(hak_ctx* c, int img, int kind, int o, int s, const float* h_src)
{
    if (tail_level_ok(c, o, s, h_src)) return 1;
    if (img < 0 || img >= c->cfg.batch) return fail("bad image index");
    const HakLayout& L = c->L;
    const HakOct oc = L.oct[o];
    HIP_TRY(hipStreamSynchronize(c->stream));
    float* arena = c->arena + (long)img * L.arena;
    if (kind == HAK_PLANE_LT) {
        HIP_TRY(hipMemcpy2D(arena + L.lt(o, s), sizeof(float) * oc.p, h_src, sizeof(float) * oc.w, sizeof(float) * oc.w, oc.h, hipMemcpyHostToDevice));
        return 0;
    }
    if (kind != HAK_PLANE_LX && kind != HAK_PLANE_LY) return fail("only the Lt, Lx and Ly planes can be set");
    // element (y, x) of the interleaved plane = {Lx, Ly} at 2 * (y * p + x): a strided 2-D copy of single floats
    float* dst = arena + L.dxy(o, s) + (kind == HAK_PLANE_LY ? 1 : 0);
    for (int y = 0; y < oc.h; y++)
        HIP_TRY(hipMemcpy2D(dst + 2L * y * oc.p, 2 * sizeof(float), h_src + (long)y * oc.w, sizeof(float), sizeof(float), oc.w, hipMemcpyHostToDevice));
    return 0;
}

extern "C" int hak_op_tail_begin(hak_ctx* c)
{
    if (!c) return fail("null context");
    c->last_fast = false;
    c->sync_stream = c->stream;
    maps_guard_begin(c);                                            // (stays set until hak_op_tail_finish has cleaned the map)
    hak_launch_reset_state(c->stream, c->state, 1);
    HIP_TRY(hipStreamSynchronize(c->stream));
    return 0;
}

extern "C" int hak_op_tail_level(hak_ctx* c, int o, int s, const float* h_src)
{
    if (tail_level_ok(c, o, s, h_src)) return 1;
    const HakLayout& L = c->L;
    const HakOct oc = L.oct[o];
    float* A = c->arena;
    float* smooth = A + L.smooth_off[o];
    HIP_TRY(hipMemcpy2D(smooth, sizeof(float) * oc.p, h_src, sizeof(float) * oc.w, sizeof(float) * oc.w, oc.h, hipMemcpyHostToDevice));
    HakBatch b = tail_batch(c);
    const int step = c->plan[(size_t)o * L.ms + s].sigma_size;
    if (!hak_launch_hessian_level(c->stream, smooth, A + L.dxy(o, s), A + L.flow_off[o], false, L.arena, oc.w, oc.h, oc.p, 1, step, &b, &L,
                                  &c->htab, o, s, c->cfg.dthreshold))
        hak_launch_extrema_level(c->stream, b, L, c->dtab, o, s, c->cfg.dthreshold, L.flow_off[o]);
    if (hipGetLastError() != hipSuccess) return fail("tail level launch failed");
    HIP_TRY(hipStreamSynchronize(c->stream));
    return 0;
}

extern "C" int hak_op_tail_det_level(hak_ctx* c, int o, int s, const float* h_det)
{
    if (tail_level_ok(c, o, s, h_det)) return 1;
    const HakLayout& L = c->L;
    const HakOct oc = L.oct[o];
    HIP_TRY(hipMemcpy2D(c->arena + L.flow_off[o], sizeof(float) * oc.p, h_det, sizeof(float) * oc.w, sizeof(float) * oc.w, oc.h,
                        hipMemcpyHostToDevice));
    hak_launch_extrema_level(c->stream, tail_batch(c), L, c->dtab, o, s, c->cfg.dthreshold, L.flow_off[o]);
    if (hipGetLastError() != hipSuccess) return fail("extrema launch failed");
    HIP_TRY(hipStreamSynchronize(c->stream));
    return 0;
}

extern "C" int hak_op_tail_seed(hak_ctx* c, const unsigned int* h_resp_bits, const int* h_layer)
{
    if (!c || !h_resp_bits || !h_layer) return fail("null argument");
    const size_t n = (size_t)c->L.oct[0].w * c->L.oct[0].h;
    unsigned* d_r = nullptr;
    int* d_l = nullptr;
    HIP_TRY(hipMalloc((void**)&d_r, sizeof(unsigned) * n));
    int rc = 0;
    if (hipMalloc((void**)&d_l, sizeof(int) * n) != hipSuccess) rc = fail("seed scratch");
    if (!rc && (hipMemcpy(d_r, h_resp_bits, sizeof(unsigned) * n, hipMemcpyHostToDevice) != hipSuccess ||
                hipMemcpy(d_l, h_layer, sizeof(int) * n, hipMemcpyHostToDevice) != hipSuccess)) rc = fail("seed upload");
    if (!rc) {
        hak_launch_seed_maps(c->stream, tail_batch(c), c->L, d_r, d_l);
        if (hipGetLastError() != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess) rc = fail("seed kernel");
    }
    (void)hipFree(d_r); (void)hipFree(d_l);
    return rc;
}

extern "C" int hak_op_tail_finish(hak_ctx* c, hak_point* d_points, int max_pts, int refine, int fast, int* num_pts)
{
    if (!c || !d_points || !num_pts || max_pts < 1) return fail("bad argument");
    hak_launch_nms_emit(c->stream, tail_batch(c), c->L, c->dtab, c->psz, d_points, max_pts, c->d_num, fast ? 1 : 0, refine ? 1 : 0);
    hak_launch_clear_maps(c->stream, tail_batch(c), c->L);
    int rc = hipGetLastError() != hipSuccess ? fail("tail finish launch failed") : 0;
    if (!rc && hipMemcpyAsync(c->h_num, c->d_num, sizeof(int), hipMemcpyDeviceToHost, c->stream) != hipSuccess) rc = fail("count download");
    if (!rc && hipStreamSynchronize(c->stream) != hipSuccess) rc = fail("sync");
    if (!rc) *num_pts = c->h_num[0];
    return maps_guard_end(c, rc);
}

extern "C" int hak_op_orient_describe(hak_ctx* c, hak_point* d_points, int n, int desc)
{
    if (!c || !d_points || n < 1) return fail("bad argument");
    HIP_TRY(hipMemcpy(&c->state[0].num_pts, &n, sizeof(int), hipMemcpyHostToDevice));
    // desc == 2: the MLDB kernel alone, rotated by the angles the records already hold
    hak_launch_describe(c->stream, tail_batch(c), c->L, c->dtab, d_points, n, c->cfg.descriptor_pattern_size, c->cfg.upright, desc ? 1 : 0,
                        c->htab.dsc_plan_ok, desc == 2 ? 0 : 1);
    if (hipGetLastError() != hipSuccess) return fail("describe launch failed");
    HIP_TRY(hipStreamSynchronize(c->stream));
    return 0;
}

extern "C" int hak_op_copy_probe(long bytes, int iters, double* gbytes_per_s)
{
    if (bytes < 16 || iters < 1 || !gbytes_per_s) return fail("bad probe argument");
    if (hak_device_count() == 0) return fail("no HIP device: libhipakaze has no CPU fallback");
    bytes &= ~15L;
    double ms = 0;
    if (hak_launch_copy_probe(bytes, iters, &ms) || ms <= 0) return fail("copy probe failed");
    *gbytes_per_s = 2.0 * (double)bytes / (ms * 1e-3) / 1e9;                      // read + write
    return 0;
}

extern "C" int hak_op_copy_probe_shapes(long bytes, int iters, double* gbytes_per_s, int n)
{
    if (bytes < 16 || iters < 1 || !gbytes_per_s || n < HAK_COPY_SHAPES) return fail("bad probe argument");
    if (hak_device_count() == 0) return fail("no HIP device: libhipakaze has no CPU fallback");
    bytes &= ~15L;
    double best = 0, ms[HAK_COPY_SHAPES] = {};
    if (hak_launch_copy_probe(bytes, iters, &best, ms)) return fail("copy probe failed");
    for (int i = 0; i < HAK_COPY_SHAPES; i++) {
        const double moved = i < HAK_COPY_SHAPES - 2 ? 2.0 * (double)bytes : (double)bytes;     // copy: read + write
        gbytes_per_s[i] = ms[i] > 0 ? moved / (ms[i] * 1e-3) / 1e9 : 0.0;
    }
    return HAK_COPY_SHAPES;
}

extern "C" int hak_op_stream_probe(int w, int h, int nimg, int nwrite, int warm_rows, int iters, double* ms_per_launch, double* gbytes_per_s)
{
    if (!ms_per_launch || !gbytes_per_s) return fail("null argument");
    if (hak_device_count() == 0) return fail("no HIP device: libhipakaze has no CPU fallback");
    double ms = 0, bytes = 0;
    if (hak_launch_stream_probe(w, h, nimg, nwrite, warm_rows, iters, &ms, &bytes) || ms <= 0) return fail("stream probe failed (w % 4, sizes, memory?)");
    *ms_per_launch = ms;
    *gbytes_per_s = bytes / (ms * 1e-3) / 1e9;
    return 0;
}

extern "C" int hak_op_hess_probe(int w, int h, int nimg, int step, int iters, double* ms_per_launch, double* gbytes_per_s)
{
    if (!ms_per_launch || !gbytes_per_s) return fail("null argument");
    if (hak_device_count() == 0) return fail("no HIP device: libhipakaze has no CPU fallback");
    double ms = 0, bytes = 0;
    if (hak_launch_hess_probe(w, h, nimg, step, iters, &ms, &bytes) || ms <= 0) return fail("hessian probe failed (w % 4, step 1..4, sizes, memory?)");
    *ms_per_launch = ms;
    *gbytes_per_s = bytes / (ms * 1e-3) / 1e9;
    return 0;
}

extern "C" int hak_op_gather_probe(long bytes, int blocks, int per_lane, int iters, double* ms_per_launch)
{
    if (bytes < 4096 || blocks < 1 || per_lane < 4 || iters < 1 || !ms_per_launch) return fail("bad probe argument");
    if (hak_device_count() == 0) return fail("no HIP device: libhipakaze has no CPU fallback");
    if (hak_launch_gather_probe(bytes & ~127L, blocks, per_lane & ~3, iters, ms_per_launch)) return fail("gather probe failed");
    return 0;
}

// the sliced matcher's per-slice summaries filled with `byte` on the context's stream (tests/stress_handoff.py: a stale read of a
// summary is only visible when the scratch does not already hold the same launch's values from the call before)
extern "C" int hak_debug_fill_match_scratch(hak_ctx* c, int byte)
{
    if (!c) return fail("null context");
    if (c->msc.part && c->msc.part_cap > 0) HIP_TRY(hipMemsetAsync(c->msc.part, byte, sizeof(uint2) * (size_t)c->msc.part_cap, c->stream));
    return 0;
}
