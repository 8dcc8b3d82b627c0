// kernels_detect.hip -- Hessian-determinant extrema, cross-scale map, disc NMS,
// deterministic compaction and sub-pixel refinement (gfx950, wave64).
//
//   hCalcExtremaMap/gCalcExtremaMap  akazed.cu:2563, 1334 -> k_extrema
//   hNmsR/gNmsRNaive                 akazed.cu:2611, 1554 -> k_nms_mark, k_row_scan, k_emit
//   hRefine/gRefine                  akazed.cu:2635, 1615 -> fused into k_emit
//
// The reference keeps three full-resolution maps (response, size, layer)
// updated by a racy compare-then-write and appends survivors in
// atomic arrival order (D6).  Here one u64 per pixel holds
//   key = response_bits << 32 | (0xFFFFFFFF - layer)
// updated with a single atomicMax: the larger response wins and, on a tie, the
// lower layer (= the reference's sequential "strict <" order).  Survivors are
// compacted in raster order: ballot -> bitmap + per-row counts -> scan -> emit.
#include "hak_internal.h"

__device__ __forceinline__ float key_resp(unsigned long long k) { return __uint_as_float((unsigned)(k >> 32)); }
__device__ __forceinline__ int key_layer(unsigned long long k) { return (int)(0xFFFFFFFFu - (unsigned)k); }

// ------------------------------------------------------------------ extrema
// Stand-alone per-level extrema (used only when the dilation is too large for the fused
// Hessian kernel, kernels_hessian.hip).  grid: (x tiles, y tiles, nimg)
__global__ __launch_bounds__(256) void k_extrema(const float* __restrict__ base, long stride, unsigned long long* maps,
                                                 long map_stride, unsigned long long* cand, long cand_cap,
                                                 HakImgState* state, HakLayout L, const HakTables* __restrict__ tab,
                                                 int octave, int s, float threshold, long det_off)
{
    const int img = blockIdx.z;
    const HakOct oc = L.oct[octave];
    const float* det = base + (long)img * stride + det_off;        // scratch plane filled by the unfused k_hessian
    const int layer = octave * L.ms + s;
    const float border = tab->borders[layer];
    const int psz = (int)tab->borders[octave * L.ms];               // akazed.cu:2572
    const int lane = threadIdx.x & 63;
    const int x = blockIdx.x * 64 + lane;
    const int y0 = blockIdx.y * 16 + (threadIdx.x >> 6);
    // akazed.cu:1346-1353
    const bool xok = x >= psz && x < oc.w && (int)(x - border + 0.5f) - 1 >= 0 && (int)(x + border + 0.5f) + 1 < oc.w;
    for (int y = y0; y < blockIdx.y * 16 + 16; y += 4) {
        bool hit = false;
        float v = 0.f;
        if (xok && y >= psz && y < oc.h && (int)(y - border + 0.5f) - 1 >= 0 && (int)(y + border + 0.5f) + 1 < oc.h) {
            const float* vp = det + (long)y * oc.p + x;
            const float* vp0 = vp - oc.p;
            const float* vp2 = vp + oc.p;
            v = *vp;
            hit = v > threshold && v > *vp0 && v > *vp2 && v > vp[-1] && v > vp[1] &&
                  v > vp0[-1] && v > vp0[1] && v > vp2[-1] && v > vp2[1];
        }
        const unsigned long long m = __ballot(hit);
        if (m) {
            int cbase = 0;
            if (lane == 0) cbase = atomicAdd(&state[img].ncand, __popcll(m));
            cbase = __builtin_amdgcn_readfirstlane(cbase);
            if (hit) {
                const int fx = x << octave, fy = y << octave;
                unsigned long long key = ((unsigned long long)__float_as_uint(v) << 32) | (0xFFFFFFFFu - (unsigned)layer);
                atomicMax(&maps[(long)img * map_stride + (long)fy * L.oct[0].p + fx], key);
                const long slot = cbase + __popcll(m & ((1ull << lane) - 1ull));
                if (slot < cand_cap)
                    cand[(long)img * cand_cap + slot] = ((unsigned long long)layer << 32) | ((unsigned)fy << 16) | (unsigned)fx;
            }
        }
    }
}

// ---------------------------------------------------------------- NMS: mark
// Candidate-driven disc NMS (akazed.cu:1554-1613): sixteen lanes per candidate of the image's list.
// A candidate proceeds only if it still owns its pixel of the key map (several levels can hit the
// same full-resolution pixel; exactly one key wins).  Survivors set their bit in the bitmap and
// bump the per-row count; order is restored by k_row_scan + k_emit.
__global__ __launch_bounds__(256) void k_nms_cand(const unsigned long long* __restrict__ maps, long map_stride,
                                                  const unsigned long long* __restrict__ cand, long cand_cap,
                                                  const HakImgState* __restrict__ state,
                                                  const HakTables* __restrict__ tab, int psz, int w, int h, int p,
                                                  unsigned long long* bitmap, int words_per_row, int* rowcount)
{
    const int img = blockIdx.y;
    const unsigned long long* map = maps + (long)img * map_stride;
    long n = state[img].ncand;
    n = n < cand_cap ? n : cand_cap;
    // Sixteen lanes per candidate: lane c looks at column x + c - isz of every row of the disc, so one load instruction covers a
    // 36..72-byte piece of one row per candidate (1-2 lines) instead of 64 unrelated lines (one thread per candidate spent the
    // kernel in the texture addresser: 81 line look-ups per candidate).  The group's verdict is the OR over its lanes (ballot).
    const int grp = threadIdx.x >> 4, c = threadIdx.x & 15;
    const int gsh = (threadIdx.x & 48);                             // the group's first bit in the wave's ballot
    for (long i0 = (long)blockIdx.x * 16; i0 < n; i0 += (long)gridDim.x * 16) {        // (block-uniform)
        const long i = i0 + grp;
        bool live = i < n;
        const unsigned long long e = live ? cand[(long)img * cand_cap + i] : 0ull;
        const int x = (int)(e & 0xFFFFu), y = (int)((e >> 16) & 0xFFFFu), layer = (int)(e >> 32);
        live = live && x >= psz && x + psz < w && y >= psz && y + psz < h;
        const float fsz = live ? tab->sizes[layer] : 0.f;
        const int isz = (int)(fsz + 0.5f);
        const int sqsz = (int)(fsz * fsz);
        // the candidate's own key (does it still own its pixel?) and the disc's rows are fetched together: one memory round trip
        // per candidate instead of one for the key and one per row
        const unsigned long long kc = live ? map[(long)y * p + x] : 0ull;
        bool hit = false;
        unsigned rc;
        if (__ballot(isz > 4) == 0ull) {                            // (wave-uniform) discs of radius <= 4: nine rows, one column per lane
            const int dj = c - isz;
            unsigned rn[9];
#pragma unroll
            for (int k = 0; k < 9; k++) {
                const int di = k - 4;
                const bool in = live && di >= -isz && di <= isz && dj <= isz && !(di == 0 && dj == 0) && di * di + dj * dj < sqsz;
                // akazed.cu:1581-1593: the reference's read cursor is not advanced by the `continue` of the centre, so on
                // the centre row every dj > 0 looks at column x + dj - 1 (dj == 1 at the centre itself) while the disc test
                // and the tie rule keep using dj -- followed literally
                const int col = dj - (int)(di == 0 && dj > 0);
                rn[k] = in ? (reinterpret_cast<const unsigned*>(map + (long)(y + di) * p + x) + 1)[2 * col] : 0u;   // high words
            }
            rc = (unsigned)(kc >> 32);              // response word: unsigned order == order of positive floats / ints
#pragma unroll
            for (int k = 0; k < 9; k++) {
                const int di = k - 4;
                const bool in = live && di >= -isz && di <= isz && dj <= isz && !(di == 0 && dj == 0) && di * di + dj * dj < sqsz;
                hit = hit || (in && (rn[k] > rc || (rn[k] == rc && di <= 0 && dj <= 0)));
            }
        } else {                                                    // any radius: row by row
            rc = (unsigned)(kc >> 32);
            for (int di = -isz; di <= isz; di++) {
                const unsigned* row = reinterpret_cast<const unsigned*>(map + (long)(y + di) * p + x) + 1;   // high words: row[2 * dj]
                for (int dj = c - isz; dj <= isz; dj += 16) {
                    const bool in = live && !(di == 0 && dj == 0) && di * di + dj * dj < sqsz;
                    const int col = dj - (int)(di == 0 && dj > 0);
                    const unsigned rn = in ? row[2 * col] : 0u;
                    hit = hit || (in && (rn > rc || (rn == rc && di <= 0 && dj <= 0)));
                }
            }
        }
        live = live && key_layer(kc) == layer;                      // (else another level won this pixel)
        const unsigned long long m = __ballot(hit);                 // (all lanes are back together here)
        const bool to_nms = ((m >> gsh) & 0xFFFFull) != 0ull;
        if (live && !to_nms && c == 0) {
            atomicOr(&bitmap[((long)img * h + y) * words_per_row + (x >> 6)], 1ull << (x & 63));
            atomicAdd(&rowcount[(long)img * h + y], 1);
        }
    }
}

// exclusive scan of the per-row survivor counts (one block per image)
__global__ __launch_bounds__(256) void k_row_scan(int* rowcount, int h, HakImgState* state, int max_pts, int cap0, int cap1, int* num_out)
{
    __shared__ int part[256];
    int* rc = rowcount + (long)blockIdx.x * h;
    const int per = (h + 255) / 256;
    const int beg = threadIdx.x * per, end = min(beg + per, h);
    int sum = 0;
    for (int i = beg; i < end; i++) sum += rc[i];
    part[threadIdx.x] = sum;
    __syncthreads();
    if (threadIdx.x == 0) {
        int run = 0;
        for (int i = 0; i < 256; i++) { int t = part[i]; part[i] = run; run += t; }
        state[blockIdx.x].total_pts = run;
        // the clamp of this image: the call's max_pts, or the image's own capacity in a pair call (never above max_pts)
        const int cap = cap0 > 0 ? (blockIdx.x == 0 ? cap0 : cap1) : max_pts;
        int n = run < cap ? run : cap;
        state[blockIdx.x].num_pts = n;
        if (num_out) num_out[blockIdx.x] = n;
    }
    __syncthreads();
    int run = part[threadIdx.x];
    for (int i = beg; i < end; i++) { int t = rc[i]; rc[i] = run; run += t; }
}

// emit survivors in raster order; one wave per row
__global__ __launch_bounds__(256) void k_emit(const float* __restrict__ base, long stride,
                                              const unsigned long long* __restrict__ maps, long map_stride,
                                              HakLayout L, const HakTables* __restrict__ tab,
                                              const unsigned long long* __restrict__ bitmap, int words_per_row,
                                              const int* __restrict__ rowstart, hak_point* points, int max_pts, int fast)
{
    const int img = blockIdx.z;
    const int h = L.oct[0].h, p0 = L.oct[0].p;
    const int y = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (y >= h) return;
    const unsigned long long* map = maps + (long)img * map_stride;
    hak_point* pts = points + (long)img * max_pts;
    int row_base = rowstart[(long)img * h + y];
    for (int w0 = 0; w0 < words_per_row; w0 += 64) {
        unsigned long long word = (w0 + lane < words_per_row) ? bitmap[((long)img * h + y) * words_per_row + w0 + lane] : 0ull;
        int cnt = __popcll(word);
        // inclusive wave prefix sum of cnt
        int incl = cnt;
        for (int o = 1; o < 64; o <<= 1) {
            int t = __shfl_up(incl, o);
            if (lane >= o) incl += t;
        }
        int idx = row_base + incl - cnt;
        while (word) {
            int bit = __ffsll((long long)word) - 1;
            word &= word - 1;
            if (idx < max_pts) {                        // (a pair call's smaller per-image clamp: the surplus records are never counted)
                int x = (w0 + lane) * 64 + bit;
                unsigned long long k = map[(long)y * p0 + x];
                int layer = key_layer(k);
                // integer position; the sub-pixel refinement (akazed.cu:1615-1662) follows in k_refine, sixteen lanes per keypoint
                const float px = (float)x, py = (float)y;
                hak_point* pt = pts + idx;
                pt->x = px;
                pt->y = py;
                pt->octave = layer;
                pt->response = fast ? (float)(int)(k >> 32) : key_resp(k);   // D8
                unsigned int* f32 = reinterpret_cast<unsigned int*>(pt->features);
#pragma unroll
                for (int q = 0; q < 16; q++) f32[q] = 0u;           // features + padding (written again by the describe kernel)
                pt->size = tab->sizes[layer];
                pt->angle = 0.f;
                pt->match = -1;
                pt->distance = -1;
                pt->match_x = -1.f;
                pt->match_y = -1.f;
            }
            idx++;
        }
        row_base += __shfl(incl, 63);
    }
}

// gRefine (akazed.cu:1615-1662) on the determinant of the winning level.  The determinant plane is not stored (HakLayout):
// the nine values of the 3x3 neighbourhood are re-evaluated from the level's derivative plane, bit-identical to what the
// Hessian kernel had in its extrema test (hak_det_at) -- one lane per value (14 gathers each), sixteen lanes per keypoint,
// so the 126 gathers of a keypoint are in flight together instead of in one lane's dependency chain.
__global__ __launch_bounds__(256) void k_refine(const float* __restrict__ base, long stride, HakLayout L,
                                                const HakTables* __restrict__ tab, const HakImgState* __restrict__ state,
                                                hak_point* points, int max_pts)
{
    const int img = blockIdx.y;
    const int n = state[img].num_pts;
    if (blockIdx.x * 16 >= n) return;                               // (block-uniform)
    const int g = threadIdx.x & 15;
    const int lane = threadIdx.x & 63;
    const int kp = blockIdx.x * 16 + (threadIdx.x >> 4);
    const bool live = kp < n;
    hak_point* pt = points + (long)img * max_pts + (live ? kp : 0);
    const int layer = live ? pt->octave : 0;
    const int o = layer / L.ms, s = layer - o * L.ms;
    const HakOct oc = L.oct[o];
    const int x = live ? (int)pt->x : 0, y = live ? (int)pt->y : 0;     // still the integer full-resolution position
    const int xx = x >> o, yy = y >> o;
    float d = 0.f;
    if (live && g < 9)
        d = hak_det_at<float>(base + (long)img * stride + L.dxy(o, s), xx + g % 3 - 1, yy + g / 3 - 1, tab->sigma_size[layer], oc.w, oc.h,
                              oc.p, tab->fac1, tab->fac2);
    const int g0 = lane & ~15;
    const float d00 = __shfl(d, g0 + 0), d01 = __shfl(d, g0 + 1), d02 = __shfl(d, g0 + 2);
    const float d10 = __shfl(d, g0 + 3), d11 = __shfl(d, g0 + 4), d12 = __shfl(d, g0 + 5);
    const float d20 = __shfl(d, g0 + 6), d21 = __shfl(d, g0 + 7), d22 = __shfl(d, g0 + 8);
    if (live && g == 0) {
        const float v2 = d11 + d11;
        const float dx = 0.5f * (d12 - d10);
        const float dy = 0.5f * (d21 - d01);
        const float dxx = d12 + d10 - v2;
        const float dyy = d21 + d01 - v2;
        const float dxy = 0.25f * (d22 + d00 - d02 - d20);
        const float dd = dxx * dyy - dxy * dxy;
        const float idd = dd != 0.f ? 1.f / dd : 0.f;
        const float dst0 = idd * (dxy * dy - dyy * dx);
        const float dst1 = idd * (dxy * dx - dxx * dy);
        const bool weak = dst0 < -1.f || dst0 > 1.f || dst1 < -1.f || dst1 > 1.f;
        if (!weak) {
            const int ratio = 1 << o;
            pt->y = ratio * (yy + dst1);
            pt->x = ratio * (xx + dst0);
        }
    }
}

void hak_launch_extrema_level(hipStream_t st, const HakBatch& b, const HakLayout& L, const HakTables* tab, int octave,
                              int s, float dthreshold, long det_off)
{
    const HakOct oc = L.oct[octave];
    dim3 grid((oc.w + 63) / 64, (oc.h + 15) / 16, b.nimg);
    k_extrema<<<grid, 256, 0, st>>>(b.base, b.stride, b.maps, b.map_stride, b.cand, b.cand_cap, b.state, L, tab,
                                    octave, s, dthreshold, det_off);
}

// The key map is sparse: only candidate pixels are ever written.  Instead of clearing the whole 8 B/px map before every
// call (2.1 GB for a 128-image 1080p batch), every call zeroes the entries its own candidates touched once the keypoints
// are emitted; the map is cleared in full only when the context is created.  The survivor bitmap and the per-row counts are
// restored to zero here as well (they used to be two memsets in front of k_nms_cand, on the critical path of every call).
__global__ __launch_bounds__(256) void k_clear_cand_maps(unsigned long long* __restrict__ maps, long map_stride,
                                                         const unsigned long long* __restrict__ cand, long cand_cap,
                                                         const HakImgState* __restrict__ state, int p,
                                                         unsigned long long* __restrict__ bitmap, long bitmap_words, int* __restrict__ rowcount, int h)
{
    const int img = blockIdx.y;
    unsigned long long* map = maps + (long)img * map_stride;
    long n = state[img].ncand;
    n = n < cand_cap ? n : cand_cap;
    for (long i = blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const unsigned long long e = cand[(long)img * cand_cap + i];
        const int x = (int)(e & 0xFFFFu), y = (int)((e >> 16) & 0xFFFFu);
        map[(long)y * p + x] = 0ull;
    }
    for (long i = blockIdx.x * 256 + threadIdx.x; i < bitmap_words; i += (long)gridDim.x * 256) bitmap[(long)img * bitmap_words + i] = 0ull;
    for (long i = blockIdx.x * 256 + threadIdx.x; i < h; i += (long)gridDim.x * 256) rowcount[(long)img * h + i] = 0;
}

// hand-made full-resolution maps -> key map + candidate list, with the extrema kernels' own key and list format
// (hak_op_tail_seed: the NMS micro-fixtures)
__global__ __launch_bounds__(256) void k_seed_maps(const unsigned* __restrict__ resp_bits, const int* __restrict__ layer, int w, int h, int p,
                                                   unsigned long long* map, unsigned long long* cand, long cand_cap, HakImgState* state)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= w || y >= h) return;
    const int l = layer[(long)y * w + x];
    if (l < 0) return;
    atomicMax(&map[(long)y * p + x], ((unsigned long long)resp_bits[(long)y * w + x] << 32) | (0xFFFFFFFFu - (unsigned)l));
    const long slot = atomicAdd(&state->ncand, 1);
    if (slot < cand_cap) cand[slot] = ((unsigned long long)l << 32) | ((unsigned)y << 16) | (unsigned)x;
}

void hak_launch_seed_maps(hipStream_t st, const HakBatch& b, const HakLayout& L, const unsigned* d_resp_bits, const int* d_layer)
{
    const int w = L.oct[0].w, h = L.oct[0].h, p = L.oct[0].p;
    k_seed_maps<<<dim3((w + 63) / 64, (h + 3) / 4), 256, 0, st>>>(d_resp_bits, d_layer, w, h, p, b.maps, b.cand, b.cand_cap, b.state);
}

void hak_launch_nms_emit(hipStream_t st, const HakBatch& b, const HakLayout& L, const HakTables* tab, int psz,
                         hak_point* points, int max_pts, int* num_out, int fast, int refine)
{
    const int w = L.oct[0].w, h = L.oct[0].h, p = L.oct[0].p;
    const int words = (w + 63) / 64;
    // (bitmap and rowcount are all zero here: hak_create, and every sequence's hak_launch_clear_maps)
    dim3 g1(b.nimg >= 8 ? 128 : 256, b.nimg);
    k_nms_cand<<<g1, 256, 0, st>>>(b.maps, b.map_stride, b.cand, b.cand_cap, b.state, tab, psz, w, h, p,
                                   b.bitmap, words, b.rowcount);
    k_row_scan<<<b.nimg, 256, 0, st>>>(b.rowcount, h, b.state, max_pts, b.nimg == 2 ? b.cap0 : 0, b.cap1, num_out);
    dim3 g3((h + 3) / 4, 1, b.nimg);
    k_emit<<<g3, 256, 0, st>>>(b.base, b.stride, b.maps, b.map_stride, L, tab, b.bitmap, words, b.rowcount, points, max_pts, fast);
    if (!fast && refine)                                            // (the FAST path refines on its int planes: k_orient<int>)
        k_refine<<<dim3((max_pts + 15) / 16, b.nimg), 256, 0, st>>>(b.base, b.stride, L, tab, b.state, points, max_pts);
}

// restores the all-zero state of the key map, the survivor bitmap and the row counts for the next sequence; needs the
// candidate list and must follow k_emit (any stream ordered after it)
void hak_launch_clear_maps(hipStream_t st, const HakBatch& b, const HakLayout& L)
{
    const int w = L.oct[0].w, h = L.oct[0].h, p = L.oct[0].p;
    const int words = (w + 63) / 64;
    k_clear_cand_maps<<<dim3(64, b.nimg), 256, 0, st>>>(b.maps, b.map_stride, b.cand, b.cand_cap, b.state, p, b.bitmap, (long)h * words, b.rowcount, h);
}

// ---- results -> pinned host memory in one launch (hak_download_batch): every image's valid prefix of point records and
// its count are stored straight into device-visible host memory over PCIe.  Replaces one D2H copy per image (each a blit
// kernel + an API call) and the extra host sync the counts needed first.
__global__ __launch_bounds__(256) void k_download(const hak_point* __restrict__ d_points, const int* __restrict__ d_num, long max_pts,
                                                  hak_point* h_points, int* h_num)
{
    const int img = blockIdx.y;
    const int n = d_num[img];
    if (blockIdx.x == 0 && threadIdx.x == 0) h_num[img] = n;
    const uint2* src = reinterpret_cast<const uint2*>(d_points + (long)img * max_pts);
    uint2* dst = reinterpret_cast<uint2*>(h_points + (long)img * max_pts);
    const long total = (long)n * (long)(sizeof(hak_point) / sizeof(uint2));
    for (long i = blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) dst[i] = src[i];
}

void hak_launch_download(hipStream_t st, const hak_point* d_points, const int* d_num, long max_pts, int nimg, hak_point* h_points,
                         int* h_num)
{
    k_download<<<dim3(8, nimg), 256, 0, st>>>(d_points, d_num, max_pts, h_points, h_num);
}

// ---- the results of a PAIR call (hak_detect_and_compute_pair): the two images' records go from the context's contiguous pair
// buffer to the caller's two device arrays (AkazeData::d_data) and, when those are pinned, straight to the two host arrays
// (AkazeData::h_data) -- one launch behind the match, counts included
__global__ __launch_bounds__(256) void k_download_pair(const hak_point* __restrict__ src_base, const int* __restrict__ d_num, long max_pts,
                                                       HakPairDst dst, int* h_num)
{
    const int img = blockIdx.y;
    const int n = min(d_num[img], dst.cap[img]);
    if (blockIdx.x == 0 && threadIdx.x == 0) h_num[img] = n;
    const uint2* src = reinterpret_cast<const uint2*>(src_base + (long)img * max_pts);
    uint2* dd = reinterpret_cast<uint2*>(dst.d[img]);
    uint2* dh = reinterpret_cast<uint2*>(dst.h[img]);
    const long total = (long)n * (long)(sizeof(hak_point) / sizeof(uint2));
    for (long i = blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const uint2 v = src[i];
        if (dd) dd[i] = v;
        if (dh) dh[i] = v;
    }
}

void hak_launch_download_pair(hipStream_t st, const hak_point* src, const int* d_num, long max_pts, const HakPairDst& dst, int* h_num)
{
    // blocks per image: the copy is a latency chain per thread (load, two stores, one of them over PCIe); 128 blocks instead of 8
    // take 7 us off the pair call (0.543 -> 0.536 ms; 16 / 32 / 64: 0.541 / 0.540 / 0.539)
    static const int nb = [] { const char* e = getenv("HAK_DOWNLOAD_BLOCKS"); const int v = e ? atoi(e) : 128; return v < 1 ? 1 : v; }();
    k_download_pair<<<dim3(nb, 2), 256, 0, st>>>(src, d_num, max_pts, dst, h_num);
}
