// kernels_describe.hip -- dominant orientation + 486-bit MLDB descriptor,
// one wave64 (one 64-thread workgroup) per keypoint, grid-strided.
//
//   hCalcOrient/gCalcOrient   akazed.cu:2655, 1665 (+ dFastAtan2 173-185)
//   hDescribe/gDescribe2      akazed.cu:2675, 1869
//   FAST path (V = int, 16.16 planes): refine akazed.cu:3600, orientation 3649, MLDB 3723 -- the same two kernels,
//   instantiated on the element type like the other heavy kernels (integer sums are exactly associative, so the
//   float path's per-lane table + tree is a valid order for them too)
//
// Bit-exactness rules taken over from the reference / the parity oracle:
//   * orientation: the 109 disc samples are summed into the 42 bins in
//     ascending sample-thread order (the reference uses racy float atomics, D7)
//   * MLDB: sample i belongs to "thread" i % 64; a thread sums its samples in
//     ascending order; threads are combined as (a_t + a_{t+32}) followed by the
//     shuffle-down tree 1,2,4,8,16 (akazed.cu:1957-1981)
//
// Both kernels issue every gather of a keypoint before any is consumed.  What bounds them (counters of the 256 x 1080p batch,
// profiles/README.md): k_describe* the texture-address path -- TA stalled by the L1 for 81 % of the kernel, 939 L1 accesses and
// 506 L2 requests per keypoint, two thirds of those L2 misses (random 64-byte sectors at 3.8 TB/s); cutting its VALU work by 40 %
// and its LDS traffic by more (k_describe_runs against k_describe) moved it by 1 %.  k_orient was bound by LDS broadcast reads
// until the bin sums were reorganised as a counting sort (0.95 -> 0.42 ms).
#include "hak_internal.h"
#include <type_traits>

#define ACC_LD 65           // padded leading dimension of the per-thread accumulator table
#define ACC_ROWS 30         // accumulator rows per round (multiple of 3): 87 rows in 3 rounds
#define MAX_SMP 7           // ceil(21*21 / 64) samples per lane for descriptor_pattern_size 10

// reduce accumulator rows [0, nrows) of the table in the reference's order:
// b_t = a_t + a_{t+32}; tree over t with strides 1,2,4,8,16; lane c owns row c
__device__ __forceinline__ float dsc_add(float a, float b) { return a + b; }
__device__ __forceinline__ int dsc_add(int a, int b) { return (int)((unsigned)a + (unsigned)b); }    // wraps like the reference
__device__ __forceinline__ int dsc_mul(int a, int b) { return (int)((unsigned)a * (unsigned)b); }

// dFastAtan2, akazed.cu:173-185 (0/0 -> 0)
__device__ __forceinline__ float dsc_fast_atan2(float yv, float xv)
{
    const float absx = fabsf(xv), absy = fabsf(yv);
    const float mn = absx < absy ? absx : absy, mx = absx < absy ? absy : absx;
    const float a = mx > 0.f ? mn / mx : 0.f;
    const float sq = a * a;
    float r = fmaf(fmaf(fmaf(-0.0464964749f, sq, 0.15931422f), sq, -0.327622764f), sq * a, a);
    r = (absy > absx ? HAK_HPI_F - r : r);
    r = (xv < 0 ? (float)(HAK_PI_D - r) : r);
    r = (yv < 0 ? -r : r);
    return r;
}

// base[byte_off / sizeof(V)] with the byte offset formed in 32 bits: with a wave-uniform base the load takes the
// `global_load v, v_off, s[base]` form -- one offset VGPR per sample for all three planes instead of a 64-bit address each
template <typename V>
__device__ __forceinline__ V dsc_ld(const V* base, unsigned byte_off)
{
    return *reinterpret_cast<const V*>(reinterpret_cast<const char*>(base) + byte_off);
}

// both first derivatives of one sample: ONE 8-byte gather from the interleaved {Lx, Ly} plane (HakLayout) -- one sector
// instead of two per sample; byte_off = element offset of the pixel * sizeof(V), as for dsc_ld
template <typename V> struct DscV2;
template <> struct DscV2<float> { using T = float2; };
template <> struct DscV2<int> { using T = int2; };
template <typename V>
__device__ __forceinline__ typename DscV2<V>::T dsc_ld2(const V* base, unsigned byte_off)
{
    return *reinterpret_cast<const typename DscV2<V>::T*>(reinterpret_cast<const char*>(base) + 2u * byte_off);
}

typedef float dsc_v2f __attribute__((ext_vector_type(2)));

template <typename V>
__device__ __forceinline__ void reduce_rows(const V* acc, V* vals, int nrows, int out_base, int lane)
{
    if (lane < nrows) {
        const V* a = acc + lane * ACC_LD;
        // the same tree, evaluated 8 leaves at a time to keep the register footprint small:
        // c_k = ((b0+b1)+(b2+b3)) + ((b4+b5)+(b6+b7)) with b_j = a[8k+j] + a[8k+j+32]; result (c0+c1)+(c2+c3)
        V c[4];
#pragma unroll 1
        for (int k = 0; k < 4; k++) {
            const V* q = a + 8 * k;
            const V b0 = dsc_add(q[0], q[32]), b1 = dsc_add(q[1], q[33]), b2 = dsc_add(q[2], q[34]), b3 = dsc_add(q[3], q[35]);
            const V b4 = dsc_add(q[4], q[36]), b5 = dsc_add(q[5], q[37]), b6 = dsc_add(q[6], q[38]), b7 = dsc_add(q[7], q[39]);
            c[k] = dsc_add(dsc_add(dsc_add(b0, b1), dsc_add(b2, b3)), dsc_add(dsc_add(b4, b5), dsc_add(b6, b7)));
        }
        vals[out_base + lane] = dsc_add(dsc_add(c[0], c[1]), dsc_add(c[2], c[3]));
    }
}

// ---- dominant orientation as its own kernel: it needs ~3 KB of LDS and few registers, so many more keypoints are in flight
// per CU than inside the descriptor kernel (whose accumulator table and sample registers cap it at 12 per CU); its gather and
// LDS latency chains then overlap.  Writes pt->angle; k_describe reads it back.  The FAST instantiation first refines the
// keypoint position on the determinant plane (akazed.cu:3600; the float path refines in k_emit) and is launched for that alone
// (do_orient = 0) when no orientation is wanted.
template <typename V>
__global__ __launch_bounds__(64) void k_orient(const V* __restrict__ base, long stride, HakLayout L,
                                               const HakTables* __restrict__ tab, const HakImgState* __restrict__ state,
                                               hak_point* points, int max_pts, int do_orient, int order, const int* __restrict__ perm)
{
    constexpr bool FAST = std::is_same<V, int>::value;
    // bin sums in sample order (D7) without every bin lane looking at every sample: the samples are counting-sorted by bin
    // (stable: place = samples of lower bins + earlier samples of the same bin, both from 64-bit lane masks OR-ed into LDS per
    // bin), then lane b adds its own run front to back.  The first version -- 42 bin lanes scanning all 109 samples through
    // 16-byte LDS broadcast reads -- was bound by exactly those reads: 112 x 768 bytes per keypoint against the CU's
    // 128 bytes/clk made up 0.6 of the kernel's 0.95 ms, whatever the VALU count.
    __shared__ unsigned long long bmask[2][42];  // [turn][bin]: lanes of that turn whose sample falls into the bin
    __shared__ int bstart[42], bfirst[42];      // samples in lower bins; samples of the bin in turn 0
    __shared__ float2 sorted[112];
    __shared__ float resx[42], resy[42], re8x[42], re8y[42];
    int img = blockIdx.y, first = blockIdx.x;
    if (order > 0) {                                                // images dealt in groups of `order`: image index fastest inside a group
        const unsigned lin = blockIdx.y * gridDim.x + blockIdx.x;
        const unsigned grp = lin / ((unsigned)order * gridDim.x), b0 = grp * order;
        const unsigned gs = min((unsigned)order, gridDim.y - b0), within = lin - grp * order * gridDim.x;
        img = b0 + within % gs;
        first = within / gs;
    }
    const int lane = threadIdx.x;
    const int npts = state[img].num_pts;
    const V* arena = base + (long)img * stride;
    hak_point* pts = points + (long)img * max_pts;
    for (int pi = first; pi < npts; pi += gridDim.x) {
        hak_point* pt = pts + (perm ? perm[(long)img * max_pts + pi] : pi);
        float ptx = pt->x, pty = pt->y;
        const float ptsize = pt->size;
        const int layer = __builtin_amdgcn_readfirstlane(pt->octave);   // one keypoint per wave: plane bases stay in SGPRs
        const int o = layer / L.ms, s = layer - o * L.ms;
        const HakOct oc = L.oct[o];
        const V* dxyd = arena + L.dxy(o, s);
        if constexpr (FAST) {
            // refinement on the integer determinant (akazed.cu:3600), re-evaluated from the derivative plane (hak_det_at): lanes
            // 0..8 take one value of the 3x3 neighbourhood each, lane 0 finishes
            int dvl = 0;
            {
                const int y = (int)pty >> o, x = (int)ptx >> o;
                if (lane < 9)
                    dvl = hak_det_at<int>(dxyd, x + lane % 3 - 1, y + lane / 3 - 1, tab->sigma_size[layer], oc.w, oc.h, oc.p, tab->ifac1, tab->ifac2);
            }
            int dv[3][3];
#pragma unroll
            for (int j = 0; j < 3; j++)
#pragma unroll
                for (int i = 0; i < 3; i++) dv[j][i] = __shfl(dvl, 3 * j + i);
            if (lane == 0) {
                const int y = (int)pty >> o, x = (int)ptx >> o;
                const int v2 = dv[1][1] + dv[1][1];
                const int dx = (dv[1][2] - dv[1][0]) >> 1, dy = (dv[2][1] - dv[0][1]) >> 1;
                const int dxx = dv[1][2] + dv[1][0] - v2, dyy = dv[2][1] + dv[0][1] - v2;
                const int dxy = (dv[2][2] + dv[0][0] - dv[0][2] - dv[2][0]) >> 2;
                const int dd = dsc_add(dsc_mul(dxx, dyy), -dsc_mul(dxy, dxy));
                const float idd = dd != 0 ? (1.f / dd) : 0.f;
                const float dst0 = idd * dsc_add(dsc_mul(dxy, dy), -dsc_mul(dyy, dx));
                const float dst1 = idd * dsc_add(dsc_mul(dxy, dx), -dsc_mul(dxx, dy));
                if (!(dst0 < -1.f || dst0 > 1.f || dst1 < -1.f || dst1 > 1.f)) {
                    const int ratio = 1 << o;
                    pty = ratio * (y + dst1);
                    ptx = ratio * (x + dst0);
                    pt->x = ptx;
                    pt->y = pty;
                }
            }
            ptx = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(ptx)));
            pty = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(pty)));
            if (!do_orient) continue;
        }
        {
            const int step = (int)(ptsize + 0.5f);
            const int x = (int)(ptx + 0.5f) >> o;
            const int y = (int)(pty + 0.5f) >> o;
            // the reference's 208 sample threads keep the 109 with r2 < 36; HakTables lists those in ascending thread order
            // (offsets and Gaussian weight per slot), so two turns of 64 lanes cover them and a sample's slot is its place in
            // the summation order (D7) -- no compaction step.  Slots 109..127 gather at the keypoint and get no bin.
            float gdx[2], gdy[2], gw[2];
#pragma unroll
            for (int q = 0; q < 2; q++) {
                const int ij = tab->orient_ij[q * 64 + lane];
                gw[q] = tab->orient_gw[q * 64 + lane];
                const int i = (int)(signed char)(ij & 0xFF), j = (int)(signed char)((ij >> 8) & 0xFF);
                const int yy = min(max(y + step * j, 0), oc.h - 1), xx = min(max(x + step * i, 0), oc.w - 1);
                const unsigned pos = (unsigned)(yy * oc.p + xx) * (unsigned)sizeof(V);
                const auto d2 = dsc_ld2(dxyd, pos);
                gdx[q] = (float)d2.x;
                gdy[q] = (float)d2.y;
            }
            int bin[2];
            float sdx[2], sdy[2];
            if (lane < 42) bmask[0][lane] = bmask[1][lane] = 0ull;  // (one wave per workgroup: its LDS operations complete in order)
#pragma unroll
            for (int q = 0; q < 2; q++) {
                sdx[q] = gw[q] * gdx[q];
                sdy[q] = gw[q] * gdy[q];
                const float ang = FAST ? dsc_fast_atan2(sdy[q], sdx[q]) : hak_atan2f(sdy[q], sdx[q]);   // akazed.cu:3685 / 1702
                int a = (int)(ang * (21 / HAK_PI_D)) + 21;
                a = a > 41 ? 41 : a;
                a = a < 0 ? 0 : a;
                bin[q] = q * 64 + lane < 109 ? a : -1;
                if (bin[q] >= 0) atomicOr(&bmask[q][bin[q]], 1ull << lane);
            }
            hak_lds_barrier();
            int cnt = 0, start;
            if (lane < 42) {
                const int c0 = __popcll(bmask[0][lane]);
                cnt = c0 + __popcll(bmask[1][lane]);
                bfirst[lane] = c0;
            }
            {
                int incl = cnt;                                      // inclusive scan over the bin lanes
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) {
                    const int up = __shfl_up(incl, d);
                    incl += lane >= d ? up : 0;
                }
                start = incl - cnt;
            }
            if (lane < 42) bstart[lane] = start;
            hak_lds_barrier();
#pragma unroll
            for (int q = 0; q < 2; q++)
                if (bin[q] >= 0) {
                    const int place = bstart[bin[q]] + (q ? bfirst[bin[q]] : 0) + __popcll(bmask[q][bin[q]] & ((1ull << lane) - 1ull));
                    sorted[place] = make_float2(sdx[q], sdy[q]);
                }
            hak_lds_barrier();
            if (lane < 42) {
                dsc_v2f r = {0.f, 0.f};
                for (int p = start; p < start + cnt; p++) {
                    const float2 v = sorted[p];
                    r.x += v.x;
                    r.y += v.y;
                }
                resx[lane] = r.x;
                resy[lane] = r.y;
            }
            hak_lds_barrier();
            if (lane < 42) {                                        // akazed.cu:1708-1717
                float ax = resx[lane], ay = resy[lane];
                for (int k = lane + 1; k < lane + 7; k++) {
                    ax += resx[k < 42 ? k : k - 42];
                    ay += resy[k < 42 ? k : k - 42];
                }
                re8x[lane] = ax;
                re8y[lane] = ay;
            }
            hak_lds_barrier();
            // first k maximising re8x^2 + re8y^2 (akazed.cu:1721-1730: strict '>' from maxr = 0)
            float rk = 0.f;
            if (lane < 42) rk = re8x[lane] * re8x[lane] + re8y[lane] * re8y[lane];
            float rmax = rk;
            for (int off = 32; off > 0; off >>= 1) rmax = fmaxf(rmax, __shfl_xor(rmax, off));
            const unsigned long long mm = __ballot(lane < 42 && rk == rmax);
            const int maxk = rmax > 0.f && mm ? __ffsll((long long)mm) - 1 : 0;
            if (lane == 0) {
                const float r = dsc_fast_atan2(re8y[maxk], re8x[maxk]);
                pt->angle = (r < 0.0f ? (float)(r + 2.0f * HAK_PI_D) : r);  // akazed.cu:1734
            }
            hak_lds_barrier();
        }
        hak_lds_barrier();                      // the scratch is reused by the next keypoint of this block
    }
}

// ---- MLDB with the sample plan of HakTables (the default descriptor_pattern_size 10 and every other pattern whose plan
// holds: <= 7 samples per lane, no lane coming back to an accumulator row it has left).  Same sums in the same order as
// k_describe below, organised around what is fixed per (lane, n):
//   * the sample's window offset and its three accumulator rows come from the plan (no division, no cell arithmetic);
//   * a lane's samples of one row are consecutive turns, so its partial sum a_t = ((0 + v) + v') + ... is carried in
//     registers and written to the LDS table ONCE (k_describe: a read-modify-write round trip per sample and row);
//   * three rounds of 30 rows over an 8 KB table, as in k_describe, but a round only walks the grids that can own its rows
//     (2x2 + 3x3, 3x3 + 4x4, 4x4: five passes over the seven samples instead of nine).  Two rounds that coincide with the grids
//     (39 + 48 rows, every (sample, grid) looked at once) need 12.8 KB: 12 instead of 16 keypoints per CU and 3 % slower;
//   * the table is cleared with 16-byte stores in linear order.  One wave per workgroup: its LDS operations complete in
//     program order, so the clear, the column writes and the row reads need no barrier between them, only the wait for the
//     read data.
#define RUN_ROWS 30         // accumulator rows per round
template <typename V>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4))) void k_describe_runs(const V* __restrict__ base, long stride, HakLayout L,
                                                      const HakTables* __restrict__ tab, const HakImgState* __restrict__ state,
                                                      hak_point* points, int max_pts, int upright, int order, const int* __restrict__ perm)
{
    __shared__ __attribute__((aligned(16))) V acc[(RUN_ROWS * ACC_LD + 3) / 4 * 4];
    __shared__ V vals[90];
    int img = blockIdx.y, first = blockIdx.x;
    if (order > 0) {                                                // images dealt in groups of `order`: image index fastest inside a group
        const unsigned lin = blockIdx.y * gridDim.x + blockIdx.x;
        const unsigned grp = lin / ((unsigned)order * gridDim.x), b0 = grp * order;
        const unsigned gs = min((unsigned)order, gridDim.y - b0), within = lin - grp * order * gridDim.x;
        img = b0 + within % gs;
        first = within / gs;
    }
    const int lane = threadIdx.x;
    const int npts = state[img].num_pts;
    const V* arena = base + (long)img * stride;
    hak_point* pts = points + (long)img * max_pts;
    if (first >= npts) return;
    const uint4 cmp = reinterpret_cast<const uint4*>(tab->comp_packed)[lane];
    const unsigned int cw[4] = {cmp.x, cmp.y, cmp.z, cmp.w};

    for (int pi = first; pi < npts; pi += gridDim.x) {
        // the lane's sample offsets, loaded with the keypoint (before the gathers: a load issued between them would have to
        // wait for them too, vmcnt counts in order); its accumulator rows are loaded behind the gathers, where they are used.
        // Per keypoint, not per block: a block runs this loop once or twice, and hoisted copies cost a wave of occupancy.
        unsigned posw[MAX_SMP], cellw[MAX_SMP];
#pragma unroll
        for (int n = 0; n < MAX_SMP; n++) posw[n] = tab->dsc_pos[n * 64 + lane];
        hak_point* pt = pts + (perm ? perm[(long)img * max_pts + pi] : pi);
        const float ptx = pt->x, pty = pt->y, ptsize = pt->size;
        const int layer = __builtin_amdgcn_readfirstlane(pt->octave);
        const int o = layer / L.ms, s = layer - o * L.ms;
        const HakOct oc = L.oct[o];
        const V* imd = arena + L.lt(o, s);
        const V* dxyd = arena + L.dxy(o, s);
        float angle = 0.f;
        if (!upright) angle = pt->angle;                            // written by k_orient

        V vim[MAX_SMP], vrx[MAX_SMP], vry[MAX_SMP];
        {
            const float iratio = 1.f / (1 << o);
            const int scale = (int)(ptsize + 0.5f);
            const float xf = ptx * iratio;
            const float yf = pty * iratio;
            float si, co;
            hak_sincosf(angle, &si, &co);
            V gdx[MAX_SMP], gdy[MAX_SMP];
#pragma unroll
            for (int n = 0; n < MAX_SMP; n++) {
                // a turn without a sample (posw = 0) gathers at the keypoint itself; it belongs to no row, so its values
                // end in a running sum that is never written
                const int l = (int)(signed char)(posw[n] & 0xFF);
                const int k = (int)(signed char)((posw[n] >> 8) & 0xFF);
                int xp = (int)(xf + scale * (k * co - l * si) + 0.5f);  // akazed.cu:1921
                int yp = (int)(yf + scale * (k * si + l * co) + 0.5f);  // akazed.cu:1922
                xp = min(max(xp, 0), oc.w - 1);
                yp = min(max(yp, 0), oc.h - 1);
                const unsigned pos = (unsigned)(yp * oc.p + xp) * (unsigned)sizeof(V);
                vim[n] = dsc_ld(imd, pos);
                const auto d2 = dsc_ld2(dxyd, pos);
                gdx[n] = d2.x;
                gdy[n] = d2.y;
            }
#pragma unroll
            for (int n = 0; n < MAX_SMP; n++) cellw[n] = tab->dsc_cell[n * 64 + lane];
#pragma unroll
            for (int n = 0; n < MAX_SMP; n++) {
                vrx[n] = (V)(-gdx[n] * si + gdy[n] * co);           // akazed.cu:1931 / 3777
                vry[n] = (V)(gdx[n] * co + gdy[n] * si);
            }
        }
        // one grid in one round: the lane's running sums per row, written when the row's last sample has been added and the
        // row belongs to the round's window [row0, row0 + nrows)
        auto runs = [&](int g, int row0, int nrows) {
            V rim = V(0), rrx = V(0), rry = V(0);
#pragma unroll
            for (int n = 0; n < MAX_SMP; n++) {
                const unsigned e = (cellw[n] >> (8 * g)) & 0xFFu;
                const bool cont = e & 0x80u;
                rim = dsc_add(cont ? rim : V(0), vim[n]);
                rrx = dsc_add(cont ? rrx : V(0), vrx[n]);
                rry = dsc_add(cont ? rry : V(0), vry[n]);
                const unsigned rr = (e & 0x7Fu) - (unsigned)row0;    // 0x7F (no row) lies beyond every window
                if (((cellw[n] >> (24 + g)) & 1u) && rr < (unsigned)nrows) {
                    V* a = acc + rr * ACC_LD + lane;
                    a[0] = rim;
                    a[ACC_LD] = rrx;
                    a[2 * ACC_LD] = rry;
                }
            }
        };
        auto clear = [&](int nrows) {
            float4* a4 = reinterpret_cast<float4*>(acc);
            const int n4 = (nrows * ACC_LD + 3) / 4;
#pragma unroll
            for (int i = 0; i < ((RUN_ROWS * ACC_LD + 3) / 4 + 63) / 64; i++)
                if (lane + 64 * i < n4) a4[lane + 64 * i] = make_float4(0.f, 0.f, 0.f, 0.f);
        };
        // rows 0..11 2x2, 12..38 3x3, 39..86 4x4 (three per cell); a cell never straddles two rounds (30 and 60 are cell starts)
        // (wave_barrier: no instruction -- it keeps the compiler from moving one phase's LDS accesses across the next one's, which
        // touch other lanes' words; the hardware runs a wave's LDS operations in order)
#define HAK_WB() __builtin_amdgcn_wave_barrier()
        clear(RUN_ROWS); HAK_WB();
        runs(0, 0, RUN_ROWS);
        runs(1, 0, RUN_ROWS); HAK_WB();
        reduce_rows(acc, vals, RUN_ROWS, 0, lane); HAK_WB();
        clear(RUN_ROWS); HAK_WB();
        runs(1, RUN_ROWS, RUN_ROWS);
        runs(2, RUN_ROWS, RUN_ROWS); HAK_WB();
        reduce_rows(acc, vals, RUN_ROWS, RUN_ROWS, lane); HAK_WB();
        clear(87 - 2 * RUN_ROWS); HAK_WB();
        runs(2, 2 * RUN_ROWS, 87 - 2 * RUN_ROWS); HAK_WB();
        reduce_rows(acc, vals, 87 - 2 * RUN_ROWS, 2 * RUN_ROWS, lane);
#undef HAK_WB
        hak_lds_barrier();
        if (lane < HAK_FLEN) {                                      // akazed.cu:1987-1999
            unsigned int desc_r = 0;
            const int nb = lane == 60 ? 6 : 8;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const unsigned int pr = (cw[i >> 1] >> (16 * (i & 1))) & 0xFFFFu;
                const int idx1 = pr & 0xFF, idx2 = pr >> 8;
                if (i < nb) desc_r |= (vals[idx1] > vals[idx2] ? 1u : 0u) << i;
            }
            pt->features[lane] = (unsigned char)desc_r;
        }
        if (lane == 0) pt->angle = angle;
        hak_lds_barrier();
    }
}

// the plan of k_describe_runs for one pattern size and k_orient's sample list (host side, at context creation)
void hak_describe_plan(HakTables* t, int patsize)
{
    int ns = 0;
    for (int tix = 0; tix < 208; tix++) {                           // akazed.cu:1676-1690: 13 x 16 threads, disc r2 < 36
        const int i = (tix & 15) - 6, j = (tix >> 4) - 6, r2 = i * i + j * j;
        if (r2 >= 36) continue;
        t->orient_ij[ns] = (i & 0xFF) | ((j & 0xFF) << 8);
        t->orient_gw[ns] = t->orient_w[r2];
        ns++;
    }
    for (; ns < 128; ns++) { t->orient_ij[ns] = 0; t->orient_gw[ns] = 0.f; }

    const int size2 = patsize;
    const int size3 = (int)ceilf(2.0f * patsize / 3.0f);            // akazed.cu:2682
    const int size4 = (int)ceilf(0.5f * patsize);                   // akazed.cu:2683
    const int winsize = std::max(3 * size3, 4 * size4);
    const int nsmp = winsize * winsize;
    t->dsc_plan_ok = 0;
    if (nsmp > MAX_SMP * 64 || size2 > 127 || winsize - size2 > 127) return;
    bool ok = true;
    for (int lane = 0; lane < 64 && ok; lane++) {
        int rows[MAX_SMP][3];
        for (int n = 0; n < MAX_SMP; n++) {
            const int i = lane + 64 * n;
            const int y = i / winsize, x = i - winsize * y;
            const int m = std::max(x, y);
            const bool have = i < nsmp;
            const int x3 = (x < size3 ? 0 : (x < 2 * size3 ? 1 : 2)), y3 = (y < size3 ? 0 : (y < 2 * size3 ? 1 : 2));
            const int x4 = (x < 2 * size4 ? (x < size4 ? 0 : 1) : (x < 3 * size4 ? 2 : 3));
            const int y4 = (y < 2 * size4 ? (y < size4 ? 0 : 1) : (y < 3 * size4 ? 2 : 3));
            rows[n][0] = have && m < 2 * size2 ? 3 * ((y < size2 ? 0 : 2) + (x < size2 ? 0 : 1)) : 0x7F;
            rows[n][1] = have && m < 3 * size3 ? 3 * (4 + y3 * 3 + x3) : 0x7F;
            rows[n][2] = have && m < 4 * size4 ? 39 + 3 * (y4 * 4 + x4) : 0x7F;
            t->dsc_pos[n * 64 + lane] = have ? ((unsigned)((x - size2) & 0xFF) | ((unsigned)((y - size2) & 0xFF) << 8) | (1u << 16)) : 0u;
        }
        for (int n = 0; n < MAX_SMP; n++) {
            unsigned w = 0;
            for (int g = 0; g < 3; g++) {
                const int r = rows[n][g];
                const bool cont = r != 0x7F && n > 0 && rows[n - 1][g] == r;
                const bool last = r != 0x7F && !(n + 1 < MAX_SMP && rows[n + 1][g] == r);
                w |= ((unsigned)r | (cont ? 0x80u : 0u)) << (8 * g);
                w |= last ? (1u << (24 + g)) : 0u;
                if (r != 0x7F && !cont)                              // a row the lane has already left?
                    for (int q = 0; q + 1 < n; q++) ok = ok && rows[q][g] != r;
            }
            t->dsc_cell[n * 64 + lane] = w;
        }
    }
    t->dsc_plan_ok = ok ? 1 : 0;
}

template <typename V>
__global__ __launch_bounds__(64) void k_describe(const V* __restrict__ base, long stride, HakLayout L,
                                                 const HakTables* __restrict__ tab, const HakImgState* __restrict__ state,
                                                 hak_point* points, int max_pts, int patsize, int upright, int desc)
{
    constexpr bool FAST = std::is_same<V, int>::value;
    __shared__ V acc[ACC_ROWS * ACC_LD];        // [cell*3+ch][thread], one round at a time
    __shared__ V vals[90];

    const int img = blockIdx.y;
    const int lane = threadIdx.x;
    const int npts = state[img].num_pts;
    const V* arena = base + (long)img * stride;
    hak_point* pts = points + (long)img * max_pts;
    if (!desc) return;

    const int size2 = patsize;
    const int size3 = (int)ceilf(2.0f * patsize / 3.0f);            // akazed.cu:2682
    const int size4 = (int)ceilf(0.5f * patsize);                   // akazed.cu:2683
    const int winsize = max(3 * size3, 4 * size4);
    const int nsmp = winsize * winsize;

    // this lane's 8 comparison pairs (akazed.cu:65-159), packed as bytes: one 16-byte load
    const uint4 cmp = reinterpret_cast<const uint4*>(tab->comp_packed)[lane];
    const unsigned int cw[4] = {cmp.x, cmp.y, cmp.z, cmp.w};

    for (int pi = blockIdx.x; pi < npts; pi += gridDim.x) {
        hak_point* pt = pts + pi;
        const float ptx = pt->x, pty = pt->y, ptsize = pt->size;
        const int layer = __builtin_amdgcn_readfirstlane(pt->octave);   // one keypoint per wave: plane bases stay in SGPRs
        const int o = layer / L.ms, s = layer - o * L.ms;
        const HakOct oc = L.oct[o];
        const V* imd = arena + L.lt(o, s);
        const V* dxyd = arena + L.dxy(o, s);
        float angle = 0.f;

        if (!upright) angle = pt->angle;                            // written by k_orient

        // ------------------------------------------------------------- MLDB
        // phase 1: positions + all gathers of this lane's samples (i = lane, lane+64, ...)
        // rotated derivatives: float path -dx*si + dy*co (akazed.cu:1931); FAST path the same in float from the 16.16 integers,
        // truncated back to int (akazed.cu:3777-3778)
        auto rot_x = [](V dx, V dy, float si, float co) -> V { return (V)(-dx * si + dy * co); };
        auto rot_y = [](V dx, V dy, float si, float co) -> V { return (V)(dx * co + dy * si); };
        V vim[MAX_SMP], vrx[MAX_SMP], vry[MAX_SMP];
        int cells[MAX_SMP];                     // per sample: its accumulator row in the 2x2 / 3x3 / 4x4 grid as three bytes (0xFF = none),
                                                // worked out once here instead of in each of the three rounds below
        {
            const float iratio = 1.f / (1 << o);
            const int scale = (int)(ptsize + 0.5f);
            const float xf = ptx * iratio;
            const float yf = pty * iratio;
            float si, co;
            hak_sincosf(angle, &si, &co);
            V gdx[MAX_SMP], gdy[MAX_SMP];
#pragma unroll
            for (int n = 0; n < MAX_SMP; n++) {
                const int i = lane + 64 * n;
                const int y = i / winsize;
                const int x = i - winsize * y;
                const bool ok = i < nsmp;
                {
                    const int m = max(x, y);
                    const int r2 = m < 2 * size2 ? 3 * ((y < size2 ? 0 : 2) + (x < size2 ? 0 : 1)) : 0xFF;
                    const int x3 = (x < size3 ? 0 : (x < 2 * size3 ? 1 : 2)), y3 = (y < size3 ? 0 : (y < 2 * size3 ? 1 : 2));
                    const int r3 = m < 3 * size3 ? 3 * (4 + y3 * 3 + x3) : 0xFF;
                    const int x4 = (x < 2 * size4 ? (x < size4 ? 0 : 1) : (x < 3 * size4 ? 2 : 3));
                    const int y4 = (y < 2 * size4 ? (y < size4 ? 0 : 1) : (y < 3 * size4 ? 2 : 3));
                    const int r4 = m < 4 * size4 ? 39 + 3 * (y4 * 4 + x4) : 0xFF;
                    cells[n] = ok ? (r2 | (r3 << 8) | (r4 << 16)) : 0xFFFFFF;
                }
                const int l = x - size2;
                const int k = y - size2;
                int xp = (int)(xf + scale * (k * co - l * si) + 0.5f);  // akazed.cu:1921
                int yp = (int)(yf + scale * (k * si + l * co) + 0.5f);  // akazed.cu:1922
                xp = min(max(xp, 0), oc.w - 1);
                yp = min(max(yp, 0), oc.h - 1);
                const unsigned pos = (unsigned)(yp * oc.p + xp) * (unsigned)sizeof(V);
                vim[n] = ok ? dsc_ld(imd, pos) : V(0);
                if (ok) {
                    const auto d2 = dsc_ld2(dxyd, pos);
                    gdx[n] = d2.x;
                    gdy[n] = d2.y;
                } else gdx[n] = gdy[n] = V(0);
            }
#pragma unroll
            for (int n = 0; n < MAX_SMP; n++) {
                vrx[n] = rot_x(gdx[n], gdy[n], si, co);
                vry[n] = rot_y(gdx[n], gdy[n], si, co);
            }
        }
        // phase 2: the 87 accumulator rows (2x2 cells: rows 0..11, 3x3: 12..38, 4x4: 39..86; three rows -- value, dx', dy' --
        // per cell) are built in rounds of ACC_ROWS rows over one small LDS table, so that more keypoints fit on a CU.
        // Cell bases and ACC_ROWS are multiples of 3: a cell never straddles two rounds.  Samples beyond MAX_SMP per
        // lane (descriptor_pattern_size > 10) are re-gathered by the tail loop of every round.
        for (int r0 = 0; r0 < 87; r0 += ACC_ROWS) {
            for (int i = lane; i < ACC_ROWS * ACC_LD; i += 64) acc[i] = V(0);
            hak_lds_barrier();
            auto add = [&](int row, V im, V rx, V ry) {
                const int rr = row - r0;
                if (rr >= 0 && rr < ACC_ROWS) {
                    acc[rr * ACC_LD + lane] = dsc_add(acc[rr * ACC_LD + lane], im);
                    acc[(rr + 1) * ACC_LD + lane] = dsc_add(acc[(rr + 1) * ACC_LD + lane], rx);
                    acc[(rr + 2) * ACC_LD + lane] = dsc_add(acc[(rr + 2) * ACC_LD + lane], ry);
                }
            };
            auto scatter = [&](int x, int y, V im, V rx, V ry) {
                const int m = max(x, y);
                if (m < 2 * size2) add(3 * ((y < size2 ? 0 : 2) + (x < size2 ? 0 : 1)), im, rx, ry);
                if (m < 3 * size3) {
                    const int x3 = (x < size3 ? 0 : (x < 2 * size3 ? 1 : 2));
                    const int y3 = (y < size3 ? 0 : (y < 2 * size3 ? 1 : 2));
                    add(3 * (4 + y3 * 3 + x3), im, rx, ry);
                }
                if (m < 4 * size4) {
                    const int x4 = (x < 2 * size4 ? (x < size4 ? 0 : 1) : (x < 3 * size4 ? 2 : 3));
                    const int y4 = (y < 2 * size4 ? (y < size4 ? 0 : 1) : (y < 3 * size4 ? 2 : 3));
                    add(39 + 3 * (y4 * 4 + x4), im, rx, ry);
                }
            };
#pragma unroll
            for (int n = 0; n < MAX_SMP; n++)
            {
                add(cells[n] & 0xFF, vim[n], vrx[n], vry[n]);            // 0xFF lies beyond every round's row window
                add((cells[n] >> 8) & 0xFF, vim[n], vrx[n], vry[n]);
                add((cells[n] >> 16) & 0xFF, vim[n], vrx[n], vry[n]);
            }
            for (int i = lane + 64 * MAX_SMP; i < nsmp; i += 64) {      // tail: only for pattern sizes > 10
                const int y = i / winsize, x = i - winsize * y;
                const float iratio = 1.f / (1 << o);
                const int scale = (int)(ptsize + 0.5f);
                float si, co;
                hak_sincosf(angle, &si, &co);
                const int l = x - size2, k = y - size2;
                int xp = (int)(ptx * iratio + scale * (k * co - l * si) + 0.5f);
                int yp = (int)(pty * iratio + scale * (k * si + l * co) + 0.5f);
                xp = min(max(xp, 0), oc.w - 1);
                yp = min(max(yp, 0), oc.h - 1);
                const unsigned pos = (unsigned)(yp * oc.p + xp) * (unsigned)sizeof(V);
                const V im = dsc_ld(imd, pos);
                const auto d2 = dsc_ld2(dxyd, pos);
                scatter(x, y, im, rot_x(d2.x, d2.y, si, co), rot_y(d2.x, d2.y, si, co));
            }
            hak_lds_barrier();
            reduce_rows(acc, vals, min(ACC_ROWS, 87 - r0), r0, lane);
            hak_lds_barrier();
        }
        if (lane < HAK_FLEN) {                                      // akazed.cu:1987-1999
            unsigned int desc_r = 0;
            const int nb = lane == 60 ? 6 : 8;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const unsigned int pr = (cw[i >> 1] >> (16 * (i & 1))) & 0xFFFFu;
                const int idx1 = pr & 0xFF, idx2 = pr >> 8;
                if (i < nb) desc_r |= (vals[idx1] > vals[idx2] ? 1u : 0u) << i;
            }
            pt->features[lane] = (unsigned char)desc_r;
        }
        if (lane == 0) pt->angle = angle;
        hak_lds_barrier();
    }
    (void)FAST;
}

// Block order of k_orient / k_describe_runs: keypoint index fastest within one image (0), or the images dealt in groups of G
// with the image index fastest inside a group (G > 0).  Measured on 256 x 1080p (describe class, ms): G = 0: 4.11, 2: 3.97,
// 4: 3.99, 8: 4.12, 16: 4.30, 32: 4.51, 256: 5.74 -- a few images side by side spread the gathers over more L2 channels, many
// lose the L2 / Infinity-Cache sharing between neighbouring keypoints.  HakKnobs::desc_order / desc_plan, from HAK_DESC_ORDER /
// HAK_DESC_PLAN (A/B runs and the alternatives test; results do not depend on either).
static HakKnobs knobs_of(const HakBatch& b) { return b.knobs ? *b.knobs : hak_knobs_from_env(); }

// Visiting order of the keypoint kernels: an image's keypoints come out of the NMS in raster order of the full-resolution map
// with all levels mixed (the reference's order: the output keeps it), so consecutive blocks gather from sixteen different plane
// pairs.  perm = the stable counting sort of the indices by level: consecutive blocks then work on neighbouring patches of ONE
// pair of planes, whose lines they share in L2 (describe class 5.08 -> 4.99 ms per 384 x 1080p images: the kernels are bound by
// the texture addresser, not by L2 misses, so the gain is small).  One block per image; in every wave lane l keeps level l's
// running offset.
#define PERM_WAVES 16
__global__ __launch_bounds__(64 * PERM_WAVES) void k_desc_perm(const HakImgState* __restrict__ state, const hak_point* __restrict__ points,
                                                              int max_pts, int* __restrict__ perm, int nlayers)
{
    // wave w takes the w-th contiguous slice of the image's keypoints (the sort is stable: slices in order, indices in order)
    __shared__ int cnts[PERM_WAVES][64];
    const int img = blockIdx.x, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int npts = min(state[img].num_pts, max_pts);
    const hak_point* pts = points + (long)img * max_pts;
    int* pm = perm + (long)img * max_pts;
    const int per = ((npts + PERM_WAVES - 1) / PERM_WAVES + 63) & ~63;
    const int i_beg = min(wv * per, npts), i_end = min(i_beg + per, npts);
    int cnt = 0;
    for (int i0 = i_beg; i0 < i_end; i0 += 64) {
        const int i = i0 + lane;
        const int layer = i < i_end ? pts[i].octave : -1;
        for (int l = 0; l < nlayers; l++) {
            const unsigned long long m = __ballot(layer == l);
            if (lane == l) cnt += __popcll(m);
        }
    }
    cnts[wv][lane] = cnt;
    __syncthreads();
    int total = 0, before = 0;                                      // level `lane`: keypoints in the whole image / in earlier slices
    for (int w = 0; w < PERM_WAVES; w++) {
        const int c = cnts[w][lane];
        total += c;
        if (w < wv) before += c;
    }
    int run = total;
    for (int d = 1; d < 64; d <<= 1) {
        const int v = __shfl_up(run, d);
        if (lane >= d) run += v;
    }
    run += before - total;                                          // where this slice's part of level `lane` starts
    for (int i0 = i_beg; i0 < i_end; i0 += 64) {
        const int i = i0 + lane;
        const int layer = i < i_end ? pts[i].octave : -1;
        for (int l = 0; l < nlayers; l++) {
            const unsigned long long m = __ballot(layer == l);
            const int base = __shfl(run, l);
            if (layer == l) pm[base + __popcll(m & ((1ull << lane) - 1ull))] = i;
            if (lane == l) run += __popcll(m);
        }
    }
}
static const int* launch_perm(hipStream_t st, const HakBatch& b, const HakLayout& L, const hak_point* points, int max_pts, const HakKnobs& kn)
{
    const int nlayers = L.noct * L.ms;
    // mode 1: batches only -- a single image's call is launch-bound and would pay ~15 us for the extra kernel
    if (!kn.desc_sort || (kn.desc_sort == 1 && b.nimg < 8) || !b.perm || max_pts > b.perm_cap || nlayers > 64) return nullptr;
    // (row stride of perm = the call's max_pts, which is at most perm_cap)
    k_desc_perm<<<b.nimg, 64 * PERM_WAVES, 0, st>>>(b.state, points, max_pts, b.perm, nlayers);
    return b.perm;
}

void hak_launch_describe(hipStream_t st, const HakBatch& b, const HakLayout& L, const HakTables* tab,
                         hak_point* points, int max_pts, int patsize, int upright, int desc, int planned, int orient)
{
    // blocks per image: k_describe_runs is fastest with one keypoint per block (2.91 ms at 4096, 3.07 at 1024: 256 x 1080p, 2181
    // keypoints per image), the short k_orient with fewer, looping blocks (0.42 -> 0.38 ms: half of 4096 would find nothing to do)
    const int gx = max_pts < 4096 ? max_pts : 4096, gxo = max_pts < 1024 ? max_pts : 1024;
    const dim3 grido(gxo, b.nimg);
    dim3 grid(gx, b.nimg);
    const HakKnobs kn = knobs_of(b);
    const int order = kn.desc_order < 0 ? 0 : (kn.desc_order > 255 ? 255 : kn.desc_order);
    const int* perm = desc ? launch_perm(st, b, L, points, max_pts, kn) : nullptr;
    if (desc && !upright && orient) k_orient<float><<<grido, 64, 0, st>>>(b.base, b.stride, L, tab, b.state, points, max_pts, 1, 0, perm);
    if (desc && (planned && kn.desc_plan)) k_describe_runs<float><<<grid, 64, 0, st>>>(b.base, b.stride, L, tab, b.state, points, max_pts, upright, order, perm);
    else if (desc) k_describe<float><<<grid, 64, 0, st>>>(b.base, b.stride, L, tab, b.state, points, max_pts, patsize, upright, desc);
}

// FAST path: k_orient<int> always runs (it carries the sub-pixel refinement of akazed.cu:3600)
void hakf_launch_describe(hipStream_t st, const HakBatch& b, const HakLayout& L, const HakTables* tab, hak_point* points, int max_pts,
                          int patsize, int upright, int desc, int planned)
{
    // blocks per image: k_describe_runs is fastest with one keypoint per block (2.91 ms at 4096, 3.07 at 1024: 256 x 1080p, 2181
    // keypoints per image), the short k_orient with fewer, looping blocks (0.42 -> 0.38 ms: half of 4096 would find nothing to do)
    const int gx = max_pts < 4096 ? max_pts : 4096, gxo = max_pts < 1024 ? max_pts : 1024;
    const dim3 grido(gxo, b.nimg);
    dim3 grid(gx, b.nimg);
    const HakKnobs kn = knobs_of(b);
    const int order = kn.desc_order < 0 ? 0 : (kn.desc_order > 255 ? 255 : kn.desc_order);
    const int* base = reinterpret_cast<const int*>(b.base);
    const int* perm = launch_perm(st, b, L, points, max_pts, kn);
    k_orient<int><<<grido, 64, 0, st>>>(base, b.stride, L, tab, b.state, points, max_pts, desc && !upright, 0, perm);
    if (desc && (planned && kn.desc_plan)) k_describe_runs<int><<<grid, 64, 0, st>>>(base, b.stride, L, tab, b.state, points, max_pts, upright, order, perm);
    else if (desc) k_describe<int><<<grid, 64, 0, st>>>(base, b.stride, L, tab, b.state, points, max_pts, patsize, upright, desc);
}
