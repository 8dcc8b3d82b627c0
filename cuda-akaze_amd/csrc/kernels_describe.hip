// kernels_describe.hip -- dominant orientation + 486-bit MLDB descriptor,
// one wave64 (one 64-thread workgroup) per keypoint, grid-strided.
//
//   hCalcOrient/gCalcOrient   akazed.cu:2655, 1665 (+ dFastAtan2 173-185)
//   hDescribe/gDescribe2      akazed.cu:2675, 1869
//
// Bit-exactness rules taken over from the reference / the parity oracle:
//   * orientation: the 109 disc samples are summed into the 42 bins in
//     ascending sample-thread order (the reference uses racy float atomics, D7)
//   * MLDB: sample i belongs to "thread" i % 64; a thread sums its samples in
//     ascending order; threads are combined as (a_t + a_{t+32}) followed by the
//     shuffle-down tree 1,2,4,8,16 (akazed.cu:1957-1981)
#include "hak_internal.h"

#define ACC_LD 65     // padded leading dimension of the per-thread accumulator table

__global__ __launch_bounds__(64) void k_describe(const float* __restrict__ base, long stride, HakLayout L,
                                                 const HakTables* __restrict__ tab, const HakImgState* __restrict__ state,
                                                 hak_point* points, int max_pts, int patsize, int upright, int desc)
{
    __shared__ float acc[90 * ACC_LD];          // [cell*3+ch][thread]
    __shared__ float vals[90];
    __shared__ float sdx[128], sdy[128];
    __shared__ int sbin[128];
    __shared__ float resx[42], resy[42];
    __shared__ float re8x[42], re8y[42];
    __shared__ float s_angle;

    const int img = blockIdx.y;
    const int lane = threadIdx.x;
    const int npts = state[img].num_pts;
    const float* arena = base + (long)img * stride;
    hak_point* pts = points + (long)img * max_pts;

    const int size2 = patsize;
    const int size3 = (int)ceilf(2.0f * patsize / 3.0f);            // akazed.cu:2682
    const int size4 = (int)ceilf(0.5f * patsize);                   // akazed.cu:2683
    const int winsize = max(3 * size3, 4 * size4);

    for (int pi = blockIdx.x; pi < npts; pi += gridDim.x) {
        hak_point* pt = pts + pi;
        const float ptx = pt->x, pty = pt->y, ptsize = pt->size;
        const int layer = pt->octave;
        const int o = layer / L.ms, s = layer - o * L.ms;
        const HakOct oc = L.oct[o];
        const float* imd = arena + L.lt(o, s);
        const float* dxd = arena + L.lx(o, s);
        const float* dyd = arena + L.ly(o, s);
        float angle = 0.f;

        if (!desc) {
            continue;
        }

        // ------------------------------------------------------ orientation
        if (!upright) {
            const int step = (int)(ptsize + 0.5f);
            const int x = (int)(ptx + 0.5f) >> o;
            const int y = (int)(pty + 0.5f) >> o;
            // the 208 sample threads of the reference, 64 at a time; valid ones
            // (r2 < 36) are compacted in ascending thread order through a ballot
            int nvalid = 0;
            for (int t0 = 0; t0 < 208; t0 += 64) {
                int tix = t0 + lane;
                int i = (tix & 15) - 6;
                int j = (tix >> 4) - 6;
                int r2 = i * i + j * j;
                bool ok = tix < 208 && r2 < 36;
                unsigned long long m = __ballot(ok);
                if (ok) {
                    int slot = nvalid + __popcll(m & ((1ull << lane) - 1ull));
                    float gw = tab->orient_w[r2];
                    int yy = min(max(y + step * j, 0), oc.h - 1), xx = min(max(x + step * i, 0), oc.w - 1);
                    long pos = (long)yy * oc.p + xx;
                    float dx = gw * dxd[pos];
                    float dy = gw * dyd[pos];
                    float ang = hak_atan2f(dy, dx);
                    int a = (int)(ang * (21 / HAK_PI_D)) + 21;      // akazed.cu:1702
                    a = a > 41 ? 41 : a;
                    a = a < 0 ? 0 : a;
                    sdx[slot] = dx;
                    sdy[slot] = dy;
                    sbin[slot] = a;
                }
                nvalid += __popcll(m);
            }
            __syncthreads();
            if (lane < 42) {
                float rx = 0.f, ry = 0.f;
                for (int n = 0; n < nvalid; n++) {
                    if (sbin[n] == lane) { rx += sdx[n]; ry += sdy[n]; }
                }
                resx[lane] = rx;
                resy[lane] = ry;
            }
            __syncthreads();
            if (lane < 42) {                                        // akazed.cu:1708-1717
                float ax = resx[lane], ay = resy[lane];
                for (int k = lane + 1; k < lane + 7; k++) {
                    ax += resx[k < 42 ? k : k - 42];
                    ay += resy[k < 42 ? k : k - 42];
                }
                re8x[lane] = ax;
                re8y[lane] = ay;
            }
            __syncthreads();
            if (lane == 0) {
                float maxr = 0.0f;
                int maxk = 0;
                for (int k = 0; k < 42; k++) {
                    float r = re8x[k] * re8x[k] + re8y[k] * re8y[k];
                    if (r > maxr) { maxr = r; maxk = k; }
                }
                // dFastAtan2 akazed.cu:173-185
                float yv = re8y[maxk], xv = re8x[maxk];
                float absx = fabsf(xv), absy = fabsf(yv);
                float mn = absx < absy ? absx : absy, mx = absx < absy ? absy : absx;
                float a = mx > 0.f ? mn / mx : 0.f;
                float sq = a * a;
                float r = fmaf(fmaf(fmaf(-0.0464964749f, sq, 0.15931422f), sq, -0.327622764f), sq * a, a);
                r = (absy > absx ? HAK_HPI_F - r : r);
                r = (xv < 0 ? (float)(HAK_PI_D - r) : r);
                r = (yv < 0 ? -r : r);
                s_angle = (r < 0.0f ? (float)(r + 2.0f * HAK_PI_D) : r);    // akazed.cu:1734
            }
            __syncthreads();
            angle = s_angle;
        }

        // ------------------------------------------------------------- MLDB
        for (int i = lane; i < 90 * ACC_LD; i += 64) acc[i] = 0.f;
        __syncthreads();
        {
            const float iratio = 1.f / (1 << o);
            const int scale = (int)(ptsize + 0.5f);
            const float xf = ptx * iratio;
            const float yf = pty * iratio;
            float si, co;
            hak_sincosf(angle, &si, &co);
            for (int i = lane; i < winsize * winsize; i += 64) {
                int y = i / winsize;
                int x = i - winsize * y;
                int m = max(x, y);
                if (m >= winsize) continue;
                int l = x - size2;
                int k = y - size2;
                int xp = (int)(xf + scale * (k * co - l * si) + 0.5f);  // akazed.cu:1921
                int yp = (int)(yf + scale * (k * si + l * co) + 0.5f);  // akazed.cu:1922
                xp = min(max(xp, 0), oc.w - 1);
                yp = min(max(yp, 0), oc.h - 1);
                long pos = (long)yp * oc.p + xp;
                float im = imd[pos];
                float dx = dxd[pos];
                float dy = dyd[pos];
                float rx = -dx * si + dy * co;
                float ry = dx * co + dy * si;
                if (m < 2 * size2) {
                    int x2 = (x < size2 ? 0 : 1);
                    int y2 = (y < size2 ? 0 : 1);
                    int c = 3 * (y2 * 2 + x2);
                    acc[c * ACC_LD + lane] += im;
                    acc[(c + 1) * ACC_LD + lane] += rx;
                    acc[(c + 2) * ACC_LD + lane] += ry;
                }
                if (m < 3 * size3) {
                    int x3 = (x < size3 ? 0 : (x < 2 * size3 ? 1 : 2));
                    int y3 = (y < size3 ? 0 : (y < 2 * size3 ? 1 : 2));
                    int c = 3 * (4 + y3 * 3 + x3);
                    acc[c * ACC_LD + lane] += im;
                    acc[(c + 1) * ACC_LD + lane] += rx;
                    acc[(c + 2) * ACC_LD + lane] += ry;
                }
                if (m < 4 * size4) {
                    int x4 = (x < 2 * size4 ? (x < size4 ? 0 : 1) : (x < 3 * size4 ? 2 : 3));
                    int y4 = (y < 2 * size4 ? (y < size4 ? 0 : 1) : (y < 3 * size4 ? 2 : 3));
                    int c = 3 * (4 + 9 + y4 * 4 + x4);
                    acc[c * ACC_LD + lane] += im;
                    acc[(c + 1) * ACC_LD + lane] += rx;
                    acc[(c + 2) * ACC_LD + lane] += ry;
                }
            }
        }
        __syncthreads();
        // transposed reduction: lane c reduces accumulator row c (and c+64) in the
        // reference's order: b_t = a_t + a_{t+32}; tree over t with strides 1,2,4,8,16
        for (int c = lane; c < 90; c += 64) {
            const float* a = acc + c * ACC_LD;
            float v[32];
#pragma unroll
            for (int t = 0; t < 32; t++) v[t] = a[t] + a[t + 32];
#pragma unroll
            for (int d = 1; d < 32; d <<= 1)
#pragma unroll
                for (int t = 0; t + d < 32; t += 2 * d) v[t] = v[t] + v[t + d];
            vals[c] = v[0];
        }
        __syncthreads();
        if (lane < HAK_FLEN) {                                      // akazed.cu:1987-1999
            unsigned int desc_r = 0;
            const int nb = lane == 60 ? 6 : 8;
            for (int i = 0; i < nb; ++i) {
                int idx1 = tab->comp1[lane * 8 + i];
                int idx2 = tab->comp2[lane * 8 + i];
                desc_r |= (vals[idx1] > vals[idx2] ? 1u : 0u) << i;
            }
            pt->features[lane] = (unsigned char)desc_r;
        }
        if (lane == 0) pt->angle = angle;
        __syncthreads();
    }
}

void hak_launch_describe(hipStream_t st, const HakBatch& b, const HakLayout& L, const HakTables* tab,
                         hak_point* points, int max_pts, int patsize, int upright, int desc)
{
    int gx = max_pts < 2048 ? max_pts : 2048;
    dim3 grid(gx, b.nimg);
    k_describe<<<grid, 64, 0, st>>>(b.base, b.stride, L, tab, b.state, points, max_pts, patsize, upright, desc);
}
