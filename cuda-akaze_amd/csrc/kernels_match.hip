// kernels_match.hip -- brute-force Hamming 1-NN matcher (gfx950, wave64).
//
//   cuMatch/hMatch/gHammingMatch/dHammingDistance2  akaze.cpp:55, akazed.cu:2758, 2144, 2125
//
// Reference semantics kept bit-for-bit: the train set is split into 16 residue
// classes j mod 16; each class keeps its first strict minimum; the match is
// accepted iff exactly one class attains the global minimum and it is < 96.
// Distances are over exactly the 61 descriptor bytes (D9: the reference reads
// 3 bytes past them); n2 < 16 and n2 == 0 are handled (D10).
//
// Mapping: a 256-thread block owns 16 queries x 16 residue classes.  Thread
// (q, c) keeps query q's 64-byte descriptor in 16 VGPRs and walks class c with
// 16 x (v_xor, v_bcnt) per train descriptor; classes are merged through LDS.
#include "hak_internal.h"

#define MQ 16      // queries per block
#define MC 16      // residue classes (X2 of akazed.cu:7)

__device__ __forceinline__ void load_desc(const hak_point* p, unsigned int d[16])
{
    // features start at byte 24 of the 104-byte record: 4-byte aligned
    const unsigned int* f = reinterpret_cast<const unsigned int*>(p->features);
#pragma unroll
    for (int i = 0; i < 15; i++) d[i] = f[i];
    d[15] = f[15] & 0xFFu;                      // byte 60 only; bytes 61..63 are struct padding
}

__global__ __launch_bounds__(256) void k_match(hak_point* pts1_base, const hak_point* pts2_base,
                                               const int* __restrict__ n1_dev, const int* __restrict__ n2_dev,
                                               int n1_host, int n2_host, long stride1, long stride2, int count_stride)
{
    __shared__ int sdist[MC][MQ];
    __shared__ int sidx[MC][MQ];
    const int pair = blockIdx.y;
    const int n1 = n1_dev ? n1_dev[pair * count_stride] : n1_host;
    const int n2 = n2_dev ? n2_dev[pair * count_stride] : n2_host;
    hak_point* pts1 = pts1_base + (long)pair * stride1;
    const hak_point* pts2 = pts2_base + (long)pair * stride2;
    const int q = threadIdx.x & (MQ - 1), c = threadIdx.x >> 4;     // lane = q + 16*(c%4): 4 classes per wave
    for (int q0 = blockIdx.x * MQ; q0 < n1; q0 += gridDim.x * MQ) {
        const int qi = q0 + q;
        unsigned int qd[16];
        if (qi < n1) load_desc(pts1 + qi, qd);
        int best = 1 << 30, besti = -1;
        if (qi < n1)
            for (int j = c; j < n2; j += MC) {
                unsigned int td[16];
                load_desc(pts2 + j, td);
                int dist = 0;
#pragma unroll
                for (int k = 0; k < 16; k += 2)
                    dist += __popcll(((unsigned long long)(qd[k + 1] ^ td[k + 1]) << 32) | (qd[k] ^ td[k]));
                if (dist < best) { best = dist; besti = j; }       // strict: first minimum of the class
            }
        sdist[c][q] = best;
        sidx[c][q] = besti;
        __syncthreads();
        if (c == 0 && qi < n1) {
            int bc = 0;
            for (int t = 1; t < MC; t++)
                if (sdist[t][q] < sdist[bc][q]) bc = t;
            const int dmin = sdist[bc][q];
            int nflag = 0;
            for (int t = 0; t < MC; t++) nflag += dmin < sdist[t][q] ? 1 : 0;       // akazed.cu:2206
            hak_point* p1 = pts1 + qi;
            const int bi = sidx[bc][q];
            if (bi >= 0 && nflag == MC - 1 && dmin < HAK_MAX_DIST) {                // akazed.cu:2223
                p1->match = bi;
                p1->distance = dmin;
                p1->match_x = pts2[bi].x;
                p1->match_y = pts2[bi].y;
            } else {
                p1->match = -1;
                p1->distance = -1;
                p1->match_x = -1.f;
                p1->match_y = -1.f;
            }
        }
        __syncthreads();
    }
}

void hak_launch_match(hipStream_t st, hak_point* pts1, const hak_point* pts2, const int* n1_dev, const int* n2_dev,
                      int n1_host, int n2_host, long pair_stride1, long pair_stride2, int npairs)
{
    int nq = n1_dev ? 0 : n1_host;
    int gx = n1_dev ? 640 : (nq + MQ - 1) / MQ;
    if (gx < 1) gx = 1;
    if (gx > 4096) gx = 4096;
    dim3 grid(gx, npairs);
    k_match<<<grid, 256, 0, st>>>(pts1, pts2, n1_dev, n2_dev, n1_host, n2_host, pair_stride1, pair_stride2, 2);
}
