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
// Two kernels with identical results: k_match_mfma (the default, below: Hamming distances as fp4 dot products on the matrix
// cores) and k_match (HAK_MATCH_VALU=1: v_xor / v_bcnt on the vector pipe, kept for A/B runs and as the second reader of the
// accept rule -- the tests run both).
//
// k_match mapping: a 256-thread block owns 32 queries x 16 residue classes.  Thread
// (q, c) keeps the 64-byte descriptors of two queries in 32 VGPRs and walks class c with
// 16 x (v_xor, v_bcnt) per distance; the train descriptors are staged through LDS
// in tiles of 128 (coalesced, once per block, the next tile's loads in flight during the compares) and read back with broadcast
// ds_read_b128, each read serving two distances; classes are merged through LDS.
#include "hak_internal.h"
#include <cstddef>

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

#define MT 128     // train descriptors staged per LDS tile (16 per residue class)

// Train descriptors go through LDS per WAVE: a wave owns four residue classes (c0 .. c0+3) for all of the block's queries, so
// it stages exactly the descriptors it compares -- records j0 + c + 16 k, k < MT / 16 -- into its own 2 KB and no block barrier
// is needed (LDS executes a wave's operations in order: the writes of tile i+1 follow the reads of tile i).  In two halves --
// global loads into registers, LDS writes -- so that the loads of tile i+1 are in flight while tile i is compared.  Lane l
// copies dword l & 15 of class c0 + (l >> 4); byte 60 masked (bytes 61..63 are struct padding).
__device__ __forceinline__ void fetch_train(const hak_point* __restrict__ pts2, int j0, int n2, unsigned int (&pre)[MT / 16], int c, int d)
{
    // byte offsets in 32 bits from the (wave-uniform) set base: one offset VGPR per load instead of a 64-bit address each
    // (n2 < 2^20 records of 104 bytes, checked by the launcher)
    const unsigned off0 = (unsigned)(j0 + c) * (unsigned)sizeof(hak_point) + (unsigned)offsetof(hak_point, features) + 4u * d;
    const unsigned msk = d == 15 ? 0xFFu : 0xFFFFFFFFu;
#pragma unroll
    for (int k = 0; k < MT / 16; k++) {
        unsigned int v = 0;
        if (j0 + c + 16 * k < n2) v = *reinterpret_cast<const unsigned int*>(reinterpret_cast<const char*>(pts2) + (off0 + 16u * k * (unsigned)sizeof(hak_point)));
        pre[k] = v & msk;
    }
}
// wt: this wave's tile [MT / 16][4 classes][16 dwords]
__device__ __forceinline__ void put_train(const unsigned int (&pre)[MT / 16], unsigned int* wt, int lane)
{
#pragma unroll
    for (int k = 0; k < MT / 16; k++) wt[k * 64 + lane] = pre[k];
}

// v_bcnt_u32_b32 computes popcount(x) + acc in ONE instruction; left to itself the compiler takes sixteen plain popcounts
// and rebuilds the sum as a tree of v_add3 (7 extra instructions per distance)
__device__ __forceinline__ unsigned bcnt_acc(unsigned x, unsigned acc)
{
    unsigned r;
    asm("v_bcnt_u32_b32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(acc));
    return r;
}

// key = (Hamming distance << 16) | (train index - j0) of one residue-class candidate, accumulated on top of `seed` = the
// index part: 16 x (v_xor, v_bcnt) + one shift-or
__device__ __forceinline__ unsigned hamming_key(const unsigned int (&q)[16], const uint4 t0, const uint4 t1, const uint4 t2,
                                                const uint4 t3, const unsigned idx)
{
    unsigned d = bcnt_acc(q[0] ^ t0.x, 0u);
    d = bcnt_acc(q[1] ^ t0.y, d); d = bcnt_acc(q[2] ^ t0.z, d); d = bcnt_acc(q[3] ^ t0.w, d);
    d = bcnt_acc(q[4] ^ t1.x, d); d = bcnt_acc(q[5] ^ t1.y, d); d = bcnt_acc(q[6] ^ t1.z, d); d = bcnt_acc(q[7] ^ t1.w, d);
    d = bcnt_acc(q[8] ^ t2.x, d); d = bcnt_acc(q[9] ^ t2.y, d); d = bcnt_acc(q[10] ^ t2.z, d); d = bcnt_acc(q[11] ^ t2.w, d);
    d = bcnt_acc(q[12] ^ t3.x, d); d = bcnt_acc(q[13] ^ t3.y, d); d = bcnt_acc(q[14] ^ t3.z, d); d = bcnt_acc(q[15] ^ t3.w, d);
    return (d << 20) | idx;
}

// Mapping: a 256-thread block owns 16 * NQ queries x 16 residue classes; thread (q, c) keeps the descriptors of NQ queries
// (q, q + 16) in 16 * NQ VGPRs and walks class c: with NQ = 2 every train descriptor read from LDS (four broadcast
// ds_read_b128) serves two distances.  Two train descriptors are in flight per iteration.  The class minimum is tracked as
// one packed key (distance << 20 | train index): an unsigned minimum keeps the smallest distance and, among equal
// distances, the smallest index -- the reference's "first strict minimum in ascending order" (akazed.cu:2176-2187) -- in
// one v_min_u32 instead of compare + two selects.  Requires n2 < 2^20 (checked by the launcher).
// NQ = 1 is used when NQ = 2 would leave the chip with fewer than ~2 blocks per CU (one big pair, e.g. 10k x 10k).
// gkey != nullptr (one big pair, blockIdx.y = train slice): the block walks only tiles [slice * tiles_per_slice, ...) and
// merges its class minima into gkey[query][class] with atomicMin -- the packed key makes the merge order-independent (smaller
// distance first, then smaller index) -- and k_match_finish applies the accept rule; the chip then sees slices x query
// blocks workgroups instead of n1 / 16.
template <int NQ>
__global__ __launch_bounds__(256, 4) void k_match(hak_point* pts1_base, const hak_point* pts2_base,
                                               const int* __restrict__ n1_dev, const int* __restrict__ n2_dev,
                                               int n1_host, int n2_host, long stride1, long stride2, int count_stride,
                                               unsigned* __restrict__ gkey, int tiles_per_slice)
{
    constexpr int QB = MQ * NQ;                          // queries per block
    __shared__ unsigned skey[MC][QB];
    __shared__ __attribute__((aligned(16))) unsigned int tile[MT * 16];   // four per-wave tiles of MT / 16 x 4 classes x 16 dwords
    const int pair = gkey ? 0 : blockIdx.y;
    const int n1 = n1_dev ? n1_dev[pair * count_stride] : n1_host;
    const int n2 = n2_dev ? n2_dev[pair * count_stride] : n2_host;
    hak_point* pts1 = pts1_base + (long)pair * stride1;
    const hak_point* pts2 = pts2_base + (long)pair * stride2;
    const int jbeg = gkey ? (int)blockIdx.y * tiles_per_slice * MT : 0;
    const int jend = gkey ? min(n2, jbeg + tiles_per_slice * MT) : n2;
    const int q = threadIdx.x & (MQ - 1), c = threadIdx.x >> 4;     // lane = q + 16*(c%4): 4 classes per wave
    const int lane = threadIdx.x & 63;
    unsigned int* wt = tile + (threadIdx.x >> 6) * (MT * 4);         // this wave's tile
    for (int q0 = blockIdx.x * QB; q0 < n1; q0 += gridDim.x * QB) {
        unsigned int qd[NQ][16];
        unsigned best[NQ];
#pragma unroll
        for (int a = 0; a < NQ; a++) {
#pragma unroll
            for (int k = 0; k < 16; k++) qd[a][k] = 0;
            if (q0 + q + MQ * a < n1) load_desc(pts1 + q0 + q + MQ * a, qd[a]);
            best[a] = 0xFFFFFFFFu;
        }
        unsigned int pre[MT / 16];
        if (jbeg < jend) fetch_train(pts2, jbeg, n2, pre, c, lane & 15);
        for (int j0 = jbeg; j0 < jend; j0 += MT) {
            __builtin_amdgcn_wave_barrier();                        // (no instruction: the other lanes' reads of the previous tile stay
            put_train(pre, wt, lane);                               //  in front of these writes, and the reads below behind them --
            __builtin_amdgcn_wave_barrier();                        //  the hardware runs a wave's LDS operations in order anyway)
            if (j0 + MT < jend) fetch_train(pts2, j0 + MT, n2, pre, c, lane & 15);   // next tile: lands during the compares below
            const int jn = min(MT, n2 - j0);
            const unsigned int* mine = wt + (lane >> 4) * 16;       // this lane's class within the wave's tile
            int k = 0;                                              // j = j0 + c + 16 k keeps the residue class: MT % MC == 0
            for (; c + MC * (k + 1) < jn; k += 2) {
                const uint4* t4 = reinterpret_cast<const uint4*>(mine + k * 64);
                const uint4* u4 = reinterpret_cast<const uint4*>(mine + (k + 1) * 64);
                const uint4 t0 = t4[0], t1 = t4[1], t2 = t4[2], t3 = t4[3];   // 16 lanes read the same 16 bytes: LDS broadcast
                const uint4 u0 = u4[0], u1 = u4[1], u2 = u4[2], u3 = u4[3];
#pragma unroll
                for (int a = 0; a < NQ; a++) {
                    best[a] = min(best[a], hamming_key(qd[a], t0, t1, t2, t3, (unsigned)(j0 + c + MC * k)));
                    best[a] = min(best[a], hamming_key(qd[a], u0, u1, u2, u3, (unsigned)(j0 + c + MC * (k + 1))));
                }
            }
            if (c + MC * k < jn) {
                const uint4* t4 = reinterpret_cast<const uint4*>(mine + k * 64);
                const uint4 t0 = t4[0], t1 = t4[1], t2 = t4[2], t3 = t4[3];
#pragma unroll
                for (int a = 0; a < NQ; a++) best[a] = min(best[a], hamming_key(qd[a], t0, t1, t2, t3, (unsigned)(j0 + c + MC * k)));
            }
        }
        if (gkey) {                                                 // (uniform) sliced search: merge, k_match_finish decides
#pragma unroll
            for (int a = 0; a < NQ; a++)
                if (q0 + q + MQ * a < n1 && best[a] != 0xFFFFFFFFu) atomicMin(&gkey[(long)(q0 + q + MQ * a) * MC + c], best[a]);
            continue;
        }
#pragma unroll
        for (int a = 0; a < NQ; a++) skey[c][q + MQ * a] = best[a];
        __syncthreads();
        if (c < NQ) {                                               // class-c threads finish queries q + 16 c
            const int qq = q + MQ * c, qi = q0 + qq;
            if (qi < n1) {
                // distances only (key >> 20): the accept rule compares class minima, not indices (akazed.cu:2190-2223)
                int bc = 0;
                for (int t = 1; t < MC; t++)
                    if ((skey[t][qq] >> 20) < (skey[bc][qq] >> 20)) bc = t;
                const unsigned kmin = skey[bc][qq];
                const int dmin = (int)(kmin >> 20);
                int nflag = 0;
                for (int t = 0; t < MC; t++) nflag += (unsigned)dmin < (skey[t][qq] >> 20) ? 1 : 0;   // akazed.cu:2206
                hak_point* p1 = pts1 + qi;
                const int bi = (int)(kmin & 0xFFFFFu);
                if (kmin != 0xFFFFFFFFu && nflag == MC - 1 && dmin < HAK_MAX_DIST) {  // akazed.cu:2223
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
        }
        __syncthreads();
    }
}

// ---- matrix-core variant.  Hamming distances are dot products: |a ^ b| = sum_k a_k (1 - 2 b_k) + |b|.  The train side supplies
// x_k = a_k, the query side y_k = 1 - 2 b_k (+1 / -1), both as fp4 (E2M1) operands of v_mfma_f32_32x32x64_f8f6f4 (the scale
// arguments 0 select the unscaled form): EIGHT matrix instructions of 32 cycles leave the exact distances of 32 train x 32 query
// descriptors (K = 512 >= 486 bits) in the accumulator -- 1 024 distances for 8 x 32 cycles of the matrix pipe instead of
// 1 024 x 32 VALU lane-operations (v_bcnt_u32_b32 is a slow-rate instruction on this part: the VALU kernel k_match above already
// sits near what the vector pipe can do with it).  Rounds 2-4 ran this on v_mfma_i32_32x32x32_i8 (a byte per k: sixteen
// instructions per tile and one v_and_b32 per FOUR k); a nibble per k halves both the matrix time and the expansion:
// 10k x 10k 0.0636 -> 0.0488 ms, the match class of the 256-pair sequence 0.51 -> 0.34 ms (same box, profiles/r04_match10k.txt).
//   E2M1 codes: 0001 = 0.5, 0010 = 1, 0100 = 2, 1000 = the sign.  To make the train side's bit -> fp4 expansion ONE v_and_b32 per
//   eight k, bit t < 3 of a nibble stays where it is (x_k = a_k 2^(t-1), mask 0x11111111 << t) and bit 3 comes down to bit 2 from a
//   staged copy shifted right by 1 (x_k = 2 a_k); the query side is y_k = (1 - 2 b_k) 2^(2-t) (1 for bit 3): every product is
//   +-2 a_k (round 4: 2^(1-t), +-a_k).  tools/fp4_hamming_probe.hip checks this arithmetic against popcounts on the device.
//   Round 4: the accumulator started at 2^23 + |b| and ended as 0x4B000000 + distance; the key epilogue attached the train index
//   to every distance (one v_lshl_add_u32 + one v_min_u32 per element).  Round 5: the accumulator comes out of the matrix unit as
//   the finished key, index included (MM_BASE below) -- one v_min3_u32 per element and PAIR of tiles.
//   A (train, rows m): lane (r = l & 31, h = l >> 5), k-step s < 8: dword 8 h + s of descriptor j0 + r and its copy >> 1
//   B (query, cols n): the same dword of query q0 + r; expanded ONCE per wave into 32 VGPRs
//   C/D: lane (n = l & 31, h) holds train rows (i & 3) + 8 (i >> 2) + 4 h, i < 16, of query n (cdna_hip_programming.md 158)
// Any assignment of descriptor bits to k positions is fine as long as A and B agree: both come from mm_* below.  All rows a register
// slot ever sees are congruent mod 16, i.e. one residue class -- so the reference's "first strict minimum per class" survives as
// the minimum of packed keys exactly as in k_match.  A wave = 32 queries x the whole train set (or its slice); the four waves of a
// block share the train descriptors through LDS.
//
// Rounds 4-5: the train set travels in CHUNKS of MM_CH descriptors (rows of 144 bytes: per lane half its 8 descriptor dwords and the
// same shifted right by 1, padding -- conflict-free ds_read_b128), double-buffered: the 8-byte loads of chunk c+1 (MM_CH / 32 per
// thread, all in flight at once; a lane fetching ITS descriptor row straight from the 104-byte records touches 32 lines per load
// instruction and ran at a third of the matrix pipe's rate) are issued before the tiles of chunk c are multiplied and written to
// the other buffer after them: ONE barrier and ONE exposed memory round trip per chunk.  Round 3 staged tile by tile (a barrier and
// a dependent load per 32 rows, two tiles ahead); for one big pair that loop was latency-bound.
// What bounds the kernel (round 5: measured with wall-clock stamps inside it, tools/mm_timing.py on a -DHAK_MM_TIMING build;
// 10k x 10k = 474 blocks, all resident, every block starts within 1.6 us):
//   a block lives 22.5 us: 3.3 us until its first chunk is in LDS (cold query + train loads), 9 chunks x 1.9-2.0 us, 1.5 us for its
//   summary + ticket; the block that draws the last ticket of a query block merges the slices and ends 3-6 us later (the kernel's
//   tail: 40 % of the blocks end after 25 us, the last at 29.7);
//   a chunk = 48 matrix instructions per wave = 1.28 us of matrix time per SIMD with its two waves at 2.4 GHz: the loop runs at
//   0.65 of the matrix rate.  What it still issues per matrix instruction: 4 v_and (the bit -> fp4 expansion), 1 v_min3_u32, the
//   LDS reads and the staging of the next chunk -- at the edge of the 6 vector instructions that are free beside one MFMA
//   (tools/probes/mfma_valu_overlap.hip).
// Round 4 read its instruction counters as "vector-issue-bound, 11 per MFMA": that was the int8 kernel; experiments that removed
// the epilogue also removed the matrix instructions (dead code) and said nothing.  What round 5 changed, in measured order:
//   the key out of the accumulator (8.25 -> 5.25 vector instructions per MFMA), three accumulator sets with the minima pinned
//   between the matrix instructions (the compiler otherwise reassociates them to the chunk's end and keeps six sets alive),
//   two chunks in flight and the next chunk staged beside the tiles instead of behind them (barrier wait 0.55 -> 0.08 us per
//   chunk -- but the chunk got as much longer: the CU is short of issue slots, not waiting), the merge's loads in one round trip:
//   kernel 37.9 -> 30.5 us, call 0.0437 -> 0.0409 ms.  Still open: a 64-query wave (each expanded train fragment feeds two
//   matrix instructions: 2 v_and per MFMA) and staggering the two resident blocks so that one's prologue meets the other's loop.
typedef int mm_v4i __attribute__((ext_vector_type(4)));
typedef int mm_v8i __attribute__((ext_vector_type(8)));
typedef float mm_v16f __attribute__((ext_vector_type(16)));
#define MM_MFMA(a, b, c) __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 4, 4, 0, 0, 0, 0)     // cbsz = blgp = 4: fp4 operands
#define MM_BCH 192      // train descriptors per LDS chunk = six tiles (two 27 KB buffers per block, two blocks per CU)
// Round 5: the accumulator is the key.  It starts at 2^10 + 2 |b| + q 2^-13 (q = the chunk's number, < 2048) and every descriptor
// bit contributes +-2, so it ends as 2^10 + 2 d + P / 4 + q 2^-13, where P / 4 comes out of the matrix unit as well: three k
// positions that hold struct padding in every descriptor (bits 8, 12, 16 of dword 15) carry the staged row's tile number P < 8 on
// the train side (fp4 value 0.5 each) against magnitudes 0.5, 1, 2 on the query side.  Every partial sum is a multiple of 2^-13
// in [2^10, 2^11): exact in fp32.  Bit pattern - bits(2^10) = d << 14 | P << 11 | q: ordered like (distance, train index) for the
// rows one accumulator slot sees (the kernel feeds the rows so that their index grows with (P, q)), hence min over the raw bit
// patterns = the reference's first minimum, and the index is decoded once per query block instead of attached to every distance.
#define MM_BASE 1024.0f
#define MM_BASE_BITS 0x44800000u
#define MM_QSTEP 0.0001220703125f        /* 2^-13 */
#ifndef MM_QT_DEFAULT
#define MM_QT_DEFAULT 1                  /* query tiles per wave (HAK_MATCH_QT overrides per call; 2 measured slower, see k_match_mfma) */
#endif
#define MM_MAX_ROWS (2047 * MM_BCH)     /* train rows one block pass can number (q < 2048): larger sets take the vector-pipe kernels */
#define MM_ROW 36       // dwords per staged train row: [w0..w7 | w0..w7 >> 1 | w8..w15 | w8..w15 >> 1 | 4 of padding]: lane half h reads its
                        // 16 dwords at 16 h (no shift in the tile loop, where every VALU instruction counts); 144-byte rows make the
                        // ds_read_b128 conflict-free
// train fragment of one descriptor dword w and its copy w1 = w >> 1
__device__ __forceinline__ mm_v8i mm_frag_a4(unsigned w, unsigned w1)
{
    mm_v8i f = {0, 0, 0, 0, 0, 0, 0, 0};
    f[0] = (int)(w & 0x11111111u); f[1] = (int)(w & 0x22222222u); f[2] = (int)(w & 0x44444444u); f[3] = (int)(w1 & 0x44444444u);
    return f;
}
// query fragment: magnitude code of 2^(2-t) (1 for bit 3: E2M1 codes 6, 4, 2, 2 = 4, 2, 1, 1), sign = the descriptor bit; against the
// train side's 2^(t-1) (2 for bit 3) every set train bit contributes +-2
__device__ __forceinline__ mm_v4i mm_frag_b4(unsigned q)
{
    mm_v4i f;
    f.x = (int)(0x66666666u | ((q & 0x11111111u) << 3));
    f.y = (int)(0x44444444u | ((q & 0x22222222u) << 2));
    f.z = (int)(0x22222222u | ((q & 0x44444444u) << 1));
    f.w = (int)(0x22222222u | (q & 0x88888888u));
    return f;
}

// accept rule of gHammingMatch on a query's 16 class minima (akazed.cu:2190-2223): distances only decide (key >> 20) -- the rule
// compares class minima, not indices; ties between classes never matter for the result (nflag rejects them)
__device__ __forceinline__ void mm_accept(hak_point* p1, const hak_point* __restrict__ pts2, const unsigned (&all)[16], const int n2)
{
    unsigned kmin = all[0];
#pragma unroll
    for (int t = 1; t < 16; t++)
        if ((all[t] >> 20) < (kmin >> 20)) kmin = all[t];
    const int dmin = (int)(kmin >> 20);
    int nflag = 0;
#pragma unroll
    for (int t = 0; t < 16; t++) nflag += (unsigned)dmin < (all[t] >> 20) ? 1 : 0;                        // akazed.cu:2206
    const int bi = min((int)(kmin & 0xFFFFFu), max(n2 - 1, 0));           // (always the index itself: belt and braces for the gather below)
    if (kmin != 0xFFFFFFFFu && nflag == MC - 1 && dmin < HAK_MAX_DIST) {                             // akazed.cu:2223
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

// Sliced search of one big pair (blockIdx.y = train slice, `ticket` != nullptr): a block merges what it found into scratch and
// takes a ticket for its query block; the block that draws the LAST ticket finishes the 128 queries -- no memset in front, no
// finish kernel behind.  The finishing block resets the ticket, so the scratch is in its initial state when the kernel ends.
//   1-NN: part[slice][query] = {this slice's smallest key, mask of the residue classes that attain its distance}
//   2-NN: part[slice][query] = this slice's two smallest keys
// (plain 8-byte stores: no initial state to restore)
// KNN = true: the 2-NN search of hak_match_knn2 on the same tiles -- per register slot the two smallest keys (min / max / min per
// element), nearest neighbour = the smallest key of all slots (smallest index among equal distances), d2 = the smallest
// distance of every OTHER train point; results go to knn_out[query] = {index, d1, d2, 0} instead of the point records.
#ifdef HAK_MM_TIMING
// variant builds only (tools/build_variant_one.sh ... -DHAK_MM_TIMING): shader-clock stamps of block (0, 0)'s wave 0 at the phase
// boundaries of its first query group
__device__ unsigned long long hak_mm_times[64];
__device__ unsigned long long hak_mm_blk[2][1024];          // every block's start / end stamp
extern "C" int hak_debug_mm_times(unsigned long long* host) { return hipMemcpyFromSymbol(host, HIP_SYMBOL(hak_mm_times), sizeof(hak_mm_times)) != hipSuccess; }
extern "C" int hak_debug_mm_blocks(unsigned long long* host) { return hipMemcpyFromSymbol(host, HIP_SYMBOL(hak_mm_blk), sizeof(hak_mm_blk)) != hipSuccess; }
#define MM_BLK(k) do { const unsigned b_ = blockIdx.y * gridDim.x + blockIdx.x; if (threadIdx.x == 0 && b_ < 1024) hak_mm_blk[k][b_] = wall_clock64(); } while (0)
#define MM_STAMP(i) do { if (blockIdx.x == 0 && blockIdx.y == HAK_MM_TIMING && threadIdx.x == 0 && (i) < 64) hak_mm_times[i] = wall_clock64(); } while (0)
#else
#define MM_STAMP(i) do { } while (0)
#define MM_BLK(k) do { } while (0)
#endif
// QT = query tiles per wave (round 5): with QT = 2 a wave owns 64 queries, every expanded train fragment feeds TWO matrix
// instructions (2 v_and per MFMA instead of 4, half the LDS reads per MFMA, two independent accumulation chains), a block 256
// queries, and the kernel runs ONE wave per SIMD out of 485 registers; QT = 1 is the 32-query wave at two waves per SIMD.
// MEASURED: QT = 2 is SLOWER (10k x 10k 0.0439 vs 0.0410 ms; a chunk of 96 matrix instructions per SIMD takes 2.7-2.9 us against
// 1.9-2.0 us with two waves of 48): beyond 256 registers the accumulators live in AccVGPRs, every minimum first copies its two
// operands out (v_accvgpr_read: 2 more vector instructions per MFMA, 5 in all again), and a lone wave cannot issue its vector
// instructions beside its own dependent matrix chain the way a second wave does.  QT = 1 is the default; HAK_MATCH_QT=2 selects
// the other instantiation (same results: the match tests run both).
template <bool KNN, int MM_CH, int QT>
__global__ __launch_bounds__(256, (QT == 2 ? 1 : 2)) void k_match_mfma(hak_point* __restrict__ pts1_base, const hak_point* __restrict__ pts2_base,
                                                       const int* __restrict__ n1_dev, const int* __restrict__ n2_dev,
                                                       int n1_host, int n2_host, long stride1, long stride2, int count_stride,
                                                       int rows_per_slice,
                                                       int4* __restrict__ knn_out_base, long knn_stride,
                                                       int* ticket, uint2* part, int n1_pad)
{
    constexpr int QPB = 128 * QT;                                            // queries per block
    // unsliced: grid (query blocks, pairs); sliced: grid (query blocks, slices, pairs)
    const bool sliced = ticket != nullptr;
    MM_STAMP(0);
    MM_BLK(0);
    const int pair = sliced ? blockIdx.z : blockIdx.y;
    const int n1 = n1_dev ? n1_dev[pair * count_stride] : n1_host;
    const int n2 = n2_dev ? n2_dev[pair * count_stride] : n2_host;
    hak_point* pts1 = pts1_base + (long)pair * stride1;
    const hak_point* __restrict__ pts2 = pts2_base + (long)pair * stride2;
    // rows_per_slice <= 0: the train count lives on the device (batched pairs): equal slices of whole 32-row tiles, cut here
    const int rps = rows_per_slice > 0 ? rows_per_slice : (((n2 + 31) >> 5) + (int)gridDim.y - 1) / (int)gridDim.y * 32;
    const int jbeg = sliced ? (int)blockIdx.y * rps : 0;                     // (multiples of 32: rows keep their residue class)
    const int jend = sliced ? min(n2, jbeg + rps) : n2;
    if (sliced) {                                                            // this pair's part of the scratch
        ticket += (long)pair * gridDim.x;
        part += (long)pair * gridDim.y * n1_pad;
    }
    const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    __shared__ __attribute__((aligned(16))) unsigned int tile[2][MM_CH * MM_ROW];
    __shared__ int s_last;
    // thread t fetches dwords 2 (t & 7), 2 (t & 7) + 1 of train descriptors j0 + (t >> 3) + 32 i, i < 6: one chunk, all six
    // loads in flight together; byte offsets in 32 bits from the (uniform) set base (n2 < 2^20 records, checked by the launcher)
    const unsigned toff = (unsigned)(threadIdx.x >> 3) * (unsigned)sizeof(hak_point) + (unsigned)offsetof(hak_point, features) +
                          8u * (threadIdx.x & 7);

    for (int qb = blockIdx.x * QPB; qb < n1; qb += gridDim.x * QPB) {                   // block-uniform: every wave takes part in the staging
        const int q0 = qb + 32 * QT * wv;                                               // (a wave past n1 computes on zeros and stores nothing)
        // ---- the order in which the slice's rows meet the matrix unit (round 5).  The slice [jbeg, jend) is cut into a MAIN part of
        // nmain = 192 NQ rows and a tail of < 192 rows.  The main part is six SIXTHS of E = 32 NQ rows; chunk q (192 rows = 6 tiles)
        // holds tile q of every sixth: tile P of chunk q = rows jbeg + P E + 32 q .. + 31.  For a fixed accumulator slot the train
        // index therefore grows with (P, q) lexicographically -- and that pair is what the accumulator itself carries below its
        // distance (see the header: P through three spare k positions of the staged row, q through the C operand), so the whole
        // key epilogue of a tile is ONE unsigned minimum per element, shared between two tiles by v_min3_u32.
        const int nrows = max(jend - jbeg, 0);
        const int NQ = nrows / MM_CH;                                                   // chunks of the main part (uniform)
        const int E = 32 * NQ;                                                          // rows per sixth
        const int jtail = jbeg + MM_CH * NQ;                                            // first row of the tail
        // two chunks travel at a time: chunk q + 2 is requested when chunk q starts and written to LDS when chunk q + 1 starts (round 5:
        // with one chunk in flight the loop was a chain of exposed memory round trips).  pre[c & 1] holds chunk c; the chunk loop is
        // unrolled by two, so the index is static.
        uint2 pre[2][MM_CH / 32];
        const int NC = NQ + (jtail < jend ? 1 : 0);                                     // chunks incl. the tail (uniform)
        // chunk c: a main chunk (every row exists) or, c == NQ, the tail (consecutive tiles, rows past jend do not exist)
        auto fetch = [&](int c, auto par) {
            constexpr int PAR = decltype(par)::value;
            if (c < NQ) {
#pragma unroll
                for (int i = 0; i < MM_CH / 32; i++)
                    pre[PAR][i] = *reinterpret_cast<const uint2*>(reinterpret_cast<const char*>(pts2) +
                                                                  ((unsigned)(jbeg + i * E + 32 * c) * (unsigned)sizeof(hak_point) + toff));
            } else {
#pragma unroll
                for (int i = 0; i < MM_CH / 32; i++) {
                    uint2 v = make_uint2(0u, 0u);
                    if (jtail + (int)(threadIdx.x >> 3) + 32 * i < jend)
                        v = *reinterpret_cast<const uint2*>(reinterpret_cast<const char*>(pts2) +
                                                            ((unsigned)(jtail + 32 * i) * (unsigned)sizeof(hak_point) + toff));
                    pre[PAR][i] = v;                                // (raw: masking here would make the wave wait for the load at once)
                }
            }
        };
        using mm_c0 = std::integral_constant<int, 0>;
        using mm_c1 = std::integral_constant<int, 1>;
        if (NC > 0) fetch(0, mm_c0{});                              // the first chunks travel while the queries are loaded and expanded
        if (NC > 1) fetch(1, mm_c1{});
        mm_v4i B[QT][8];
        float c0[QT];                                               // 2^10 + 2 |b|: the accumulators' start without the chunk number
#pragma unroll
        for (int t = 0; t < QT; t++) {
            // the lane half's eight dwords 8 h .. 8 h + 7 of query q0 + 32 t + r (features start at byte 24 of the record: 8-byte aligned)
            unsigned int qd[8];
#pragma unroll
            for (int i = 0; i < 8; i++) qd[i] = 0;
            if (q0 + 32 * t + r < n1) {
                const uint2* f = reinterpret_cast<const uint2*>(reinterpret_cast<const char*>(pts1 + q0 + 32 * t + r) + offsetof(hak_point, features)) + 4 * h;
#pragma unroll
                for (int i = 0; i < 4; i++) { const uint2 v = f[i]; qd[2 * i] = v.x; qd[2 * i + 1] = v.y; }
                if (h) qd[7] &= 0xFFu;                              // byte 60 only; bytes 61..63 are struct padding
            }
            unsigned pb = 0;
#pragma unroll
            for (int i = 0; i < 8; i++) pb = bcnt_acc(qd[i], pb);
            pb += (unsigned)__shfl_xor((int)pb, 32);
#pragma unroll
            for (int s = 0; s < 8; s++) B[t][s] = mm_frag_b4(qd[s]);
            // the three index positions (bits 8, 12, 16 of descriptor dword 15: struct padding, zero in every query): magnitudes 0.5,
            // 1, 2 against the staged row's 0.5 -> P / 4
            if (h) B[t][7].x = (int)(((unsigned)B[t][7].x & ~0x000FFF00u) | 0x00042100u);
            c0[t] = MM_BASE + 2.0f * (float)pb;
        }
        unsigned best[QT][16], sec[QT][KNN ? 16 : 1];
#pragma unroll
        for (int t = 0; t < QT; t++) {
#pragma unroll
            for (int i = 0; i < 16; i++) best[t][i] = 0xFFFFFFFFu;
#pragma unroll
            for (int i = 0; i < (KNN ? 16 : 1); i++) sec[t][i] = 0xFFFFFFFFu;
        }
        // the lane's 16 dwords of row r of tile K of chunk buffer BUF -> TD
#define MM_READ(BUF, K, TD)                                                                                 \
        {                                                                                                   \
            const uint4* row = reinterpret_cast<const uint4*>(tile[BUF] + (32 * (K) + r) * MM_ROW + 16 * h); \
            _Pragma("unroll") for (int c = 0; c < 4; c++) {                                                 \
                const uint4 v = row[c];                                                                     \
                TD[4 * c] = v.x; TD[4 * c + 1] = v.y; TD[4 * c + 2] = v.z; TD[4 * c + 3] = v.w;             \
            }                                                                                               \
        }
        // two finished tiles X, Y of query tile T -> the slot's smallest (and, 2-NN, second smallest) key: one v_min3_u32 per element
        // and pair of tiles (2-NN: five instructions).  The accumulators ARE the keys: positive floats order like their bit patterns.
#define MM_EPI2(T, X, Y, I)                                                                                 \
        {                                                                                                   \
            const unsigned ka = __float_as_uint(X[T][I]), kb = __float_as_uint(Y[T][I]);                    \
            if constexpr (KNN) {                                                                            \
                const unsigned lo = min(ka, kb), hi = max(ka, kb);                                          \
                sec[T][I] = min(min(sec[T][I], hi), max(best[T][I], lo));                                   \
                best[T][I] = min(best[T][I], lo);                                                           \
                asm volatile("" : "+v"(sec[T][I]));                                                         \
            } else best[T][I] = min(min(best[T][I], ka), kb);                                               \
            /* (the minimum is associative and the compiler knows it: left alone it keeps SIX accumulator sets alive and takes   \
               all minima of a chunk at its end -- 96 registers and no overlap with the matrix instructions) */                  \
            asm volatile("" : "+v"(best[T][I]));                                                            \
        }
        // the eight matrix instructions per query tile of one train tile (rows from TD) into CUR, with the epilogue of the finished
        // tiles EX, EY between them when EPI.  The train fragment is expanded once and feeds all QT query tiles.
#define MM_PIPE(TD, CUR, EPI, EX, EY)                                                                       \
        {                                                                                                   \
            _Pragma("unroll") for (int s = 0; s < 8; s++) {                                                 \
                const mm_v8i af = mm_frag_a4(TD[s], TD[8 + s]);                                             \
                _Pragma("unroll") for (int t = 0; t < QT; t++) {                                            \
                    const mm_v4i bs = B[t][s];                                                              \
                    const mm_v8i bf = {bs.x, bs.y, bs.z, bs.w, 0, 0, 0, 0};                                 \
                    if (s == 0) CUR[t] = MM_MFMA(af, bf, cinit[t]);                                         \
                    else CUR[t] = MM_MFMA(af, bf, CUR[t]);                                                  \
                }                                                                                           \
                if (EPI) {                                                                                  \
                    _Pragma("unroll") for (int t = 0; t < QT; t++) { MM_EPI2(t, EX, EY, 2 * s) MM_EPI2(t, EX, EY, 2 * s + 1) } \
                }                                                                                           \
                __builtin_amdgcn_sched_barrier(0);                                                          \
            }                                                                                               \
        }
        // rows of chunk buffer c & 1: [w0..w7 | w0..w7 >> 1 | w8..w15 | w8..w15 >> 1 | pad]: lane half h reads its 16 dwords at 16 h.
        // The thread that holds dword 15 plants the tile's number P (tail: 7) into its bits 8, 12, 16.
        auto stage = [&](int c, auto par) {                         // chunk c (held in pre[c & 1]) -> tile[c & 1]
            constexpr int PAR = decltype(par)::value;
            const bool tail = c >= NQ;
#pragma unroll
            for (int i = 0; i < MM_CH / 32; i++) {
                const unsigned P = tail ? 7u : (unsigned)i;
                const unsigned pbits = ((P & 1u) << 8) | ((P & 2u) << 11) | ((P & 4u) << 14);
                const uint2 v = make_uint2(pre[PAR][i].x, (threadIdx.x & 7) == 7 ? (pre[PAR][i].y & 0xFFu) | pbits : pre[PAR][i].y);
                unsigned int* row = tile[PAR] + ((threadIdx.x >> 3) + 32 * i) * MM_ROW + 16 * ((threadIdx.x & 7) >> 2) + 2 * (threadIdx.x & 3);
                *reinterpret_cast<uint2*>(row) = v;
                *reinterpret_cast<uint2*>(row + 8) = make_uint2(v.x >> 1, v.y >> 1);
            }
        };
        __syncthreads();                                            // (the previous query group's chunks have been read)
        MM_STAMP(1);
        if (NC > 0) stage(0, mm_c0{});
        __syncthreads();
        MM_STAMP(2);
        mm_v16f cinit[QT], X0[QT], X1[QT], X2[QT];                  // three accumulator sets per query tile: train tile k lives in set k mod 3
#pragma unroll
        for (int t = 0; t < QT; t++) {
#pragma unroll
            for (int i = 0; i < 16; i++) cinit[t][i] = c0[t];
            X0[t] = cinit[t]; X1[t] = cinit[t]; X2[t] = cinit[t];
        }
        bool pend = false;                                          // X1, X2 hold the last two tiles of the previous chunk (uniform)
        // one main chunk q out of tile[q & 1]: six tiles; the minima of tiles (4, 5) of the previous chunk ride on tile 0, those of
        // (0, 1) on tile 2, those of (2, 3) on tile 4
        auto chunk = [&](int q, auto par) {
            constexpr int PAR = decltype(par)::value;
            using nxt = std::integral_constant<int, PAR ^ 1>;
            // chunk q + 1 (requested two chunks ago) goes to the other buffer FIRST -- its last readers passed the barrier at the end of
            // chunk q - 1 -- so that the LDS writes run beside this chunk's matrix instructions and the barrier below only collects
            // stragglers; then chunk q + 2 is requested
            if (q + 1 < NC) stage(q + 1, nxt{});
            if (q + 2 < NC) fetch(q + 2, par);                      // (pre[PAR] went to LDS one chunk ago)
            unsigned int ta[16], tb[16];
            MM_READ(PAR, 0, ta)
            MM_READ(PAR, 1, tb)
            if (pend) MM_PIPE(ta, X0, true, X1, X2)
            else MM_PIPE(ta, X0, false, X1, X2)
            MM_READ(PAR, 2, ta)
            MM_PIPE(tb, X1, false, X0, X0)
            MM_READ(PAR, 3, tb)
            MM_PIPE(ta, X2, true, X0, X1)
            MM_READ(PAR, 4, ta)
            MM_PIPE(tb, X0, false, X1, X1)
            MM_READ(PAR, 5, tb)
            MM_PIPE(ta, X1, true, X2, X0)
            MM_PIPE(tb, X2, false, X0, X0)
            pend = true;
#pragma unroll
            for (int t = 0; t < QT; t++) {
#pragma unroll
                for (int i = 0; i < 16; i++) cinit[t][i] += MM_QSTEP;   // the next chunk's number (exact: one ulp of the accumulators' binade)
            }
            MM_STAMP(3 + 2 * q);
            __syncthreads();
            MM_STAMP(4 + 2 * q);
        };
        for (int q = 0; q < NQ; q += 2) {
            chunk(q, mm_c0{});
            if (q + 1 < NQ) chunk(q + 1, mm_c1{});
        }
        if (pend) {
#pragma unroll
            for (int t = 0; t < QT; t++) {
#pragma unroll
                for (int i = 0; i < 16; i++) MM_EPI2(t, X1, X2, i)
            }
        }
        // the tail: up to six consecutive tiles, tile k carries (P, q) = (7, k); rows past jend do not exist
        if (jtail < jend) {
            const int nt = (jend - jtail + 31) >> 5;
            const unsigned int* tbase = tile[NQ & 1];
            for (int k = 0; k < nt; k++) {
                unsigned int ta[16];
                {
                    const uint4* row = reinterpret_cast<const uint4*>(tbase + (32 * k + r) * MM_ROW + 16 * h);
#pragma unroll
                    for (int c = 0; c < 4; c++) { const uint4 v = row[c]; ta[4 * c] = v.x; ta[4 * c + 1] = v.y; ta[4 * c + 2] = v.z; ta[4 * c + 3] = v.w; }
                }
#pragma unroll
                for (int t = 0; t < QT; t++) {
#pragma unroll
                    for (int i = 0; i < 16; i++) cinit[t][i] = c0[t] + (float)k * MM_QSTEP;
                }
                MM_PIPE(ta, X0, false, X0, X0)
                const int jb = jtail + 32 * k + 4 * h;
#pragma unroll
                for (int t = 0; t < QT; t++) {
#pragma unroll
                    for (int i = 0; i < 16; i++) {
                        const unsigned key = jb + (i & 3) + 8 * (i >> 2) < jend ? __float_as_uint(X0[t][i]) : 0xFFFFFFFFu;
                        if constexpr (KNN) sec[t][i] = min(sec[t][i], max(best[t][i], key));
                        best[t][i] = min(best[t][i], key);
                    }
                }
            }
            __syncthreads();                                        // (the tail's buffer is free for the next query group's first chunk)
        }
        MM_STAMP(40);
#undef MM_PIPE
#undef MM_EPI2
#undef MM_READ
        // decode: bits - bits(2^10) = d 2^14 + P 2^11 + q  ->  d << 20 | first row of the slot's lane half in that tile
        auto decode = [&](unsigned k) -> unsigned {
            if (k == 0xFFFFFFFFu) return k;
            const unsigned u = k - MM_BASE_BITS;
            const unsigned P = (u >> 11) & 7u, q = u & 2047u;
            const unsigned row = (unsigned)jbeg + min(P, 6u) * (unsigned)E + 32u * q + 4u * (unsigned)h;
            return ((u >> 14) << 20) + row;
        };
#pragma unroll
        for (int t = 0; t < QT; t++) {
#pragma unroll
            for (int i = 0; i < 16; i++) best[t][i] = decode(best[t][i]);
#pragma unroll
            for (int i = 0; i < (KNN ? 16 : 0); i++) sec[t][i] = decode(sec[t][i]);
        }
        // ---- per query: this block's result (unsliced: final; sliced: the slice's summary for the merge below)
        int4* out = KNN ? knn_out_base + (long)pair * knn_stride : nullptr;
#pragma unroll
        for (int t = 0; t < QT; t++) {
            const int qi = q0 + 32 * t + r;
            if constexpr (KNN) {
                // slot i saw rows (i & 3) + 8 (i >> 2) + 4 h (+ 32 per tile): add the slot's row offset to both of its keys; the query's
                // other sixteen slots sit in lane ^ 32
                unsigned b1[32], b2[32];
#pragma unroll
                for (int i = 0; i < 16; i++) {
                    const unsigned off = (unsigned)((i & 3) + 8 * (i >> 2));
                    b1[i] = best[t][i] == 0xFFFFFFFFu ? best[t][i] : best[t][i] + off;
                    b2[i] = sec[t][i] == 0xFFFFFFFFu ? sec[t][i] : sec[t][i] + off;
                    b1[16 + i] = (unsigned)__shfl_xor((int)b1[i], 32);
                    b2[16 + i] = (unsigned)__shfl_xor((int)b2[i], 32);
                }
                unsigned m1 = b1[0];
#pragma unroll
                for (int i = 1; i < 32; i++) m1 = min(m1, b1[i]);
                unsigned m2 = 0xFFFFFFFFu;                          // the keys are distinct (they carry the index): one slot holds m1
#pragma unroll
                for (int i = 0; i < 32; i++) m2 = min(m2, b1[i] == m1 ? b2[i] : b1[i]);
                if (!sliced) {
                    if (h == 0 && qi < n1)
                        out[qi] = m1 == 0xFFFFFFFFu ? make_int4(-1, 512, 512, 0)
                                                    : make_int4((int)(m1 & 0xFFFFFu), (int)(m1 >> 20), m2 == 0xFFFFFFFFu ? 512 : (int)(m2 >> 20), 0);
                } else if (h == 0) {
                    // sliced: this slice's two smallest keys of the query; the last block of the query block merges the slices:
                    // nearest = the smallest m1, second = the smallest of the other slices' m1 and the winning slice's m2
                    __hip_atomic_store(reinterpret_cast<unsigned long long*>(part + (long)blockIdx.y * n1_pad + qi), ((unsigned long long)m2 << 32) | m1,
                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            } else {
                // slot i saw rows (i & 3) + 8 (i >> 2) + 4 h (+ 32 per tile): add the slot's row offset, then slots i and i + 8 (rows 16
                // apart) are one residue class: cls[k], k < 8 = class (k & 3) + 8 (k >> 2) + 4 h
                unsigned cls[8];
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    const unsigned lo = best[t][k] == 0xFFFFFFFFu ? best[t][k] : best[t][k] + (unsigned)((k & 3) + 8 * (k >> 2));
                    const unsigned hi = best[t][k + 8] == 0xFFFFFFFFu ? best[t][k + 8] : best[t][k + 8] + (unsigned)((k & 3) + 8 * (k >> 2) + 16);
                    cls[k] = min(lo, hi);
                }
                if (sliced) {                                       // (uniform)
                    // What the accept rule needs of a slice is little: the smallest key (distance << 20 | index) and WHICH classes attain
                    // its distance -- the rule asks whether exactly one class attains the global minimum distance (akazed.cu:2206, 2223).
                    // Plain 8-byte stores; atomicMin on 16 class keys per query ran into the atomic units' throughput (2 M lane-atomics:
                    // 35-85 us for 10k x 10k).
                    unsigned kloc = cls[0];
#pragma unroll
                    for (int k = 1; k < 8; k++) kloc = min(kloc, cls[k]);
                    const unsigned kmin = min(kloc, (unsigned)__shfl_xor((int)kloc, 32));
                    unsigned mloc = 0;
#pragma unroll
                    for (int k = 0; k < 8; k++) mloc |= (cls[k] >> 20) == (kmin >> 20) ? 1u << ((k & 3) + 8 * (k >> 2) + 4 * h) : 0u;
                    const unsigned mask = mloc | (unsigned)__shfl_xor((int)mloc, 32);
                    if (h == 0)
                        __hip_atomic_store(reinterpret_cast<unsigned long long*>(part + (long)blockIdx.y * n1_pad + qi),
                                           ((unsigned long long)(kmin == 0xFFFFFFFFu ? 0u : mask) << 32) | kmin, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                } else {
                    // the other half of the query's classes sits in lane ^ 32
                    unsigned all[16];
#pragma unroll
                    for (int k = 0; k < 8; k++) { all[k] = cls[k]; all[8 + k] = (unsigned)__shfl_xor((int)cls[k], 32); }
                    if (h == 0 && qi < n1) mm_accept(pts1 + qi, pts2, all, n2);
                }
            }
        }
        if (!sliced) continue;                                      // (uniform)
        // ---- sliced: the ticket; the block that draws the last ticket of its query block merges the slices' summaries.
        // (no agent-scope fence: on this part that is a write-back of the whole L2 per block -- 30-100 us for 10k x 10k.  The summaries
        // are written and read with agent-scope atomic accesses, which go to the coherence point themselves; the barrier's
        // s_waitcnt orders them in front of the ticket.
        // HARDWARE ASSUMPTION, not the HIP memory model: a workgroup-scope fence does not formally synchronise with another
        // workgroup, so the finisher's reads of `part` have no happens-before edge on paper.  What orders them on gfx950: an
        // agent-scope atomic store is issued sc1 (write-through to the device coherence point), __syncthreads() waits for
        // vmcnt(0) -- the stores have been acknowledged there -- before thread 0's ticket RMW (performed at the same
        // coherence point) is issued, and the finisher's agent-scope atomic loads bypass its own L1 / non-coherent L2 lines.
        // tests/test_gpu_pipeline.py::test_match_sliced_handoff_stress repeats 10k x 10k and the sliced pair path against
        // the VALU kernel to catch a compiler or cache-policy change that breaks this.)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __syncthreads();
        MM_STAMP(41);
        if (threadIdx.x == 0) {
            const int tk = __hip_atomic_fetch_add(&ticket[qb / QPB], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_last = tk == (int)gridDim.y - 1;
            if (s_last) __hip_atomic_store(&ticket[qb / QPB], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        MM_STAMP(42);
        if (s_last) {                                               // (block-uniform) the last block of the query block decides
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            const int q = qb + (int)threadIdx.x;
            if ((int)threadIdx.x < QPB && q < n1) {
                // (eight slices' summaries in flight at a time: one memory round trip instead of one per slice -- the finishing
                // blocks are the kernel's tail, round 5: they ended 4-8 us after the others)
                if constexpr (KNN) {
                    unsigned M1 = 0xFFFFFFFFu, M2 = 0xFFFFFFFFu;
                    for (int s0 = 0; s0 < (int)gridDim.y; s0 += 8) {
                        unsigned long long pv[8];
#pragma unroll
                        for (int j = 0; j < 8; j++)
                            pv[j] = s0 + j < (int)gridDim.y
                                        ? __hip_atomic_load(reinterpret_cast<const unsigned long long*>(part + (long)(s0 + j) * n1_pad + q),
                                                            __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                        : 0xFFFFFFFFFFFFFFFFull;
#pragma unroll
                        for (int j = 0; j < 8; j++) {
                            const unsigned a1 = (unsigned)pv[j], a2 = (unsigned)(pv[j] >> 32);
                            // union of {M1, M2} and {a1, a2}, each ascending: its two smallest
                            const unsigned lo = min(M1, a1), hi = max(M1, a1);
                            M2 = min(hi, lo == M1 ? M2 : a2);
                            M1 = lo;
                        }
                    }
                    out[q] = M1 == 0xFFFFFFFFu ? make_int4(-1, 512, 512, 0)
                                               : make_int4((int)(M1 & 0xFFFFFu), (int)(M1 >> 20), M2 == 0xFFFFFFFFu ? 512 : (int)(M2 >> 20), 0);
                } else {
                    unsigned K = 0xFFFFFFFFu, M = 0u;
                    for (int s0 = 0; s0 < (int)gridDim.y; s0 += 8) {
                        unsigned long long pv[8];
#pragma unroll
                        for (int j = 0; j < 8; j++)
                            pv[j] = s0 + j < (int)gridDim.y
                                        ? __hip_atomic_load(reinterpret_cast<const unsigned long long*>(part + (long)(s0 + j) * n1_pad + q),
                                                            __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                        : 0xFFFFFFFFull;
#pragma unroll
                        for (int j = 0; j < 8; j++) {
                            const unsigned k2 = (unsigned)pv[j], m2 = (unsigned)(pv[j] >> 32);
                            if ((k2 >> 20) < (K >> 20)) M = m2;      // a smaller distance: its classes alone attain it
                            else if ((k2 >> 20) == (K >> 20)) M |= m2;
                            K = min(K, k2);
                        }
                    }
                    hak_point* p1 = pts1 + q;
                    const int dmin = (int)(K >> 20);
                    const int bi = min((int)(K & 0xFFFFFu), max(n2 - 1, 0));
                    if (K != 0xFFFFFFFFu && __popc(M) == 1 && dmin < HAK_MAX_DIST) {                   // akazed.cu:2206, 2223
                        p1->match = bi; p1->distance = dmin; p1->match_x = pts2[bi].x; p1->match_y = pts2[bi].y;
                    } else {
                        p1->match = -1; p1->distance = -1; p1->match_x = -1.f; p1->match_y = -1.f;
                    }
                }
            }
        }
    }
    MM_BLK(1);
}

// accept rule of gHammingMatch (akazed.cu:2190-2223) on the merged class minima of the sliced search of the VALU kernel
// (HAK_MATCH_VALU=1; k_match_mfma finishes inside its last block); one thread per query
__global__ __launch_bounds__(256) void k_match_finish(hak_point* pts1, const hak_point* pts2, int n1, unsigned* __restrict__ gkey)
{
    const int qi = blockIdx.x * 256 + threadIdx.x;
    if (qi >= n1) return;
    uint4* k4 = reinterpret_cast<uint4*>(gkey + (long)qi * MC);
    unsigned k[MC];
#pragma unroll
    for (int t = 0; t < MC / 4; t++) {
        const uint4 v = k4[t]; k[4 * t] = v.x; k[4 * t + 1] = v.y; k[4 * t + 2] = v.z; k[4 * t + 3] = v.w;
        k4[t] = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);     // the scratch goes back to its empty state (HakMatchScratch)
    }
    int bc = 0;
#pragma unroll
    for (int t = 1; t < MC; t++)
        if ((k[t] >> 20) < (k[bc] >> 20)) bc = t;
    const unsigned kmin = k[bc];
    const int dmin = (int)(kmin >> 20);
    int nflag = 0;
#pragma unroll
    for (int t = 0; t < MC; t++) nflag += (unsigned)dmin < (k[t] >> 20) ? 1 : 0;
    hak_point* p1 = pts1 + qi;
    const int bi = (int)(kmin & 0xFFFFFu);
    if (kmin != 0xFFFFFFFFu && nflag == MC - 1 && dmin < HAK_MAX_DIST) {
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

// ------------------------------------------------------------------ match post-processing
// 2-NN search: for every point a of set A its nearest neighbour in B (smallest index among ties), the
// distance d1 to it and the distance d2 to the nearest OTHER point of B (512 when B has fewer than two
// points -- the initial score of the reference's unused gMatch, akazed.cu:2028-2122, whose accept rule
// "d1 < d2 && d1 < MAX_DIST" is the ratio test with ratio 1).  Same 16 x 16 block mapping as k_match.
__global__ __launch_bounds__(256) void k_knn2(const hak_point* ptsA_base, const hak_point* ptsB_base,
                                              const int* __restrict__ nA_dev, const int* __restrict__ nB_dev,
                                              int nA_host, int nB_host, long strideA, long strideB, int count_stride,
                                              int4* __restrict__ out_base, long out_stride)
{
    __shared__ int sd1[MC][MQ], sd2[MC][MQ], si1[MC][MQ];
    const int pair = blockIdx.y;
    const int nA = nA_dev ? nA_dev[pair * count_stride] : nA_host;
    const int nB = nB_dev ? nB_dev[pair * count_stride] : nB_host;
    const hak_point* A = ptsA_base + (long)pair * strideA;
    const hak_point* B = ptsB_base + (long)pair * strideB;
    int4* out = out_base + (long)pair * out_stride;
    const int q = threadIdx.x & (MQ - 1), c = threadIdx.x >> 4;
    for (int q0 = blockIdx.x * MQ; q0 < nA; q0 += gridDim.x * MQ) {
        const int qi = q0 + q;
        unsigned int qd[16];
        if (qi < nA) load_desc(A + qi, qd);
        int best = 512, second = 512, besti = -1;
        if (qi < nA)
            for (int j = c; j < nB; j += MC) {
                unsigned int td[16];
                load_desc(B + j, td);
                int dist = 0;
#pragma unroll
                for (int k = 0; k < 16; k += 2)
                    dist += __popcll(((unsigned long long)(qd[k + 1] ^ td[k + 1]) << 32) | (qd[k] ^ td[k]));
                if (dist < best) { second = best; best = dist; besti = j; }
                else if (dist < second) second = dist;
            }
        sd1[c][q] = best; sd2[c][q] = second; si1[c][q] = besti;
        __syncthreads();
        if (c == 0 && qi < nA) {
            int bc = -1;
            for (int t = 0; t < MC; t++) {
                if (si1[t][q] < 0) continue;
                if (bc < 0 || sd1[t][q] < sd1[bc][q] || (sd1[t][q] == sd1[bc][q] && si1[t][q] < si1[bc][q])) bc = t;
            }
            int d2 = 512;
            for (int t = 0; t < MC; t++) {
                const int v = (t == bc) ? sd2[t][q] : sd1[t][q];
                d2 = v < d2 ? v : d2;
            }
            out[qi] = bc < 0 ? make_int4(-1, 512, 512, 0) : make_int4(si1[bc][q], sd1[bc][q], d2, 0);
        }
        __syncthreads();
    }
}

// accept rule + compaction in query order.  One 1024-thread block per pair walks the queries in chunks;
// accepted matches are appended at the running base through a ballot/popcount block scan, so the output
// order (ascending query index) is deterministic.
__global__ __launch_bounds__(1024) void k_knn2_finish(hak_point* pts1_base, const hak_point* pts2_base,
                                                      const int* __restrict__ n1_dev, int n1_host, long stride1, long stride2,
                                                      int count_stride, const int4* __restrict__ fwd_base,
                                                      const int4* __restrict__ rev_base, long knn_stride, int ratio_num,
                                                      int ratio_den, int cross, int max_dist, hak_match_pair* out_base,
                                                      long out_stride, int* __restrict__ out_count, int count_out_stride)
{
    __shared__ int wsum[16];
    __shared__ int sbase;
    const int pair = blockIdx.x;
    const int n1 = n1_dev ? n1_dev[pair * count_stride] : n1_host;
    hak_point* pts1 = pts1_base + (long)pair * stride1;
    const hak_point* pts2 = pts2_base + (long)pair * stride2;
    const int4* fwd = fwd_base + (long)pair * knn_stride;
    const int4* rev = rev_base ? rev_base + (long)pair * knn_stride : nullptr;
    hak_match_pair* out = out_base ? out_base + (long)pair * out_stride : nullptr;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (threadIdx.x == 0) sbase = 0;
    __syncthreads();
    for (int i0 = 0; i0 < n1; i0 += 1024) {
        const int i = i0 + threadIdx.x;
        bool ok = false;
        int4 f = make_int4(-1, 512, 512, 0);
        if (i < n1) {
            f = fwd[i];
            ok = f.x >= 0 && f.y < max_dist && (long)f.y * ratio_den < (long)f.z * ratio_num;
            if (ok && cross) ok = rev[f.x].x == i;
            hak_point* p1 = pts1 + i;
            if (ok) {
                p1->match = f.x; p1->distance = f.y;
                p1->match_x = pts2[f.x].x; p1->match_y = pts2[f.x].y;
            } else {
                p1->match = -1; p1->distance = -1; p1->match_x = -1.f; p1->match_y = -1.f;
            }
        }
        const unsigned long long m = __ballot(ok);
        if (lane == 0) wsum[wv] = __popcll(m);
        __syncthreads();
        int before = sbase;
        for (int t = 0; t < wv; t++) before += wsum[t];
        if (ok && out) {
            const int pos = before + __popcll(m & ((1ull << lane) - 1ull));
            hak_match_pair r;
            r.query = i; r.train = f.x; r.distance = f.y; r.second = f.z;
            r.x1 = pts1[i].x; r.y1 = pts1[i].y; r.x2 = pts2[f.x].x; r.y2 = pts2[f.x].y;
            out[pos] = r;
        }
        __syncthreads();
        if (threadIdx.x == 0) { int tot = 0; for (int t = 0; t < 16; t++) tot += wsum[t]; sbase += tot; }
        __syncthreads();
    }
    if (threadIdx.x == 0 && out_count) out_count[pair * count_out_stride] = sbase;
}

// ---- big single pairs: accept rule + compaction in two multi-block passes (k_knn2_finish above walks a pair with ONE block:
// right for a batch of pairs, 35 us for 10k queries).  Pass A applies the rule, writes the match fields and counts the accepted
// matches of its 1024 queries; pass B re-derives the rule's outcome, adds the counts of the blocks in front of it and writes the
// match list in ascending query order.
__device__ __forceinline__ bool knn2_rule(const int i, const int n1, const int4* __restrict__ fwd, const int4* __restrict__ rev,
                                          const int ratio_num, const int ratio_den, const int cross, const int max_dist, int4& f)
{
    f = make_int4(-1, 512, 512, 0);
    if (i >= n1) return false;
    f = fwd[i];
    bool ok = f.x >= 0 && f.y < max_dist && (long)f.y * ratio_den < (long)f.z * ratio_num;
    if (ok && cross) ok = rev[f.x].x == i;
    return ok;
}
__global__ __launch_bounds__(1024) void k_knn2_finish_a(hak_point* pts1, const hak_point* __restrict__ pts2, int n1,
                                                        const int4* __restrict__ fwd, const int4* __restrict__ rev, int ratio_num,
                                                        int ratio_den, int cross, int max_dist, int* __restrict__ blk)
{
    __shared__ int wsum[16];
    const int i = blockIdx.x * 1024 + threadIdx.x;
    int4 f;
    const bool ok = knn2_rule(i, n1, fwd, rev, ratio_num, ratio_den, cross, max_dist, f);
    if (i < n1) {
        hak_point* p1 = pts1 + i;
        if (ok) { p1->match = f.x; p1->distance = f.y; p1->match_x = pts2[f.x].x; p1->match_y = pts2[f.x].y; }
        else { p1->match = -1; p1->distance = -1; p1->match_x = -1.f; p1->match_y = -1.f; }
    }
    const unsigned long long m = __ballot(ok);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = __popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) { int tot = 0; for (int t = 0; t < 16; t++) tot += wsum[t]; blk[blockIdx.x] = tot; }
}
__global__ __launch_bounds__(1024) void k_knn2_finish_b(const hak_point* __restrict__ pts1, const hak_point* __restrict__ pts2, int n1,
                                                        const int4* __restrict__ fwd, const int4* __restrict__ rev, int ratio_num,
                                                        int ratio_den, int cross, int max_dist, const int* __restrict__ blk,
                                                        hak_match_pair* __restrict__ out, int* __restrict__ out_count, int* __restrict__ h_count)
{
    __shared__ int wsum[16];
    __shared__ int sbase;
    const int i = blockIdx.x * 1024 + threadIdx.x;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int4 f;
    const bool ok = knn2_rule(i, n1, fwd, rev, ratio_num, ratio_den, cross, max_dist, f);
    const unsigned long long m = __ballot(ok);
    if (lane == 0) wsum[wv] = __popcll(m);
    if (wv == 0) {                                                  // accepted matches of all blocks in front of this one
        int part = 0;
        for (int b = lane; b < (int)blockIdx.x; b += 64) part += blk[b];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o);
        if (lane == 0) sbase = part;
    }
    __syncthreads();
    int before = sbase;
    for (int t = 0; t < wv; t++) before += wsum[t];
    if (ok && out) {
        hak_match_pair r;
        r.query = i; r.train = f.x; r.distance = f.y; r.second = f.z;
        r.x1 = pts1[i].x; r.y1 = pts1[i].y; r.x2 = pts2[f.x].x; r.y2 = pts2[f.x].y;
        out[before + __popcll(m & ((1ull << lane) - 1ull))] = r;
    }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) {
        int tot = sbase;
        for (int t = 0; t < 16; t++) tot += wsum[t];
        if (out_count) *out_count = tot;
        if (h_count) *h_count = tot;                                // pinned host word: the count is there when the stream is idle
    }
}

// ---- scratch of the sliced searches (hak_internal.h)
static bool grow(void** p, long* cap, long want, size_t elem, int fill, hipStream_t st)
{
    if (want <= *cap) return true;
    if (*p) { (void)hipStreamSynchronize(st); (void)hipFree(*p); }
    *p = nullptr; *cap = 0;
    if (hipMalloc(p, elem * (size_t)want) != hipSuccess) { (void)hipGetLastError(); *p = nullptr; return false; }
    // (on the stream of the search itself: hipMemset works on the NULL stream and may return before the fill has run, and a context's
    // stream is non-blocking -- nothing would order the fill in front of the kernel that takes the tickets)
    if (fill >= 0 && hipMemsetAsync(*p, fill, elem * (size_t)want, st) != hipSuccess) { (void)hipGetLastError(); (void)hipFree(*p); *p = nullptr; return false; }
    *cap = want;
    return true;
}
bool hak_match_scratch_reserve(HakMatchScratch* sc, hipStream_t st, long keys, long tickets, long parts, long knn, long blks)
{
    if (sc->device < 0) (void)hipGetDevice(&sc->device);
    bool ok = grow((void**)&sc->keys, &sc->keys_cap, keys, sizeof(unsigned), 0xFF, st);
    ok = grow((void**)&sc->ticket, &sc->ticket_cap, tickets, sizeof(int), 0, st) && ok;
    ok = grow((void**)&sc->part, &sc->part_cap, parts, sizeof(uint2), -1, st) && ok;
    ok = grow((void**)&sc->knn, &sc->knn_cap, knn, sizeof(int4), -1, st) && ok;
    ok = grow((void**)&sc->blk, &sc->blk_cap, blks, sizeof(int), -1, st) && ok;
    if (ok && (knn > 0 || blks > 0) && !sc->d_cnt) {
        ok = hipMalloc((void**)&sc->d_cnt, sizeof(int)) == hipSuccess && hipHostMalloc((void**)&sc->h_cnt, sizeof(int)) == hipSuccess;
        if (!ok) (void)hipGetLastError();
    }
    return ok;
}
void hak_match_scratch_free(HakMatchScratch* sc)
{
    void* bufs[] = {sc->keys, sc->ticket, sc->part, sc->knn, sc->blk, sc->d_cnt};
    for (void* b : bufs) if (b) (void)hipFree(b);
    if (sc->h_cnt) (void)hipHostFree(sc->h_cnt);
    *sc = HakMatchScratch();
}

// slices of a big pair's train set for the matrix-core kernel: query blocks x slices ~ want_blocks, every slice a whole number of
// 32-row tiles, none empty.  Two blocks per CU are resident (launch bounds, 2 x 72 KB of LDS): 512 blocks fill the chip once, and
// 10k x 10k (79 query blocks) is fastest at 6 slices = 474 blocks for both searches (1-NN, ms per call at 4 5 6 7 8 10 13 slices:
// 0.0541 0.0491 0.0462 0.0532 0.0504 0.0504 0.0510).  HAK_MATCH_SLICES overrides the count (tuning).
static int mfma_slices(int gx, int n2, int* rows_per_slice, int want_blocks)
{
    static const int env = [] { const char* e = getenv("HAK_MATCH_SLICES"); return e ? atoi(e) : 0; }();
    const int tiles = (n2 + 31) / 32;
    if (tiles < 16) { *rows_per_slice = tiles * 32; return 1; }
    int slices = env > 0 ? env : (want_blocks + gx / 2) / gx;
    if (slices > tiles / 8) slices = tiles / 8;                     // at least 8 tiles (one LDS chunk) per slice
    if (slices < 1) slices = 1;
    int tps = (tiles + slices - 1) / slices;
    tps = (tps + 5) / 6 * 6;                                        // whole 6-tile chunks: only the last slice has a tail
    *rows_per_slice = tps * 32;
    return (tiles + tps - 1) / tps;
}

// query tiles per wave of k_match_mfma (HAK_MATCH_QT = 1 | 2, read per call like HAK_MATCH_VALU: A/B runs and the tests drive both)
static int mm_query_tiles()
{
    const char* e = getenv("HAK_MATCH_QT");
    const int v = e ? atoi(e) : MM_QT_DEFAULT;
    return v == 1 ? 1 : 2;
}
template <bool KNN, typename... A>
static void mm_launch(int qt, dim3 grid, hipStream_t st, A... a)
{
    if (qt == 2) k_match_mfma<KNN, MM_BCH, 2><<<grid, 256, 0, st>>>(a...);
    else k_match_mfma<KNN, MM_BCH, 1><<<grid, 256, 0, st>>>(a...);
}

void hak_launch_knn2(hipStream_t st, const hak_point* ptsA, const hak_point* ptsB, const int* nA_dev, const int* nB_dev,
                     int nA_host, int nB_host, long strideA, long strideB, int npairs, int4* out, long out_stride, HakMatchScratch* sc)
{
    const char* env_valu = getenv("HAK_MATCH_VALU");               // (read per call, as in hak_launch_match)
    // (with device-side counts nA_host / nB_host carry the CAPACITY of the sets)
    if (!(env_valu && atoi(env_valu) != 0) && nB_host <= MM_MAX_ROWS) {
        // the matrix-core kernel with its 2-NN epilogue (the point records are only read: ptsA is not written)
        const int qt = mm_query_tiles(), qpb = 128 * qt;
        int gx = nA_dev ? (qt == 2 ? 43 : 83) : (nA_host + qpb - 1) / qpb;
        if (gx < 1) gx = 1;
        if (gx > 4096) gx = 4096;
        // one big pair with host-side counts whose query blocks alone cannot fill the chip: slice the train set
        if (sc && !nA_dev && npairs == 1 && gx < 384 && (long)gx * qpb >= nA_host) {
            int rps = 0;
            const int slices = mfma_slices(gx, nB_host, &rps, qt == 2 ? 256 : 512);
            if (slices > 1 && hak_match_scratch_reserve(sc, st, 0, gx, (long)slices * gx * qpb, 0, 0)) {
                mm_launch<true>(qt, dim3(gx, slices), st, const_cast<hak_point*>(ptsA), ptsB, (const int*)nullptr, (const int*)nullptr, nA_host, nB_host, 0L, 0L,
                                2, rps, out, 0L, sc->ticket, sc->part, gx * qpb);
                return;
            }
        }
        mm_launch<true>(qt, dim3(gx, npairs), st, const_cast<hak_point*>(ptsA), ptsB, nA_dev, nB_dev, nA_host, nB_host, strideA, strideB,
                        2, 0, out, out_stride, (int*)nullptr, (uint2*)nullptr, 0);
        return;
    }
    int gx = nA_dev ? 640 : (nA_host + MQ - 1) / MQ;
    if (gx < 1) gx = 1;
    if (gx > 4096) gx = 4096;
    k_knn2<<<dim3(gx, npairs), 256, 0, st>>>(ptsA, ptsB, nA_dev, nB_dev, nA_host, nB_host, strideA, strideB, 2, out, out_stride);
}

void hak_launch_knn2_finish(hipStream_t st, hak_point* pts1, const hak_point* pts2, const int* n1_dev, int n1_host, long stride1,
                            long stride2, int npairs, const int4* fwd, const int4* rev, long knn_stride, int ratio_num,
                            int ratio_den, int cross, int max_dist, hak_match_pair* out, long out_stride, int* out_count,
                            HakMatchScratch* sc)
{
    const int nb = (n1_host + 1023) / 1024;
    if (sc && !n1_dev && npairs == 1 && nb > 1 && hak_match_scratch_reserve(sc, st, 0, 0, 0, 0, nb)) {
        k_knn2_finish_a<<<nb, 1024, 0, st>>>(pts1, pts2, n1_host, fwd, rev, ratio_num, ratio_den, cross, max_dist, sc->blk);
        k_knn2_finish_b<<<nb, 1024, 0, st>>>(pts1, pts2, n1_host, fwd, rev, ratio_num, ratio_den, cross, max_dist, sc->blk, out, out_count,
                                             sc->h_cnt);
        return;
    }
    k_knn2_finish<<<npairs, 1024, 0, st>>>(pts1, pts2, n1_dev, n1_host, stride1, stride2, 2, fwd, rev, knn_stride, ratio_num,
                                           ratio_den, cross, max_dist, out, out_stride, out_count, 1);
}

// Big single pairs (hak_match with host-side counts, e.g. 10k x 10k) are SLICED: the train set is cut so that query blocks x
// slices fill the chip; the slices merge through `scratch` (HakMatchScratch: owned by the calling context or handed out by
// hak_api.hip's per-device pool -- never a process-wide pointer that could belong to another device).
void hak_launch_match(hipStream_t st, hak_point* pts1, const hak_point* pts2, const int* n1_dev, const int* n2_dev,
                      int n1_host, int n2_host, long pair_stride1, long pair_stride2, int npairs, HakMatchScratch* sc)
{
    // k_match_mfma: a wave = 32 QT queries x the train set, four waves per block.  HAK_MATCH_VALU=1: the VALU / LDS kernel k_match
    const char* env_valu = getenv("HAK_MATCH_VALU");               // (read per call: the tests run both kernels in one process)
    const bool valu = (env_valu && atoi(env_valu) != 0) || n2_host > MM_MAX_ROWS;       // (device-side counts: n2_host = the capacity)
    const int qt = mm_query_tiles(), qpb = 128 * qt;
    const int nq = n1_dev ? 0 : n1_host;
    const bool two = n1_dev ? npairs >= 8 : (long)((nq + 2 * MQ - 1) / (2 * MQ)) * npairs >= 2048;    // (k_match only: queries per thread)
    const int qb = valu ? (two ? 2 * MQ : MQ) : qpb;                // queries per block
    // device-side counts: k_match loops over the queries; k_match_mfma gets blocks for 10 240 queries (waves past n1 leave at once)
    // (83 / 43, not 80 / 40: blocks go to the eight XCDs by linear index mod 8, and with a multiple of 8 per pair the blocks of every
    // pair that find queries would land on the same XCDs pair after pair -- two XCDs with three of them, six with two: 1.33 x the mean)
    int gx = n1_dev ? (valu ? (two ? 320 : 640) : (qt == 2 ? 43 : 83)) : (nq + qb - 1) / qb;
    if (gx < 1) gx = 1;
    if (gx > 4096) gx = 4096;
    // with device-side counts n1_host carries the CAPACITY of a query set (the context's max_pts; 0: unknown).  The sliced path
    // below gives every query block its own ticket and partial-result rows, so its grid must cover the capacity -- the plain
    // kernel's blocks loop over the queries and need no such bound.
    const int cap_blocks = n1_dev && n1_host > 0 ? (n1_host + qpb - 1) / qpb : 0;
    // one pair with host-side counts whose query blocks alone cannot fill the chip: slice the train set as well
    if (sc && !n1_dev && npairs == 1 && (long)gx * qb >= nq) {
        if (!valu && gx < 384) {
            int rps = 0;
            const int slices = mfma_slices(gx, n2_host, &rps, qt == 2 ? 256 : 512);
            if (slices > 1 && hak_match_scratch_reserve(sc, st, 0, gx, (long)slices * gx * qpb, 0, 0)) {
                mm_launch<false>(qt, dim3(gx, slices), st, pts1, pts2, (const int*)nullptr, (const int*)nullptr, n1_host, n2_host, 0L, 0L, 2, rps,
                                 (int4*)nullptr, 0L, sc->ticket, sc->part, gx * qpb);
                return;
            }
        }
        const int tiles = (n2_host + MT - 1) / MT;
        if (valu && !two && gx < 2048 && tiles >= 4) {
            int slices = (2048 + gx - 1) / gx;
            if (slices > tiles / 2) slices = tiles / 2;
            const int tps = (tiles + slices - 1) / slices;
            slices = (tiles + tps - 1) / tps;
            if (slices > 1 && hak_match_scratch_reserve(sc, st, (long)n1_host * MC, 0, 0, 0, 0)) {
                k_match<1><<<dim3(gx, slices), 256, 0, st>>>(pts1, pts2, nullptr, nullptr, n1_host, n2_host, 0, 0, 2, sc->keys, tps);
                k_match_finish<<<(n1_host + 255) / 256, 256, 0, st>>>(pts1, pts2, n1_host, sc->keys);
                return;
            }
        }
    }
    // few pairs with device-side counts (the pair call, batches of a handful of images): few of a pair's blocks find queries,
    // each walks the whole train set while most of the chip idles -- slice the train sets as for one big pair
    if (!valu && sc && n1_dev && npairs <= 12 && cap_blocks > 0 && cap_blocks <= 4096) {
        if (cap_blocks > gx) gx = cap_blocks | 3;                   // (odd, as 83: not a multiple of the eight XCDs)
        const int slices = (npairs <= 2 ? 8 : npairs <= 4 ? 4 : npairs <= 8 ? 3 : 2) * (qt == 2 ? 2 : 1);
        if (hak_match_scratch_reserve(sc, st, 0, (long)npairs * gx, (long)npairs * slices * gx * qpb, 0, 0)) {
            mm_launch<false>(qt, dim3(gx, slices, npairs), st, pts1, pts2, n1_dev, n2_dev, n1_host, n2_host, pair_stride1, pair_stride2,
                             2, 0, (int4*)nullptr, 0L, sc->ticket, sc->part, gx * qpb);
            return;
        }
    }
    dim3 grid(gx, npairs);
    if (!valu) mm_launch<false>(qt, grid, st, pts1, pts2, n1_dev, n2_dev, n1_host, n2_host, pair_stride1, pair_stride2, 2, 0,
                                (int4*)nullptr, 0L, (int*)nullptr, (uint2*)nullptr, 0);
    else if (two) k_match<2><<<grid, 256, 0, st>>>(pts1, pts2, n1_dev, n2_dev, n1_host, n2_host, pair_stride1, pair_stride2, 2, nullptr, 0);
    else k_match<1><<<grid, 256, 0, st>>>(pts1, pts2, n1_dev, n2_dev, n1_host, n2_host, pair_stride1, pair_stride2, 2, nullptr, 0);
}
