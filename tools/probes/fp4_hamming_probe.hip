#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
// one 32x32 tile: train rows (A) x query cols (B), 512 bits each (16 dwords); out[row*32+col] = hamming
template <bool SCALED>
__global__ void k(const unsigned* __restrict__ tr, const unsigned* __restrict__ qu, unsigned* out)
{
    const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
    unsigned pb = 0;
    for (int i = 0; i < 16; i++) pb += __popc(qu[r * 16 + i]);
    v16f acc;
    for (int i = 0; i < 16; i++) acc[i] = 8388608.0f + (float)pb;
    for (int s = 0; s < 8; s++) {
        const unsigned w = tr[r * 16 + 2 * s + h], q = qu[r * 16 + 2 * s + h];
        v8i a = {0,0,0,0,0,0,0,0}, b = {0,0,0,0,0,0,0,0};
        a[0] = (int)(w & 0x11111111u);            // bit 0 of each nibble: 0.5
        a[1] = (int)(w & 0x22222222u);            // 1.0
        a[2] = (int)(w & 0x44444444u);            // 2.0
        a[3] = (int)((w >> 1) & 0x44444444u);     // bit 3 -> 2.0
        // query: (1 - 2 b) * {2, 1, 0.5, 0.5}: fp4 codes 2.0 = 0100, 1.0 = 0010, 0.5 = 0001, sign = 1000
        b[0] = (int)(0x44444444u | ((q & 0x11111111u) << 3));
        b[1] = (int)(0x22222222u | ((q & 0x22222222u) << 2));
        b[2] = (int)(0x11111111u | ((q & 0x44444444u) << 1));
        b[3] = (int)(0x11111111u | (q & 0x88888888u));
        if constexpr (SCALED) acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc, 4, 4, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
        else acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc, 4, 4, 0, 0, 0, 0);     // folds to the unscaled v_mfma_f32_32x32x64_f8f6f4
    }
    for (int i = 0; i < 16; i++) {
        const int row = (i & 3) + 8 * (i >> 2) + 4 * h;
        out[row * 32 + r] = __float_as_uint(acc[i]);
    }
}
// round 5: the accumulator as the finished key.  Query magnitudes doubled (every descriptor bit contributes +-2), the accumulator
// starts at 2^10 + 2 |b| + q 2^-13, and three k positions that hold struct padding in every descriptor (bits 8, 12, 16 of dword 15:
// both sets are masked to byte 60 there) carry the train row's tile number P (0.5 on the train side against 0.5 / 1 / 2 on the query
// side): bit pattern - bits(2^10) must be d << 14 | P << 11 | q for every pair
__global__ void k5(const unsigned* __restrict__ tr, const unsigned* __restrict__ qu, unsigned* out, int q_chunk)
{
    const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
    unsigned pb = 0;
    for (int i = 0; i < 16; i++) pb += __popc(qu[r * 16 + i] & (i == 15 ? 0xFFu : 0xFFFFFFFFu));     // (the kernel masks dword 15 to byte 60)
    v16f acc;
    for (int i = 0; i < 16; i++) acc[i] = 1024.0f + 2.0f * (float)pb + (float)q_chunk * 0.0001220703125f;
    const unsigned P = (unsigned)(r & 7);                         // train row r carries P = r % 8
    for (int s = 0; s < 8; s++) {
        unsigned w = tr[r * 16 + 2 * s + h];
        unsigned q = qu[r * 16 + 2 * s + h];
        if (2 * s + h == 15) { w &= 0xFFu; q &= 0xFFu; }
        if (2 * s + h == 15) w |= ((P & 1u) << 8) | ((P & 2u) << 11) | ((P & 4u) << 14);
        v8i a = {0,0,0,0,0,0,0,0}, b = {0,0,0,0,0,0,0,0};
        a[0] = (int)(w & 0x11111111u); a[1] = (int)(w & 0x22222222u); a[2] = (int)(w & 0x44444444u); a[3] = (int)((w >> 1) & 0x44444444u);
        b[0] = (int)(0x66666666u | ((q & 0x11111111u) << 3));     // 4, 2, 1, 1
        b[1] = (int)(0x44444444u | ((q & 0x22222222u) << 2));
        b[2] = (int)(0x22222222u | ((q & 0x44444444u) << 1));
        b[3] = (int)(0x22222222u | (q & 0x88888888u));
        if (2 * s + h == 15) b[0] = (int)(((unsigned)b[0] & ~0x000FFF00u) | 0x00042100u);
        acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc, 4, 4, 0, 0, 0, 0);
    }
    for (int i = 0; i < 16; i++) {
        const int row = (i & 3) + 8 * (i >> 2) + 4 * h;
        out[row * 32 + r] = __float_as_uint(acc[i]);
    }
}
int main()
{
    unsigned tr[512], qu[512], *dt, *dq, *dout, out[1024];
    srand(1);
    for (int i = 0; i < 512; i++) { tr[i] = (unsigned)rand() ^ ((unsigned)rand() << 16); qu[i] = (unsigned)rand() ^ ((unsigned)rand() << 16); }
    for (int r = 0; r < 32; r++) { tr[r * 16 + 15] &= 0xFF; qu[r * 16 + 15] &= 0xFF; }
    for (int i = 0; i < 16; i++) { tr[3 * 16 + i] = 0; qu[5 * 16 + i] = 0xFFFFFFFFu; tr[7*16+i] = 0xFFFFFFFFu; }
    hipMalloc(&dt, 2048); hipMalloc(&dq, 2048); hipMalloc(&dout, 4096);
    hipMemcpy(dt, tr, 2048, hipMemcpyHostToDevice); hipMemcpy(dq, qu, 2048, hipMemcpyHostToDevice);
    int total = 0;
    for (int scaled = 0; scaled < 2; scaled++) {
    if (scaled) k<true><<<1, 64>>>(dt, dq, dout); else k<false><<<1, 64>>>(dt, dq, dout);
    hipMemcpy(out, dout, 4096, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int row = 0; row < 32; row++) for (int col = 0; col < 32; col++) {
        unsigned d = 0;
        for (int i = 0; i < 16; i++) d += __builtin_popcount(tr[row * 16 + i] ^ qu[col * 16 + i]);
        const unsigned got = out[row * 32 + col];
        if (got != 0x4B000000u + d) { if (bad < 8) printf("row %d col %d want %u got %08x (%d)\n", row, col, d, got, (int)(got - 0x4B000000u)); bad++; }
    }
    printf("fp4 mfma hamming (%s): %d mismatches of 1024\n", scaled ? "scale operands 0x7F" : "scale 0 = unscaled form", bad);
    total += bad;
    }
    // the dword order of this probe is (2 s + h); the staged row of the kernel is (8 h + s): any assignment is fine as long as A and B agree
    for (int qc = 0; qc < 2048; qc += 2047) {
        k5<<<1, 64>>>(dt, dq, dout, qc);
        hipMemcpy(out, dout, 4096, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int row = 0; row < 32; row++) for (int col = 0; col < 32; col++) {
            unsigned d = 0;
            for (int i = 0; i < 16; i++) d += __builtin_popcount((tr[row * 16 + i] ^ qu[col * 16 + i]) & (i == 15 ? 0xFFu : 0xFFFFFFFFu));
            const unsigned want = 0x44800000u + (d << 14) + ((unsigned)(row & 7) << 11) + (unsigned)qc;
            const unsigned got = out[row * 32 + col];
            if (got != want) { if (bad < 8) printf("key: row %d col %d want %08x got %08x\n", row, col, want, got); bad++; }
        }
        printf("fp4 mfma key = 2^10 + 2 d + P / 4 + q 2^-13 (q = %d): %d mismatches of 1024\n", qc, bad);
        total += bad;
    }
    return total != 0;
}
