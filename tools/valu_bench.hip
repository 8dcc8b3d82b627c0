// tools/valu_bench.hip -- calibrates VALU issue cost on gfx950: cycles per wave-instruction per SIMD for
// independent v_fma_f32, v_pk_fma_f32, v_mul_lo_u32, v_mad_u64_u32 at 1..8 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#define N_IT 4096
template <int MODE>
__global__ void k(float* out, int n)
{
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const float b = 1.0001f, c = 0.5f;
    unsigned u0 = threadIdx.x, u1 = u0 + 1, u2 = u0 + 2, u3 = u0 + 3;
    unsigned long long w0 = threadIdx.x, w1 = 3;
    for (int i = 0; i < n; i++) {
        if (MODE == 0) {
            a0 = fmaf(a0, b, c); a1 = fmaf(a1, b, c); a2 = fmaf(a2, b, c); a3 = fmaf(a3, b, c);
            a4 = fmaf(a4, b, c); a5 = fmaf(a5, b, c); a6 = fmaf(a6, b, c); a7 = fmaf(a7, b, c);
        } else if (MODE == 1) {
            typedef float v2 __attribute__((ext_vector_type(2)));
            v2 x = {a0, a1}, y = {a2, a3}, z = {a4, a5}, t = {a6, a7}, bb = {b, b}, cc = {c, c};
            x = __builtin_elementwise_fma(x, bb, cc); y = __builtin_elementwise_fma(y, bb, cc);
            z = __builtin_elementwise_fma(z, bb, cc); t = __builtin_elementwise_fma(t, bb, cc);
            a0 = x.x; a1 = x.y; a2 = y.x; a3 = y.y; a4 = z.x; a5 = z.y; a6 = t.x; a7 = t.y;
        } else if (MODE == 2) {
            u0 = u0 * 2654435761u + 1; u1 = u1 * 2654435761u + 1; u2 = u2 * 2654435761u + 1; u3 = u3 * 2654435761u + 1;
        } else {
            w0 = w0 * 6364136223846793005ull + w1; w1 = w1 * 6364136223846793005ull + w0;
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + u0 + u1 + u2 + u3 + (float)(w0 + w1);
}
template <int MODE>
void run(const char* name, int ops_per_it)
{
    float* d; hipMalloc(&d, 256 * 8 * 256 * 8 * sizeof(float));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int wps = 1; wps <= 8; wps *= 2) {
        // 256 CUs x 4 SIMDs x wps waves: blocks of 256 threads (1 wave per SIMD), wps blocks per CU
        int blocks = 256 * wps;
        k<MODE><<<blocks, 256>>>(d, 64);
        hipEventRecord(e0); k<MODE><<<blocks, 256>>>(d, N_IT); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double inst_per_simd = (double)N_IT * ops_per_it * wps;
        printf("%-14s waves/SIMD %d: %.3f ms -> %.2f ns per wave-instr per SIMD (%.2f cycles @2.4GHz)\n", name, wps, ms,
               ms * 1e6 / inst_per_simd, ms * 1e6 / inst_per_simd * 2.4);
    }
    hipFree(d);
}
int main() { run<0>("v_fma_f32", 8); run<1>("v_pk_fma_f32", 4); run<2>("v_mul_lo+add", 8); run<3>("u64 mul+add", 2); return 0; }
