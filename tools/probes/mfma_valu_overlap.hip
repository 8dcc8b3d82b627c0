// mfma_valu_overlap.hip -- how much independent vector work hides beside the matcher's matrix instruction: per wave a chain of
// v_mfma_f32_32x32x64_f8f6f4 (fp4 operands) with NV independent v_and_b32 between two of them, 1-3 waves per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 -w -o build/mfma_valu_overlap tools/probes/mfma_valu_overlap.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
// the matcher's own mix: per MFMA four v_and_b32 that feed its A fragment plus NE x (v_lshl_add_u32 on an element of the OTHER
// accumulator + v_min_u32 into a running minimum)
template <int NE>
__global__ __launch_bounds__(256) void kmix(const int* in, float* out, int n)
{
    v8i a = {0,0,0,0,0,0,0,0}, b = {0,0,0,0,0,0,0,0};
    for (int i = 0; i < 4; i++) { a[i] = in[threadIdx.x + 64 * i]; b[i] = in[threadIdx.x + 256 + 64 * i]; }
    int f[8];
    for (int i = 0; i < 8; i++) f[i] = in[threadIdx.x + 8 * i];
    unsigned best[16];
    for (int i = 0; i < 16; i++) best[i] = 0xFFFFFFFFu;
    v16f c0 = {0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0}, c1 = c0;
    unsigned jb = in[threadIdx.x];
    for (int i = 0; i < n; i += 16) {
#pragma unroll
        for (int t = 0; t < 2; t++) {
#pragma unroll
            for (int u = 0; u < 8; u++) {
#pragma unroll
                for (int v = 0; v < 4; v++) asm volatile("v_and_b32 %0, %1, %2" : "=v"(a[v]) : "v"(f[(u + v) & 7]), "v"(0x11111111 << v));
                if (t == 0) c0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c0, 4, 4, 0, 0, 0, 0);
                else c1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c1, 4, 4, 0, 0, 0, 0);
#pragma unroll
                for (int e = 0; e < NE; e++) {
                    const int idx = (2 * u + e) & 15;
                    const unsigned key = (__float_as_uint(t == 0 ? c1[idx] : c0[idx]) << 20) + jb;
                    best[idx] = min(best[idx], key);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    float s = 0;
    for (int i = 0; i < 16; i++) s += c0[i] + c1[i] + (float)best[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NE> void runmix(int blocks, int n)
{
    int* in; float* out;
    hipMalloc(&in, 1024 * 4); hipMemset(in, 0x22, 1024 * 4); hipMalloc(&out, (size_t)blocks * 256 * 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    kmix<NE><<<blocks, 256>>>(in, out, n);
    hipEventRecord(a);
    kmix<NE><<<blocks, 256>>>(in, out, n);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double wps = blocks * 4.0 / 1024.0;
    printf("matcher mix: 4 v_and + %d x (v_lshl_add_u32 + v_min_u32) per MFMA, %.0f wave(s) per SIMD: %.1f ns per MFMA and SIMD\n", NE, wps, ms * 1e6 / (n * wps));
    hipFree(in); hipFree(out);
}
template <int NV, bool DEP>
__global__ __launch_bounds__(256) void k(const int* in, float* out, int n)
{
    v8i a = {0,0,0,0,0,0,0,0}, b = {0,0,0,0,0,0,0,0};
    for (int i = 0; i < 4; i++) { a[i] = in[threadIdx.x + 64 * i]; b[i] = in[threadIdx.x + 256 + 64 * i]; }
    int f[8];
    for (int i = 0; i < 8; i++) f[i] = in[threadIdx.x + 8 * i];
    v16f c = {0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0};
    for (int i = 0; i < n; i += 8) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            if (DEP) {      // the matcher's shape: the A fragment of every MFMA comes from 4 fresh v_and (the rest of NV independent)
#pragma unroll
                for (int v = 0; v < 4 && v < NV; v++) asm volatile("v_and_b32 %0, %1, %2" : "=v"(a[v]) : "v"(f[(u + v) & 7]), "v"(0x11111111 << v));
#pragma unroll
                for (int v = 4; v < NV; v++) asm volatile("v_and_b32 %0, %1, %0" : "+v"(f[v & 7]) : "v"(0x7FFFFFFF));
            } else {
#pragma unroll
                for (int v = 0; v < NV; v++) asm volatile("v_and_b32 %0, %1, %0" : "+v"(f[v & 7]) : "v"(0x7FFFFFFF));
            }
            c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 4, 4, 0, 0, 0, 0);
        }
    }
    float s = 0;
    for (int i = 0; i < 16; i++) s += c[i];
    for (int i = 0; i < 8; i++) s += (float)f[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NV, bool DEP> void run(int blocks, int n)
{
    int* in; float* out;
    hipMalloc(&in, 1024 * 4); hipMemset(in, 0x22, 1024 * 4); hipMalloc(&out, (size_t)blocks * 256 * 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k<NV, DEP><<<blocks, 256>>>(in, out, n);
    hipEventRecord(a);
    k<NV, DEP><<<blocks, 256>>>(in, out, n);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double wps = blocks * 4.0 / 1024.0;
    printf("%s NV = %2d vector instructions per MFMA, %.0f wave(s) per SIMD: %.1f ns per MFMA and SIMD = %.1f cycles at 2.4 GHz\n",
           DEP ? "feeding " : "beside  ", NV, wps, ms * 1e6 / (n * wps), ms * 1e6 / (n * wps) * 2.4);
    hipFree(in); hipFree(out);
}
int main()
{
    for (int blocks : {256, 512, 768}) {
        run<0, false>(blocks, 16000); run<2, false>(blocks, 16000); run<4, false>(blocks, 16000); run<6, false>(blocks, 16000);
        run<8, false>(blocks, 16000); run<12, false>(blocks, 16000); run<16, false>(blocks, 16000);
        run<4, true>(blocks, 16000); run<8, true>(blocks, 16000);
        runmix<0>(blocks, 16000); runmix<1>(blocks, 16000); runmix<2>(blocks, 16000); runmix<4>(blocks, 16000);
    }
    return 0;
}
