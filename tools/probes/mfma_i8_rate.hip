// mfma_i8_rate.hip -- cycles per v_mfma_i32_32x32x32_i8 / v_mfma_i32_16x16x64_i8 (one accumulation chain per wave), 1-3 waves per SIMD
// Build: hipcc --offload-arch=gfx950 -O3 -w -o build/mfma_i8_rate tools/probes/mfma_i8_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
template <int KIND>
__global__ __launch_bounds__(256) void k(const v4i* in, int* out, int n, long long* cyc)
{
    v4i a = in[threadIdx.x], b = in[threadIdx.x + 256];
    v16i c16 = {0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0};
    v4i c4 = {0,0,0,0};
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; i += 16) {
#pragma unroll
        for (int u = 0; u < 16; u++) {
            if (KIND == 0) c16 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c16, 0, 0, 0);
            else c4 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c4, 0, 0, 0);
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    int s = 0;
    for (int i = 0; i < 16; i++) s += c16[i];
    for (int i = 0; i < 4; i++) s += c4[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int KIND> void run(const char* name, int blocks, int n)
{
    v4i* in; int* out; long long* cyc;
    hipMalloc(&in, 512 * 16); hipMemset(in, 1, 512 * 16); hipMalloc(&out, (size_t)blocks * 256 * 4); hipMalloc(&cyc, 8);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k<KIND><<<blocks, 256>>>(in, out, n, cyc);
    hipEventRecord(a);
    k<KIND><<<blocks, 256>>>(in, out, n, cyc);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    const double waves_per_simd = blocks * 4.0 / 1024.0;
    printf("%-12s %.0f wave(s) per SIMD, one accumulation chain each, n = %d per wave: %.3f ms = %.1f ns per MFMA and SIMD = %.1f cycles at 2.4 GHz\n", name,
           waves_per_simd, n, ms, ms * 1e6 / (n * waves_per_simd), ms * 1e6 / (n * waves_per_simd) * 2.4);
    hipFree(in); hipFree(out); hipFree(cyc);
}
int main()
{
    for (int blocks : {256, 512, 768}) { run<0>("32x32x32_i8", blocks, 32000); run<1>("16x16x64_i8", blocks, 32000); }
    return 0;
}
