// Exhaustive check: for every float d in [1, 2^64) compare the IEEE quotient 1.0f / d (hipcc's default correctly rounded
// division: v_div_scale / v_rcp / fma chain / v_div_fmas / v_div_fixup, ~11 instructions) with short reciprocal sequences.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/rcp_exhaustive.hip -o /tmp/rcp_exh && /tmp/rcp_exh
#include <hip/hip_runtime.h>
#include <cstdio>

__device__ __forceinline__ float cand_a(float d)
{
    const float y0 = __builtin_amdgcn_rcpf(d);
    const float e = fmaf(-d, y0, 1.0f);
    return fmaf(e, y0, y0);
}
__device__ __forceinline__ float cand_b(float d)
{
    const float y1 = cand_a(d);
    const float e = fmaf(-d, y1, 1.0f);
    return fmaf(e, y1, y1);
}

__global__ void k_check(unsigned lo, unsigned hi, unsigned long long* bad)
{
    unsigned long long na = 0, nb = 0, nr = 0;
    for (unsigned long long b = lo + blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x; b < hi;
         b += (unsigned long long)gridDim.x * blockDim.x) {
        const float d = __uint_as_float((unsigned)b);
        const float ref = 1.0f / d;
        na += __float_as_uint(cand_a(d)) != __float_as_uint(ref);
        nb += __float_as_uint(cand_b(d)) != __float_as_uint(ref);
        nr += __float_as_uint(__builtin_amdgcn_rcpf(d)) != __float_as_uint(ref);
    }
    atomicAdd(&bad[0], na);
    atomicAdd(&bad[1], nb);
    atomicAdd(&bad[2], nr);
}

int main()
{
    unsigned long long* d_bad;
    unsigned long long h_bad[3] = {0, 0, 0};
    hipMalloc(&d_bad, sizeof(h_bad));
    hipMemset(d_bad, 0, sizeof(h_bad));
    const unsigned lo = 0x3F800000u, hi = 0x5F800000u;          // [1, 2^64)
    k_check<<<4096, 256>>>(lo, hi, d_bad);
    hipDeviceSynchronize();
    hipMemcpy(h_bad, d_bad, sizeof(h_bad), hipMemcpyDeviceToHost);
    printf("values %u  mismatches: one-step %llu  two-step %llu  raw v_rcp %llu\n", hi - lo, h_bad[0], h_bad[1], h_bad[2]);
    return 0;
}
