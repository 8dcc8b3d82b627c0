// write_shapes.hip -- write-only bandwidth by store shape (which store instruction, how many bytes per wave-instruction, grid shape).
// Build: hipcc --offload-arch=gfx950 -O3 -o build/write_shapes tools/probes/write_shapes.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int W, bool NT> __device__ __forceinline__ void st(float* p, float v)
{
    if constexpr (W == 1) { if (NT) __builtin_nontemporal_store(v, p); else *p = v; }
    else if constexpr (W == 2) { float2 q = make_float2(v, v); if (NT) { __builtin_nontemporal_store(q.x, p); __builtin_nontemporal_store(q.y, p + 1); } else *reinterpret_cast<float2*>(p) = q; }
    else { typedef float f4 __attribute__((ext_vector_type(4))); f4 q = {v, v, v, v}; if (NT) __builtin_nontemporal_store(q, reinterpret_cast<f4*>(p)); else *reinterpret_cast<f4*>(p) = q; }
}
// every wave writes chunks of 4 KB: W floats per lane and instruction, 16 / W instructions per chunk, each instruction contiguous
template <int W, bool NT, bool SEQ = false>
__global__ __launch_bounds__(256) void k_write(float* __restrict__ dst, long nfloat, int chunks_per_wave)
{
    const long wave = ((long)blockIdx.x * 256 + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    const long nwaves = (long)gridDim.x * 4;
    for (int c = 0; c < chunks_per_wave; c++) {
        // consecutive waves write consecutive chunks (grid-stride), or SEQ: every wave writes its own contiguous region front to back
        const long chunk = SEQ ? wave * chunks_per_wave + c : (long)c * nwaves + wave;
        float* base = dst + chunk * 1024;
        if ((chunk + 1) * 1024 > nfloat) return;
#pragma unroll
        for (int j = 0; j < 16 / W; j++) st<W, NT>(base + j * 64 * W + lane * W, 1.f);
    }
}
template <int W, bool NT, bool SEQ = false> double run(float* d, long nfloat, int blocks, int iters)
{
    const long nwaves = (long)blocks * 4;
    const int cpw = (int)(nfloat / 1024 / nwaves);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k_write<W, NT, SEQ><<<blocks, 256>>>(d, nfloat, cpw);
    hipEventRecord(a);
    for (int i = 0; i < iters; i++) k_write<W, NT, SEQ><<<blocks, 256>>>(d, nfloat, cpw);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms = 0; hipEventElapsedTime(&ms, a, b);
    hipEventDestroy(a); hipEventDestroy(b);
    return (double)cpw * nwaves * 4096.0 * iters / (ms * 1e-3) / 1e9;
}
int main()
{
    const long nfloat = 1L << 30;                               // 4 GiB
    float* d = nullptr;
    if (hipMalloc((void**)&d, nfloat * 4) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(d, 0, nfloat * 4);
    const int grids[] = {256 * 2, 256 * 4, 256 * 8, 256 * 16, 256 * 64, 256 * 1024};
    printf("%-10s", "blocks");
    for (const char* n : {"x1", "x1 nt", "x2", "x2 nt", "x4", "x4 nt", "x4 seq", "x4 nt seq"}) printf("%10s", n);
    printf("   GB/s (write only, 4 GiB, 256-thread blocks)\n");
    for (int g : grids) {
        printf("%-10d", g);
        printf("%10.0f", run<1, false>(d, nfloat, g, 5)); printf("%10.0f", run<1, true>(d, nfloat, g, 5));
        printf("%10.0f", run<2, false>(d, nfloat, g, 5)); printf("%10.0f", run<2, true>(d, nfloat, g, 5));
        printf("%10.0f", run<4, false>(d, nfloat, g, 5)); printf("%10.0f", run<4, true>(d, nfloat, g, 5));
        printf("%10.0f", run<4, false, true>(d, nfloat, g, 5)); printf("%10.0f", run<4, true, true>(d, nfloat, g, 5));
        printf("\n");
    }
    hipFree(d);
    return 0;
}
