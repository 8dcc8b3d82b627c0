// Is hipMemset on device memory complete when it returns, and is it ordered in front of work on a NON-BLOCKING stream?
// (the library's contexts run on non-blocking streams; hipMemset fills on the NULL stream).  A 1 GiB fill takes ~200 us; a kernel
// launched on a non-blocking stream right after hipMemset returns reads words at the end of the buffer.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
__global__ void k_read(const unsigned* p, size_t n, unsigned* out) { out[threadIdx.x] = p[n - 1 - threadIdx.x * 4096]; }
__global__ void k_read16(const unsigned* p, unsigned* out) { out[threadIdx.x] = p[1023 - 16 * threadIdx.x]; }
__global__ void k_read2d(const unsigned* p, unsigned* out) { out[threadIdx.x] = p[(size_t)(32767 - 37 * threadIdx.x) * 2048 + 1023]; }
int main()
{
    const size_t bytes = 1ull << 30, n = bytes / 4;
    unsigned *buf, *out, h[64];
    hipStream_t st;
    hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    hipMalloc(&buf, bytes); hipMalloc(&out, 256);
    int stale_total = 0;
    for (int rep = 0; rep < 20; rep++) {
        hipMemset(buf, rep & 1 ? 0x11 : 0xEE, bytes);      // (previous contents: the other pattern)
        hipDeviceSynchronize();
        const unsigned want = rep & 1 ? 0xEEEEEEEEu : 0x11111111u;
        const auto t0 = std::chrono::steady_clock::now();
        hipMemset(buf, rep & 1 ? 0xEE : 0x11, bytes);
        const auto t1 = std::chrono::steady_clock::now();
        k_read<<<1, 64, 0, st>>>(buf, n, out);
        hipStreamSynchronize(st);
        hipMemcpy(h, out, 256, hipMemcpyDeviceToHost);
        int stale = 0;
        for (int i = 0; i < 64; i++) stale += h[i] != want;
        stale_total += stale;
        if (rep < 4 || stale) printf("rep %d: hipMemset(1 GiB) returned after %.0f us; %d of 64 words read on a non-blocking stream right after were STALE\n",
                                     rep, std::chrono::duration<double, std::micro>(t1 - t0).count(), stale);
    }
    printf("hipMemset then a kernel on a non-blocking stream: %d stale words in 20 x 64\n", stale_total);
    // small fills (the size of the matcher's ticket array): 4 KiB, the kernel reads word 1023 - 16 t of the page
    int small_stale = 0;
    for (int rep = 0; rep < 20000; rep++) {
        hipMemsetAsync(buf, 0x77, 4096, st);
        hipStreamSynchronize(st);
        hipMemset(buf, 0, 4096);
        k_read16<<<1, 64, 0, st>>>(buf, out);
        hipStreamSynchronize(st);
        hipMemcpy(h, out, 256, hipMemcpyDeviceToHost);
        int stale = 0;
        for (int i = 0; i < 64; i++) stale += h[i] != 0;
        small_stale += stale != 0;
    }
    printf("hipMemset(4 KiB) then a kernel on a non-blocking stream: stale reads in %d of 20000 launches\n", small_stale);
    // ... and the blocking copies a caller uploads images with: hipMemcpy / hipMemcpy2D from pageable host memory
    const size_t cb = 256u << 20;
    unsigned* hostbuf = (unsigned*)malloc(cb);
    int copy_stale = 0;
    for (int rep = 0; rep < 10; rep++) {
        for (size_t i = 0; i < cb / 4; i++) hostbuf[i] = 0xA0000000u + rep;
        hipMemcpy(buf, hostbuf, cb, hipMemcpyHostToDevice);
        k_read<<<1, 64, 0, st>>>(buf, cb / 4, out);
        hipStreamSynchronize(st);
        hipMemcpy(h, out, 256, hipMemcpyDeviceToHost);
        for (int i = 0; i < 64; i++) copy_stale += h[i] != 0xA0000000u + rep;
        hipMemcpy2D(buf, 8192, hostbuf, 4096, 4096, 32768, hipMemcpyHostToDevice);      // 128 MiB in rows of 4 KiB at a pitch of 8 KiB
        k_read2d<<<1, 64, 0, st>>>(buf, out);
        hipStreamSynchronize(st);
        hipMemcpy(h, out, 256, hipMemcpyDeviceToHost);
        for (int i = 0; i < 64; i++) copy_stale += h[i] != 0xA0000000u + rep;
    }
    printf("hipMemcpy / hipMemcpy2D (pageable host -> device) then a kernel on a non-blocking stream: %d stale words in 10 x 128\n", copy_stale);
    // ... and small ones (a context's parameter tables are a few KiB): 4 KiB, the kernel reads word 1023 - 16 t
    int small_copy_stale = 0;
    for (int rep = 0; rep < 20000; rep++) {
        for (int i = 0; i < 1024; i++) hostbuf[i] = 0xB0000000u + rep;
        hipMemcpy(buf, hostbuf, 4096, hipMemcpyHostToDevice);
        k_read16<<<1, 64, 0, st>>>(buf, out);
        hipStreamSynchronize(st);
        hipMemcpy(h, out, 256, hipMemcpyDeviceToHost);
        int stale = 0;
        for (int i = 0; i < 64; i++) stale += h[i] != 0xB0000000u + rep;
        small_copy_stale += stale != 0;
    }
    printf("hipMemcpy(4 KiB, pageable host -> device) then a kernel on a non-blocking stream: stale reads in %d of 20000 launches\n", small_copy_stale);
    return 0;
}
