// hip_utils.h -- HIP counterpart of the reference's cuda_utils.h (same spellings: CHECK, CheckMsg,
// initDevice, GpuTimer, cpuTimer, iAlignUp, iDivUp, iExp2UpP) so a main.cpp-style caller ports by
// changing one include and cuda* -> hip* runtime calls.  Host-side only.
#pragma once
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <hip/hip_runtime.h>

#define H_PI 1.5707963267948966f                                   // cuda_utils.h:7

#define CHECK(err) hak_utils::check((err), __FILE__, __LINE__)       // cuda_utils.h:9, 18-25
#define CheckMsg(msg) hak_utils::check_msg((msg), __FILE__, __LINE__) // cuda_utils.h:10, 27-37

namespace hak_utils
{
    inline void check(hipError_t err, const char* file, int line)
    {
        if (err == hipSuccess) return;
        fprintf(stderr, "CHECK() Runtime API error in file <%s>, line %i : %s.\n", file, line, hipGetErrorString(err));
        exit(-1);
    }
    inline void check_msg(const char* msg, const char* file, int line)
    {
        hipError_t err = hipGetLastError();
        if (err == hipSuccess) return;
        fprintf(stderr, "CheckMsg() HIP error: %s in file <%s>, line %i : %s.\n", msg, file, line, hipGetErrorString(err));
        exit(-1);
    }
}

// cuda_utils.h:41-67
inline bool initDevice(int dev)
{
    int n = 0;
    CHECK(hipGetDeviceCount(&n));
    if (n == 0) { fprintf(stderr, "HIP error: no devices supporting HIP.\n"); return false; }
    dev = std::max(0, std::min(dev, n - 1));
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, dev));
    CHECK(hipSetDevice(dev));
    int drv = 0, rt = 0;
    CHECK(hipDriverGetVersion(&drv));
    CHECK(hipRuntimeGetVersion(&rt));
    fprintf(stderr, "Using Device %d: %s (%s), HIP Driver Version: %d, Runtime Version: %d\n", dev, prop.name,
            prop.gcnArchName, drv, rt);
    return true;
}

// cuda_utils.h:71-77
inline long long cpuTimer()
{
    return std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::system_clock::now().time_since_epoch()).count();
}

// cuda_utils.h:81-108: event pair on a stream, read() = ms since construction
class GpuTimer
{
public:
    explicit GpuTimer(hipStream_t s = 0) : stream(s)
    {
        (void)hipEventCreate(&start);
        (void)hipEventCreate(&stop);
        (void)hipEventRecord(start, stream);
    }
    ~GpuTimer() { (void)hipEventDestroy(start); (void)hipEventDestroy(stop); }
    float read()
    {
        (void)hipEventRecord(stop, stream);
        (void)hipEventSynchronize(stop);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, start, stop);
        return ms;
    }
private:
    hipEvent_t start, stop;
    hipStream_t stream;
};

inline int iAlignUp(const int a, const int b) { return (a % b != 0) ? (a - a % b + b) : a; }   // cuda_utils.h:160
inline int iDivUp(int a, int b) { return (a % b != 0) ? (a / b + 1) : (a / b); }                // cuda_utils.h:167
inline int iExp2UpP(const int a) { int p = 0, v = 1; while (v < a) { v <<= 1; p++; } return p; } // cuda_utils.h:174
