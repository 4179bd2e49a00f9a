// How long does "launch one tiny kernel and wait for it" take, by the way of waiting?  (speckv_access on a miss)
//   hipcc --offload-arch=gfx950 -O2 sync_latency.hip -o sync_latency && ./sync_latency
//   MI355X, ROCm 7.2: 10.9-11.6 us through the runtime (stream / event synchronize, spinning on a query), 7.1 us spinning on a flag
//   the kernel writes to pinned host memory after __threadfence_system().
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <algorithm>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void k_copy(const uint4* src, uint4* dst, volatile uint32_t* flag, uint32_t seq)
{
    dst[threadIdx.x] = src[threadIdx.x];                 // one 4 KiB page by 256 threads
    if (flag) {
        __threadfence_system();
        __syncthreads();
        if (threadIdx.x == 0) *flag = seq;
    }
}
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
    uint4 *src, *dst; CK(hipMalloc(&src, 4096)); CK(hipMalloc(&dst, 4096));
    uint32_t* flag; CK(hipHostMalloc(&flag, 64, hipHostMallocMapped)); *flag = 0;
    uint32_t* dflag; CK(hipHostGetDevicePointer(reinterpret_cast<void**>(&dflag), flag, 0));
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipEvent_t ev; CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    hipEvent_t evb; CK(hipEventCreateWithFlags(&evb, hipEventDisableTiming | hipEventBlockingSync));
    const int reps = 2000;
    uint32_t seq = 0;
    auto stat = [&](const char* name, std::vector<double>& v) {
        std::sort(v.begin(), v.end());
        printf("%-58s median %.2f us  p10 %.2f  p90 %.2f\n", name, v[v.size() / 2], v[v.size() / 10], v[v.size() * 9 / 10]);
    };
    for (int mode = 0; mode < 6; ++mode) {
        std::vector<double> v;
        for (int i = 0; i < reps + 50; ++i) {
            const double t0 = now_us();
            switch (mode) {
            case 0: hipLaunchKernelGGL(k_copy, dim3(1), dim3(256), 0, s, src, dst, nullptr, 0u); CK(hipStreamSynchronize(s)); break;
            case 1: hipLaunchKernelGGL(k_copy, dim3(1), dim3(256), 0, s, src, dst, nullptr, 0u); CK(hipEventRecord(ev, s)); CK(hipEventSynchronize(ev)); break;
            case 2: hipLaunchKernelGGL(k_copy, dim3(1), dim3(256), 0, s, src, dst, nullptr, 0u); CK(hipEventRecord(ev, s));
                    while (hipEventQuery(ev) == hipErrorNotReady) {} break;
            case 3: hipLaunchKernelGGL(k_copy, dim3(1), dim3(256), 0, s, src, dst, nullptr, 0u); while (hipStreamQuery(s) == hipErrorNotReady) {} break;
            case 4: ++seq; hipLaunchKernelGGL(k_copy, dim3(1), dim3(256), 0, s, src, dst, dflag, seq);
                    while (*reinterpret_cast<volatile uint32_t*>(flag) != seq) {} break;
            case 5: hipLaunchKernelGGL(k_copy, dim3(1), dim3(256), 0, s, src, dst, nullptr, 0u); CK(hipEventRecord(evb, s)); CK(hipEventSynchronize(evb)); break;
            }
            const double t1 = now_us();
            if (i >= 50) v.push_back(t1 - t0);
        }
        CK(hipStreamSynchronize(s));
        const char* names[] = {"launch + hipStreamSynchronize", "launch + hipEventRecord + hipEventSynchronize", "launch + hipEventRecord + spin on hipEventQuery",
                               "launch + spin on hipStreamQuery", "launch + spin on a flag the kernel writes to pinned host memory", "launch + record + sync of a BlockingSync event"};
        stat(names[mode], v);
    }
    return 0;
}
