#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <chrono>
#include <vector>
#include <atomic>
__global__ void k_sum(const uint32_t* p, uint32_t n, unsigned long long* out)
{
    unsigned long long s = 0;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) s += p[i];
    atomicAdd(out, s);
}
int main()
{
    const uint32_t n = 102400;   // 400 KB
    uint32_t* d = nullptr; uint32_t* dc = nullptr; unsigned long long* out = nullptr;
    hipExtMallocWithFlags(reinterpret_cast<void**>(&d), n * 4, hipDeviceMallocFinegrained);
    hipMalloc(&dc, n * 4);
    hipMalloc(&out, 8);
    hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    std::vector<uint32_t> src(n);
    int bad = 0;
    for (uint32_t r = 1; r <= 200; ++r) {
        for (auto& v : src) v = r;
        auto t0 = std::chrono::steady_clock::now();
        memcpy(d, src.data(), n * 4);
        std::atomic_thread_fence(std::memory_order_seq_cst);
        auto t1 = std::chrono::steady_clock::now();
        hipMemsetAsync(out, 0, 8, s);
        hipEventRecord(e0, s);
        hipLaunchKernelGGL(k_sum, dim3(256), dim3(256), 0, s, d, n, out);
        hipEventRecord(e1, s);
        unsigned long long h = 0; hipMemcpyAsync(&h, out, 8, hipMemcpyDeviceToHost, s); hipStreamSynchronize(s);
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        if (h != static_cast<unsigned long long>(r) * n) ++bad;
        if (r <= 3 || r == 200) printf("iter %u: host write %.1f us, kernel over fine-grained %.1f us, sum %s\n", r,
                                       std::chrono::duration<double, std::micro>(t1 - t0).count(), ms * 1e3, h == (unsigned long long)r * n ? "ok" : "STALE");
    }
    printf("stale results: %d / 200\n", bad);
    hipMemcpy(dc, src.data(), n * 4, hipMemcpyHostToDevice);
    hipMemsetAsync(out, 0, 8, s); hipEventRecord(e0, s);
    hipLaunchKernelGGL(k_sum, dim3(256), dim3(256), 0, s, dc, n, out);
    hipEventRecord(e1, s); hipStreamSynchronize(s);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    printf("kernel over ordinary device memory %.1f us\n", ms * 1e3);
    return 0;
}
