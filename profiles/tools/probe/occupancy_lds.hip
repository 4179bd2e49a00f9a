// How many 256-thread workgroups with 40 960 B of LDS and ~110 VGPRs does an MI355X CU hold at once?  (k_attend_int4_wg's footprint:
// is it 3 per CU -- amdgpu_waves_per_eu(3,3) -- or 4, which LDS and registers allow?)   Every workgroup notes its CU (HW_ID) and the
// time it starts and ends; the host counts the overlap per CU.     hipcc --offload-arch=gfx950 -O2 occupancy_lds.hip -o occupancy_lds
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
#include <map>
#include <algorithm>
template <int LDS>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void k(uint64_t* rec, float* sink, uint32_t spin)
{
    __shared__ float lds[LDS / 4];
    float r[52];                                        // register pressure: ~110 VGPRs
#pragma unroll
    for (int i = 0; i < 52; ++i) r[i] = threadIdx.x * 0.001f + i;
    const uint64_t t0 = __builtin_readcyclecounter();
    uint32_t hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    uint32_t xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    for (uint32_t it = 0; it < spin; ++it) {
#pragma unroll
        for (int i = 0; i < 52; ++i) r[i] = r[i] * 1.0001f + 0.5f;
        lds[(threadIdx.x + it) % (LDS / 4)] = r[it % 52];
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 52; ++i) s += r[i];
    const uint64_t t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) { rec[3 * blockIdx.x] = t0; rec[3 * blockIdx.x + 1] = t1; rec[3 * blockIdx.x + 2] = (uint64_t(xcc & 15u) << 32) | hw; }
    if (s == 1.2345f) sink[0] = s + lds[threadIdx.x];
}
template <int LDS> void run(const char* name)
{
    const uint32_t n = 4096;
    uint64_t* d; float* sink; hipMalloc(&d, n * 24); hipMalloc(&sink, 4);
    hipLaunchKernelGGL(k<LDS>, dim3(n), dim3(256), 0, 0, d, sink, 2000u);
    hipDeviceSynchronize();
    std::vector<uint64_t> h(3 * n);
    hipMemcpy(h.data(), d, n * 24, hipMemcpyDeviceToHost);
    std::map<uint64_t, std::vector<std::pair<uint64_t, int>>> per_cu;         // (xcc, se, cu) -> events
    for (uint32_t i = 0; i < n; ++i) {
        const uint64_t id = h[3 * i + 2];
        const uint32_t hw = uint32_t(id), cu = (hw >> 8) & 15u, sh = (hw >> 12) & 1u, se = (hw >> 13) & 7u;
        const uint64_t key = (id >> 32) << 16 | se << 8 | sh << 4 | cu;
        per_cu[key].push_back({h[3 * i], +1});
        per_cu[key].push_back({h[3 * i + 1], -1});
    }
    int worst = 0; std::map<int, int> hist;
    for (auto& kv : per_cu) {
        auto& ev = kv.second; std::sort(ev.begin(), ev.end());
        int cur = 0, mx = 0;
        for (auto& e : ev) { cur += e.second; mx = std::max(mx, cur); }
        hist[mx]++; worst = std::max(worst, mx);
    }
    printf("%s: %zu CUs seen; max concurrent workgroups per CU:", name, per_cu.size());
    for (auto& kv : hist) printf("  %d on %d CUs", kv.first, kv.second);
    printf("\n");
    hipFree(d); hipFree(sink);
}
int main()
{
    run<40960>("LDS 40960 B (k_attend_int4_wg<false>)");
    run<41024>("LDS 41024 B (the striped form: + 64 B)");
    run<20480>("LDS 20480 B");
    return 0;
}
