// Write-only and copy ceilings of the chip by access shape: how fast can 1 GiB be filled / copied when a wave moves 1, 4 or 16 KiB,
// in one piece or interleaved with the other waves of its workgroup?  (all-zero / flat-run blocks decode to almost pure stores; the
// block decoder's wave reads a record and writes its own 4 KiB block)
//   hipcc --offload-arch=gfx950 -O3 -o fill_rate fill_rate.hip && ./fill_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
// MODE 0: a wave owns PER consecutive KiB.  MODE 1: the workgroup's 4 waves own 4 PER consecutive KiB and sweep them together
// (wave w takes KiB 4 j + w of the workgroup's stretch in step j).
template <bool NT, int PER, int MODE, bool COPY>
__global__ __launch_bounds__(256) void k_move(const u32x4* __restrict__ src, u32x4* __restrict__ dst, uint64_t n16, uint32_t v)
{
    const uint32_t lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
    u32x4 x[PER];
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const uint64_t kib = MODE == 0 ? (static_cast<uint64_t>(blockIdx.x) * 4u + w) * PER + j : static_cast<uint64_t>(blockIdx.x) * 4u * PER + 4u * j + w;
        const uint64_t i = kib * 64u + lane;
        x[j] = u32x4{v, v, v, v};
        if (COPY && i < n16) x[j] = NT ? __builtin_nontemporal_load(src + i) : src[i];
    }
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const uint64_t kib = MODE == 0 ? (static_cast<uint64_t>(blockIdx.x) * 4u + w) * PER + j : static_cast<uint64_t>(blockIdx.x) * 4u * PER + 4u * j + w;
        const uint64_t i = kib * 64u + lane;
        if (i < n16) { if (NT) __builtin_nontemporal_store(x[j], dst + i); else dst[i] = x[j]; }
    }
}
// the same with workgroups of WAVES waves (a wave owns PER consecutive KiB)
template <int PER, int WAVES, bool COPY>
__global__ __launch_bounds__(64 * WAVES) void k_move_wg(const u32x4* __restrict__ src, u32x4* __restrict__ dst, uint64_t n16, uint32_t v)
{
    const uint32_t lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
    u32x4 x[PER];
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const uint64_t i = ((static_cast<uint64_t>(blockIdx.x) * WAVES + w) * PER + j) * 64u + lane;
        x[j] = u32x4{v, v, v, v};
        if (COPY && i < n16) x[j] = __builtin_nontemporal_load(src + i);
    }
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const uint64_t i = ((static_cast<uint64_t>(blockIdx.x) * WAVES + w) * PER + j) * 64u + lane;
        if (i < n16) __builtin_nontemporal_store(x[j], dst + i);
    }
}
// read 4 KiB per wave, write 4 / DIV KiB of it (the compress kernels' mixes: INT4 4096 -> 1152, INT8 / FP8 4096 -> 2052): what
// does the chip deliver in TOTAL when most of the traffic is reads?
template <int DIV>
__global__ __launch_bounds__(256) void k_shrink(const u32x4* __restrict__ src, u32x4* __restrict__ dst, uint64_t n16)
{
    const uint32_t lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
    const uint64_t wave = static_cast<uint64_t>(blockIdx.x) * 4u + w;
    u32x4 x[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) x[j] = __builtin_nontemporal_load(src + (wave * 4u + j) * 64u + lane);
    if (DIV == 0) {                                   // read only: keep the loads alive
        const u32x4 t = x[0] ^ x[1] ^ x[2] ^ x[3];
        if ((t.x ^ t.y ^ t.z ^ t.w) == 0x12345678u) dst[0] = t;
        return;
    }
    // 4 / DIV KiB out: lanes [0, 64 * 4 / DIV) of the wave's four pieces, folded
    if (DIV == 4) { const u32x4 t = x[0] ^ x[1] ^ x[2] ^ x[3]; __builtin_nontemporal_store(t, dst + wave * 64u + lane); }
    if (DIV == 2) { __builtin_nontemporal_store(x[0] ^ x[1], dst + (wave * 2u) * 64u + lane); __builtin_nontemporal_store(x[2] ^ x[3], dst + (wave * 2u + 1u) * 64u + lane); }
}
int main()
{
    const uint64_t bytes = 1ull << 30, n16 = bytes / 16;
    void *d, *s2; (void)hipMalloc(&d, bytes); (void)hipMalloc(&s2, bytes);
    (void)hipMemset(s2, 1, bytes);
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    auto time = [&](const char* name, double moved, auto fn) {
        for (int i = 0; i < 20; ++i) fn();
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(a);
        for (int i = 0; i < 20; ++i) fn();
        (void)hipEventRecord(b); (void)hipEventSynchronize(b);
        float ms; (void)hipEventElapsedTime(&ms, a, b); ms /= 20;
        printf("%-46s %.4f ms  %.0f GB/s  %.3f of 8 TB/s\n", name, ms, moved / (ms * 1e-3) / 1e9, moved / (ms * 1e-3) / 8e12);
    };
    const uint32_t kib = bytes / 1024;
#define RUN(NAME, NT, PER, MODE, COPY) time(NAME, (COPY ? 2.0 : 1.0) * bytes, [&] { hipLaunchKernelGGL((k_move<NT, PER, MODE, COPY>), dim3(kib / 4 / PER), dim3(256), 0, 0, (const u32x4*)s2, (u32x4*)d, n16, 0u); })
    RUN("fill nt, 1 KiB per wave", true, 1, 0, false);
    RUN("fill nt, 4 KiB per wave", true, 4, 0, false);
    RUN("fill nt, 4 KiB per wave, swept by the workgroup", true, 4, 1, false);
    RUN("fill plain, 4 KiB per wave", false, 4, 0, false);
    RUN("fill nt, 16 KiB per wave", true, 16, 0, false);
    RUN("fill nt, 16 KiB per wave, swept", true, 16, 1, false);
    time("hipMemsetAsync", 1.0 * bytes, [&] { (void)hipMemsetAsync(d, 0, bytes, 0); });
    RUN("copy nt, 1 KiB per wave", true, 1, 0, true);
    RUN("copy nt, 4 KiB per wave", true, 4, 0, true);
    RUN("copy nt, 4 KiB per wave, swept", true, 4, 1, true);
    RUN("copy plain, 4 KiB per wave", false, 4, 0, true);
    RUN("copy nt, 2 KiB per wave", true, 2, 0, true);
#define RUNW(NAME, PER, WAVES, COPY) time(NAME, (COPY ? 2.0 : 1.0) * bytes, [&] { hipLaunchKernelGGL((k_move_wg<PER, WAVES, COPY>), dim3(kib / WAVES / PER), dim3(64 * WAVES), 0, 0, (const u32x4*)s2, (u32x4*)d, n16, 0u); })
    RUNW("fill nt, 4 KiB per wave, 1-wave workgroups", 4, 1, false);
    RUNW("fill nt, 4 KiB per wave, 2-wave workgroups", 4, 2, false);
    RUNW("fill nt, 4 KiB per wave, 16-wave workgroups", 4, 16, false);
    RUNW("fill nt, 1 KiB per wave, 1-wave workgroups", 1, 1, false);
    RUNW("fill nt, 1 KiB per wave, 16-wave workgroups", 1, 16, false);
    RUNW("copy nt, 4 KiB per wave, 1-wave workgroups", 4, 1, true);
    RUNW("copy nt, 4 KiB per wave, 2-wave workgroups", 4, 2, true);
    RUNW("copy nt, 4 KiB per wave, 16-wave workgroups", 4, 16, true);
    RUNW("copy nt, 1 KiB per wave, 1-wave workgroups", 1, 1, true);
    RUNW("copy nt, 1 KiB per wave, 16-wave workgroups", 1, 16, true);
    time("read 4 KiB per wave, write nothing", 1.0 * bytes, [&] { hipLaunchKernelGGL(k_shrink<0>, dim3(kib / 16), dim3(256), 0, 0, (const u32x4*)s2, (u32x4*)d, n16); });
    time("read 4 KiB per wave, write 1 KiB (INT4 mix)", 1.25 * bytes, [&] { hipLaunchKernelGGL(k_shrink<4>, dim3(kib / 16), dim3(256), 0, 0, (const u32x4*)s2, (u32x4*)d, n16); });
    time("read 4 KiB per wave, write 2 KiB (INT8 mix)", 1.5 * bytes, [&] { hipLaunchKernelGGL(k_shrink<2>, dim3(kib / 16), dim3(256), 0, 0, (const u32x4*)s2, (u32x4*)d, n16); });
    time("hipMemcpyAsync d2d", 2.0 * bytes, [&] { (void)hipMemcpyAsync(d, s2, bytes, hipMemcpyDeviceToDevice, 0); });
    return 0;
}
