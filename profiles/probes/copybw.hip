// scratch probe: copy bandwidth vs burst shape (not part of the product)
#include <hip/hip_runtime.h>
#include <cstdint>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
// each wave copies chunks of UNR KiB: UNR loads (1 KiB each) then UNR stores
template <int UNR, bool NT>
__global__ __launch_bounds__(256) void k_copy(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, uint64_t bytes)
{
    const uint64_t wave = (blockIdx.x * 256ull + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    const uint64_t nw = (gridDim.x * 256ull) >> 6;
    const uint64_t chunk = 1024ull * UNR;
    for (uint64_t off = wave * chunk; off + chunk <= bytes; off += nw * chunk) {
        u32x4 v[UNR];
#pragma unroll
        for (int i = 0; i < UNR; ++i) {
            const u32x4* q = reinterpret_cast<const u32x4*>(src + off + 1024ull * i + 16 * lane);
            v[i] = NT ? __builtin_nontemporal_load(q) : *q;
        }
#pragma unroll
        for (int i = 0; i < UNR; ++i) {
            u32x4* q = reinterpret_cast<u32x4*>(dst + off + 1024ull * i + 16 * lane);
            if (NT) __builtin_nontemporal_store(v[i], q); else *q = v[i];
        }
    }
}
extern "C" int probe_copy(const void* s, void* d, uint64_t bytes, int unr, int nt, int wgs, void* stream)
{
    hipStream_t st = (hipStream_t)stream;
#define L(U, N) hipLaunchKernelGGL((k_copy<U, N>), dim3(wgs), dim3(256), 0, st, (const uint8_t*)s, (uint8_t*)d, bytes)
    if (unr == 1) { if (nt) L(1, true); else L(1, false); }
    else if (unr == 4) { if (nt) L(4, true); else L(4, false); }
    else if (unr == 8) { if (nt) L(8, true); else L(8, false); }
    else { if (nt) L(16, true); else L(16, false); }
    return (int)hipGetLastError();
}
