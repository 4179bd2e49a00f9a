// scratch probe: read-only HBM bandwidth for a few access shapes (not part of the product)
#include <hip/hip_runtime.h>
#include <cstdint>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
// mode 0: linear, each wave-instruction 1 KiB contiguous, UNR loads in flight per wave
template <int UNR, bool NT>
__global__ __launch_bounds__(256) void k_linear(const uint8_t* __restrict__ p, uint64_t bytes, uint32_t* out)
{
    const uint64_t wave = (blockIdx.x * 256ull + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    const uint64_t nw = (gridDim.x * 256ull) >> 6;
    u32x4 acc = {0, 0, 0, 0};
    const uint64_t chunk = 1024ull * UNR;
    for (uint64_t off = wave * chunk; off + chunk <= bytes; off += nw * chunk) {
        u32x4 v[UNR];
#pragma unroll
        for (int i = 0; i < UNR; ++i) {
            const u32x4* q = reinterpret_cast<const u32x4*>(p + off + 1024ull * i + 16 * lane);
            v[i] = NT ? __builtin_nontemporal_load(q) : *q;
        }
#pragma unroll
        for (int i = 0; i < UNR; ++i) acc ^= v[i];
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) out[0] = 1;
}
// mode 1: the attend shape: wave = (layer, head, split); per tile 32 K rows + 32 V rows of 128 B at 1 KiB stride
__global__ __launch_bounds__(256) void k_heads(const uint8_t* __restrict__ p, uint32_t T, uint32_t n_splits, uint32_t tiles_per_split, uint32_t rot, uint32_t* out)
{
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t split = blockIdx.x, layer = blockIdx.y >> 1, head = (blockIdx.y & 1) * 4 + wave;
    const uint8_t* kbase = p + (uint64_t)layer * T * 2048ull + head * 128u;     // K region: T rows of 1 KiB; V region follows
    const uint8_t* vbase = kbase + (uint64_t)T * 1024ull;
    u32x4 acc = {0, 0, 0, 0};
    const uint32_t r0 = rot ? (split * 5u + blockIdx.y * 3u) % tiles_per_split : 0u;
    for (uint32_t i = 0; i < tiles_per_split; ++i) {
        const uint32_t tile = split * tiles_per_split + (i + r0) % tiles_per_split;
        const uint64_t row0 = (uint64_t)tile * 32u;
        u32x4 v[8];
#pragma unroll
        for (int j = 0; j < 4; ++j) {      // 8 rows per instruction: lane -> (row = lane/8, 16-byte chunk = lane%8)
            v[j] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(kbase + (row0 + 8 * j + (lane >> 3)) * 1024ull + 16 * (lane & 7)));
            v[4 + j] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(vbase + (row0 + 8 * j + (lane >> 3)) * 1024ull + 16 * (lane & 7)));
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) acc ^= v[j];
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) out[0] = 1;
}
extern "C" int probe_linear(const void* p, uint64_t bytes, int unr, int nt, int wgs, void* out, void* stream)
{
    hipStream_t s = (hipStream_t)stream;
#define L(U, N) hipLaunchKernelGGL((k_linear<U, N>), dim3(wgs), dim3(256), 0, s, (const uint8_t*)p, bytes, (uint32_t*)out)
    if (unr == 1) { if (nt) L(1, true); else L(1, false); }
    else if (unr == 4) { if (nt) L(4, true); else L(4, false); }
    else { if (nt) L(8, true); else L(8, false); }
    return (int)hipGetLastError();
}
extern "C" int probe_heads(const void* p, uint32_t T, uint32_t layers, uint32_t n_splits, uint32_t rot, void* out, void* stream)
{
    const uint32_t tps = T / 32 / n_splits;
    hipLaunchKernelGGL(k_heads, dim3(n_splits, layers * 2), dim3(256), 0, (hipStream_t)stream, (const uint8_t*)p, T, n_splits, tps, rot, (uint32_t*)out);
    return (int)hipGetLastError();
}
