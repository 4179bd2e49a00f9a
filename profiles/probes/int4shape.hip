// scratch probe: read-only bandwidth of the INT4 fused attention's address pattern and of variations of it (not part of the product)
//   A workgroup (4 waves) reads, per tile of 16 records of `stride` bytes, the 16-byte chunks the host lists for it (offsets relative to the
//   tile's first K record; V-region chunks carry the region distance already), NI wave instructions per wave and tile, DEPTH tiles in flight.
#include <hip/hip_runtime.h>
#include <cstdint>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
template <int NI, int DEPTH>
__global__ __launch_bounds__(256) void k_shape(const uint8_t* __restrict__ p, const uint64_t* __restrict__ table, uint32_t groups, uint64_t layer_bytes, uint32_t tile_bytes,
                                               uint32_t tiles_per_split, uint32_t* out)
{
    extern __shared__ uint8_t dyn_lds[];
    if (tiles_per_split == 0xFFFFFFFFu) dyn_lds[threadIdx.x] = 1;        // (keeps the allocation: occupancy control)
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t split = blockIdx.x, layer = blockIdx.y / groups, grp = blockIdx.y % groups;
    const uint8_t* base = p + (uint64_t)layer * layer_bytes + (uint64_t)split * tiles_per_split * tile_bytes;
    uint64_t off[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) off[i] = table[((grp * 4u + wave) * NI + i) * 64u + lane];
    u32x4 acc = {0, 0, 0, 0};
    if (DEPTH == 3) {                                   // rolling: two tiles in flight, the older one consumed and re-issued (the fused kernel's pipeline)
        u32x4 A[NI], B[NI];
        auto ld = [&](u32x4 (&r)[NI], uint32_t t) {
            const uint64_t to = (uint64_t)(t < tiles_per_split ? t : tiles_per_split - 1u) * tile_bytes;
#pragma unroll
            for (int i = 0; i < NI; ++i) r[i] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(base + to + off[i]));
        };
        auto eat = [&](const u32x4 (&r)[NI]) {
#pragma unroll
            for (int i = 0; i < NI; ++i) acc ^= r[i];
        };
        ld(A, 0); ld(B, 1);
#pragma unroll 1
        for (uint32_t t = 0; t + 2u < tiles_per_split; t += 2) { eat(A); ld(A, t + 2u); eat(B); ld(B, t + 3u); }
        eat(A); eat(B);
        if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) out[0] = 1;
        return;
    }
    for (uint32_t t = 0; t < tiles_per_split; t += DEPTH) {
        u32x4 v[DEPTH][NI];
#pragma unroll
        for (int d = 0; d < DEPTH; ++d)
#pragma unroll
            for (int i = 0; i < NI; ++i)
                v[d][i] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(base + (uint64_t)(t + d) * tile_bytes + off[i]));
#pragma unroll
        for (int d = 0; d < DEPTH; ++d)
#pragma unroll
            for (int i = 0; i < NI; ++i) acc ^= v[d][i];
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) out[0] = 1;
}
extern "C" int probe_shape(const void* p, const void* table, uint32_t ni, uint32_t depth, uint32_t groups, uint64_t layer_bytes, uint32_t tile_bytes, uint32_t tiles_per_layer,
                           uint32_t layers, uint32_t n_splits, uint32_t lds_bytes, void* out, void* stream)
{
    const uint32_t tps = tiles_per_layer / n_splits;
    dim3 g(n_splits, layers * groups);
#define GO(N, D) hipLaunchKernelGGL((k_shape<N, D>), g, dim3(256), lds_bytes, (hipStream_t)stream, (const uint8_t*)p, (const uint64_t*)table, groups, layer_bytes, tile_bytes, tps, (uint32_t*)out)
#define BOTH(N) if (ni == N) { if (depth == 1) GO(N, 1); else if (depth == 2) GO(N, 2); else if (depth == 3) GO(N, 3); else GO(N, 4); return (int)hipGetLastError(); }
    BOTH(4) BOTH(5) BOTH(8) BOTH(9) BOTH(10)
    return -1;
}
