// Is v_mfma_f32_16x16x32_f16 as fast with SUBNORMAL fp16 inputs as with normal ones (gfx950)?
//   hipcc --offload-arch=gfx950 -O2 mfma_denorm_rate.hip -o mfma_denorm_rate && ./mfma_denorm_rate
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k(uint32_t pattern, uint32_t iters, float* out)
{
    // eight independent accumulators so the matrix pipe is kept busy; A operand = the given bit pattern (+ lane noise in the low nibble)
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t w = pattern | ((lane & 15u) * 0x00010001u);
    const f16x8 a = __builtin_bit_cast(f16x8, u32x4{w, w ^ 0x00030003u, w ^ 0x00050005u, w ^ 0x00060006u});
    const f16x8 b = __builtin_bit_cast(f16x8, u32x4{0x3C003C00u, 0x3C003800u, 0x40003C00u, 0x3C004000u});
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (uint32_t it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main()
{
    float* d; hipMalloc(&d, 1024 * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const uint32_t iters = 20000;
    const struct { const char* name; uint32_t pat; } cases[] = {
        {"normal    (1024 + u: 0x6400 | u)", 0x64006400u}, {"subnormal (u * 2^-24: 0x0000 | u)", 0x00000000u},
        {"normal    (1024 + u) again", 0x64006400u}, {"subnormal again", 0x00000000u}};
    for (auto& c : cases) {
        hipLaunchKernelGGL(k, dim3(1024), dim3(256), 0, 0, c.pat, 100u, d);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(1024), dim3(256), 0, 0, c.pat, iters, d);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double flops = 1024.0 * 4 * iters * 8 * 2.0 * 16 * 16 * 32;
        printf("%-36s %.3f ms  %.1f TFLOP/s\n", c.name, ms, flops / ms / 1e9);
    }
    return 0;
}
