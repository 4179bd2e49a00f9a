// Rate of the fp32 matrix instructions on gfx950: dependent chains (one accumulator) and independent ones.
//   hipcc --offload-arch=gfx950 -O2 mfma_f32_rate.hip -o mfma_f32_rate && ./mfma_f32_rate   (MI355X: 143-156 TFLOP/s, 64-70 clocks per 32x32x2, 32-34 per 16x16x4)
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ __launch_bounds__(256) void k32(uint32_t iters, float* out)
{
    const float a = 1.0f + (threadIdx.x & 7), b = 0.5f;
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int v = 0; v < 16; ++v) acc[i][v] = 0.f;
    for (uint32_t it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8 / NACC; ++r)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) for (int v = 0; v < 16; ++v) s += acc[i][v];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC>
__global__ __launch_bounds__(256) void k16(uint32_t iters, float* out)
{
    const float a = 1.0f + (threadIdx.x & 7), b = 0.5f;
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (uint32_t it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8 / NACC; ++r)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <typename K>
void run(const char* name, K kern, int wgs, double flops_per_mfma)
{
    float* d; (void)hipMalloc(&d, 4096 * 256 * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const uint32_t iters = 4000;
    hipLaunchKernelGGL(kern, dim3(wgs), dim3(256), 0, 0, 100u, d);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(wgs), dim3(256), 0, 0, iters, d);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double n_mfma = double(wgs) * 4 * iters * 8;
    printf("%-44s %d wgs  %.3f ms  %.1f TFLOP/s  %.1f clk/mfma/SIMD at 2.4 GHz\n", name, wgs, ms, n_mfma * flops_per_mfma / ms / 1e9,
           ms * 1e-3 * 2.4e9 / (n_mfma / 1024.0));
    (void)hipFree(d);
}
int main()
{
    for (int wgs : {256, 512, 1024}) {
        run("32x32x2 f32, 1 accumulator (dependent)", k32<1>, wgs, 4096.0);
        run("32x32x2 f32, 2 accumulators", k32<2>, wgs, 4096.0);
        run("32x32x2 f32, 4 accumulators", k32<4>, wgs, 4096.0);
        run("16x16x4 f32, 1 accumulator (dependent)", k16<1>, wgs, 2048.0);
        run("16x16x4 f32, 4 accumulators", k16<4>, wgs, 2048.0);
    }
    return 0;
}
