// does v_mfma_f32_16x16x32_f16 keep fp16 subnormal inputs?  A[m][k] = bits 0x000u (u = 1..15 -> u * 2^-24), B = 1.0
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(float* out)
{
    const uint32_t lane = threadIdx.x;
    uint16_t abits[8]; f16x8 A, B;
    for (int e = 0; e < 8; ++e) { abits[e] = (uint16_t)((lane + e) & 15); A[e] = __builtin_bit_cast(_Float16, abits[e]); B[e] = (_Float16)1.0f; }
    f32x4 acc = {0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(A, B, acc, 0, 0, 0);
    for (int i = 0; i < 4; ++i) out[lane * 4 + i] = acc[i];
}
int main()
{
    float* d; hipMalloc(&d, 64 * 4 * sizeof(float));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    float h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    // expected: row m = 4*kb + i of column c: sum over k (32 values: lanes (m, kb'=0..3), e=0..7) of ((m + 16 kb' + e) & 15) * 2^-24
    int bad = 0;
    for (int lane = 0; lane < 64; ++lane) for (int i = 0; i < 4; ++i) {
        int m = 4 * (lane >> 4) + i; double want = 0;
        for (int kb = 0; kb < 4; ++kb) for (int e = 0; e < 8; ++e) want += (double)(((m + 16 * kb) + e) & 15) / 16777216.0;
        if (h[lane * 4 + i] != (float)want) { if (bad < 4) printf("lane %d i %d got %g want %g\n", lane, i, h[lane * 4 + i], want); ++bad; }
    }
    printf("mfma f16 subnormal inputs: %s (%d mismatches), sample %g\n", bad ? "FLUSHED or wrong" : "exact", bad, h[0]);
    return 0;
}
