// profiles/probes/mxprobe3.hip -- which bits of its scale operand v_cvt_scalef32_pk_f16_fp4 / _pk_f32_fp4 read (the attention kernel
// hands it a rotated word whose bits 30..23 are the E8M0 code and whose other bits are neighbouring codes).
//   hipcc --offload-arch=gfx950 -O2 profiles/probes/mxprobe3.hip -o scratch/probe/mxprobe3
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
__global__ void k(const uint32_t* scales, int n, float* out16, float* out32)
{
    const int i = threadIdx.x;
    if (i >= n) return;
    const float s = __uint_as_float(scales[i]);
    const uint32_t w = 0x0000F7A2u;       // nibbles 2 (1.0), A (-1.0), 7 (6.0), F (-6.0)
    const f16x2 a = __builtin_amdgcn_cvt_scalef32_pk_f16_fp4(w, s, 0), b = __builtin_amdgcn_cvt_scalef32_pk_f16_fp4(w, s, 1);
    out16[4 * i] = (float)a.x; out16[4 * i + 1] = (float)a.y; out16[4 * i + 2] = (float)b.x; out16[4 * i + 3] = (float)b.y;
    const f32x2 c = __builtin_amdgcn_cvt_scalef32_pk_f32_fp4(w, s, 0), d = __builtin_amdgcn_cvt_scalef32_pk_f32_fp4(w, s, 1);
    out32[4 * i] = c.x; out32[4 * i + 1] = c.y; out32[4 * i + 2] = d.x; out32[4 * i + 3] = d.y;
}
int main()
{
    const uint32_t sc[] = {0x40000000u /* 2 */, 0xC0000000u /* -2 */, 0x40123456u, 0xC07FFFFFu, 0x3F800000u, 0x00000000u, 0x00400000u, 0x80000000u,
                           0x00800000u /* 2^-126 */, 0x7F000000u /* 2^127 */, 0x7F800000u /* inf */, 0x7FC00000u /* nan */, 0x47000000u /* 2^15 */,
                           0x33800000u /* 2^-24 */, 0x32000000u /* 2^-27 */, 0x48000000u /* 2^17 */};
    const int n = sizeof(sc) / 4;
    uint32_t* d; float *o16, *o32;
    hipMalloc(&d, sizeof(sc)); hipMalloc(&o16, n * 16); hipMalloc(&o32, n * 16);
    hipMemcpy(d, sc, sizeof(sc), hipMemcpyHostToDevice);
    k<<<1, 64>>>(d, n, o16, o32);
    float h16[64], h32[64];
    hipMemcpy(h16, o16, n * 16, hipMemcpyDeviceToHost); hipMemcpy(h32, o32, n * 16, hipMemcpyDeviceToHost);
    for (int i = 0; i < n; ++i)
        printf("scale bits %08x: nibbles (1, -1, 6, -6) -> f16 %g %g %g %g | f32 %g %g %g %g\n", sc[i], h16[4 * i], h16[4 * i + 1], h16[4 * i + 2], h16[4 * i + 3],
               h32[4 * i], h32[4 * i + 1], h32[4 * i + 2], h32[4 * i + 3]);
    return 0;
}
