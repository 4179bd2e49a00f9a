// Does v_mfma_f32_16x16x32_f16 on gfx950 honour fp16 SUBNORMAL inputs?  (A nibble u in the low mantissa bits of an fp16
// with exponent 0 is the exact value u * 2^-24: if the matrix core keeps subnormals, INT4 records can be fed to it
// without any bias term.)   hipcc --offload-arch=gfx950 -O2 mfma_denorm.hip -o mfma_denorm && ./mfma_denorm
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cmath>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__global__ void k(const uint16_t* a_bits, const uint16_t* b_bits, float* out)
{
    // A[16][32], B[32][16] row-major fp16 bit patterns; lane (c, kb): A row c, k = 8kb..8kb+7; B col c, same k
    const uint32_t lane = threadIdx.x, c = lane & 15u, kb = lane >> 4;
    f16x8 a, b;
    for (int e = 0; e < 8; ++e) {
        a[e] = __builtin_bit_cast(_Float16, a_bits[c * 32 + kb * 8 + e]);
        b[e] = __builtin_bit_cast(_Float16, b_bits[(kb * 8 + e) * 16 + c]);
    }
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0);
    for (int i = 0; i < 4; ++i) out[(4 * kb + i) * 16 + c] = acc[i];
}

static float h2f(uint16_t h)
{
    const uint32_t s = (h >> 15) & 1u, e = (h >> 10) & 31u, m = h & 1023u;
    float v = e == 0 ? ldexpf((float)m, -24) : ldexpf((float)(m + 1024u), (int)e - 25);
    return s ? -v : v;
}
static uint16_t f2h_exact(float f)   // only for values that are exactly representable normals here
{
    _Float16 h = (_Float16)f; uint16_t u; __builtin_memcpy(&u, &h, 2); return u;
}

int main()
{
    uint16_t ha[16 * 32], hb[32 * 16];
    uint32_t seed = 12345u;
    auto rnd = [&]() { seed = seed * 1664525u + 1013904223u; return seed >> 8; };
    for (int i = 0; i < 16 * 32; ++i) {
        const uint32_t u = rnd() & 15u;
        ha[i] = (i & 1) ? (uint16_t)(u << 4) : (uint16_t)u;          // low plane u*2^-24, high plane u*2^-20: both subnormal
    }
    for (int i = 0; i < 32 * 16; ++i) hb[i] = f2h_exact((float)((int)(rnd() % 2001) - 1000) / 256.0f);
    uint16_t *da, *db; float* dout;
    hipMalloc(&da, sizeof(ha)); hipMalloc(&db, sizeof(hb)); hipMalloc(&dout, 256 * 4);
    hipMemcpy(da, ha, sizeof(ha), hipMemcpyHostToDevice); hipMemcpy(db, hb, sizeof(hb), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, da, db, dout);
    float out[256];
    hipMemcpy(out, dout, sizeof(out), hipMemcpyDeviceToHost);
    double worst = 0.0; int zero_rows = 0;
    for (int m = 0; m < 16; ++m)
        for (int n = 0; n < 16; ++n) {
            double want = 0.0;
            for (int kk = 0; kk < 32; ++kk) want += (double)h2f(ha[m * 32 + kk]) * (double)h2f(hb[kk * 16 + n]);
            const double err = fabs(out[m * 16 + n] - want) / (fabs(want) + 1e-30);
            if (err > worst) worst = err;
            if (out[m * 16 + n] == 0.0f && want != 0.0) ++zero_rows;
        }
    printf("mfma_f32_16x16x32_f16 with subnormal A: worst relative error %.3g, outputs flushed to zero %d / 256 -> %s\n",
           worst, zero_rows, (worst < 1e-6) ? "SUBNORMALS HONOURED" : "SUBNORMALS FLUSHED OR WRONG");
    return 0;
}
