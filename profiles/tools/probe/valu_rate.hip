// Issue rate of the vector instructions the INT4 dequant is made of (gfx950), per SIMD, with 1 / 2 / 4 waves per SIMD:
//   hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip && ./valu_rate
// One workgroup per CU of 256 * W threads (W waves on every SIMD); every wave runs REPS x 64 independent instructions of one
// kind over 8 registers; cycles from s_memtime around the loop.  Printed: SIMD cycles per wave-instruction = cycles / (64 * REPS * W).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define BODY8(INS) INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7)
#define BODY64(INS) BODY8(INS) BODY8(INS) BODY8(INS) BODY8(INS) BODY8(INS) BODY8(INS) BODY8(INS) BODY8(INS)
#define KERNEL(NAME, ASM)                                                                                          \
__global__ void NAME(uint32_t reps, uint32_t seed, unsigned long long* out)                                        \
{                                                                                                                  \
    uint32_t r[8]; uint32_t b = seed | 0x3C003C00u, c = 0x38003800u;                                               \
    for (int i = 0; i < 8; ++i) r[i] = seed + threadIdx.x + i;                                                     \
    unsigned long long t0, t1;                                                                                     \
    __syncthreads();                                                                                               \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");                                   \
    for (uint32_t k = 0; k < reps; ++k) {                                                                          \
        asm volatile(ASM : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]) : "v"(b), "v"(c)); \
    }                                                                                                              \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");                                   \
    uint32_t x = 0; for (int i = 0; i < 8; ++i) x ^= r[i];                                                         \
    if (x == 0x12345u) out[1] = x;                                                                                 \
    if ((threadIdx.x & 63) == 0) out[2 + blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;                          \
}
#define I_PKMUL(i)  "v_pk_mul_f16 %" #i ", %" #i ", %8\n\t"
#define I_PKADD(i)  "v_pk_add_f16 %" #i ", %" #i ", %8\n\t"
#define I_PKFMA(i)  "v_pk_fma_f16 %" #i ", %" #i ", %8, %9\n\t"
#define I_BITOP(i)  "v_bitop3_b32 %" #i ", %" #i ", %8, %9 bitop3:0x6c\n\t"
#define I_PERM(i)   "v_perm_b32 %" #i ", %" #i ", %8, %9\n\t"
#define I_FMA32(i)  "v_fma_f32 %" #i ", %" #i ", %8, %9\n\t"
#define I_LSHR(i)   "v_lshrrev_b32 %" #i ", 8, %" #i "\n\t"
#define I_MOV(i)    "v_mov_b32 %" #i ", %8\n\t"
#define I_EXP(i)    "v_exp_f32 %" #i ", %" #i "\n\t"
#define I_CVT(i)    "v_cvt_f32_f16 %" #i ", %" #i "\n\t"
#define I_PKMULF32(i) ""
KERNEL(k_pkmul, BODY64(I_PKMUL))
KERNEL(k_pkadd, BODY64(I_PKADD))
KERNEL(k_pkfma, BODY64(I_PKFMA))
KERNEL(k_bitop, BODY64(I_BITOP))
KERNEL(k_perm, BODY64(I_PERM))
KERNEL(k_fma32, BODY64(I_FMA32))
KERNEL(k_lshr, BODY64(I_LSHR))
KERNEL(k_mov, BODY64(I_MOV))
KERNEL(k_exp, BODY64(I_EXP))
KERNEL(k_cvt, BODY64(I_CVT))
typedef void (*kern_t)(uint32_t, uint32_t, unsigned long long*);
int main()
{
    unsigned long long* d; hipMalloc(&d, (2 + 256 * 16) * 8);
    struct { const char* n; kern_t k; } ks[] = {{"v_pk_mul_f16", k_pkmul}, {"v_pk_add_f16", k_pkadd}, {"v_pk_fma_f16", k_pkfma}, {"v_bitop3_b32", k_bitop},
        {"v_perm_b32", k_perm}, {"v_fma_f32", k_fma32}, {"v_lshrrev_b32", k_lshr}, {"v_mov_b32", k_mov}, {"v_exp_f32", k_exp}, {"v_cvt_f32_f16", k_cvt}};
    const uint32_t reps = 2000;
    for (auto& e : ks) {
        printf("%-16s", e.n);
        for (int W : {1, 2, 4}) {
            for (int it = 0; it < 3; ++it) hipLaunchKernelGGL(e.k, dim3(256), dim3(256 * W), 0, 0, reps, 7u, d);
            hipDeviceSynchronize();
            std::vector<unsigned long long> h(2 + 256 * 16);
            hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
            double s = 0; int n = 0;
            for (int b = 0; b < 256; ++b) for (int w = 0; w < 4 * W; ++w) { s += (double)h[2 + b * 16 + w]; ++n; }
            // s_memtime ticks at 100 MHz on gfx9? it counts shader clocks here (MI355X_MICROARCH.md): report ticks per instruction per wave-on-SIMD
            printf("  W=%d: %.2f", W, s / n / (64.0 * reps * W));
        }
        printf("   (s_memtime ticks per wave-instruction per SIMD)\n");
    }
    return 0;
}
