// profiles/probes/mxprobe2.hip -- second pass over v_mfma_scale_f32_16x16x128_f8f6f4: where the k index of an 8-bit operand
// lives (mxprobe.hip showed that the e2m1 operand is k = 32*(lane>>4) + nibble and that an e4m3 operand is NOT k = 32*(lane>>4) + byte),
// which lane's scale byte applies to which k block, then a full random check of both mixed arrangements under the map found.
//   hipcc --offload-arch=gfx950 -O2 -Wno-unused-value profiles/probes/mxprobe2.hip -o scratch/probe/mxprobe2
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef int v8i __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(2); } } while (0)

// FA / FB: 0 = e4m3, 4 = e2m1
template <int FA, int FB>
__global__ void k_mfma(const v8i* a, const v8i* b, const int* sa, const int* sb, f32x4* d, int n)
{
    const int l = threadIdx.x;
    for (int t = blockIdx.x; t < n; t += gridDim.x) {
        f32x4 acc = {0, 0, 0, 0};
        acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[t * 64 + l], b[t * 64 + l], acc, FA, FB, 0, sa[t * 64 + l], 0, sb[t * 64 + l]);
        d[t * 64 + l] = acc;
    }
}
static float e2m1(int n) { static const float t[8] = {0, 0.5f, 1, 1.5f, 2, 3, 4, 6}; return (n & 8) ? -t[n & 7] : t[n & 7]; }
static float e4m3(int b)
{
    const int s = b >> 7, e = (b >> 3) & 15, m = b & 7;
    float v = (e == 0) ? std::ldexp((float)m, -9) : std::ldexp(1.0f + m / 8.0f, e - 7);
    return s ? -v : v;
}
template <class T> static T* dev(const std::vector<T>& h)
{
    T* p; CK(hipMalloc(&p, h.size() * sizeof(T) + 16)); CK(hipMemcpy(p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice)); return p;
}
template <int FA, int FB>
static std::vector<f32x4> run(const std::vector<v8i>& a, const std::vector<v8i>& b, const std::vector<int>& sa, const std::vector<int>& sb)
{
    const int n = (int)a.size() / 64;
    v8i *da = dev(a), *db = dev(b); int *dsa = dev(sa), *dsb = dev(sb);
    f32x4* dd; CK(hipMalloc(&dd, a.size() * sizeof(f32x4)));
    k_mfma<FA, FB><<<n < 256 ? n : 256, 64>>>(da, db, dsa, dsb, dd, n);
    CK(hipDeviceSynchronize());
    std::vector<f32x4> d(a.size()); CK(hipMemcpy(d.data(), dd, a.size() * sizeof(f32x4), hipMemcpyDeviceToHost));
    hipFree(da); hipFree(db); hipFree(dsa); hipFree(dsb); hipFree(dd);
    return d;
}
static void put_fp4(v8i& r, int i, int code) { uint8_t* p = reinterpret_cast<uint8_t*>(&r); p[i >> 1] = (uint8_t)((p[i >> 1] & ~(0xF << (4 * (i & 1)))) | (code << (4 * (i & 1)))); }

int main()
{
    srand(11);
    int kmap[4][32];            // k of byte i of lane group kb of an e4m3 operand
    // (a) e4m3 as A: one-hot 1.0 at (row 0, lane group kb, byte i); B (e2m1) column j = bit j of k, column 7 = ones
    for (int side = 0; side < 2; ++side) {
        std::vector<v8i> a(128 * 64), b(128 * 64);
        std::vector<int> sa(128 * 64, 127), sb(128 * 64, 127);
        memset(a.data(), 0, a.size() * sizeof(v8i)); memset(b.data(), 0, b.size() * sizeof(v8i));
        for (int kb = 0; kb < 4; ++kb)
            for (int i = 0; i < 32; ++i) {
                const int t = kb * 32 + i;
                std::vector<v8i>& X8 = side == 0 ? a : b;     // the e4m3 operand
                std::vector<v8i>& X4 = side == 0 ? b : a;     // the e2m1 operand
                reinterpret_cast<uint8_t*>(&X8[t * 64 + 16 * kb + 0])[i] = 0x38;       // row / col 0 = lane 16*kb
                for (int l = 0; l < 64; ++l) {
                    const int j = l & 15, kbl = l >> 4;
                    for (int n = 0; n < 32; ++n) {
                        const int k = 32 * kbl + n;
                        const int on = j < 7 ? (k >> j) & 1 : (j == 7 ? 1 : 0);
                        put_fp4(X4[t * 64 + l], n, on ? 2 : 0);
                    }
                }
            }
        const std::vector<f32x4> d = side == 0 ? run<0, 4>(a, b, sa, sb) : run<4, 0>(a, b, sa, sb);
        printf("[e4m3 as %s] k index of byte i of lane group kb (rows: kb, columns: byte 0..31):\n", side == 0 ? "A" : "B");
        for (int kb = 0; kb < 4; ++kb) {
            printf("   kb %d:", kb);
            for (int i = 0; i < 32; ++i) {
                const int t = kb * 32 + i;
                int k = 0, ones;
                // side 0: D[row 0][col j] -> lane j (kb 0), reg 0.  side 1: D[row j][col 0] -> lane 16*(j/4), reg j%4
                for (int j = 0; j < 7; ++j) {
                    const float v = side == 0 ? d[t * 64 + j][0] : d[t * 64 + 16 * (j / 4)][j % 4];
                    k |= (v == 1.0f ? 1 : 0) << j;
                    if (v != 0.0f && v != 1.0f) k = -999;
                }
                ones = (int)(side == 0 ? d[t * 64 + 7][0] : d[t * 64 + 16][3]);
                printf(" %3d%s", k, ones == 1 ? "" : "!");
                if (side == 0) kmap[kb][i] = k;
            }
            printf("\n");
        }
    }
    // (b) which lane's scale byte scales which k block: e4m3 A all ones in row 0, B column j = indicator of block j; scale 128 in lane group kbS
    for (int side = 0; side < 4; ++side) {      // 0: scale_a of an e4m3 A; 1: scale_b of an e2m1 B; 2: scale_a of an e2m1 A; 3: scale_b of an e4m3 B
        const bool a8 = side == 0 || side == 1;     // A is e4m3 (else e2m1), B the other
        std::vector<v8i> a(4 * 64), b(4 * 64);
        std::vector<int> sa(4 * 64, 127), sb(4 * 64, 127);
        memset(a.data(), 0, a.size() * sizeof(v8i)); memset(b.data(), 0, b.size() * sizeof(v8i));
        for (int t = 0; t < 4; ++t) {
            for (int l = 0; l < 64; ++l) {
                // A: ones everywhere (all rows); B column j: ones where k / 32 == j (j < 4)
                uint8_t* pa = reinterpret_cast<uint8_t*>(&a[t * 64 + l]);
                if (a8) memset(pa, 0x38, 32); else memset(pa, 0x22, 16);
                const int j = l & 15, kbl = l >> 4;
                if (j < 4) {
                    uint8_t* pb = reinterpret_cast<uint8_t*>(&b[t * 64 + l]);
                    if (a8) { if (kbl == j) memset(pb, 0x22, 16); }                 // B e2m1: k = 32*kbl + n
                    else for (int i = 0; i < 32; ++i) if (kmap[kbl][i] / 32 == j) pb[i] = 0x38;      // B e4m3: by the map found above
                }
            }
            const bool on_a = side == 0 || side == 2;
            for (int c = 0; c < 16; ++c) (on_a ? sa : sb)[t * 64 + 16 * t + c] = 128;     // lane group t carries 2.0
        }
        const std::vector<f32x4> d = a8 ? run<0, 4>(a, b, sa, sb) : run<4, 0>(a, b, sa, sb);
        printf("[%s] scale 2.0 in lane group g, the others 1.0 -> D[0][block j] / 32 for j = 0..3:\n",
               side == 0 ? "scale_a, A = e4m3" : side == 1 ? "scale_b, B = e2m1" : side == 2 ? "scale_a, A = e2m1" : "scale_b, B = e4m3");
        for (int t = 0; t < 4; ++t)
            printf("   g %d: %g %g %g %g\n", t, d[t * 64 + 0][0] / 32, d[t * 64 + 1][0] / 32, d[t * 64 + 2][0] / 32, d[t * 64 + 3][0] / 32);
    }
    // (c) full random check of both mixed arrangements with the map found (scale of lane (c, kb) = its row / column, block kb)
    for (int form = 0; form < 2; ++form) {
        std::vector<int> M8(16 * 128), M4(16 * 128), S8(16 * 4), S4(16 * 4);
        for (auto& v : M8) { int e = 5 + rand() % 5, m = rand() % 8; v = ((rand() & 1) << 7) | (e << 3) | m; }
        for (auto& v : M4) v = rand() & 15;
        for (auto& v : S8) v = 124 + rand() % 7;
        for (auto& v : S4) v = 120 + rand() % 12;
        std::vector<v8i> x8(64), x4(64);
        std::vector<int> s8(64), s4(64);
        for (int l = 0; l < 64; ++l) {
            const int c = l & 15, kb = l >> 4;
            uint8_t p8[32], p4[32];
            for (int i = 0; i < 32; ++i) { p8[i] = (uint8_t)M8[c * 128 + kmap[kb][i]]; p4[i] = (uint8_t)rand(); }
            for (int i = 0; i < 16; ++i) p4[i] = (uint8_t)(M4[c * 128 + 32 * kb + 2 * i] | (M4[c * 128 + 32 * kb + 2 * i + 1] << 4));
            memcpy(&x8[l], p8, 32); memcpy(&x4[l], p4, 32);
            s8[l] = S8[c * 4 + kb] | (rand() << 8); s4[l] = S4[c * 4 + kb] | (rand() << 8);
        }
        const std::vector<f32x4> d = form == 0 ? run<0, 4>(x8, x4, s8, s4) : run<4, 0>(x4, x8, s4, s8);
        int bad = 0; double worst = 0;
        for (int l = 0; l < 64; ++l)
            for (int r = 0; r < 4; ++r) {
                const int row = 4 * (l >> 4) + r, col = l & 15;
                const int i8 = form == 0 ? row : col, i4 = form == 0 ? col : row;      // index into the 8-bit / 4-bit matrix
                double s = 0;
                for (int k = 0; k < 128; ++k)
                    s += (double)e4m3(M8[i8 * 128 + k]) * std::ldexp(1.0, S8[i8 * 4 + k / 32] - 127) * (double)e2m1(M4[i4 * 128 + k]) * std::ldexp(1.0, S4[i4 * 4 + k / 32] - 127);
                const double err = std::fabs(d[l][r] - s);
                worst = std::fmax(worst, err / (std::fabs(s) + 1e-3));
                if (err > 2e-4 * (std::fabs(s) + 1.0)) ++bad;
            }
        printf("[random data, %s, map above, scale of lane (c, kb) = (row|col c, block kb)] %s (mismatches %d / 256, worst rel %.3g)\n",
               form == 0 ? "A = e4m3, B = e2m1" : "A = e2m1, B = e4m3", bad ? "NO" : "YES", bad, worst);
    }
    // (d) accumulation: exact integer data, is the result the exact sum (fp32) -- 128 products of magnitude up to 448*6
    {
        std::vector<v8i> a(64), b(64);
        std::vector<int> sa(64, 127), sb(64, 127);
        for (int l = 0; l < 64; ++l) { memset(&a[l], 0x7E, 32); memset(&b[l], 0x77, 32); }          // 448 x 6 everywhere
        const std::vector<f32x4> d = run<0, 4>(a, b, sa, sb);
        printf("[accumulate] 128 x (448 x 6) = %.1f, got %.1f\n", 128.0 * 448 * 6, d[0][0]);
    }
    return 0;
}
