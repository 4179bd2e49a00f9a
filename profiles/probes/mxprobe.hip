// profiles/probes/mxprobe.hip -- what gfx950's block-scaled matrix instruction, the 4-bit transpose read and the fp4
// conversions really do (no ISA document in the image: measured with exact data instead).
//   hipcc --offload-arch=gfx950 -O2 profiles/probes/mxprobe.hip -o scratch/probe/mxprobe && scratch/probe/mxprobe
// Prints, for each question, the hypothesis and whether the hardware agrees (and the raw mapping where it does not).
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef int v8i __attribute__((ext_vector_type(8)));
typedef int v2i __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(2); } } while (0)

// ---- 1. v_mfma_scale_f32_16x16x128_f8f6f4: A = e4m3 (cbsz 0), B = e2m1 (blgp 4) --------------------------------------
template <int OPA, int OPB>
__global__ void k_mfma(const v8i* a, const v8i* b, const int* sa, const int* sb, f32x4* d)
{
    const int l = threadIdx.x;
    f32x4 acc = {0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[l], b[l], acc, 0, 4, OPA, sa[l], OPB, sb[l]);
    d[l] = acc;
}
// both operands e2m1
__global__ void k_mfma44(const v8i* a, const v8i* b, const int* sa, const int* sb, f32x4* d)
{
    const int l = threadIdx.x;
    f32x4 acc = {0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[l], b[l], acc, 4, 4, 0, sa[l], 0, sb[l]);
    d[l] = acc;
}
// A = e2m1, B = e4m3
__global__ void k_mfma40(const v8i* a, const v8i* b, const int* sa, const int* sb, f32x4* d)
{
    const int l = threadIdx.x;
    f32x4 acc = {0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[l], b[l], acc, 4, 0, 0, sa[l], 0, sb[l]);
    d[l] = acc;
}

static float e2m1(int n)
{
    static const float t[8] = {0.0f, 0.5f, 1.0f, 1.5f, 2.0f, 3.0f, 4.0f, 6.0f};
    return (n & 8) ? -t[n & 7] : t[n & 7];
}
static float e4m3(int b)
{
    const int s = b >> 7, e = (b >> 3) & 15, m = b & 7;
    float v = (e == 0) ? std::ldexp((float)m, -9) : std::ldexp(1.0f + m / 8.0f, e - 7);
    return s ? -v : v;
}

// ---- 2. ds_read_b64_tr_b4 --------------------------------------------------------------------------------------------
__global__ void k_tr4(const uint8_t* src, int nbytes, const int* lane_addr, v2i* out)
{
    __shared__ __attribute__((aligned(16))) uint8_t lds[8192];
    for (int i = threadIdx.x; i < nbytes; i += 64) lds[i] = src[i];
    __syncthreads();
    out[threadIdx.x] = __builtin_amdgcn_ds_read_tr4_b64_v2i32((v2i __attribute__((address_space(3)))*)(lds + lane_addr[threadIdx.x]));
}
__global__ void k_tr8(const uint8_t* src, int nbytes, const int* lane_addr, v2i* out)
{
    __shared__ __attribute__((aligned(16))) uint8_t lds[8192];
    for (int i = threadIdx.x; i < nbytes; i += 64) lds[i] = src[i];
    __syncthreads();
    out[threadIdx.x] = __builtin_amdgcn_ds_read_tr8_b64_v2i32((v2i __attribute__((address_space(3)))*)(lds + lane_addr[threadIdx.x]));
}

// ---- 3. fp4 conversions ---------------------------------------------------------------------------------------------
__global__ void k_cvt(const float* x, int n, float scale, uint32_t* q, float* back, _Float16* back16)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t w = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(0u, x[2 * i], x[2 * i + 1], scale, 0);
    q[i] = w;
    const f32x2 f = __builtin_amdgcn_cvt_scalef32_pk_f32_fp4(w, scale, 0);
    back[2 * i] = f.x; back[2 * i + 1] = f.y;
    const f16x2 h = __builtin_amdgcn_cvt_scalef32_pk_f16_fp4(w, scale, 0);
    back16[2 * i] = h.x; back16[2 * i + 1] = h.y;
}
__global__ void k_cvt_sel(uint32_t w, float scale, float* out)
{
    f32x2 f;
    f = __builtin_amdgcn_cvt_scalef32_pk_f32_fp4(w, scale, 0); out[0] = f.x; out[1] = f.y;
    f = __builtin_amdgcn_cvt_scalef32_pk_f32_fp4(w, scale, 1); out[2] = f.x; out[3] = f.y;
    f = __builtin_amdgcn_cvt_scalef32_pk_f32_fp4(w, scale, 2); out[4] = f.x; out[5] = f.y;
    f = __builtin_amdgcn_cvt_scalef32_pk_f32_fp4(w, scale, 3); out[6] = f.x; out[7] = f.y;
    uint32_t o = 0xAAAAAAAAu;
    o = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(o, 1.0f, -2.0f, 1.0f, 2);
    out[8] = __uint_as_float(o);
}
// e4m3 from f32 with a scale (for the MXFP8 query)
__global__ void k_cvt8(const float* x, int n, float scale, uint32_t* q)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    typedef short v2s __attribute__((ext_vector_type(2)));
    v2s old = {0, 0};
    v2s r = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(old, x[2 * i], x[2 * i + 1], scale, false);
    q[i] = (uint16_t)r.x;
}

template <class T> static T* dev(const std::vector<T>& h)
{
    T* p; CK(hipMalloc(&p, h.size() * sizeof(T) + 16)); CK(hipMemcpy(p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice)); return p;
}

static int host_e2m1_rne(float x)      // round to nearest even onto {0,.5,1,1.5,2,3,4,6}, saturating
{
    const int s = std::signbit(x) ? 8 : 0;
    const float a = std::fabs(x);
    if (std::isnan(a)) return s | 7;
    static const float t[8] = {0.0f, 0.5f, 1.0f, 1.5f, 2.0f, 3.0f, 4.0f, 6.0f};
    if (a >= 6.0f) return s | 7;
    int best = 0;
    for (int i = 0; i < 7; ++i) {
        if (a >= t[i] && a <= t[i + 1]) {
            const float mid = 0.5f * (t[i] + t[i + 1]);
            if (a < mid) best = i; else if (a > mid) best = i + 1; else best = (i & 1) ? i + 1 : i;      // tie -> even mantissa bit
            break;
        }
    }
    return s | best;
}

int main()
{
    srand(7);
    // ---------------- 1. MFMA layout -------------------------------------------------------------------------------
    {
        // logical matrices: A[16][128] e4m3 codes, B[128][16] e2m1 codes, scales sa[16][4], sb[4][16] (E8M0)
        std::vector<int> Ac(16 * 128), Bc(128 * 16), SA(16 * 4), SB(4 * 16);
        for (auto& v : Ac) { int e = 5 + rand() % 5, m = rand() % 8; v = ((rand() & 1) << 7) | (e << 3) | m; }     // normal values around 1
        for (auto& v : Bc) v = rand() & 15;
        for (auto& v : SA) v = 125 + rand() % 5;
        for (auto& v : SB) v = 124 + rand() % 7;
        std::vector<double> ref(16 * 16, 0.0);
        for (int m = 0; m < 16; ++m)
            for (int n = 0; n < 16; ++n) {
                double s = 0;
                for (int k = 0; k < 128; ++k)
                    s += (double)e4m3(Ac[m * 128 + k]) * std::ldexp(1.0, SA[m * 4 + k / 32] - 127) * (double)e2m1(Bc[k * 16 + n]) * std::ldexp(1.0, SB[(k / 32) * 16 + n] - 127);
                ref[m * 16 + n] = s;
            }
        // hypothesis: lane l: A row l&15, k = 32*(l>>4) + byte; B col l&15, k = 32*(l>>4) + nibble (low nibble first) in VGPR 0..3;
        // scale of lane l = its (row|col, k block l>>4), byte OPSEL of the scale register; D[row 4*(l>>4)+r][col l&15]
        std::vector<v8i> a(64), b(64);
        std::vector<int> sa(64), sb(64);
        for (int opsel = 0; opsel < 4; ++opsel) {
            for (int l = 0; l < 64; ++l) {
                uint8_t ab[32], bb[32];
                memset(bb, 0, 32);
                for (int i = 0; i < 32; ++i) ab[i] = (uint8_t)Ac[(l & 15) * 128 + 32 * (l >> 4) + i];
                for (int i = 0; i < 16; ++i)
                    bb[i] = (uint8_t)(Bc[(32 * (l >> 4) + 2 * i) * 16 + (l & 15)] | (Bc[(32 * (l >> 4) + 2 * i + 1) * 16 + (l & 15)] << 4));
                for (int i = 16; i < 32; ++i) bb[i] = (uint8_t)rand();      // upper four registers: must be ignored for e2m1
                memcpy(&a[l], ab, 32); memcpy(&b[l], bb, 32);
                uint32_t ra = (uint32_t)rand() | ((uint32_t)rand() << 16), rb = (uint32_t)rand() | ((uint32_t)rand() << 16);
                ra = (ra & ~(0xFFu << (8 * opsel))) | ((uint32_t)SA[(l & 15) * 4 + (l >> 4)] << (8 * opsel));
                rb = (rb & ~(0xFFu << (8 * opsel))) | ((uint32_t)SB[(l >> 4) * 16 + (l & 15)] << (8 * opsel));
                sa[l] = (int)ra; sb[l] = (int)rb;
            }
            v8i *da = dev(a), *db = dev(b); int *dsa = dev(sa), *dsb = dev(sb);
            f32x4* dd; CK(hipMalloc(&dd, 64 * sizeof(f32x4)));
            switch (opsel) {
            case 0: k_mfma<0, 0><<<1, 64>>>(da, db, dsa, dsb, dd); break;
            case 1: k_mfma<1, 1><<<1, 64>>>(da, db, dsa, dsb, dd); break;
            case 2: k_mfma<2, 2><<<1, 64>>>(da, db, dsa, dsb, dd); break;
            default: k_mfma<3, 3><<<1, 64>>>(da, db, dsa, dsb, dd); break;
            }
            CK(hipDeviceSynchronize());
            std::vector<f32x4> d(64); CK(hipMemcpy(d.data(), dd, 64 * sizeof(f32x4), hipMemcpyDeviceToHost));
            int bad = 0; double worst = 0;
            for (int l = 0; l < 64; ++l)
                for (int r = 0; r < 4; ++r) {
                    const double want = ref[(4 * (l >> 4) + r) * 16 + (l & 15)];
                    const double err = std::fabs(d[l][r] - want);
                    worst = std::fmax(worst, err / (std::fabs(want) + 1e-3));
                    if (err > 1e-4 * (std::fabs(want) + 1.0)) ++bad;
                }
            printf("[mfma_scale 16x16x128 A=e4m3 B=e2m1 opsel=%d] hypothesis (row/col = lane&15, k = 32*(lane>>4)+i, low nibble first, scale byte = opsel, "
                   "2^(s-127), D row 4*(lane>>4)+r col lane&15): %s (mismatches %d / 256, worst rel %.3g)\n", opsel, bad ? "NO" : "YES", bad, worst);
            if (bad && opsel == 0) {
                for (int l = 0; l < 4; ++l) printf("   lane %d: got %g %g %g %g want %g %g %g %g\n", l, d[l][0], d[l][1], d[l][2], d[l][3],
                                                   ref[(4 * (l >> 4) + 0) * 16 + (l & 15)], ref[(4 * (l >> 4) + 1) * 16 + (l & 15)],
                                                   ref[(4 * (l >> 4) + 2) * 16 + (l & 15)], ref[(4 * (l >> 4) + 3) * 16 + (l & 15)]);
            }
            hipFree(da); hipFree(db); hipFree(dsa); hipFree(dsb); hipFree(dd);
        }
        // both e2m1, and A e2m1 x B e4m3
        for (int form = 0; form < 2; ++form) {
            std::vector<int> A4(16 * 128), B8(128 * 16);
            for (auto& v : A4) v = rand() & 15;
            for (auto& v : B8) { int e = 5 + rand() % 5, m = rand() % 8; v = ((rand() & 1) << 7) | (e << 3) | m; }
            std::vector<double> r2(256, 0.0);
            for (int m = 0; m < 16; ++m)
                for (int n = 0; n < 16; ++n) {
                    double s = 0;
                    for (int k = 0; k < 128; ++k) {
                        const double bv = form == 0 ? (double)e2m1(Bc[k * 16 + n]) : (double)e4m3(B8[k * 16 + n]);
                        s += (double)e2m1(A4[m * 128 + k]) * std::ldexp(1.0, SA[m * 4 + k / 32] - 127) * bv * std::ldexp(1.0, SB[(k / 32) * 16 + n] - 127);
                    }
                    r2[m * 16 + n] = s;
                }
            for (int l = 0; l < 64; ++l) {
                uint8_t ab[32], bb[32];
                for (int i = 0; i < 32; ++i) { ab[i] = (uint8_t)rand(); bb[i] = (uint8_t)rand(); }
                for (int i = 0; i < 16; ++i)
                    ab[i] = (uint8_t)(A4[(l & 15) * 128 + 32 * (l >> 4) + 2 * i] | (A4[(l & 15) * 128 + 32 * (l >> 4) + 2 * i + 1] << 4));
                if (form == 0)
                    for (int i = 0; i < 16; ++i)
                        bb[i] = (uint8_t)(Bc[(32 * (l >> 4) + 2 * i) * 16 + (l & 15)] | (Bc[(32 * (l >> 4) + 2 * i + 1) * 16 + (l & 15)] << 4));
                else
                    for (int i = 0; i < 32; ++i) bb[i] = (uint8_t)B8[(32 * (l >> 4) + i) * 16 + (l & 15)];
                memcpy(&a[l], ab, 32); memcpy(&b[l], bb, 32);
                sa[l] = SA[(l & 15) * 4 + (l >> 4)] | 0x55aa3300; sb[l] = SB[(l >> 4) * 16 + (l & 15)] | 0x11227700;
            }
            v8i *da = dev(a), *db = dev(b); int *dsa = dev(sa), *dsb = dev(sb);
            f32x4* dd; CK(hipMalloc(&dd, 64 * sizeof(f32x4)));
            if (form == 0) k_mfma44<<<1, 64>>>(da, db, dsa, dsb, dd); else k_mfma40<<<1, 64>>>(da, db, dsa, dsb, dd);
            CK(hipDeviceSynchronize());
            std::vector<f32x4> d(64); CK(hipMemcpy(d.data(), dd, 64 * sizeof(f32x4), hipMemcpyDeviceToHost));
            int bad = 0;
            for (int l = 0; l < 64; ++l)
                for (int r = 0; r < 4; ++r)
                    if (std::fabs(d[l][r] - r2[(4 * (l >> 4) + r) * 16 + (l & 15)]) > 1e-4 * (std::fabs(r2[(4 * (l >> 4) + r) * 16 + (l & 15)]) + 1.0)) ++bad;
            printf("[mfma_scale 16x16x128 %s] same hypothesis: %s (mismatches %d / 256)\n", form == 0 ? "A=e2m1 B=e2m1" : "A=e2m1 B=e4m3", bad ? "NO" : "YES", bad);
            hipFree(da); hipFree(db); hipFree(dsa); hipFree(dsb); hipFree(dd);
        }
        // special scale bytes: 0, 254, 255 with simple data
        for (int sv : {0, 1, 254, 255, 127}) {
            for (int l = 0; l < 64; ++l) {
                uint8_t ab[32], bb[32];
                memset(ab, 0, 32); memset(bb, 0, 32);
                ab[0] = 0x38;                    // e4m3 1.0 at k = 32*(l>>4)
                bb[0] = 0x02;                    // e2m1 1.0 at the same k
                memcpy(&a[l], ab, 32); memcpy(&b[l], bb, 32);
                sa[l] = sv; sb[l] = 127;
            }
            v8i *da = dev(a), *db = dev(b); int *dsa = dev(sa), *dsb = dev(sb);
            f32x4* dd; CK(hipMalloc(&dd, 64 * sizeof(f32x4)));
            k_mfma<0, 0><<<1, 64>>>(da, db, dsa, dsb, dd);
            CK(hipDeviceSynchronize());
            std::vector<f32x4> d(64); CK(hipMemcpy(d.data(), dd, 64 * sizeof(f32x4), hipMemcpyDeviceToHost));
            printf("[mfma_scale] scale byte %3d on A, 4 products of 1.0 x 1.0: D[0][0] = %g (2^(s-127)*4 would be %g)\n", sv, d[0][0], std::ldexp(4.0, sv - 127));
            hipFree(da); hipFree(db); hipFree(dsa); hipFree(dsb); hipFree(dd);
        }
    }
    // ---------------- 2. ds_read_b64_tr_b4 / tr_b8: where does every received element come from --------------------------
    for (int form = 0; form < 3; ++form) {
        // form 0: lane address = 8 * lane (a dense 512-byte image); form 1: rows of 64 bytes, lane -> row (lane & 15), 8-byte piece lane >> 4;
        // form 2 (tr8): dense
        const int nbytes = 4096;
        std::vector<int> addr(64);
        for (int l = 0; l < 64; ++l) addr[l] = form == 1 ? (l & 15) * 64 + (l >> 4) * 8 : 8 * l;
        int* daddr = dev(addr);
        const int elems_per_lane = form == 2 ? 8 : 16, bits = form == 2 ? 8 : 4;
        std::vector<int> srcidx(64 * elems_per_lane, 0);
        const int n_elem = nbytes * 8 / bits;
        int passes = 0; while ((1 << passes) < n_elem) ++passes;
        const int per_pass = form == 2 ? 8 : 4;           // index bits carried per pass
        for (int p = 0; p * per_pass < passes; ++p) {
            std::vector<uint8_t> img(nbytes, 0);
            for (int e = 0; e < n_elem; ++e) {
                const int v = (e >> (p * per_pass)) & ((1 << per_pass) - 1);
                if (bits == 4) img[e >> 1] |= (uint8_t)(v << (4 * (e & 1))); else img[e] = (uint8_t)v;
            }
            uint8_t* dimg = dev(img);
            v2i* dout; CK(hipMalloc(&dout, 64 * sizeof(v2i)));
            if (form == 2) k_tr8<<<1, 64>>>(dimg, nbytes, daddr, dout); else k_tr4<<<1, 64>>>(dimg, nbytes, daddr, dout);
            CK(hipDeviceSynchronize());
            std::vector<v2i> o(64); CK(hipMemcpy(o.data(), dout, 64 * sizeof(v2i), hipMemcpyDeviceToHost));
            for (int l = 0; l < 64; ++l) {
                const uint64_t w = (uint32_t)o[l].x | ((uint64_t)(uint32_t)o[l].y << 32);
                for (int i = 0; i < elems_per_lane; ++i)
                    srcidx[l * elems_per_lane + i] |= (int)((w >> (bits * i)) & ((1u << bits) - 1)) << (p * per_pass);
            }
            hipFree(dimg); hipFree(dout);
        }
        printf("[ds_read_b64_tr_b%d, %s] element i of lane l comes from LDS element index (byte*%d + slot):\n", bits,
               form == 1 ? "lane address = (lane&15)*64 + (lane>>4)*8" : "lane address = 8*lane", 8 / bits);
        for (int l = 0; l < 64; ++l) {
            if (l >= 20 && l < 60 && (l & 15) > 1) continue;
            printf("   lane %2d:", l);
            for (int i = 0; i < elems_per_lane; ++i) printf(" %4d", srcidx[l * elems_per_lane + i]);
            printf("\n");
        }
        hipFree(daddr);
    }
    // ---------------- 3. conversions ---------------------------------------------------------------------------------
    {
        std::vector<float> x;
        for (int i = 0; i < 4096; ++i) x.push_back((float)((rand() / (double)RAND_MAX) * 16.0 - 8.0));
        static const float t[8] = {0.0f, 0.5f, 1.0f, 1.5f, 2.0f, 3.0f, 4.0f, 6.0f};
        for (int i = 0; i < 7; ++i) for (int s = -1; s <= 1; s += 2) {          // exact ties and their neighbours
            const float mid = 0.5f * (t[i] + t[i + 1]);
            x.push_back(s * mid); x.push_back(s * std::nextafterf(mid, 0.0f)); x.push_back(s * std::nextafterf(mid, 10.0f)); x.push_back(s * t[i]);
        }
        for (float v : {6.0f, 6.5f, 7.0f, 8.0f, 100.0f, 1e30f, INFINITY, -INFINITY, 0.0f, -0.0f, 0.25f, 0.2499999f, 0.2500001f, 1e-30f, -1e-30f}) x.push_back(v);
        if (x.size() & 1) x.push_back(0.0f);
        const int n = (int)x.size() / 2;
        float* dx = dev(x);
        uint32_t* dq; float* dbk; _Float16* dbk16;
        CK(hipMalloc(&dq, n * 4)); CK(hipMalloc(&dbk, n * 8)); CK(hipMalloc(&dbk16, n * 4));
        for (float scale : {1.0f, 2.0f, 0.25f, 3.0f, 1.5f}) {
            k_cvt<<<(n + 63) / 64, 64>>>(dx, n, scale, dq, dbk, dbk16);
            CK(hipDeviceSynchronize());
            std::vector<uint32_t> q(n); std::vector<float> bk(2 * n); std::vector<_Float16> bk16(2 * n);
            CK(hipMemcpy(q.data(), dq, n * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(bk.data(), dbk, n * 8, hipMemcpyDeviceToHost));
            CK(hipMemcpy(bk16.data(), dbk16, n * 4, hipMemcpyDeviceToHost));
            // hypothesis: code = rne_e2m1(x / 2^floor(log2(scale))) saturating; back = e2m1(code) * 2^floor(log2 scale)
            int e; std::frexp(scale, &e); const float p2 = std::ldexp(1.0f, e - 1);
            int bad_q = 0, bad_q_full = 0, bad_b = 0, bad_h = 0, shown = 0;
            for (int i = 0; i < 2 * n; ++i) {
                const int code = (q[i / 2] >> (4 * (i & 1))) & 15;
                const int want = host_e2m1_rne(x[i] / p2), want_full = host_e2m1_rne(x[i] / scale);
                if (code != want) { ++bad_q; if (shown++ < 6) printf("      x=%.9g scale=%g: code %d, rne(x/2^e) gives %d\n", x[i], scale, code, want); }
                if (code != want_full) ++bad_q_full;
                if (bk[i] != e2m1(code) * p2) ++bad_b;
                if ((float)bk16[i] != e2m1(code) * p2) ++bad_h;
            }
            printf("[cvt_scalef32_pk_fp4_f32 scale=%g] code == rne_e2m1_sat(x / 2^floor(log2 scale)): %s (%d bad of %d); == rne(x / scale): %d bad; "
                   "pk_f32_fp4 back == e2m1*2^floor(log2 scale): %d bad; pk_f16_fp4: %d bad; upper 24 bits of the packed word of pair 0: %06x\n",
                   scale, bad_q ? "NO" : "YES", bad_q, 2 * n, bad_q_full, bad_b, bad_h, q[0] >> 8);
        }
        float* dout; CK(hipMalloc(&dout, 64));
        k_cvt_sel<<<1, 1>>>(0x76543210u, 1.0f, dout);
        CK(hipDeviceSynchronize());
        float o[9]; CK(hipMemcpy(o, dout, 36, hipMemcpyDeviceToHost));
        uint32_t ow; memcpy(&ow, &o[8], 4);
        printf("[cvt_scalef32_pk_f32_fp4 of 0x76543210] sel0 = %g %g, sel1 = %g %g, sel2 = %g %g, sel3 = %g %g (codes 0..7 = 0 .5 1 1.5 2 3 4 6); "
               "pk_fp4_f32(old=0xAAAAAAAA, 1, -2, sel 2) = %08x\n", o[0], o[1], o[2], o[3], o[4], o[5], o[6], o[7], ow);
        // e4m3 with scale
        uint32_t* dq8; CK(hipMalloc(&dq8, n * 4));
        for (float scale : {1.0f, 4.0f, 0.125f}) {
            k_cvt8<<<(n + 63) / 64, 64>>>(dx, n, scale, dq8);
            CK(hipDeviceSynchronize());
            std::vector<uint32_t> q(n); CK(hipMemcpy(q.data(), dq8, n * 4, hipMemcpyDeviceToHost));
            int bad = 0, shown = 0;
            for (int i = 0; i < 2 * n; ++i) {
                const int code = (q[i / 2] >> (8 * (i & 1))) & 255;
                const float v = x[i] / scale;
                // nearest e4m3 (saturating to 448) by search
                int best = 0; float bd = INFINITY;
                for (int c = 0; c < 256; ++c) {
                    if ((c & 0x7f) == 0x7f) continue;
                    const float d = std::fabs(e4m3(c) - (std::isinf(v) ? std::copysign(448.0f, v) : v));
                    if (d < bd || (d == bd && !(c & 1) && std::signbit(e4m3(c)) == std::signbit(v))) { bd = d; best = c; }
                }
                if (e4m3(code) != e4m3(best) && !(std::isinf(x[i]))) { ++bad; if (shown++ < 4) printf("      x=%.9g scale=%g: code %02x (%g), nearest %02x (%g)\n", x[i], scale, code, e4m3(code), best, e4m3(best)); }
            }
            printf("[cvt_scalef32_pk_fp8_f32 scale=%g] == nearest-even e4m3 of x/scale, saturating: %s (%d bad of %d)\n", scale, bad ? "NO" : "YES", bad, 2 * n);
        }
    }
    return 0;
}
