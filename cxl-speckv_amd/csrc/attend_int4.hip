// cxl-speckv_amd/csrc/attend_int4.hip -- decode attention straight from INT4_G32 pool
// records (the 4:1 format of BASELINE config 5; SURVEY 8a row A22, no reference counterpart,
// parity against oracle/orc_attend_f16 over the decompressed pages).
//
// Record of one 4 KiB page (2 positions x 8 kv heads x 128 d, element e = (slot*8 + head)*128 + d):
// 64 fp16 group scales (group = 32 consecutive elements) then 1024 nibble bytes, low nibble = even
// element, two's complement.  A head's row of one position = 64 nibble bytes + 4 scales.
//
// Semantics: K and V are dequantised exactly as fetch+decompress does (fp16(q4 * scale), one
// rounding: v_pk_mul_f16 of two exactly represented factors), the query stays fp16, both products
// run on v_mfma_f32_16x16x32_f16 with fp32 accumulation, softmax weights are rounded to f16.  So
// the result is the attention over the decompressed fp16 pages without ever writing them.
//
// One wave = one kv head x one split of the positions, tiles of 32 positions.  Two kernels:
//   k_attend_int4_wg8   LINEAR form with 8 kv heads (records of the allocation in one run: record p at lin_base + p*1152,
//                       never-written records zero bytes, addresses are arithmetic): 8 waves = the 8 heads of a tile, whole
//                       records per workgroup, global -> LDS by LDS-DMA (see there)
//   k_attend_int4_wg    four neighbouring heads per workgroup, whole 128-byte lines by LDS-DMA, two tiles deep: the LINEAR
//                       form for other head counts, the STRIPED form (computed addresses over 2..8 pools) and the TABLE form
//                       (record addresses from the page-table entries, never-written pages read a zero page, look-ups
//                       clamped to the range: migrated pools, ranges whose last tile would leave the region)
// (Rounds 1-3 also had a per-wave page-table kernel with register-staged loads, 0.37 of HBM peak; the table form replaced it.)
// Operand mapping (both kernels):
//   scores S^T = K . q^T: lane (c, kb) feeds row c = position 16b + c, d = 32kb + 8*step + e --
//     exactly group kb of that row: 16 nibble bytes and one scale per lane and block.
//   output O^T = V^T . P^T: rows c = d columns 8c + t, k-slots = the lane's 8 positions (as in
//     attend.hip): the V tile goes through LDS and is read back as "8 nibbles of one position" dwords.
#include "kernels.hpp"

namespace speckv {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

namespace {

// Plain (temporal) loads on purpose: a head's row of one position is 64 B, HALF a 128-byte line, and the other half
// belongs to the neighbouring head = the neighbouring wave of this workgroup.  With non-temporal loads the line did not
// stay in L2 for the partner: PMC showed 19.1 M L2 misses for 11.8 M distinct lines (1.6x the record bytes from HBM,
// which then ran at 0.70 of its peak while the kernel delivered 0.43).  profiles/r02_int4_mem_pmc.json
__device__ __forceinline__ uint4 ldg16(const uint8_t* p)
{
    // (explicit global address space: record pointers read from page-table entries would otherwise make FLAT loads, which
    // count in lgkmcnt as well and serialise against every LDS wait -- attend.hip)
    typedef const u32x4 __attribute__((address_space(1)))* gp;
#ifdef SPECKV_INT4_NT_LOADS
    const u32x4 v = __builtin_nontemporal_load((gp)(reinterpret_cast<uintptr_t>(p)));
#else
    const u32x4 v = *(gp)(reinterpret_cast<uintptr_t>(p));
#endif
    return make_uint4(v.x, v.y, v.z, v.w);
}
template <typename T> __device__ __forceinline__ T ldg_small(const uint8_t* p)
{
    typedef const T __attribute__((address_space(1)))* gp;
    return *(gp)(reinterpret_cast<uintptr_t>(p));
}
__device__ __forceinline__ float max_over_kb(float v)
{
    // lane ^ 16 and lane ^ 32 through gfx950's row / half swaps (v_permlane16_swap / v_permlane32_swap: both operands the same
    // register -> the two rows, then the two halves, side by side), not through the LDS crossbar (ds_bpermute)
    const uint32_t u = __float_as_uint(v);
    const auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    const uint32_t m = __float_as_uint(fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1])));
    const auto b = __builtin_amdgcn_permlane32_swap(m, m, false, false);
    return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}
__device__ __forceinline__ float sum_over_kb(float v)
{
    const uint32_t u = __float_as_uint(v);
    const auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    const uint32_t m = __float_as_uint(__uint_as_float(a[0]) + __uint_as_float(a[1]));
    const auto b = __builtin_amdgcn_permlane32_swap(m, m, false, false);
    return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}
__device__ __forceinline__ f16x2 as_h2(uint32_t u) { return __builtin_bit_cast(f16x2, u); }

// The kernel is VALU-bound on the nibble -> f16 conversion (rocprofv3: 426 VALU per 32-position tile and wave at first,
// VALU issue 63 % busy, MFMA 9 %; profiles/r02_int4_pmc_before.json), so the conversion is written for instruction count.
// A two's-complement nibble t, xor 8, is q + 8 (0..15); placed in the low mantissa bits of fp16 1024.0 (ulp 1) it reads
// as the exact fp16 integer 1032 + q, and a nibble four bits higher, in the mantissa of 64.0 (that bit weighs 1), as
// 72 + q.  So ONE v_bitop3_b32 -- (w & mask) ^ magic, the xor-8 folded into the magic -- turns a dword into a PAIR of fp16
// integers, nibbles 16 bits apart; subtracting the bias is exact and the product with the group scale rounds once:
//     fp16(q) * s  =  fp16(q * s),   bit for bit the value fetch + decompress stores,
// for every scale (no -8 s term that could overflow: groups with scales beyond 8188 need no path of their own).
// Per dword of 8 nibbles: 1 shift + 4 bitop3 + 4 packed subtracts + 4 packed multiplies = 13 instructions, against 16
// with byte-selecting converts (v_cvt_f16_u16_sdwa per value) and a packed FMA per pair.
template <int HI> __device__ __forceinline__ f16x2 deq_pair(uint32_t w, f16x2 s2)
{
    const uint32_t u = HI ? ((w & 0x00F000F0u) ^ 0x54805480u) : ((w & 0x000F000Fu) ^ 0x64086408u);
    const f16x2 bias = HI ? f16x2{static_cast<_Float16>(72.0f), static_cast<_Float16>(72.0f)}
                          : f16x2{static_cast<_Float16>(1032.0f), static_cast<_Float16>(1032.0f)};
    return (as_h2(u) - bias) * s2;
}
// 8 nibbles of one row (two's complement, nibble p = element p) times the row's group scale, in the order
// e0 e4 e1 e5 e2 e6 e3 e7 (pairs are 16 bits apart); the query operand is loaded in the same order (q_operand)
__device__ __forceinline__ f16x8 deq_row8(uint32_t w, f16x2 s2)
{
    const uint32_t w8 = w >> 8;
    const f16x2 a = deq_pair<0>(w, s2), b = deq_pair<1>(w, s2), c = deq_pair<0>(w8, s2), d = deq_pair<1>(w8, s2);
    return f16x8{a.x, a.y, b.x, b.y, c.x, c.y, d.x, d.y};
}
// 8 consecutive fp16 query elements q0..q7 -> the operand order of deq_row8
__device__ __forceinline__ f16x8 q_operand(uint4 t)
{
    const uint32_t w0 = __builtin_amdgcn_perm(t.z, t.x, 0x05040100u), w1 = __builtin_amdgcn_perm(t.z, t.x, 0x07060302u);   // q0 q4 | q1 q5
    const uint32_t w2 = __builtin_amdgcn_perm(t.w, t.y, 0x05040100u), w3 = __builtin_amdgcn_perm(t.w, t.y, 0x07060302u);   // q2 q6 | q3 q7
    return __builtin_bit_cast(f16x8, make_uint4(w0, w1, w2, w3));
}

// ---- the arithmetic of one 32-position tile, shared by the two kernels below ----------------------------------------
// scores of one block of 16 positions: S^T = K . q^T (raw dot products)
__device__ __forceinline__ f32x4 score_block(const uint4& kx, uint32_t ks16, const f16x8 (&qv)[4])
{
    const _Float16 sh = __builtin_bit_cast(_Float16, static_cast<uint16_t>(ks16));
    const f16x2 s2 = {sh, sh};
    const uint32_t w[4] = {kx.x, kx.y, kx.z, kx.w};
    f32x4 s = {0.0f, 0.0f, 0.0f, 0.0f};
#ifdef SPECKV_ABL_NO_QK
    s[0] = __uint_as_float((w[0] ^ w[1] ^ w[2] ^ w[3]) & 0x3F000000u) + static_cast<float>(s2.x);
#else
#pragma unroll
    for (int st = 0; st < 4; ++st)
        s = __builtin_amdgcn_mfma_f32_16x16x32_f16(deq_row8(w[st], s2), qv[st], s, 0, 0, 0);
#endif
    return s;
}

// online softmax of query row c over the tile's 8 scores of this lane.  The running reference m_run only moves when the
// tile's maximum passes it by more than 2^kLazy: until then the weights are exp2(x - m_run) <= 2^kLazy (fine for f16,
// the sums are fp32) and the accumulators need no rescaling -- the result is the same after normalisation.
constexpr float kLazy = 8.0f;
__device__ __forceinline__ f16x8 softmax_tile(const float (&sc)[8], float qscale, float& m_run, float& l_run, f32x4 (&acc)[8])
{
    float mx = sc[0];
#pragma unroll
    for (int j = 1; j < 8; ++j) mx = fmaxf(mx, sc[j]);
    mx = max_over_kb(mx) * qscale;                                 // qscale > 0
    const bool grow = mx > m_run + kLazy;                         // also true for the first tile (m_run = -inf)
    if (__builtin_amdgcn_ballot_w64(grow) != 0ull) {              // wave-uniform: rare after the first tiles
        const float m_new = grow ? mx : m_run;
        const float m_use = (m_new == -INFINITY) ? 0.0f : m_new;
        const float f = __builtin_amdgcn_exp2f(m_run - m_use);    // 0 for the first tile, 1 for rows that keep theirs
        m_run = m_new;
        l_run *= f;
#pragma unroll
        for (int t = 0; t < 8; ++t) acc[t] = acc[t] * f;
    }
    const float m_sub = (m_run == -INFINITY) ? 0.0f : -m_run;
    float psum = 0.0f;
    f16x8 P;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(sc[j], qscale, m_sub));
        psum += p;
        P[j] = static_cast<_Float16>(p);
    }
    l_run += psum;
    return P;
}

// out^T += V^T . P^T, accumulated in place.  vw[j] = 8 nibbles (two's complement) of position slot j at d = 8c..8c+7,
// vs16[j] = that slot's group scale (group c/4).  The MFMA of column d = 8c + t wants nibble t of all 8 slots, as pairs
// (slot 2jp, slot 2jp+1): two byte permutes per slot pair put the low halves of both dwords (nibbles 0..3) into one
// register and the high halves into another, nibbles of the two slots 16 bits apart, as deq_pair wants them.
__device__ __forceinline__ void pv_tile(const uint32_t (&vw)[8], const uint32_t (&vs16)[8], const f16x8& P, f32x4 (&acc)[8])
{
#ifdef SPECKV_ABL_NO_PV
    {
        uint32_t x = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) x ^= vw[j] ^ vs16[j];
        acc[0][0] += __uint_as_float(x & 0x3F000000u) + static_cast<float>(P[0]);
        return;
    }
#endif
    f16x2 s2[4];
#pragma unroll
    for (int jp = 0; jp < 4; ++jp) s2[jp] = as_h2(vs16[2 * jp] | (vs16[2 * jp + 1] << 16));
    // one nibble plane pair is live at a time: t = 0, 1 from the low halves, 2, 3 from the same shifted by 8, then the
    // high halves (the planes replace the dwords they came from)
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        uint32_t x[4];
#pragma unroll
        for (int jp = 0; jp < 4; ++jp) x[jp] = __builtin_amdgcn_perm(vw[2 * jp + 1], vw[2 * jp], half ? 0x07060302u : 0x05040100u);
#pragma unroll
        for (int sh = 0; sh < 2; ++sh) {
            if (sh) {
#pragma unroll
                for (int jp = 0; jp < 4; ++jp) x[jp] >>= 8;
            }
#pragma unroll
            for (int hi = 0; hi < 2; ++hi) {
                const int t = 4 * half + 2 * sh + hi;
                f16x8 V;
#pragma unroll
                for (int jp = 0; jp < 4; ++jp) {
                    const f16x2 v = hi ? deq_pair<1>(x[jp], s2[jp]) : deq_pair<0>(x[jp], s2[jp]);
                    V[2 * jp] = v.x;
                    V[2 * jp + 1] = v.y;
                }
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(V, P, acc[t], 0, 0, 0);
                // V stays live past the MFMA: otherwise the register allocator writes the result over V, away from
                // acc[t], and moves all 32 accumulators back at the end of every iteration
                asm volatile("" :: "v"(V));
            }
        }
    }
}

// the split's partial result: running maximum, sum and the un-normalised accumulators
__device__ __forceinline__ void store_partial(const AttendArgs& a, uint64_t part, uint64_t row, uint32_t my_splits, uint32_t c, uint32_t kb, float m_run, float l_run, const f32x4 (&acc)[8])
{
#ifdef SPECKV_ABL_NO_STORE
    if (l_run != 12345.0f) return;                                         // timing only (profiles/tools/int4_where.sh)
#endif
    const float l_tot = sum_over_kb(l_run);
    if (a.direct_out && (!a.direct_per_seq || my_splits == 1u)) {                               // single split per row: the final result (AttendArgs::direct_out)
        if (c < a.g) {
            const float w = l_tot > 0.0f ? 1.0f / l_tot : 0.0f;
            float* dst = a.direct_out + (row * a.g + c) * 128u + 32u * kb;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                *reinterpret_cast<f32x4*>(dst + 8 * i) = f32x4{acc[0][i], acc[1][i], acc[2][i], acc[3][i]} * w;
                *reinterpret_cast<f32x4*>(dst + 8 * i + 4) = f32x4{acc[4][i], acc[5][i], acc[6][i], acc[7][i]} * w;
            }
            if (a.direct_lse && kb == 0)
                a.direct_lse[row * a.g + c] = l_tot > 0.0f ? (m_run + log2f(l_tot)) * 0.6931471805599453f : -INFINITY;
        }
        return;
    }
    if (kb == 0) {
        a.part_ml[part * 32u + c] = m_run;
        a.part_ml[part * 32u + 16u + c] = l_tot;
    }
    if (c < a.g) {
        float* dst = a.part_acc + (part * 16u + c) * 128u + 32u * kb;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            *reinterpret_cast<f32x4*>(dst + 8 * i) = f32x4{acc[0][i], acc[1][i], acc[2][i], acc[3][i]};
            *reinterpret_cast<f32x4*>(dst + 8 * i + 4) = f32x4{acc[4][i], acc[5][i], acc[6][i], acc[7][i]};
        }
    }
}

} // namespace

constexpr uint32_t kWgHeads = 4;     // kv heads (= waves) per workgroup of k_attend_int4_wg

// ---------------------------------------------------------------------------------------------------------------------
// Linear form, workgroup-cooperative LDS-DMA.  Measured on the per-wave kernels: with all arithmetic removed they ran no
// faster (0.55 -> 0.57 of HBM peak), with two tiles in flight per wave instead of one no faster either -- but with every
// wave instruction covering WHOLE 128-byte lines the same loop ran at 0.665.  A head's row is 64 B, half a line, and the
// other half belongs to the neighbouring wave: fetched per head, every line is requested twice, by different waves at
// different times.  Here the four waves of a workgroup fetch the tile for their four heads together: per position row
// one 256-byte span (2 lines), per page one 128-byte line of scales, 5 DMA instructions of 1 KiB per wave and tile.
//   LDS, per buffer (two buffers):  K rows [32][256 B] | V rows [32][256 B] | K scale lines [16][128 B] | V scale lines
//   rows are stored with their sixteen 16-byte pieces XOR-ed by (row & 15) (chosen on the SOURCE side: lane l fetches
//   piece (l & 15) ^ row): the K operand reads (ds_read_b128, 16 rows x one piece) and the V reads (ds_read_b32, rows
//   4 kb + j) are then conflict-free; K scale lines likewise by (page & 7), V scale lines by (page / 2) & 3.
//   iteration t:  own DMAs of tile t landed (vmcnt) + barrier -> K operand, scores, softmax, V dwords
//                 -> barrier (everybody is done with the buffer) -> DMAs of tile t+2 into it -> PV
namespace {
constexpr uint32_t kWgBuf = 20480u, kWgV = 8192u, kWgKs = 16384u, kWgVs = 18432u;

// one LDS-DMA: lane l's 16 bytes at base + voff land at lds_dst + 16 l.  M0 (the destination) belongs to the compiler:
// saved and restored inside the statement.
__device__ __forceinline__ void dma16(uint32_t lds_dst, const uint8_t* base, uint32_t voff)
{
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(lds_dst), "v"(voff), "s"(base) : "memory");
}

// the same with a full 64-bit address per lane (striped form: a lane's rows may sit in any pool's run)
__device__ __forceinline__ void dma16v(uint32_t lds_dst, const uint8_t* addr)
{
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(lds_dst), "v"(addr) : "memory");
}

// this wave's 5 DMAs of the current tile have landed (the 5 of the next tile may still be in flight); then everybody's
// (SPECKV_ABL_* macros: timing-only ablation builds, results are garbage -- profiles/tools/int4_ablate.sh)
#ifdef SPECKV_ABL_NO_BARRIER
#define SPECKV_WG_BARRIER ""
#else
#define SPECKV_WG_BARRIER "\n\ts_barrier"
#endif
__device__ __forceinline__ void wg_landed(bool is_last)
{
    if (is_last) asm volatile("s_waitcnt vmcnt(0)" SPECKV_WG_BARRIER ::: "memory");
    else         asm volatile("s_waitcnt vmcnt(5)" SPECKV_WG_BARRIER ::: "memory");
}
__device__ __forceinline__ void wg_take_k(uint32_t rd, uint32_t rs, u32x4& k0, u32x4& k1, uint32_t& s0, uint32_t& s1)
{
    asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:4096\n\t"
                 "ds_read_u16 %2, %5\n\tds_read_u16 %3, %5 offset:1024\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(k0), "=&v"(k1), "=&v"(s0), "=&v"(s1) : "v"(rd), "v"(rs) : "memory");
}
// rd[j] = address of the dword of row 4 kb + j; rows 16 + 4 kb + j are 4096 B further.  Scales: page 2 kb + j/2 (+ 8),
// slot j % 2 = 64 B further
__device__ __forceinline__ void wg_take_v(const uint32_t (&rd)[4], uint32_t rs, uint32_t (&w)[8], uint32_t (&sc)[8])
{
    asm volatile("ds_read_b32 %0, %16\n\tds_read_b32 %1, %17\n\tds_read_b32 %2, %18\n\tds_read_b32 %3, %19\n\t"
                 "ds_read_b32 %4, %16 offset:4096\n\tds_read_b32 %5, %17 offset:4096\n\tds_read_b32 %6, %18 offset:4096\n\tds_read_b32 %7, %19 offset:4096\n\t"
                 "ds_read_u16 %8, %20\n\tds_read_u16 %9, %20 offset:64\n\tds_read_u16 %10, %20 offset:128\n\tds_read_u16 %11, %20 offset:192\n\t"
                 "ds_read_u16 %12, %20 offset:1024\n\tds_read_u16 %13, %20 offset:1088\n\tds_read_u16 %14, %20 offset:1152\n\tds_read_u16 %15, %20 offset:1216\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(w[0]), "=&v"(w[1]), "=&v"(w[2]), "=&v"(w[3]), "=&v"(w[4]), "=&v"(w[5]), "=&v"(w[6]), "=&v"(w[7]),
                   "=&v"(sc[0]), "=&v"(sc[1]), "=&v"(sc[2]), "=&v"(sc[3]), "=&v"(sc[4]), "=&v"(sc[5]), "=&v"(sc[6]), "=&v"(sc[7])
                 : "v"(rd[0]), "v"(rd[1]), "v"(rd[2]), "v"(rd[3]), "v"(rs) : "memory");
}
} // namespace

// 3 waves per SIMD: 4 fit (109 VGPRs, 4 x 40 KiB of LDS) and measure the same (-DSPECKV_INT4_WG_WAVES=4)
#ifndef SPECKV_INT4_WG_WAVES
#define SPECKV_INT4_WG_WAVES 3
#endif
// STRIPED: the allocation is striped regularly over several pools (AttendArgs::stripe_bases): the five DMA addresses of a
// wave and tile are computed per lane from the page number (multiply-high, LDS read of the run base, 64-bit multiply-add)
// and the DMAs take a full address per lane.  Single-sequence form only.
// TABLE: an allocation without a regular placement (pages migrated one by one; AttendArgs::table_form), tile-aligned range:
// the five record addresses of a wave and tile come from page-table entries.  The entries are fetched by hand-counted loads
// one tile ahead of the DMAs that need them, so that neither waits for the other:
//     after tile t is consumed:  s_waitcnt vmcnt(5)  (entries of t+2 are here; the 5 DMAs of t+1 may fly)
//                                entry loads of t+3, DMAs of t+2
//     top of iteration t+1:      s_waitcnt vmcnt(10) (DMAs of t+1 have landed; entries of t+3 and DMAs of t+2 may fly)
// The schedule is uniform -- tile numbers beyond the split are clamped (their DMAs land in a buffer nobody reads again), an
// odd split gets one extra step whose scores are -inf -- because the counts are constants.  Single-sequence form only.
template <bool STRIPED, bool TABLE = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(SPECKV_INT4_WG_WAVES, SPECKV_INT4_WG_WAVES))) void k_attend_int4_wg(AttendArgs a)
{
    static_assert(kWgHeads == 4, "the cooperative kernel is laid out for 4 heads per workgroup");
    __shared__ __attribute__((aligned(16))) uint8_t lds[2 * kWgBuf];
    __shared__ uint64_t s_bases[8];
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t c = lane & 15u, kb = lane >> 4;
    const uint32_t split = a.rows_first ? blockIdx.y : blockIdx.x, by = a.rows_first ? blockIdx.x : blockIdx.y;
    if (a.rows_first && by >= a.rows_real) return;                      // (padding row: see AttendArgs::rows_real)
    const uint32_t hq = a.heads / 4u;
    uint32_t layer = by / hq;                                            // batch form: the sequence index
    if (a.seqs && a.order) layer = a.order[layer];                       // (workgroup-uniform)
    const uint32_t head0 = (by % hq) * 4u;                               // first head of the workgroup
    const uint32_t head = head0 + wave;
    const uint64_t row = static_cast<uint64_t>(layer) * a.heads + head;
    uint64_t part = row * a.n_splits + split;
    uint32_t my_splits = a.n_splits;
    if (a.seqs) {                                                        // workgroup-uniform: per-sequence geometry
        const AttendSeq sq = a.seqs[layer];
        if (split >= sq.n_splits) {
            if (sq.n_splits == 0u && split == 0u && a.direct_out && a.direct_per_seq == 2u) attend_zero_rows(a.direct_out, a.direct_lse, a.g, row, lane);
            return;
        }
        a.lin_base = sq.lin_base;
        a.k_first = sq.k_first;
        a.v_first = sq.v_first;
        a.n_pages = sq.n_pages;
        a.tiles_per_split = sq.tiles_per_split;
        a.k_first += static_cast<uint64_t>(a.batch_layer) * sq.layer_pages;
        a.v_first += static_cast<uint64_t>(a.batch_layer) * sq.layer_pages;
        my_splits = sq.n_splits;
        part = sq.part_base + static_cast<uint64_t>(head) * sq.n_splits + split;
        layer = 0;
        if (TABLE) a.entries = reinterpret_cast<const PageEntry*>(sq.lin_base);
        if (STRIPED) {                                                   // the sequence's own placement
            a.stripe_bases = sq.stripe_bases;
            a.stripe_n = sq.stripe_n;
            a.stripe_magic = sq.stripe_n > 1u ? static_cast<uint32_t>((1ull << 32) / sq.stripe_n + 1u) : 0u;
        }
    }
    if (STRIPED) {
        if (threadIdx.x < 8u) s_bases[threadIdx.x] = a.stripe_bases[threadIdx.x];
        __syncthreads();
    }

    f16x8 qv[4];
    if (TABLE) {                                                         // (the table form asks for its query rows first: its entry look-ups are counted by hand)
        const uint16_t* q16 = reinterpret_cast<const uint16_t*>(a.q8) + (row * a.g + c) * 128u + kb * 32u;
#pragma unroll
        for (int st = 0; st < 4; ++st) {
            uint4 t = make_uint4(0u, 0u, 0u, 0u);
            if (c < a.g) t = *reinterpret_cast<const uint4*>(q16 + 8 * st);
            qv[st] = q_operand(t);
        }
    }
    const float qscale = a.scale_log2e;
    const uint32_t n_tiles = (a.n_pages + 15u) / 16u;
    const uint32_t t0 = split * a.tiles_per_split;
    const uint32_t t1 = min(t0 + a.tiles_per_split, n_tiles);
    float m_run = -INFINITY, l_run = 0.0f;
    f32x4 acc[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) acc[t] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};

    if (t0 < t1) {                                                       // workgroup-uniform
        const uint32_t tile_bytes = 16u * kInt4RecBytes;
        const uint8_t* kreg = nullptr;
        const uint8_t* vreg = nullptr;
        if (!STRIPED && !TABLE) {
            kreg = a.lin_base + (a.k_first + layer * a.layer_stride) * kInt4RecBytes;      // tile tt: + tt * tile_bytes
            vreg = a.lin_base + (a.v_first + layer * a.layer_stride) * kInt4RecBytes;
        }
        // ---- this wave's share of the fetch: rows 8w .. 8w+7 of K and of V (two instructions of 4 rows each), and the
        // scale lines of 8 pages of K (waves 0, 1) or V (waves 2, 3)
        auto row_in = [&](uint32_t r, uint32_t piece) { return 128u + ((r & 1u) * 8u + head0) * 64u + piece * 16u; };   // offset inside the record
        const uint32_t r0 = 8u * wave + (lane >> 4), r1 = r0 + 4u;
        const uint32_t in0 = row_in(r0, (lane & 15u) ^ (r0 & 15u)), in1 = row_in(r1, (lane & 15u) ^ (r1 & 15u));
        const uint32_t gr0 = (r0 >> 1) * kInt4RecBytes + in0, gr1 = (r1 >> 1) * kInt4RecBytes + in1;
        const uint32_t spage = 8u * (wave & 1u) + (lane >> 3);
        // (V scale lines: pieces xor-ed by (page / 2) & 3 -- the reader's four position groups kb read pages 2 kb + j / 2, 256 B
        // apart, i.e. the same banks: PMC showed 13.1 M LDS conflict cycles per launch, 20 per tile and wave, all from these 8 reads)
        const uint32_t sin = ((lane & 7u) ^ ((wave < 2u) ? (spage & 7u) : ((spage >> 1) & 3u))) * 16u;
        const uint32_t gs = spage * kInt4RecBytes + sin;
        const uint32_t lbase = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(reinterpret_cast<uintptr_t>(&lds[0])));
        const uint32_t dr0 = 8u * wave * 256u, ds_ = ((wave < 2u) ? kWgKs : kWgVs) + (wave & 1u) * 1024u;   // destinations inside a buffer
        const uint8_t* sreg = (wave < 2u) ? kreg : vreg;
        const uint32_t last = t1 - 1u;
        // striped form: first page of the K / V region of this layer; the lane's pages of tile tt are + 16 tt + ...
        const uint32_t kpage0 = static_cast<uint32_t>(a.k_first + layer * a.layer_stride), vpage0 = static_cast<uint32_t>(a.v_first + layer * a.layer_stride);
        const uint32_t spage0 = ((wave < 2u) ? kpage0 : vpage0) + spage;
        auto rec = [&](uint32_t page) { return attend_stripe_rec(s_bases, page, a.stripe_n, a.stripe_magic, kInt4RecBytes); };
        auto issue = [&](uint32_t tt, uint32_t buf) {
            const uint32_t dst = lbase + buf * kWgBuf;
            if (STRIPED) {
                const uint32_t pg = tt * 16u + (r0 >> 1);                 // rows r0 and r0 + 4: pages pg and pg + 2
                dma16v(dst + dr0, rec(kpage0 + pg) + in0);
                dma16v(dst + dr0 + 1024u, rec(kpage0 + pg + 2u) + in1);
                dma16v(dst + kWgV + dr0, rec(vpage0 + pg) + in0);
                dma16v(dst + kWgV + dr0 + 1024u, rec(vpage0 + pg + 2u) + in1);
                dma16v(dst + ds_, rec(spage0 + tt * 16u) + sin);
            } else {
                const uint64_t to = static_cast<uint64_t>(tt) * tile_bytes;
                dma16(dst + dr0, kreg + to, gr0);
                dma16(dst + dr0 + 1024u, kreg + to, gr1);
                dma16(dst + kWgV + dr0, vreg + to, gr0);
                dma16(dst + kWgV + dr0 + 1024u, vreg + to, gr1);
                dma16(dst + ds_, sreg + to, gs);
            }
        };
        // ---- reader addresses inside buffer 0
        const uint32_t rdk = lbase + c * 256u + (((wave * 4u + kb) ^ c) * 16u);                       // row 16 b + c: + 4096 b
        const uint32_t h1 = head >> 1;
        const uint32_t rsk = lbase + kWgKs + (c >> 1) * 128u + ((((c & 1u) * 4u + h1) ^ (c >> 1)) * 16u) + (head & 1u) * 8u + kb * 2u;
        uint32_t rdv[4];
#pragma unroll
        for (uint32_t j = 0; j < 4; ++j)
            rdv[j] = lbase + kWgV + (4u * kb + j) * 256u + ((((wave ^ kb) * 4u) + ((c >> 2) ^ j)) * 16u) + (c & 3u) * 4u;
        const uint32_t rsv = lbase + kWgVs + 2u * kb * 128u + ((h1 ^ kb) * 16u) + (head & 1u) * 8u + 2u * (c >> 2);   // pages 2 kb + j / 2 (+ 8): (page / 2) & 3 = kb
        // table form: the query rows must have arrived before the first DMA is issued: the compiler would otherwise place its own
        // vmcnt(0) for them at their first use, inside the loop, and drain the pipeline there in every iteration
        if (TABLE) asm volatile("" :: "v"(qv[0]), "v"(qv[1]), "v"(qv[2]), "v"(qv[3]));
#ifdef SPECKV_ABL_PLAIN_LOADS
        // timing only: the same bytes at the same addresses by plain 16-byte loads into registers, two tiles in flight, nothing else
        if (!STRIPED) {
            uint4 A[5], B[5], sink = make_uint4(0u, 0u, 0u, 0u);
            auto ld = [&](uint4 (&r)[5], uint32_t tt) {
                const uint64_t to = static_cast<uint64_t>(tt < last ? tt : last) * tile_bytes;
                r[0] = ldg16(kreg + to + gr0); r[1] = ldg16(kreg + to + gr1); r[2] = ldg16(vreg + to + gr0);
                r[3] = ldg16(vreg + to + gr1); r[4] = ldg16(sreg + to + gs);
            };
            auto eat = [&](const uint4 (&r)[5]) {
#pragma unroll
                for (int i = 0; i < 5; ++i) { sink.x ^= r[i].x; sink.y ^= r[i].y; sink.z ^= r[i].z; sink.w ^= r[i].w; }
            };
            ld(A, t0); ld(B, t0 + 1u);
#pragma unroll 1
            for (uint32_t tile = t0; tile < t1; tile += 2u) { eat(A); ld(A, tile + 2u); eat(B); ld(B, tile + 3u); }
            eat(A); eat(B);
            l_run += __uint_as_float((sink.x ^ sink.y ^ sink.z ^ sink.w) & 1u);
            store_partial(a, part, row, my_splits, c, kb, m_run, l_run, acc);
            return;
        }
#endif
        const bool ragged = (a.n_pages & 15u) != 0u;
        if (TABLE) {
            // the lane's five page-table entries of tile tt: K pages of its rows r0 / r1, the same of V, the scale line's page
            const uint32_t k_end = kpage0 + a.n_pages - 1u, v_end = vpage0 + a.n_pages - 1u;
            auto entry_load = [&](u32x4& e, uint32_t page, uint32_t end) {
                const PageEntry* p = a.entries + min(page, end);
                asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(e) : "v"(p) : "memory");
            };
            auto lookups = [&](u32x4 (&e)[5], uint32_t tt) {
                const uint32_t t_ = min(tt, last), pg = t_ * 16u + (r0 >> 1);
                entry_load(e[0], kpage0 + pg, k_end);
                entry_load(e[1], kpage0 + pg + 2u, k_end);
                entry_load(e[2], vpage0 + pg, v_end);
                entry_load(e[3], vpage0 + pg + 2u, v_end);
                entry_load(e[4], spage0 + t_ * 16u, (wave < 2u) ? k_end : v_end);
            };
            auto base_of = [&](const u32x4& e) -> const uint8_t* {       // {address lo, hi, record bytes, scale}; never written: zeros
                const uint8_t* r = reinterpret_cast<const uint8_t*>(static_cast<uint64_t>(e.x) | (static_cast<uint64_t>(e.y) << 32));
                return e.z >= kInt4RecBytes ? r : a.zero_page;
            };
            auto dmas = [&](const u32x4 (&e)[5], uint32_t buf) {
                const uint32_t dst = lbase + buf * kWgBuf;
                dma16v(dst + dr0, base_of(e[0]) + in0);
                dma16v(dst + dr0 + 1024u, base_of(e[1]) + in1);
                dma16v(dst + kWgV + dr0, base_of(e[2]) + in0);
                dma16v(dst + kWgV + dr0 + 1024u, base_of(e[3]) + in1);
                dma16v(dst + ds_, base_of(e[4]) + sin);
            };
            // one tile: `mine` holds the entries of tile + 2 (landing), `other` takes those of tile + 3
            auto step = [&](uint32_t tile, uint32_t buf, u32x4 (&mine)[5], u32x4 (&other)[5]) {
                const uint32_t bo = buf * kWgBuf;
                asm volatile("s_waitcnt vmcnt(10)\n\ts_barrier" ::: "memory");
                u32x4 k0, k1;
                uint32_t ks0, ks1;
                wg_take_k(rdk + bo, rsk + bo, k0, k1, ks0, ks1);
                float sc[8];
                {
                    const f32x4 s0 = score_block(make_uint4(k0.x, k0.y, k0.z, k0.w), ks0, qv);
                    const f32x4 s1 = score_block(make_uint4(k1.x, k1.y, k1.z, k1.w), ks1, qv);
#pragma unroll
                    for (int i = 0; i < 4; ++i) { sc[i] = s0[i]; sc[4 + i] = s1[i]; }
                }
                if ((ragged && tile + 1u == n_tiles) || tile > last) {      // workgroup-uniform: positions beyond the range / the extra step
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const uint32_t pg = tile * 16u + ((j >> 1) < 2 ? 2u * kb + (j >> 1) : 8u + 2u * kb + ((j >> 1) - 2));
                        if (pg >= a.n_pages || tile > last) sc[j] = -INFINITY;
                    }
                }
                const f16x8 P = softmax_tile(sc, qscale, m_run, l_run, acc);
                uint32_t vw[8], vs16[8];
                const uint32_t rdvb[4] = {rdv[0] + bo, rdv[1] + bo, rdv[2] + bo, rdv[3] + bo};
                wg_take_v(rdvb, rsv + bo, vw, vs16);
                // buffer free; the entries of tile + 2 are here.  (They are operands of the wait: their uses below must not be
                // scheduled in front of it -- to the compiler the loads that produced them were complete when issued.)
                asm volatile("s_barrier\n\ts_waitcnt vmcnt(5)" : "+v"(mine[0]), "+v"(mine[1]), "+v"(mine[2]), "+v"(mine[3]), "+v"(mine[4]) :: "memory");
                lookups(other, tile + 3u);
                dmas(mine, buf);
                pv_tile(vw, vs16, P, acc);
            };
            u32x4 A[5], B[5];
            lookups(A, t0);
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(A[0]), "+v"(A[1]), "+v"(A[2]), "+v"(A[3]), "+v"(A[4]) :: "memory");
            lookups(B, t0 + 1u);
            dmas(A, 0u);                                                  // tile t0
            asm volatile("s_waitcnt vmcnt(5)" : "+v"(B[0]), "+v"(B[1]), "+v"(B[2]), "+v"(B[3]), "+v"(B[4]) :: "memory");   // entries of t0 + 1 (the DMAs of t0 may fly)
            lookups(A, t0 + 2u);
            dmas(B, 1u);                                                  // tile t0 + 1
            const uint32_t t_end = t0 + ((t1 - t0 + 1u) & ~1u);
#pragma unroll 1
            for (uint32_t tile = t0; tile < t_end; tile += 2u) {
                step(tile, 0u, A, B);                                     // A: entries of tile + 2; B <- tile + 3
                step(tile + 1u, 1u, B, A);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // (the clamped DMAs of the last two steps)
            store_partial(a, part, row, my_splits, c, kb, m_run, l_run, acc);
            return;
        }
        // first tile, THEN the query rows (inline assembly: not the compiler's to count), then the second tile -- or the first once
        // more, so that the wait's count holds for a run of one tile: descriptor -> {tile, query, tile} -> scores (see k_attend_int4_wg8)
        issue(t0, 0u);
        {
            u32x4 tq0, tq1, tq2, tq3;
            const uint32_t cq = c < a.g ? c : a.g - 1u;
            const uint16_t* q16 = reinterpret_cast<const uint16_t*>(a.q8) + (row * a.g + cq) * 128u + kb * 32u;
            asm volatile("global_load_dwordx4 %0, %4, off\n\tglobal_load_dwordx4 %1, %4, off offset:16\n\t"
                         "global_load_dwordx4 %2, %4, off offset:32\n\tglobal_load_dwordx4 %3, %4, off offset:48"
                         : "=&v"(tq0), "=&v"(tq1), "=&v"(tq2), "=&v"(tq3) : "v"(q16) : "memory");
            issue(t0 < last ? t0 + 1u : t0, 1u);
            asm volatile("s_waitcnt vmcnt(5)" : "+v"(tq0), "+v"(tq1), "+v"(tq2), "+v"(tq3) :: "memory");
            const bool qlive = c < a.g;
            const u32x4 z = {0u, 0u, 0u, 0u};
            tq0 = qlive ? tq0 : z; tq1 = qlive ? tq1 : z; tq2 = qlive ? tq2 : z; tq3 = qlive ? tq3 : z;
            qv[0] = q_operand(make_uint4(tq0.x, tq0.y, tq0.z, tq0.w)); qv[1] = q_operand(make_uint4(tq1.x, tq1.y, tq1.z, tq1.w));
            qv[2] = q_operand(make_uint4(tq2.x, tq2.y, tq2.z, tq2.w)); qv[3] = q_operand(make_uint4(tq3.x, tq3.y, tq3.z, tq3.w));
        }
#pragma unroll 1
        for (uint32_t tile = t0; tile < t1; ++tile) {
            const uint32_t buf = (tile - t0) & 1u;
            const uint32_t bo = buf * kWgBuf;
            wg_landed(tile == last);
            u32x4 k0, k1;
            uint32_t ks0, ks1;
#ifdef SPECKV_ABL_NO_LDSREAD
            k0 = u32x4{tile, lane, 3u, 4u}; k1 = u32x4{lane, tile, 7u, 8u}; ks0 = 0x3C00u; ks1 = 0x3C00u;
#else
            wg_take_k(rdk + bo, rsk + bo, k0, k1, ks0, ks1);
#endif
            float sc[8];
            {
                const f32x4 s0 = score_block(make_uint4(k0.x, k0.y, k0.z, k0.w), ks0, qv);
                const f32x4 s1 = score_block(make_uint4(k1.x, k1.y, k1.z, k1.w), ks1, qv);
#pragma unroll
                for (int i = 0; i < 4; ++i) { sc[i] = s0[i]; sc[4 + i] = s1[i]; }
            }
            if (ragged && tile + 1u == n_tiles) {                         // workgroup-uniform: positions beyond the range
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const uint32_t pg = tile * 16u + ((j >> 1) < 2 ? 2u * kb + (j >> 1) : 8u + 2u * kb + ((j >> 1) - 2));
                    if (pg >= a.n_pages) sc[j] = -INFINITY;
                }
            }
            const f16x8 P = softmax_tile(sc, qscale, m_run, l_run, acc);
            uint32_t vw[8], vs16[8];
            const uint32_t rdvb[4] = {rdv[0] + bo, rdv[1] + bo, rdv[2] + bo, rdv[3] + bo};
#ifdef SPECKV_ABL_NO_LDSREAD
#pragma unroll
            for (int j = 0; j < 8; ++j) { vw[j] = tile + j + lane; vs16[j] = 0x3C00u; }
            (void)rdvb;
#else
            wg_take_v(rdvb, rsv + bo, vw, vs16);
#endif
#ifndef SPECKV_ABL_NO_BARRIER
            asm volatile("s_barrier" ::: "memory");                       // every wave has taken what it needs from this buffer
#endif
            if (tile + 2u <= last) issue(tile + 2u, buf);                 // (taking V earlier, to issue earlier, measured the same)
            pv_tile(vw, vs16, P, acc);
        }
    }
    store_partial(a, part, row, my_splits, c, kb, m_run, l_run, acc);
}

// ---------------------------------------------------------------------------------------------------------------------
// Linear form, WHOLE RECORDS per workgroup: 8 waves = the 8 kv heads.  The 4-head kernel above asks memory for 256-byte
// segments of 1152-byte records (the other four heads' segments are fetched by another workgroup at another time): a
// read probe of that address pattern tops out at 0.80 of HBM peak, whole records at 0.85-0.88
// (profiles/r03_int4_ablation.txt, third series).  Here a tile's K records (16 x 1152 B) and V records are consumed
// inside ONE workgroup: every 128-byte line is requested once, a page's 1 KiB of nibbles by one DMA instruction (lines
// 1..8 of its record), its scale line together with 3 others'.
//   per wave and tile: 4 full DMAs (K and V nibbles of pages 2w, 2w+1) and one with 32 active lanes (4 scale lines) = 5
//   vector-memory operations, the same in every wave, so the vmcnt arithmetic is uniform;
//   two LDS stages of 36 KiB per workgroup, two workgroups (16 waves) per CU: the workgroups of a CU run out of phase, so
//   while one consumes a tile the other's DMAs (and its own next tile's) are in flight.  (Three stages in one workgroup
//   per CU -- DMAs of tile t+2 issued before tile t is consumed, one barrier per tile -- measured 0.64 against 0.68 for
//   the 4-head kernel: with two waves per SIMD the arithmetic of a tile, ~350 instructions per wave in one dependent
//   chain, takes longer than the tile's bytes take to arrive.  profiles/r04_int4_wg8.txt)
// LDS stage:  K rows [32][512 B] | V rows [32][512 B] | K scale lines [16][128 B] | V scale lines [16][128 B]
//   a position row = 8 heads x 64 B; its thirty-two 16-byte pieces are stored XOR-ed by (row & 15) (chosen on the SOURCE
//   side of the DMA), K scale lines by (page & 7), V scale lines by (page / 2) & 3: the operand reads of the arithmetic
//   above (ds_read_b128 of 16 rows x one piece, ds_read_b32 of rows 4 kb + j, ds_read_u16 of scales) stay conflict-free.
// STREAM form (AttendArgs::stream.n_wgs != 0; many layers of one sequence): the launch's tiles in layer-major order are cut
// into n_wgs contiguous pieces of equal length, one per workgroup, n_wgs = what is resident at once.  A workgroup streams
// through its piece -- across a layer boundary the DMA pipeline just goes on; the finished layer's partial is stored, the
// next layer's query rows are loaded -- so the launch has ONE pipeline fill per workgroup, no partial last round, and a
// layer gets ceil(n_tiles / len) + 1 partials at most instead of one per split of a fixed grid (the partial stores and the
// merge launch cost 9 % of the fixed-grid launch at 16 splits x 80 layers).  The merge derives each row's partial count
// from the same arithmetic (attend_stream_wg_of).
// The arithmetic of a tile is the one of the other kernels (score_block, softmax_tile, pv_tile): one wave = one head.
namespace {
constexpr uint32_t kW8V = 16384u, kW8Ks = 32768u, kW8Vs = 34816u, kW8Stage = 36864u;

__device__ __forceinline__ void w8_take_k(uint32_t rd, uint32_t rs, u32x4& k0, u32x4& k1, uint32_t& s0, uint32_t& s1)
{
    asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:8192\n\t"
                 "ds_read_u16 %2, %5\n\tds_read_u16 %3, %5 offset:1024\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(k0), "=&v"(k1), "=&v"(s0), "=&v"(s1) : "v"(rd), "v"(rs) : "memory");
}
__device__ __forceinline__ void w8_take_v(const uint32_t (&rd)[4], uint32_t rs, uint32_t (&w)[8], uint32_t (&sc)[8])
{
    asm volatile("ds_read_b32 %0, %16\n\tds_read_b32 %1, %17\n\tds_read_b32 %2, %18\n\tds_read_b32 %3, %19\n\t"
                 "ds_read_b32 %4, %16 offset:8192\n\tds_read_b32 %5, %17 offset:8192\n\tds_read_b32 %6, %18 offset:8192\n\tds_read_b32 %7, %19 offset:8192\n\t"
                 "ds_read_u16 %8, %20\n\tds_read_u16 %9, %20 offset:64\n\tds_read_u16 %10, %20 offset:128\n\tds_read_u16 %11, %20 offset:192\n\t"
                 "ds_read_u16 %12, %20 offset:1024\n\tds_read_u16 %13, %20 offset:1088\n\tds_read_u16 %14, %20 offset:1152\n\tds_read_u16 %15, %20 offset:1216\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(w[0]), "=&v"(w[1]), "=&v"(w[2]), "=&v"(w[3]), "=&v"(w[4]), "=&v"(w[5]), "=&v"(w[6]), "=&v"(w[7]),
                   "=&v"(sc[0]), "=&v"(sc[1]), "=&v"(sc[2]), "=&v"(sc[3]), "=&v"(sc[4]), "=&v"(sc[5]), "=&v"(sc[6]), "=&v"(sc[7])
                 : "v"(rd[0]), "v"(rd[1]), "v"(rd[2]), "v"(rd[3]), "v"(rs) : "memory");
}
} // namespace

#ifndef SPECKV_INT4_W8_WAVES
#define SPECKV_INT4_W8_WAVES 4
#endif
__device__ __forceinline__ const uint8_t* w8_uniform_ptr(const uint8_t* p)
{
    const uint64_t v = reinterpret_cast<uint64_t>(p);
    const uint32_t lo = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(v)), hi = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(v >> 32));
    return reinterpret_cast<const uint8_t*>((static_cast<uint64_t>(hi) << 32) | lo);
}
// HALVES = 2 (fixed-grid and batch launches): 16 waves, the workgroup's run of tiles cut in two, waves 8..15 take the second
// half with LDS stages of their own; at the end their (m, l, accumulators) cross over through LDS and waves 0..7 store the
// merged result -- a row that fits one workgroup needs no partials and no merge launch, and a batch of 256 sequences fills
// 256 CUs with 16 waves each.  One workgroup per CU then (144 KiB of LDS).  The stream form keeps HALVES = 1, two per CU.
// CLS (round 6): an allocation striped regularly over 2..8 pools (AttendArgs::stripe_bases; BASELINE configs[3]: page % 7).  The
// range's pages are taken by residue CLASS, as k_attend_mx4's striped form does: class c = the pages j of the range with j % n == c
// -- consecutive records of ONE run for K and of one run for V -- every class in tiles of 16, every class with the tile count of the
// largest (mx4_class_tiles); the run of tiles is class-major.  A tile is then fetched exactly as in the linear form, from a base
// that is recomputed per tile instead of advanced; rows past a class's end are masked (their fetch stays inside the run: the
// engine allocates INT4 runs with 15 records of slack), a tile past a class's end fetches the class's last one.  Attention does
// not care in which order it meets the positions.  Fixed-grid and batch launches (no stream form).
template <int HALVES, bool CLS = false>
__global__ __launch_bounds__(512 * HALVES) __attribute__((amdgpu_waves_per_eu(SPECKV_INT4_W8_WAVES, SPECKV_INT4_W8_WAVES))) void k_attend_int4_wg8(AttendArgs a)
{
    __shared__ __attribute__((aligned(16))) uint8_t lds[HALVES * 2 * kW8Stage];
    __shared__ uint64_t s_bases[8];
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t head = wave & 7u, half = wave >> 3;                   // wave = kv head (x half of the run)
    const uint32_t c = lane & 15u, kb = lane >> 4;
    const uint32_t n_tiles = CLS ? mx4_striped_tiles(a.n_pages, a.stripe_n) : (a.n_pages + 15u) / 16u;       // (batch form: set per sequence below)
    // ---- this workgroup's run of tiles: `count` tiles from tile `ct` of layer `cl` on, in layer-major order
    uint32_t cl, ct, count, slot = 0u, my_splits = a.n_splits;
    uint64_t part = 0u;
    const bool stream = a.stream.n_wgs != 0u;
    uint32_t tiles_in_layer = n_tiles;
    if (stream) {
        const uint64_t g0 = attend_stream_begin(blockIdx.x, a.stream.len, a.stream.rem), g1 = attend_stream_begin(blockIdx.x + 1u, a.stream.len, a.stream.rem);
        cl = static_cast<uint32_t>(g0 / n_tiles);
        ct = static_cast<uint32_t>(g0 - static_cast<uint64_t>(cl) * n_tiles);
        count = static_cast<uint32_t>(g1 - g0);
        slot = blockIdx.x - attend_stream_wg_of(static_cast<uint64_t>(cl) * n_tiles, a.stream.len, a.stream.rem);
        my_splits = 0u;                                                   // (never the direct output: a layer has several pieces)
    } else {
        const uint32_t split = a.rows_first ? blockIdx.y : blockIdx.x;
        cl = a.rows_first ? blockIdx.x : blockIdx.y;                      // batch form: the sequence index
        if (a.rows_first && cl >= a.rows_real) return;                    // (padding row: see AttendArgs::rows_real)
        if (a.seqs && a.order) cl = a.order[cl];                          // (workgroup-uniform)
        part = (static_cast<uint64_t>(cl) * 8u + head) * a.n_splits + split;
        if (a.seqs) {                                                    // workgroup-uniform: per-sequence geometry
            const AttendSeq sq = a.seqs[cl];
            if (split >= sq.n_splits) {
                if (sq.n_splits == 0u && split == 0u && a.direct_out && a.direct_per_seq == 2u)
                    attend_zero_rows(a.direct_out, a.direct_lse, a.g, static_cast<uint64_t>(cl) * 8u + head, lane);
                return;
            }
            a.lin_base = sq.lin_base;
            a.k_first = sq.k_first + static_cast<uint64_t>(a.batch_layer) * sq.layer_pages;
            a.v_first = sq.v_first + static_cast<uint64_t>(a.batch_layer) * sq.layer_pages;
            a.n_pages = sq.n_pages;
            a.tiles_per_split = sq.tiles_per_split;
            a.layer_stride = 0u;                                         // the sequence's own region: no layer offset
            my_splits = sq.n_splits;
            part = sq.part_base + static_cast<uint64_t>(head) * sq.n_splits + split;
            tiles_in_layer = (sq.n_pages + 15u) / 16u;
            if (CLS) { a.stripe_bases = sq.stripe_bases; a.stripe_n = sq.stripe_n; }
        }
        if (CLS) tiles_in_layer = mx4_striped_tiles(a.n_pages, a.stripe_n);
        ct = split * a.tiles_per_split;
        count = ct < tiles_in_layer ? min(a.tiles_per_split, tiles_in_layer - ct) : 0u;
    }
    if (CLS) {
        if (threadIdx.x < 8u) s_bases[threadIdx.x] = ldg_small<uint64_t>(reinterpret_cast<const uint8_t*>(a.stripe_bases + threadIdx.x));
        __syncthreads();
    }
    const uint32_t out_row0 = cl;                                        // (batch form: the row index is the sequence, the addresses use layer 0)
    // every wave runs `iters` iterations (the barriers are the workgroup's); this half's own tiles are the first `count` of them
    uint32_t iters = count;
    if (HALVES == 2) {
        iters = (count + 1u) / 2u;
        ct += half * iters;
        count = half ? count - iters : iters;
    }
    const float qscale = a.scale_log2e;
    float m_run = -INFINITY, l_run = 0.0f;
    f32x4 acc[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) acc[t] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    f16x8 qv[4];
    auto load_q = [&](uint32_t row_layer) {
        const uint16_t* q16 = reinterpret_cast<const uint16_t*>(a.q8) + ((static_cast<uint64_t>(row_layer) * 8u + head) * a.g + c) * 128u + kb * 32u;
#pragma unroll
        for (int st = 0; st < 4; ++st) {
            uint4 t = make_uint4(0u, 0u, 0u, 0u);
            if (c < a.g) t = *reinterpret_cast<const uint4*>(q16 + 8 * st);
            qv[st] = q_operand(t);
        }
        // the rows must have arrived before the next DMA is issued: the compiler would otherwise place its own vmcnt(0)
        // for them at their first use, inside the loop, and drain the pipeline there in every iteration
        asm volatile("" :: "v"(qv[0]), "v"(qv[1]), "v"(qv[2]), "v"(qv[3]));
    };
    if (iters != 0u) {                                                   // workgroup-uniform
        const uint32_t tile_bytes = 16u * kInt4RecBytes;
        const uint32_t addr_layer = a.seqs ? 0u : cl;
        const uint64_t layer_bytes = a.layer_stride * kInt4RecBytes;
        // issue cursor (scalar): K / V record pointers of the next tile to request, and that tile's index in its layer
        const uint8_t* kptr = a.lin_base + (a.k_first + addr_layer * a.layer_stride) * kInt4RecBytes + static_cast<uint64_t>(ct) * tile_bytes;
        const uint8_t* vptr = a.lin_base + (a.v_first + addr_layer * a.layer_stride) * kInt4RecBytes + static_cast<uint64_t>(ct) * tile_bytes;
        uint32_t itile = ct;
        // class form: pages per class (class c holds jq + (c < jr) of the range's pages), tiles per class, the next tile to request as
        // (class, tile of the class) and the tile the arithmetic is at
        const uint32_t cls_n = CLS ? (a.stripe_n ? a.stripe_n : 1u) : 1u, cls_m = CLS ? mx4_class_tiles(a.n_pages, cls_n) : 1u;
        const uint32_t jq = a.n_pages / cls_n, jr = a.n_pages - jq * cls_n;
        uint32_t kpage0 = static_cast<uint32_t>(a.k_first + addr_layer * a.layer_stride), vpage0 = static_cast<uint32_t>(a.v_first + addr_layer * a.layer_stride);      // (the layer the requests are at)
        uint32_t iq_cls = CLS ? ct / cls_m : 0u, iq_m = CLS ? ct - iq_cls * cls_m : 0u;
        uint32_t cc_cls = iq_cls, cc_m = iq_m;
        // what the class the requests are in resolves to -- the K / V records of its first page, its page count -- looked up when the
        // requests ENTER a class, not per tile (a division, an LDS read and the wait behind it in front of every request cost the
        // class forms 3-4 points of the roofline: attend_mx4.hip)
        const uint8_t* cls_k0 = nullptr; const uint8_t* cls_v0 = nullptr;
        uint32_t cls_last_tile = 0u;
        auto cls_enter = [&]() {
            uint32_t cls = min(iq_cls, cls_n - 1u);
            uint32_t cnt = jq + (cls < jr ? 1u : 0u);
            if (cnt == 0u) { cls = 0u; cnt = 1u; }                        // (fewer pages than runs: an empty class fetches the range's first record, all masked)
            const uint32_t pk = kpage0 + cls, rk = pk / cls_n, pv = vpage0 + cls, rv = pv / cls_n;
            cls_k0 = w8_uniform_ptr(reinterpret_cast<const uint8_t*>(s_bases[pk - rk * cls_n]) + static_cast<uint64_t>(rk) * kInt4RecBytes);
            cls_v0 = w8_uniform_ptr(reinterpret_cast<const uint8_t*>(s_bases[pv - rv * cls_n]) + static_cast<uint64_t>(rv) * kInt4RecBytes);
            cls_last_tile = __builtin_amdgcn_readfirstlane((cnt - 1u) >> 4);
        };
        if (CLS) cls_enter();
        // ---- this wave's share of the fetch.  Nibbles: pages 2w and 2w+1 of K and of V, one instruction per page: lanes
        // 0..31 -> the page's slot 0 (row 4w or 4w+2), lanes 32..63 -> slot 1; LDS piece y of row r holds source piece y ^ (r & 15)
        const uint32_t pslot = lane >> 5, y = lane & 31u;
        const uint32_t ra = 4u * head + pslot, rb = ra + 2u;
        const uint32_t ga = (2u * head) * kInt4RecBytes + 128u + pslot * 512u + ((y ^ (ra & 15u)) * 16u);
        const uint32_t gb = (2u * head + 1u) * kInt4RecBytes + 128u + pslot * 512u + ((y ^ (rb & 15u)) * 16u);
        // scale lines (lanes 0..31 only): waves 0..3 the K lines of pages 4w .. 4w+3, waves 4..7 the V lines of pages 4(w-4) ..
        const uint32_t spage = 4u * (head & 3u) + ((lane & 31u) >> 3);
        const uint32_t gs = spage * kInt4RecBytes + (((lane & 7u) ^ ((head < 4u) ? (spage & 7u) : ((spage >> 1) & 3u))) * 16u);
        const uint32_t lbase = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(reinterpret_cast<uintptr_t>(&lds[0]))) + half * 2u * kW8Stage;
        const uint32_t dn = 4u * head * 512u, dsl = ((head < 4u) ? kW8Ks : kW8Vs) + 4u * (head & 3u) * 128u;    // destinations inside a stage
        auto issue = [&](uint32_t stage_off) {
            const uint32_t dst = lbase + stage_off;
            if (CLS) {                                                   // the tile's bases from its (class, tile of the class); wave-uniform
                const uint64_t toff = static_cast<uint64_t>(min(iq_m, cls_last_tile)) * tile_bytes;      // (a tile past the class's end: its last one again, all masked)
                kptr = w8_uniform_ptr(cls_k0 + toff);
                vptr = w8_uniform_ptr(cls_v0 + toff);
                if (++iq_m == cls_m) {
                    iq_m = 0u;
                    if (++iq_cls == cls_n && stream) {                    // (stream form: on into the next layer's regions)
                        iq_cls = 0u;
                        kpage0 += static_cast<uint32_t>(a.layer_stride); vpage0 += static_cast<uint32_t>(a.layer_stride);
                    }
                    cls_enter();
                }
            }
            dma16(dst + dn, kptr, ga);
            dma16(dst + dn + 1024u, kptr, gb);
            dma16(dst + kW8V + dn, vptr, ga);
            dma16(dst + kW8V + dn + 1024u, vptr, gb);
            if (lane < 32u) dma16(dst + dsl, (head < 4u) ? kptr : vptr, gs);
            if (CLS) return;
            kptr += tile_bytes; vptr += tile_bytes;
            if (++itile == tiles_in_layer) {                             // (stream form: on into the next layer's regions)
                itile = 0u;
                kptr += layer_bytes - static_cast<uint64_t>(tiles_in_layer) * tile_bytes;
                vptr += layer_bytes - static_cast<uint64_t>(tiles_in_layer) * tile_bytes;
            }
        };
        // ---- reader addresses inside stage 0 (see the 4-head kernel: the same maps with 512-byte rows and heads 0..7)
        const uint32_t h1 = head >> 1;
        const uint32_t rdk = lbase + c * 512u + (((head * 4u + kb) ^ c) * 16u);                              // row 16 b + c: + 8192 b
        const uint32_t rsk = lbase + kW8Ks + (c >> 1) * 128u + ((((c & 1u) * 4u + h1) ^ (c >> 1)) * 16u) + (head & 1u) * 8u + kb * 2u;
        uint32_t rdv[4];
#pragma unroll
        for (uint32_t j = 0; j < 4; ++j)
            rdv[j] = lbase + kW8V + (4u * kb + j) * 512u + (((head * 4u + (c >> 2)) ^ (4u * kb + j)) * 16u) + (c & 3u) * 4u;
        const uint32_t rsv = lbase + kW8Vs + 2u * kb * 128u + ((h1 ^ kb) * 16u) + (head & 1u) * 8u + 2u * (c >> 2);
        const bool ragged = (a.n_pages & 15u) != 0u;
        // The first tile's requests go out BEFORE the query rows are asked for, the second tile's right behind them: descriptor ->
        // {tile 0, query, tile 1} -> first scores is two round trips where descriptor -> query -> tiles -> scores was three (a
        // launch of 256 x 1k spends 70 us, its fixed part 8).  The query loads are inline assembly like the DMAs, so that the
        // compiler neither counts them nor waits for them by itself; ONE wait statement follows, with the query registers as its
        // operands (nothing that reads them can be placed in front of it): at most the five requests of the second tile stay in
        // flight behind it.  A run of a single tile requests that tile twice (the second copy is never read) so that the same
        // count holds; a half without tiles asks for nothing.
        if (count != 0u) {
            const uint8_t* const k0p = kptr;
            const uint8_t* const v0p = vptr;
            const uint32_t t0i = itile;
            issue(0u);
            u32x4 tq0, tq1, tq2, tq3;
            const uint32_t cq = c < a.g ? c : a.g - 1u;                   // (lanes behind the last query row: any row, zeroed below)
            const uint16_t* q16 = reinterpret_cast<const uint16_t*>(a.q8) + ((static_cast<uint64_t>(cl) * 8u + head) * a.g + cq) * 128u + kb * 32u;
            asm volatile("global_load_dwordx4 %0, %4, off\n\tglobal_load_dwordx4 %1, %4, off offset:16\n\t"
                         "global_load_dwordx4 %2, %4, off offset:32\n\tglobal_load_dwordx4 %3, %4, off offset:48"
                         : "=&v"(tq0), "=&v"(tq1), "=&v"(tq2), "=&v"(tq3) : "v"(q16) : "memory");
            if (count == 1u) { kptr = k0p; vptr = v0p; itile = t0i; if (CLS) { iq_cls = cc_cls; iq_m = cc_m; } }
            issue(kW8Stage);
            asm volatile("s_waitcnt vmcnt(5)" : "+v"(tq0), "+v"(tq1), "+v"(tq2), "+v"(tq3) :: "memory");
            const bool qlive = c < a.g;
            const u32x4 z = {0u, 0u, 0u, 0u};
            tq0 = qlive ? tq0 : z; tq1 = qlive ? tq1 : z; tq2 = qlive ? tq2 : z; tq3 = qlive ? tq3 : z;
            qv[0] = q_operand(make_uint4(tq0.x, tq0.y, tq0.z, tq0.w)); qv[1] = q_operand(make_uint4(tq1.x, tq1.y, tq1.z, tq1.w));
            qv[2] = q_operand(make_uint4(tq2.x, tq2.y, tq2.z, tq2.w)); qv[3] = q_operand(make_uint4(tq3.x, tq3.y, tq3.z, tq3.w));
        }
#pragma unroll 1
        for (uint32_t i = 0; i < iters; ++i) {
            const uint32_t bo = (i & 1u) * kW8Stage;
            // this wave's 5 DMAs of the tile have landed (the 5 of the next tile may still be in flight); then everybody's
            if (i + 1u >= count) asm volatile("s_waitcnt vmcnt(0)" SPECKV_WG_BARRIER ::: "memory");
            else                 asm volatile("s_waitcnt vmcnt(5)" SPECKV_WG_BARRIER ::: "memory");
            if (HALVES == 2 && i >= count) {                              // the shorter half's spare iteration: the barriers only
#ifndef SPECKV_ABL_NO_BARRIER
                asm volatile("s_barrier" ::: "memory");
#endif
                continue;
            }
            u32x4 k0, k1;
            uint32_t ks0, ks1;
#ifdef SPECKV_ABL_NO_LDSREAD
            k0 = u32x4{i, lane, 3u, 4u}; k1 = u32x4{lane, i, 7u, 8u}; ks0 = 0x3C00u; ks1 = 0x3C00u;
#else
            w8_take_k(rdk + bo, rsk + bo, k0, k1, ks0, ks1);
#endif
            float sc[8];
            {
                const f32x4 s0 = score_block(make_uint4(k0.x, k0.y, k0.z, k0.w), ks0, qv);
                const f32x4 s1 = score_block(make_uint4(k1.x, k1.y, k1.z, k1.w), ks1, qv);
#pragma unroll
                for (int j = 0; j < 4; ++j) { sc[j] = s0[j]; sc[4 + j] = s1[j]; }
            }
            if (CLS) {
                const uint32_t cnt = cc_cls < cls_n ? jq + (cc_cls < jr ? 1u : 0u) : 0u;
                if (16u * (cc_m + 1u) > cnt) {                            // workgroup-uniform: the class ends inside (or in front of) this tile
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const uint32_t pi = 16u * cc_m + ((j >> 1) < 2 ? 2u * kb + (j >> 1) : 8u + 2u * kb + ((j >> 1) - 2));
                        if (pi >= cnt) sc[j] = -INFINITY;
                    }
                }
                if (++cc_m == cls_m) { cc_m = 0u; if (++cc_cls == cls_n && stream) cc_cls = 0u; }
            } else if (ragged && ct + 1u == tiles_in_layer) {             // workgroup-uniform: positions beyond the range
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const uint32_t pg = ct * 16u + ((j >> 1) < 2 ? 2u * kb + (j >> 1) : 8u + 2u * kb + ((j >> 1) - 2));
                    if (pg >= a.n_pages) sc[j] = -INFINITY;
                }
            }
            const f16x8 P = softmax_tile(sc, qscale, m_run, l_run, acc);
            uint32_t vw[8], vs16[8];
            const uint32_t rdvb[4] = {rdv[0] + bo, rdv[1] + bo, rdv[2] + bo, rdv[3] + bo};
#ifdef SPECKV_ABL_NO_LDSREAD
#pragma unroll
            for (int j = 0; j < 8; ++j) { vw[j] = i + j + lane; vs16[j] = 0x3C00u; }
            (void)rdvb;
#else
            w8_take_v(rdvb, rsv + bo, vw, vs16);
#endif
#ifndef SPECKV_ABL_NO_BARRIER
            asm volatile("s_barrier" ::: "memory");                       // every wave has taken what it needs from this stage
#endif
            if (i + 2u < count) issue(bo);
            pv_tile(vw, vs16, P, acc);
            if (++ct == tiles_in_layer && i + 1u < count) {               // stream form: the layer is finished, the piece goes on
                store_partial(a, (static_cast<uint64_t>(cl) * 8u + head) * a.stream.max_slots + slot, static_cast<uint64_t>(cl) * 8u + head, 0u, c, kb, m_run, l_run, acc);
                ct = 0u; ++cl; slot = 0u;                                 // (this workgroup is the first of the next layer)
                m_run = -INFINITY; l_run = 0.0f;
#pragma unroll
                for (int t = 0; t < 8; ++t) acc[t] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
                load_q(cl);
            }
        }
    }
    if (stream) part = (static_cast<uint64_t>(cl) * 8u + head) * a.stream.max_slots + slot;
    if (HALVES == 2) {
        // waves 8..15 hand their state to waves 0..7 of the same head: [head][8 accumulators x 64 lanes x 16 B | 64 lanes x 8 B]
        // in the second half's stages (every DMA has landed and been consumed: nobody reads them any more)
        const uint32_t xb = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(reinterpret_cast<uintptr_t>(&lds[0]))) + 2u * kW8Stage + head * 8704u + lane * 16u;
        const uint32_t xm = xb - lane * 16u + 8192u + lane * 8u;
        if (half == 1u) {
#pragma unroll
            for (int t = 0; t < 8; ++t) asm volatile("ds_write_b128 %0, %1 offset:%2" :: "v"(xb), "v"(acc[t]), "n"(t * 1024) : "memory");
            asm volatile("ds_write_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" :: "v"(xm), "v"(make_float2(m_run, l_run)) : "memory");
        }
        __syncthreads();
        if (half == 1u) return;
        f32x4 o[8];
        float2 ml;
#pragma unroll
        for (int t = 0; t < 8; ++t) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(o[t]) : "v"(xb), "n"(t * 1024) : "memory");
        asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(ml) : "v"(xm) : "memory");
        asm volatile("" : "+v"(o[0]), "+v"(o[1]), "+v"(o[2]), "+v"(o[3]), "+v"(o[4]), "+v"(o[5]), "+v"(o[6]), "+v"(o[7]));     // (used behind the wait)
        const float m_new = fmaxf(m_run, ml.x), m_use = (m_new == -INFINITY) ? 0.0f : m_new;
        const float f0 = __builtin_amdgcn_exp2f(m_run - m_use), f1 = __builtin_amdgcn_exp2f(ml.x - m_use);      // 0 for an empty half
        m_run = m_new;
        l_run = l_run * f0 + ml.y * f1;
#pragma unroll
        for (int t = 0; t < 8; ++t) acc[t] = acc[t] * f0 + o[t] * f1;
    }
    store_partial(a, part, static_cast<uint64_t>(stream ? cl : out_row0) * 8u + head, my_splits, c, kb, m_run, l_run, acc);
}

hipError_t launch_attend_int4(const AttendArgs& a_in, uint32_t n_layers, hipStream_t s)
{
    AttendArgs a = a_in;
    if ((a.n_pages == 0 && !a.seqs) || n_layers == 0 || a.n_splits == 0) return hipSuccess;   // batch form: geometry per sequence
    const bool w8 = a.wg8 && a.heads == 8u && !a.table_form && (a.lin_base || a.stripe_bases);
    a.rows_real = w8 ? n_layers : n_layers * (a.heads / 4u);            // (rows-first grids are padded to an odd number of rows: AttendArgs::rows_real)
    const dim3 wg_grid = a.rows_first ? dim3((n_layers * (a.heads / 4u)) | 1u, a.n_splits) : dim3(a.n_splits, n_layers * (a.heads / 4u));
    if (a.table_form) { hipLaunchKernelGGL((k_attend_int4_wg<false, true>), wg_grid, dim3(256), 0, s, a); return hipGetLastError(); }
    if (a.wg8 && !a.lin_base && a.stripe_bases && a.heads == 8u) {       // striped regularly: the same kernel over residue classes
        const dim3 grid8 = a.stream.n_wgs ? dim3(a.stream.n_wgs) : a.rows_first ? dim3(n_layers | 1u, a.n_splits) : dim3(a.n_splits, n_layers);
        if (a.stream.n_wgs || a.wg8 == 2u) hipLaunchKernelGGL((k_attend_int4_wg8<1, true>), grid8, dim3(512), 0, s, a);
        else hipLaunchKernelGGL((k_attend_int4_wg8<2, true>), grid8, dim3(1024), 0, s, a);
        return hipGetLastError();
    }
    if (a.wg8 && a.lin_base && a.heads == 8u) {                          // whole records per workgroup: one workgroup per (layer | sequence, split)
        const dim3 grid8 = a.stream.n_wgs ? dim3(a.stream.n_wgs) : a.rows_first ? dim3(n_layers | 1u, a.n_splits) : dim3(a.n_splits, n_layers);
        if (a.stream.n_wgs || a.wg8 == 2u) hipLaunchKernelGGL(k_attend_int4_wg8<1>, grid8, dim3(512), 0, s, a);
        else hipLaunchKernelGGL(k_attend_int4_wg8<2>, grid8, dim3(1024), 0, s, a);
        return hipGetLastError();
    }
    if (a.lin_base) hipLaunchKernelGGL(k_attend_int4_wg<false>, wg_grid, dim3(256), 0, s, a);
    else if (a.stripe_bases) hipLaunchKernelGGL(k_attend_int4_wg<true>, wg_grid, dim3(256), 0, s, a);
    else return hipErrorInvalidValue;                                    // (the engine always names a form: linear, striped or table)
    return hipGetLastError();
}

} // namespace speckv
