// cxl-speckv_amd/csrc/attend_mx4.hip -- decode attention straight from MXFP4 pool records (scheme 5: OCP MX v1.0, E2M1
// elements with one E8M0 scale per 32; BASELINE configs[4] "int4/fp8 KV compression path (CDNA4 fp8 MFMA dequant), 4:1 ratio";
// no reference counterpart, parity is against oracle/orc_attend_mx4).
//
// The K half runs on the instruction the format was made for: v_mfma_scale_f32_16x16x128_f8f6f4 contracts the WHOLE head
// dimension (K = 128 = 4 blocks of 32) in one instruction and applies both operands' block scales in hardware -- K nibbles and
// E8M0 codes feed it exactly as they lie in the record, the query is quantised to MXFP8 (e4m3 + E8M0 per 32) in the prologue.
// No vector instruction touches a K element.  The V half cannot use it: p.V contracts over POSITIONS, and the instruction wants
// 32 consecutive k of one row in a lane (a nibble transpose: ds_read_b64_tr_b4 does it out of LDS, profiles/probes/mxprobe.hip,
// but the softmax weights would have to be e4m3).  V is widened to f16 by v_cvt_scalef32_pk_f16_fp4 -- ONE instruction per element
// pair, and it applies the group's E8M0 scale itself (the pair = two d of one position = one scale group) -- and meets the f16
// weights on v_mfma_f32_16x16x32_f16: the attention over the decompressed fp16 pages, as the INT4 path defines it.
//
// Operand maps as measured on the hardware (profiles/probes/mxprobe.hip, mxprobe2.hip; lane = (c = lane%16, kb = lane/16)):
//   e2m1 operand  row/col c, k = 32 kb + nibble (low nibble of byte 0 first), registers 0..3
//   e4m3 operand  row/col c, bytes 0..15 -> k = 16 kb + i, bytes 16..31 -> k = 64 + 16 kb + (i-16)
//   scale operand byte 0 of lane (c, kb) scales (row/col c, k block kb), factor 2^(code-127)
//   D             lane (c, kb) register r = D[row 4 kb + r][col c]
//
// A wave takes HPW kv heads at once (HPW = 16 / (query rows per head rounded up to 4, 8, 16)), so that all 16 columns of the
// score MFMA are live query rows -- with GQA 8 a one-head wave would run its softmax with half its lanes dead:
//   scores  S^T = K . q^T : A row i = (position i / HPW of the MFMA's 16 / HPW positions, head i % HPW)
//                           B col c = (query row c % QG, head c / QG),  QG = 16 / HPW
//           lane (c, kb) finds its own head's rows among D rows 4 kb .. 4 kb + 3: 4 / HPW position slots per MFMA, 8 per
//           32-position tile: slot j = position PPM (j / SPM) + 4 kb / HPW + j % SPM   (PPM = 16 / HPW, SPM = 4 / HPW)
//   output  O^T = V^T . P^T, one MFMA per (head a, t): A row c = d column 8 c + t of head a, k-slot j = the lane group's
//           position slot j; B = P with the columns of the other heads zeroed, so all heads of the wave accumulate into ONE
//           set of 32 accumulator registers: lane (c, kb) ends with out[query row][32 kb + 8 r + t] in acc[t][r].
// Memory shape: the 8 / HPW waves of a workgroup read whole 1088-byte records between them; a K request is 16 bytes per lane,
// HPW x 64 contiguous bytes per position; V is requested as dwords (8 nibbles = 8 d of one position).
#include "kernels.hpp"

namespace speckv {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef int v8i __attribute__((ext_vector_type(8)));

namespace {

#define MX_GP(T, p) ((const T __attribute__((address_space(1)))*)(reinterpret_cast<uintptr_t>(p)))
__device__ __forceinline__ u32x4 ldg16(const uint8_t* p) { return __builtin_nontemporal_load(MX_GP(u32x4, p)); }
__device__ __forceinline__ uint32_t ldg4(const uint8_t* p) { return __builtin_nontemporal_load(MX_GP(uint32_t, p)); }
__device__ __forceinline__ uint32_t ldg1(const uint8_t* p) { return __builtin_nontemporal_load(MX_GP(uint8_t, p)); }
// the E8M0 codes of HPW heads of one position (HPW x 4 bytes)
template <int HPW> struct Codes { uint32_t w[HPW]; };
template <int HPW> __device__ __forceinline__ Codes<HPW> ldg_codes(const uint8_t* p)
{
    Codes<HPW> r;
    if constexpr (HPW == 1) { r.w[0] = ldg4(p); }
    else if constexpr (HPW == 2) { const u32x2 v = __builtin_nontemporal_load(MX_GP(u32x2, p)); r.w[0] = v.x; r.w[1] = v.y; }
    else { const u32x4 v = ldg16(p); r.w[0] = v.x; r.w[1] = v.y; r.w[2] = v.z; r.w[3] = v.w; }
    return r;
}
__device__ __forceinline__ float max_over_kb(float v)
{
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
// an fp16 query element as the MX conversion sees it: NaN counts as 0, inf as 65504 (oracle: orc_quantize_rows_mxfp8)
__device__ __forceinline__ float q_clean(uint32_t half_bits)
{
    const float x = static_cast<float>(__builtin_bit_cast(_Float16, static_cast<uint16_t>(half_bits)));
    return (x == x) ? fminf(fmaxf(x, -65504.0f), 65504.0f) : 0.0f;
}
constexpr uint32_t kRec = kMx4RecBytes;          // 1088

} // namespace

// FORM 0: records in one run (record p at lin_base + p * 1088; never-written records are zero bytes = zeros with code 0)
// FORM 1: striped regularly over 2..8 pools (AttendArgs::stripe_bases)
// FORM 2: no regular placement: every record address from its page-table entry (never-written pages read the zero page)
template <int HPW, int FORM>
__global__ __launch_bounds__(64 * (8 / HPW)) void k_attend_mx4(AttendArgs a)
{
    constexpr int PPM = 16 / HPW;        // positions per score MFMA
    constexpr int NM = 2 * HPW;          // score MFMAs per 32-position tile
    constexpr int SPM = 4 / HPW;         // position slots of a lane per score MFMA
    constexpr int QG = 16 / HPW;         // query-row columns per head
    __shared__ uint64_t s_bases[8];
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t c = lane & 15u, kb = lane >> 4;
    const uint32_t split = blockIdx.x;
    uint32_t layer = blockIdx.y;                                         // batch form: the sequence index
    const uint32_t h0 = wave * HPW;                                      // first kv head of this wave
    const uint32_t b = c / QG, q = c % QG;                               // this lane's column: query row q of head h0 + b
    const uint32_t head = h0 + b;
    uint64_t row = static_cast<uint64_t>(layer) * a.heads + head;        // query / output row block of the lane's column
    uint64_t part = row * a.n_splits + split;
    uint32_t my_splits = a.n_splits;
    if (a.seqs) {                                                        // workgroup-uniform: per-sequence geometry
        const AttendSeq sq = a.seqs[layer];
        if (split >= sq.n_splits) {
            if (sq.n_splits == 0u && split == 0u && a.direct_out && a.direct_per_seq == 2u) {
#pragma unroll
                for (int hh = 0; hh < HPW; ++hh)
                    attend_zero_rows(a.direct_out, a.direct_lse, a.g, static_cast<uint64_t>(layer) * a.heads + h0 + hh, lane);
            }
            return;
        }
        a.lin_base = sq.lin_base;
        a.k_first = sq.k_first + static_cast<uint64_t>(a.batch_layer) * sq.layer_pages;
        a.v_first = sq.v_first + static_cast<uint64_t>(a.batch_layer) * sq.layer_pages;
        a.n_pages = sq.n_pages;
        a.tiles_per_split = sq.tiles_per_split;
        my_splits = sq.n_splits;
        part = sq.part_base + static_cast<uint64_t>(head) * sq.n_splits + split;
        if (FORM == 2) a.entries = reinterpret_cast<const PageEntry*>(sq.lin_base);
        if (FORM == 1) {
            a.stripe_bases = sq.stripe_bases;
            a.stripe_n = sq.stripe_n;
            a.stripe_magic = sq.stripe_n > 1u ? static_cast<uint32_t>((1ull << 32) / sq.stripe_n + 1u) : 0u;
        }
    } else {
        a.k_first += static_cast<uint64_t>(layer) * a.layer_stride;
        a.v_first += static_cast<uint64_t>(layer) * a.layer_stride;
    }
    if (FORM == 1) {
        if (threadIdx.x < 8u) s_bases[threadIdx.x] = a.stripe_bases[threadIdx.x];
        __syncthreads();
    }

    const uint32_t n_tiles = (a.n_pages + 15u) / 16u;
    const uint32_t t0 = split * a.tiles_per_split;
    const uint32_t t1 = min(t0 + a.tiles_per_split, n_tiles);
    float m_run = -INFINITY, l_run = 0.0f;
    f32x4 acc[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) acc[t] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};

    if (t0 < t1) {                                                       // wave-uniform
        // ---- where the lane's pieces of a tile sit, relative to the tile's first K / V record (linear form: bytes)
        const uint32_t kpos0 = c / HPW, khead = h0 + c % HPW;            // K row of score MFMA 0: position kpos0 of the tile, head khead
        const uint32_t koff = (kpos0 & 1u) * 512u + khead * 64u + kb * 16u;                     // inside the record (MFMA m: page + m * PPM / 2)
        const uint32_t ksoff = 1024u + ((kpos0 & 1u) * 8u + khead) * 4u + kb;
        const uint32_t vpos0 = 4u * kb / HPW;                            // position slot 0
        const uint32_t voff = (vpos0 & 1u) * 512u + h0 * 64u + 4u * c;  // head a: + 64 a; slot j: + slot_bytes(j)
        const uint32_t vsoff = 1024u + ((vpos0 & 1u) * 8u + h0) * 4u;
        // slot j relative to slot 0: positions dpos = PPM (j / SPM) + j % SPM further on (vpos0 is even unless HPW == 4, where dpos is)
        auto slot_page = [](int j) { return static_cast<uint32_t>((PPM * (j / SPM) + j % SPM) >> 1); };
        auto slot_half = [](int j) { return static_cast<uint32_t>((PPM * (j / SPM) + j % SPM) & 1); };
        const uint32_t kpage0 = static_cast<uint32_t>(a.k_first) + (kpos0 >> 1);               // + 16 tile + m * PPM / 2
        const uint32_t vpage0 = static_cast<uint32_t>(a.v_first) + (vpos0 >> 1);               // + 16 tile + slot_page(j)
        const uint32_t k_end = static_cast<uint32_t>(a.k_first) + a.n_pages - 1u, v_end = static_cast<uint32_t>(a.v_first) + a.n_pages - 1u;
        auto rec = [&](uint32_t page, uint32_t end) -> const uint8_t* {
            if (FORM == 0) return a.lin_base + static_cast<uint64_t>(page) * kRec;
            if (FORM == 1) return attend_stripe_rec(s_bases, page, a.stripe_n, a.stripe_magic, kRec);
            const u32x4 e = *MX_GP(u32x4, a.entries + min(page, end));                         // {address lo, hi, record bytes, scale}
            const uint8_t* r = reinterpret_cast<const uint8_t*>(static_cast<uint64_t>(e.x) | (static_cast<uint64_t>(e.y) << 32));
            return e.z >= kRec ? r : a.zero_page;
        };

        // ---- query operand: MXFP8 rows of this lane's column (query row q of head h0 + b), quantised here
        v8i QB;
        int q_code;
        {
            const bool live = q < a.g;
            const uint16_t* qrow = a.q16 + (row * a.g + min(q, a.g - 1u)) * 128u;
            // the lane's own scale block kb (d = 32 kb .. 32 kb + 31): its E8M0 code is this lane's scale operand
            float amax = 0.0f;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const u32x4 w = *MX_GP(u32x4, qrow + 32u * kb + 8u * i);
                const uint32_t ws[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
                for (int k = 0; k < 8; ++k) amax = fmaxf(amax, fabsf(q_clean((ws[k >> 1] >> (16 * (k & 1))) & 0xFFFFu)));
            }
            const uint32_t code = (live && amax > 0.0f) ? (__float_as_uint(amax) >> 23) - 8u : 0u;       // floor(log2 amax) - 8 + 127
            q_code = static_cast<int>(code);
            // the lane's data bytes belong to two other blocks: d = 16 kb .. 16 kb + 15 (block kb / 2) and 64 + 16 kb .. (block 2 + kb / 2)
            const uint32_t code_lo = __shfl(code, c + 16u * (kb >> 1)), code_hi = __shfl(code, c + 16u * (2u + (kb >> 1)));
            uint32_t qd[8];
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const float mul = __uint_as_float((254u - (half ? code_hi : code_lo)) << 23);           // 2^(127 - code), exact
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const u32x4 w = *MX_GP(u32x4, qrow + 64u * half + 16u * kb + 8u * i);
                    const uint32_t ws[4] = {w.x, w.y, w.z, w.w};
                    float v[8];
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        const float x = q_clean((ws[k >> 1] >> (16 * (k & 1))) & 0xFFFFu) * mul;
                        v[k] = live ? fminf(fmaxf(x, -448.0f), 448.0f) : 0.0f;
                    }
                    int pk = __builtin_amdgcn_cvt_pk_fp8_f32(v[0], v[1], 0, false);
                    pk = __builtin_amdgcn_cvt_pk_fp8_f32(v[2], v[3], pk, true);
                    qd[4 * half + 2 * i] = static_cast<uint32_t>(pk);
                    pk = __builtin_amdgcn_cvt_pk_fp8_f32(v[4], v[5], 0, false);
                    pk = __builtin_amdgcn_cvt_pk_fp8_f32(v[6], v[7], pk, true);
                    qd[4 * half + 2 * i + 1] = static_cast<uint32_t>(pk);
                }
            }
            QB = v8i{static_cast<int>(qd[0]), static_cast<int>(qd[1]), static_cast<int>(qd[2]), static_cast<int>(qd[3]),
                     static_cast<int>(qd[4]), static_cast<int>(qd[5]), static_cast<int>(qd[6]), static_cast<int>(qd[7])};
        }

        u32x4 kx[NM];
        uint32_t ks[NM];
        uint32_t vx[HPW][8];
        Codes<HPW> vs[8];
        auto issue_k = [&](uint32_t tile) {
#pragma unroll
            for (int m = 0; m < NM; ++m) {
                const uint8_t* r = rec(kpage0 + tile * 16u + static_cast<uint32_t>(m * PPM / 2), k_end);
                kx[m] = ldg16(r + koff);
                ks[m] = ldg1(r + ksoff);
            }
        };
        auto issue_v = [&](uint32_t tile) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if (HPW != 4 && (j & 1)) continue;                      // (slots 2i, 2i+1 share a page unless HPW == 4)
                const uint8_t* r = rec(vpage0 + tile * 16u + slot_page(j), v_end);
#pragma unroll
                for (int jj = j; jj < (HPW != 4 ? j + 2 : j + 1); ++jj) {
                    const uint32_t hb = (HPW != 4 ? slot_half(jj) : 0u) * 512u;
#pragma unroll
                    for (int hh = 0; hh < HPW; ++hh) vx[hh][jj] = ldg4(r + voff + hb + 64u * hh);
                    vs[jj] = ldg_codes<HPW>(r + vsoff + (HPW != 4 ? slot_half(jj) : 0u) * 32u);
                }
            }
        };
        __builtin_amdgcn_sched_barrier(0);
        issue_k(t0);
        __builtin_amdgcn_sched_barrier(0);
        issue_v(t0);
        __builtin_amdgcn_sched_barrier(0);
        const bool ragged = (a.n_pages & 15u) != 0u;
        const uint32_t n_pos = 2u * a.n_pages, skip_pos = 2u * a.skip_pages;
        const uint32_t gsel = 8u * (c >> 2);                             // bit offset of the lane's V scale group (d = 8c .. 8c+7 -> group c / 4) in a head's code word
#pragma unroll 1
        for (uint32_t tile = t0; tile < t1; ++tile) {
            const uint32_t nxt = (tile + 1u < t1) ? tile + 1u : tile;    // the last iteration re-requests its own tile: one basic block
            // ---- scores: one block-scaled MFMA per 16 / HPW positions, the whole head dimension at once
            float sc[8];
#pragma unroll
            for (int m = 0; m < NM; ++m) {
                const v8i A = {static_cast<int>(kx[m].x), static_cast<int>(kx[m].y), static_cast<int>(kx[m].z), static_cast<int>(kx[m].w), 0, 0, 0, 0};
                const f32x4 s = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(A, QB, f32x4{0.0f, 0.0f, 0.0f, 0.0f}, 4 /* A: e2m1 */, 0 /* B: e4m3 */,
                                                                                  0, static_cast<int>(ks[m]), 0, q_code);
#pragma unroll
                for (int u = 0; u < SPM; ++u) {
                    float v;
                    if (HPW == 1) v = s[u];
                    else if (HPW == 2) v = b ? s[2 * u + 1] : s[2 * u];
                    else v = b == 0u ? s[0] : b == 1u ? s[1] : b == 2u ? s[2] : s[3];
                    sc[m * SPM + u] = v * a.scale_log2e;
                }
            }
            if ((ragged && tile + 1u == n_tiles) || (a.skip_pages && tile == 0u)) {             // wave-uniform: positions beyond / in front of the range
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const uint32_t pos = tile * 32u + static_cast<uint32_t>(PPM * (j / SPM) + j % SPM) + vpos0;
                    if (pos >= n_pos || pos < skip_pos) sc[j] = -INFINITY;
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            issue_k(nxt);
            __builtin_amdgcn_sched_barrier(0);
            // ---- online softmax of the lane's column (its 32 positions sit in the four lanes {c, c+16, c+32, c+48})
            float mx = sc[0];
#pragma unroll
            for (int j = 1; j < 8; ++j) mx = fmaxf(mx, sc[j]);
            mx = max_over_kb(mx);
            const float m_new = fmaxf(m_run, mx);
            const float m_use = (m_new == -INFINITY) ? 0.0f : m_new;
            const float f = __builtin_amdgcn_exp2f(m_run - m_use);
            m_run = m_new;
            float psum = 0.0f;
            f16x8 P;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float p = __builtin_amdgcn_exp2f(sc[j] - m_use);
                psum += p;
                P[j] = static_cast<_Float16>(p);
            }
            l_run = l_run * f + psum;
            // ---- out^T += V^T . P^T: V widened to f16 with its group scale by the conversion instruction itself
#pragma unroll
            for (int hh = 0; hh < HPW; ++hh) {
                f16x8 Pa = P;
                if (HPW > 1) {
                    const u32x4 pw = __builtin_bit_cast(u32x4, P);
                    const bool mine = b == static_cast<uint32_t>(hh);
                    Pa = __builtin_bit_cast(f16x8, u32x4{mine ? pw.x : 0u, mine ? pw.y : 0u, mine ? pw.z : 0u, mine ? pw.w : 0u});
                }
                uint32_t lo[4][4], hi[4][4];                            // [slot pair][byte of the dword = d pair]: the pair's even / odd d, positions 2jp | 2jp+1
#pragma unroll
                for (int jp = 0; jp < 4; ++jp) {
                    // scale operand of the conversion: a float whose exponent field is the group's E8M0 code
                    const float s0 = __uint_as_float(((vs[2 * jp].w[hh] >> gsel) & 0xFFu) << 23);
                    const float s1 = __uint_as_float(((vs[2 * jp + 1].w[hh] >> gsel) & 0xFFu) << 23);
                    const uint32_t w0 = vx[hh][2 * jp], w1 = vx[hh][2 * jp + 1];
#define MX_PAIR(SEL)                                                                                                   \
    {                                                                                                                  \
        const uint32_t e0 = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_f16_fp4(w0, s0, SEL));        \
        const uint32_t e1 = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_f16_fp4(w1, s1, SEL));        \
        lo[jp][SEL] = __builtin_amdgcn_perm(e1, e0, 0x05040100u);                                                       \
        hi[jp][SEL] = __builtin_amdgcn_perm(e1, e0, 0x07060302u);                                                       \
    }
                    MX_PAIR(0) MX_PAIR(1) MX_PAIR(2) MX_PAIR(3)
#undef MX_PAIR
                }
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    const u32x4 vw = (t & 1) ? u32x4{hi[0][t >> 1], hi[1][t >> 1], hi[2][t >> 1], hi[3][t >> 1]}
                                             : u32x4{lo[0][t >> 1], lo[1][t >> 1], lo[2][t >> 1], lo[3][t >> 1]};
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, vw), Pa, hh == 0 ? acc[t] * f : acc[t], 0, 0, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            issue_v(nxt);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // ---- partial result of this split (or, single split: the final result)
    const float l_tot = sum_over_kb(l_run);
    if (a.direct_out && (!a.direct_per_seq || my_splits == 1u)) {
        if (q < a.g) {
            const float w = l_tot > 0.0f ? 1.0f / l_tot : 0.0f;
            float* dst = a.direct_out + (row * a.g + q) * 128u + 32u * kb;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                *reinterpret_cast<f32x4*>(dst + 8 * i) = f32x4{acc[0][i], acc[1][i], acc[2][i], acc[3][i]} * w;
                *reinterpret_cast<f32x4*>(dst + 8 * i + 4) = f32x4{acc[4][i], acc[5][i], acc[6][i], acc[7][i]} * w;
            }
            if (a.direct_lse && kb == 0)
                a.direct_lse[row * a.g + q] = l_tot > 0.0f ? (m_run + log2f(l_tot)) * 0.6931471805599453f : -INFINITY;
        }
        return;
    }
    if (kb == 0) {
        a.part_ml[part * 32u + q] = m_run;
        a.part_ml[part * 32u + 16u + q] = l_tot;
    }
    if (q < a.g) {
        float* dst = a.part_acc + (part * 16u + q) * 128u + 32u * kb;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            *reinterpret_cast<f32x4*>(dst + 8 * i) = f32x4{acc[0][i], acc[1][i], acc[2][i], acc[3][i]};
            *reinterpret_cast<f32x4*>(dst + 8 * i + 4) = f32x4{acc[4][i], acc[5][i], acc[6][i], acc[7][i]};
        }
    }
}

// a.lin_base set: linear form; else a.stripe_bases: striped; else a.table_form: page-table form.  Grid (splits, layers | sequences),
// workgroup = 8 / HPW waves covering the 8 kv heads.  Writes the final rows itself when a.direct_out allows it, else the split
// partials followed by launch_attend_combine.
hipError_t launch_attend_mx4(const AttendArgs& a, uint32_t n_rows, float* d_out, float* d_lse, hipStream_t s)
{
    if (n_rows == 0 || a.n_splits == 0 || a.heads != 8u) return a.heads != 8u ? hipErrorInvalidValue : hipSuccess;
    if (!a.seqs && a.n_pages == 0) return hipSuccess;
    const int form = a.lin_base ? 0 : a.stripe_bases ? 1 : a.table_form ? 2 : -1;
    if (form < 0) return hipErrorInvalidValue;
    const dim3 grid(a.n_splits, n_rows);
#define MX_LAUNCH(HPW)                                                                                                             \
    do {                                                                                                                           \
        if (form == 0) hipLaunchKernelGGL((k_attend_mx4<HPW, 0>), grid, dim3(64 * (8 / HPW)), 0, s, a);                             \
        else if (form == 1) hipLaunchKernelGGL((k_attend_mx4<HPW, 1>), grid, dim3(64 * (8 / HPW)), 0, s, a);                        \
        else hipLaunchKernelGGL((k_attend_mx4<HPW, 2>), grid, dim3(64 * (8 / HPW)), 0, s, a);                                       \
    } while (0)
    if (a.g <= 4u) MX_LAUNCH(4);
    else if (a.g <= 8u) MX_LAUNCH(2);
    else MX_LAUNCH(1);
#undef MX_LAUNCH
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    const bool all_final = a.direct_out && (!a.direct_per_seq || a.direct_per_seq == 2u);
    if (all_final) return hipSuccess;
    return launch_attend_combine(a, n_rows, d_out, d_lse, s);
}

} // namespace speckv
