// cxl-speckv_amd/csrc/attend.hip -- decode attention straight from the FP8_E4M3 pool
// records (BASELINE config 5 "fused dequant-matvec", SURVEY 8a row A22: both halves,
// q.K^T and p.V; no reference counterpart, parity is against oracle/orc_attend_fp8).
//
// No fp16 K or V is ever materialised: K bytes feed v_mfma_f32_16x16x32_fp8_fp8 as they
// lie in the record; V bytes are widened to f16 in registers (exact) and feed
// v_mfma_f32_16x16x32_f16 against the softmax weights.
//
// Work split: one wave = one kv head of one layer over one contiguous split of the
// positions, walked in tiles of 32 positions (16 pages).  In the shim layout
// [layer][kind][pos][head][128] a head's row of one position is exactly one 128-byte
// line of the FP8 record, so per-head waves still move whole lines.  The four waves of
// a workgroup take four heads of the same split (same pages, adjacent lines).
//
// MFMA operand maps (16x16x32; lane = (c = lane%16, kb = lane/16), 8 k-slots per lane):
//   scores  S^T = K . q^T : A = K  (row c = position 16b+c of the tile, k-slot = d),
//                           B = q  (column c = query row),  d = 32kb + 8*step + e
//           -> lane (c, kb) holds, for query row c, the positions 16b + 4kb + i.
//   output  O^T = V^T . P^T: B = P (column c = query row, k-slot j = position slot),
//                           A = V^T (row c = a d column, k-slot j)
//           position slot (kb, j): j < 4 -> 4kb + j, j >= 4 -> 16 + 4kb + (j-4): exactly
//           the positions whose scores the lane already holds, so P never crosses lanes
//           and the running max / sum of query row c live in the lanes that use them.
//           d column of (blk, t, row c) = 64blk + 4c + t: the lane fetches V as dwords
//           (4 consecutive d of one position) and splits them with v_perm_b32.
//           -> lane (c, kb) holds out[query row c][64blk + 16kb + 4i + t].
#include "kernels.hpp"
#include "tuning.hpp"
#include "codec_device.hpp"          // the exact reciprocal divide of the codecs (div_by_scale and friends)
#include <cstdlib>

namespace speckv {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

namespace {

__device__ __forceinline__ long pack64(uint32_t lo, uint32_t hi)
{
    return static_cast<long>(static_cast<uint64_t>(lo) | (static_cast<uint64_t>(hi) << 32));
}
// Record loads go through an explicit GLOBAL-address-space pointer.  A pointer the compiler cannot trace back to a kernel argument
// (one that is assigned under a template condition, or read from a page-table entry) otherwise becomes a FLAT load: flat loads
// count in lgkmcnt as well as vmcnt, the waits in front of the loop's scalar and LDS reads then drain every record load in
// flight, and the compiler's own vmcnt(N) bookkeeping collapses to vmcnt(0) -- the linear FP8 kernel lost 15 % that way when
// its pointers were initialised as nullptr for the striped instantiation (128 x 2k batch: 0.73 -> 0.62 of HBM peak).
#ifdef SPECKV_ABL_FLAT_LDG      // (A/B only: the loads as they were, flat wherever the pointer's origin is not visible)
#define SPECKV_GP(T, p) reinterpret_cast<const T*>(p)
#else
#define SPECKV_GP(T, p) ((const T __attribute__((address_space(1)))*)(reinterpret_cast<uintptr_t>(p)))
#endif
__device__ __forceinline__ uint4 ldg16(const uint8_t* p)
{
    typedef const u32x4 __attribute__((address_space(1)))* gp;
    (void)sizeof(gp);
    const u32x4 v = __builtin_nontemporal_load(SPECKV_GP(u32x4, p));
    return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ uint2 ldg8(const uint8_t* p)
{
    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
    const u32x2 v = __builtin_nontemporal_load(SPECKV_GP(u32x2, p));
    return make_uint2(v.x, v.y);
}
__device__ __forceinline__ uint32_t ldg4(const uint8_t* p)
{
    return __builtin_nontemporal_load(SPECKV_GP(uint32_t, p));
}
// four floats of a scale table (temporal: the table is small and every workgroup of a layer reads it), global address space as above
__device__ __forceinline__ f32x4 ldg_f4(const float* p)
{
    typedef const f32x4 __attribute__((address_space(1)))* gp;
    return *(gp)(reinterpret_cast<uintptr_t>(p));
}
// max over the four lanes {c, c+16, c+32, c+48}
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

// Registers of one 32-position tile as loaded, and the per-wave running state.
struct AttTile {
    uint4 kx[2][2];        // K: block b, bytes [32kb, 32kb+32) of row 16b + c
    uint32_t vx[2][8];     // V: d block blk (64 wide), position slot j: dword at d = 64blk + 4c
    float ks[4], vs[4];    // page scales of the lane's position slots (slot j -> page j/2)
};
struct AttState {
    float m_run, l_run;    // running max (log2 domain) and sum of query row c
    f32x4 acc[2][4];       // out[row c][64blk + 16kb + 4i + t] in acc[blk][t][i], unnormalised
};

// Phase 1 of a tile: scores of the lane's 8 position slots (log2 domain) from the K registers.
// inr[r]: slot page r lies inside the range.
__device__ __forceinline__ void att_scores(const uint4 (&kx)[2][2], const float (&ks)[4], const bool (&inr)[4],
                                           const uint32_t (&qd)[8], float qscale, float (&sc)[8])
{
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const uint32_t kd[8] = {kx[b][0].x, kx[b][0].y, kx[b][0].z, kx[b][0].w, kx[b][1].x, kx[b][1].y, kx[b][1].z, kx[b][1].w};
        f32x4 s = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int st = 0; st < 4; ++st)
            s = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(pack64(kd[2 * st], kd[2 * st + 1]), pack64(qd[2 * st], qd[2 * st + 1]), s, 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int j = 4 * b + i;
            sc[j] = inr[j >> 1] ? s[i] * ks[j >> 1] * qscale : -INFINITY;
        }
    }
}
// Phase 2: online softmax of query row c (its 32 positions sit in lanes c, c+16, c+32, c+48),
// then out^T += V^T . P^T from the V registers.
__device__ __forceinline__ void att_softmax_pv(const float (&sc)[8], const uint32_t (&vx)[2][8], const float (&vs)[4], AttState& S)
{
    float mx = sc[0];
#pragma unroll
    for (int j = 1; j < 8; ++j) mx = fmaxf(mx, sc[j]);
    mx = max_over_kb(mx);
    const float m_new = fmaxf(S.m_run, mx);
    const float m_use = (m_new == -INFINITY) ? 0.0f : m_new;              // a fully masked row stays at weight 0
    const float alpha = __builtin_amdgcn_exp2f(S.m_run - m_use);
    S.m_run = m_new;
    float vmx = fmaxf(fmaxf(vs[0], vs[1]), fmaxf(vs[2], vs[3]));
    vmx = max_over_kb(vmx);                                               // the tile's largest V page scale
    const float vinv = vmx > 0.0f ? 1.0f / vmx : 0.0f;
    float psum = 0.0f;
    f16x8 P;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float p = __builtin_amdgcn_exp2f(sc[j] - m_use);
        psum += p;
        P[j] = static_cast<_Float16>(p * (vs[j >> 1] * vinv));            // V page scale rides on the weight, <= 1
    }
    S.l_run = S.l_run * alpha + psum;
#pragma unroll
    for (int blk = 0; blk < 2; ++blk) {
        uint32_t w01[4], w23[4];
#pragma unroll
        for (int jp = 0; jp < 4; ++jp) {
            // [lo.b0, hi.b0, lo.b1, hi.b1] / [lo.b2, hi.b2, lo.b3, hi.b3]: two positions of one d column per word
            w01[jp] = __builtin_amdgcn_perm(vx[blk][2 * jp + 1], vx[blk][2 * jp], 0x05010400u);
            w23[jp] = __builtin_amdgcn_perm(vx[blk][2 * jp + 1], vx[blk][2 * jp], 0x07030602u);
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            f16x8 V;
#pragma unroll
            for (int jp = 0; jp < 4; ++jp) {
                const uint32_t w = (t < 2) ? w01[jp] : w23[jp];
                const f16x2 h = (t & 1) ? __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(w, 1.0f, true)
                                        : __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(w, 1.0f, false);
                V[2 * jp] = h.x;
                V[2 * jp + 1] = h.y;
            }
            const f32x4 o = __builtin_amdgcn_mfma_f32_16x16x32_f16(V, P, f32x4{0.0f, 0.0f, 0.0f, 0.0f}, 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 4; ++i) S.acc[blk][t][i] = S.acc[blk][t][i] * alpha + o[i] * vmx;
        }
    }
}
__device__ __forceinline__ void attend_tile(const AttTile& T, const bool (&inr)[4], const uint32_t (&qd)[8], float qscale,
                                            AttState& S)
{
    float sc[8];
    att_scores(T.kx, T.ks, inr, qd, qscale, sc);
    att_softmax_pv(sc, T.vx, T.vs, S);
}

__device__ __forceinline__ void att_init(AttState& S)
{
    S.m_run = -INFINITY;
    S.l_run = 0.0f;
#pragma unroll
    for (int blk = 0; blk < 2; ++blk)
#pragma unroll
        for (int t = 0; t < 4; ++t) S.acc[blk][t] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
}
// partial result of one split: part_acc [layers][heads][splits][16][128] (unnormalised),
// part_ml [layers][heads][splits][2][16] (running max in the log2 domain, running sum)
__device__ __forceinline__ void att_store(const AttendArgs& a, const AttState& S, uint64_t part, uint32_t c, uint32_t kb)
{
    const float l_tot = sum_over_kb(S.l_run);
    if (kb == 0) {
        a.part_ml[part * 32u + c] = S.m_run;
        a.part_ml[part * 32u + 16u + c] = l_tot;
    }
    if (c < a.g) {
        float* dst = a.part_acc + (part * 16u + c) * 128u;
#pragma unroll
        for (int blk = 0; blk < 2; ++blk)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const f32x4 o = {S.acc[blk][0][i], S.acc[blk][1][i], S.acc[blk][2][i], S.acc[blk][3][i]};
                *reinterpret_cast<f32x4*>(dst + 64 * blk + 16 * kb + 4 * i) = o;
            }
    }
}

// Query operand of the score MFMAs for lane (c, kb): 32 values of fp16 row c, d = 16kb .. 16kb+15 and 64+16kb .. 64+16kb+15
// (qsrc = row + 16 kb), quantised exactly as k_quantize_q_e4m3 does (row scale = max|q|/448, 1 if zero; e4m3 of
// clamp(q/scale)); a dead row (c >= g) is zero.  The split of d over the four kb lanes follows the K loads: lane kb takes
// bytes [16kb, 16kb+16) of BOTH 64-byte halves of a K row, so that each of the two 16-byte load instructions reads 64
// contiguous bytes per row.  (With bytes [32kb, 32kb+32) per lane each instruction read four separate 16-byte pieces per
// row: PMC showed 30 cache accesses per load instruction and the texture addresser 80 % busy at 0.70 of HBM peak.)
__device__ __forceinline__ void quantize_query_operand(const uint16_t* qsrc, bool live, uint32_t (&qd)[8], float& row_scale)
{
    float xq[32];
    float mx = 0.0f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint4 w = *reinterpret_cast<const uint4*>(qsrc + 8 * (i & 1) + 64 * (i >> 1));
        const uint32_t ws[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            xq[8 * i + k] = static_cast<float>(__builtin_bit_cast(_Float16, static_cast<uint16_t>(ws[k >> 1] >> (16 * (k & 1)))));
            mx = fmaxf(mx, fabsf(xq[8 * i + k]));
        }
    }
    mx = max_over_kb(mx);
    const float sc = (mx > 0.0f) ? (mx / 448.0f) : 1.0f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        float v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = live ? fminf(fmaxf(xq[4 * i + k] / sc, -448.0f), 448.0f) : 0.0f;
        int pk = __builtin_amdgcn_cvt_pk_fp8_f32(v[0], v[1], 0, false);
        pk = __builtin_amdgcn_cvt_pk_fp8_f32(v[2], v[3], pk, true);
        qd[i] = static_cast<uint32_t>(pk);
    }
    row_scale = live ? sc : 1.0f;
}
// The same from the four 16-byte pieces of the row already in registers (the fast kernels ask for them behind their first tile's
// requests), with the 32 quotients through the row's reciprocal: div_by_scale is bit-identical to the IEEE divide for an fp16
// dividend and a scale m / 448 (exhaustive device check: test_fast_division_is_exact); rows with inf / NaN keep the divide.
__device__ __forceinline__ void quantize_query_rows(const u32x4 (&w4)[4], bool live, uint32_t (&qd)[8], float& row_scale)
{
    float xq[32];
    float mx = 0.0f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint32_t ws[4] = {w4[i].x, w4[i].y, w4[i].z, w4[i].w};
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            xq[8 * i + k] = static_cast<float>(__builtin_bit_cast(_Float16, static_cast<uint16_t>(ws[k >> 1] >> (16 * (k & 1)))));
            mx = fmaxf(mx, fabsf(xq[8 * i + k]));
        }
    }
    mx = max_over_kb(mx);
    const bool finite = mx < INFINITY;
    const float sc = (mx > 0.0f) ? (finite ? div448_of_f16_value(mx) : mx / 448.0f) : 1.0f;
    const float rcp = finite ? rcp_of_scale(sc) : 0.0f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        float v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float x = xq[4 * i + k];
            const float qt = finite ? __builtin_copysignf(div_by_scale(x, sc, rcp), x) : x / sc;
            v[k] = live ? fminf(fmaxf(qt, -448.0f), 448.0f) : 0.0f;
        }
        int pk = __builtin_amdgcn_cvt_pk_fp8_f32(v[0], v[1], 0, false);
        pk = __builtin_amdgcn_cvt_pk_fp8_f32(v[2], v[3], pk, true);
        qd[i] = static_cast<uint32_t>(pk);
    }
    row_scale = live ? sc : 1.0f;
}

} // namespace

// General form: every page goes through its page-table entry (pool address, validity, scale), so
// records may sit anywhere (striped over pool GPUs, migrated, fragmented).  Two dependent
// memory round trips per tile; the linear form below is the fast one.
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_attend_fp8(AttendArgs a)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t c = lane & 15u, kb = lane >> 4;
    const uint32_t split = blockIdx.x;
    const uint32_t hq = a.heads / 4u;                                   // workgroups per (layer, split)
    const uint32_t layer = blockIdx.y / hq;
    const uint32_t head = (blockIdx.y % hq) * 4u + wave;
    const PageEntry* kent = a.entries + a.k_first + layer * a.layer_stride;
    const PageEntry* vent = a.entries + a.v_first + layer * a.layer_stride;
    const uint64_t row = static_cast<uint64_t>(layer) * a.heads + head;

    // query operand: bytes [32kb, 32kb+32) of e4m3 row c; its scale carries sm_scale*log2(e)
    uint32_t qd[8];
    float qscale;
    quantize_query_operand(a.q16 + (row * a.g + min(c, a.g - 1u)) * 128u + kb * 16u, c < a.g, qd, qscale);
    qscale *= a.scale_log2e;

    const uint32_t n_tiles = (a.n_pages + 15u) / 16u;
    const uint32_t t0 = split * a.tiles_per_split;
    const uint32_t t1 = min(t0 + a.tiles_per_split, n_tiles);
    AttState S;
    att_init(S);
    // page descriptors of one tile: the two K rows this lane feeds, and its four position-slot pages.  They are looked up
    // ONE TILE AHEAD (after the current tile's data loads have been issued), so a tile costs one memory round trip, not a
    // page-table entry and then the record behind it.
    struct Desc { const uint8_t* kaddr[2]; const uint8_t* vaddr[4]; float ks[4], vs[4]; };
    auto lookup = [&](uint32_t tile) {
        Desc d;
        const uint32_t pg0 = tile * 16u;
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const uint32_t pg = pg0 + 8u * b + (c >> 1);
            d.kaddr[b] = a.zero_page + kb * 16u;
            if (pg < a.n_pages) {
                const PageEntry e = kent[pg];
                if (e.rec_bytes >= kBlockElems)
                    d.kaddr[b] = reinterpret_cast<const uint8_t*>(e.pool_addr) + (c & 1u) * 1024u + head * 128u + kb * 16u;
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const uint32_t pg = pg0 + (r < 2 ? 2u * kb + r : 8u + 2u * kb + (r - 2));
            d.vaddr[r] = a.zero_page + 4u * c;
            d.ks[r] = 0.0f;
            d.vs[r] = 0.0f;
            if (pg < a.n_pages) {
                const PageEntry ke = kent[pg], ve = vent[pg];
                if (ke.rec_bytes >= kBlockElems) d.ks[r] = ke.scale;
                if (ve.rec_bytes >= kBlockElems) {
                    d.vs[r] = ve.scale;
                    d.vaddr[r] = reinterpret_cast<const uint8_t*>(ve.pool_addr) + head * 128u + 4u * c;
                }
            }
        }
        return d;
    };
    Desc cur = lookup(t0);                                               // (t0 >= t1: nothing below runs)
#pragma unroll 1
    for (uint32_t tile = t0; tile < t1; ++tile) {
        const uint32_t pg0 = tile * 16u;
        AttTile T;
        bool inr[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            inr[r] = pg0 + (r < 2 ? 2u * kb + r : 8u + 2u * kb + (r - 2)) < a.n_pages;
            T.ks[r] = cur.ks[r];
            T.vs[r] = cur.vs[r];
        }
        // ---- data: K 2 x 32 B per lane, V 16 dwords per lane
#pragma unroll
        for (int b = 0; b < 2; ++b) { T.kx[b][0] = ldg16(cur.kaddr[b]); T.kx[b][1] = ldg16(cur.kaddr[b] + 64); }
#pragma unroll
        for (int blk = 0; blk < 2; ++blk)
#pragma unroll
            for (int j = 0; j < 8; ++j) T.vx[blk][j] = ldg4(cur.vaddr[j >> 1] + (j & 1) * 1024 + 64 * blk);
        cur = lookup(tile + 1u);                                          // pages beyond the range resolve to the zero page
        attend_tile(T, inr, qd, qscale, S);
    }
    att_store(a, S, row * a.n_splits + split, c, kb);
}

// ---- linear form ------------------------------------------------------------------------------
// The allocation's records lie in one run (record p at lin_base + p*2048, the engine's default
// placement) and never-written records are zero bytes (the engine zero-fills FP8 pools), so every
// data address is arithmetic.  The page scales come from the allocation's scale_tab (kept in tile
// order by k_compress: one 16-byte load gives a lane the scales of its four slot pages); nothing
// gates the data loads any more and the call needs no helper launch: the query rows are quantised
// in the kernel prologue (each lane converts exactly the 32 values it feeds to the score MFMAs).
//
// The texture addresser, not HBM, bounds the page-table form (28 loads per tile, TA 80 % busy at
// 5.1 TB/s, profiles/): here a tile costs 15 loads -- K 4 x 16 B, V 8 x 8 B (whole 128-byte lines
// per instruction), tables 3.  V as 8-byte pieces changes the d map of the output MFMAs to
// d = 8c + t (t = 0..7), i.e. lane (c, kb) holds out[row c][32kb + 8i + t] in acc[t][i].
//
// The output MFMAs accumulate in place (C = acc).  acc is kept in units of the current tile's V
// reference scale vref_t (its largest V page scale; unchanged when that is 0), so the V page scales
// ride on the f16 weights as vs/vref_t <= 1 and the only per-tile fix-up of acc is one multiply by
// alpha * vref_{t-1}/vref_t.

// scale_tab of an allocation: the block scale of every page, stored in "tile order" -- for each aligned group of 16
// pages of a (layer, kind) region the order [kb][r] = pages 2kb, 2kb+1, 8+2kb, 9+2kb -- so that lane group kb reads the
// scales of its four slot pages with one 16-byte load.  k_compress keeps it current (CodecArgs::scale_tab); this kernel
// (re)builds it from the page table when the layout becomes known.  Never-written pages hold 0.
__global__ __launch_bounds__(256) void k_build_scale_tab(const PageEntry* __restrict__ entries, uint64_t n_pages,
                                                         uint32_t region_pages, float* __restrict__ tab, uint32_t scale_run)
{
    const uint64_t p = static_cast<uint64_t>(blockIdx.x) * 256u + threadIdx.x;
    if (p >= n_pages) return;
    const PageEntry e = entries[p];
    const uint32_t j = static_cast<uint32_t>(p % region_pages) & 15u;
    const float sc = e.rec_bytes >= kBlockElems ? e.scale : 0.0f;
    tab[p - j + attend_tile_slot(j)] = sc;
    if (scale_run) tab[scale_run_index(p, scale_run)] = sc;             // (the run-order part in front of the table: CodecArgs::scale_run)
}

// STRIPED: the same loop for an allocation striped regularly over several pools (AttendArgs::stripe_bases): the six record
// addresses a lane needs per tile (2 K pages, 4 V pages) are recomputed from the page number -- one multiply-high, one
// LDS read of the run base, one 64-bit multiply-add each -- instead of advancing one pointer.  Single-sequence form only.
// TABLE: the same loop for an allocation without a regular placement (AttendArgs::table_form): the six record addresses come
// from the page-table entries, looked up ONE REQUEST AHEAD (a request is then one round trip, not an entry and the record
// behind it); never-written pages read the zero page.  Single-sequence form only.
// CLS (round 6): regular striping taken by residue class (see k_attend_fp8_dma<2>): the pages j = c, c + D, ... of the range are
// consecutive records of one run, so the lane's pointers advance through a class exactly as through a linear region and are set
// anew only where the requests enter the next class; the four page scales of a lane group come from table entries D pages apart.
template <bool STRIPED, bool TABLE = false, bool CLS = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(TABLE ? 3 : 4, TABLE ? 3 : 4))) void k_attend_fp8_linear(AttendArgs a)
{
    __shared__ uint64_t s_bases[8];
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t c = lane & 15u, kb = lane >> 4;
    const uint32_t split = a.rows_first ? blockIdx.y : blockIdx.x, by = a.rows_first ? blockIdx.x : blockIdx.y;     // (rows_first: batches of members of different lengths)
    if (a.rows_first && by >= a.rows_real) return;                      // (padding row: see AttendArgs::rows_real)
    const uint32_t hq = a.heads / 4u;
    uint32_t layer = by / hq;                                            // batch form: the sequence index
    if (a.seqs && a.order) layer = a.order[layer];                       // (workgroup-uniform)
    const uint32_t head = (by % hq) * 4u + wave;
    const uint64_t row = static_cast<uint64_t>(layer) * a.heads + head;  // query / output row block
    uint64_t part = row * a.n_splits + split;
    uint32_t my_splits = a.n_splits;
    if (a.seqs) {                                                        // workgroup-uniform: per-sequence geometry
        const AttendSeq sq = a.seqs[layer];
        if (split >= sq.n_splits) {
            if (sq.n_splits == 0u && split == 0u && a.direct_out && a.direct_per_seq == 2u) attend_zero_rows(a.direct_out, a.direct_lse, a.g, row, lane);
            return;
        }
        a.lin_base = sq.lin_base;
        a.scale_tab = sq.scale_tab;
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
        if (STRIPED || CLS) {                                            // the sequence's own placement
            a.stripe_bases = sq.stripe_bases;
            a.stripe_n = sq.stripe_n;
            a.stripe_magic = sq.stripe_n > 1u ? static_cast<uint32_t>((1ull << 32) / sq.stripe_n + 1u) : 0u;
        }
    }
    if (STRIPED || CLS) {
        if (threadIdx.x < 8u) s_bases[threadIdx.x] = *reinterpret_cast<const uint64_t __attribute__((address_space(1)))*>(reinterpret_cast<uintptr_t>(a.stripe_bases + threadIdx.x));
        __syncthreads();
    }

    // query operand: row c of this head, d = 32kb + 8*step + e, quantised here exactly as k_quantize_q_e4m3 does
    // (scale = max|q|/448 over the row, 1 if zero; e4m3 of clamp(q/scale)); rows >= g are zero.  Called BEHIND the first tile's
    // requests: descriptor -> {tile, query} -> scores instead of descriptor -> query -> tile -> scores, and the 32 quotients through
    // the row's reciprocal (div_by_scale: bit-identical for these operands, exhaustive check in test_fast_division_is_exact)
    // instead of 32 IEEE divides -- a launch of 256 x 1k lasts 100 us and this stood in front of every workgroup's first load.
    uint32_t qd[8];
    float qscale = 1.0f;
    auto quantise_query = [&]() {
        float xq[32];
        float mx = 0.0f;
        const uint16_t* qsrc = a.q16 + (row * a.g + min(c, a.g - 1u)) * 128u + kb * 16u;     // d split: see quantize_query_operand
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint4 w = *reinterpret_cast<const uint4*>(qsrc + 8 * (i & 1) + 64 * (i >> 1));
            const uint32_t ws[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                xq[8 * i + k] = static_cast<float>(__builtin_bit_cast(_Float16, static_cast<uint16_t>(ws[k >> 1] >> (16 * (k & 1)))));
                mx = fmaxf(mx, fabsf(xq[8 * i + k]));
            }
        }
        mx = max_over_kb(mx);
        const bool finite = mx < INFINITY;                                // (inf / NaN rows: the IEEE divide, as before)
        const float sc = (mx > 0.0f) ? (finite ? div448_of_f16_value(mx) : mx / 448.0f) : 1.0f;
        const float rcp = finite ? rcp_of_scale(sc) : 0.0f;
        const bool live = c < a.g;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            float v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float x = xq[4 * i + k];
                const float qt = finite ? __builtin_copysignf(div_by_scale(x, sc, rcp), x) : x / sc;
                v[k] = live ? fminf(fmaxf(qt, -448.0f), 448.0f) : 0.0f;
            }
            int pk = __builtin_amdgcn_cvt_pk_fp8_f32(v[0], v[1], 0, false);
            pk = __builtin_amdgcn_cvt_pk_fp8_f32(v[2], v[3], pk, true);
            qd[i] = static_cast<uint32_t>(pk);
        }
        qscale = (live ? sc : 1.0f) * a.scale_log2e;
    };

    const uint32_t n_tiles = CLS ? mx4_striped_tiles(a.n_pages, a.stripe_n) : (a.n_pages + 15u) / 16u;
    const uint32_t t0 = split * a.tiles_per_split;
    const uint32_t t1 = min(t0 + a.tiles_per_split, n_tiles);
    float m_run = -INFINITY, l_run = 0.0f, vref = 1.0f;
    f32x4 acc[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) acc[t] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};

    if (t0 < t1) {                                                       // wave-uniform
        // running pointers of the tile being requested
        const uint8_t* kp = nullptr;
        const uint8_t* vp = nullptr;
        // CLS: the K and the V requests each walk (class, tile of the class); class c holds jq + (c < jr) of the range's pages
        const uint32_t cls_n = CLS ? (a.stripe_n ? a.stripe_n : 1u) : 1u, cls_m = CLS ? mx4_class_tiles(a.n_pages, cls_n) : 1u;
        const uint32_t jq = a.n_pages / cls_n, jr = a.n_pages - jq * cls_n;
        const uint32_t kfirst32 = static_cast<uint32_t>(a.k_first + layer * a.layer_stride), vfirst32 = static_cast<uint32_t>(a.v_first + layer * a.layer_stride);
        uint32_t qk_cls = CLS ? t0 / cls_m : 0u, qk_m = CLS ? t0 - qk_cls * cls_m : 0u, qv_cls = qk_cls, qv_m = qk_m, cc_cls = qk_cls, cc_m = qk_m;
        uint32_t qk_ic = 0u, qk_cnt = 1u, qv_ic = 0u, qv_cnt = 1u;
        const uint8_t* kp_c = nullptr; const uint8_t* vp_c = nullptr;      // the lane's row pointers in tile 0 of the class the requests are in
        uint32_t qk_run = 0u, qv_run = 0u;                                 // (the class's first entry in the run-order scale table: AttendArgs::scale_run)
        const uint32_t run_cap = a.scale_run >> 4;
        auto cls_enter = [&](uint32_t cls_i, uint32_t first32, uint32_t lane_off, const uint8_t*& p_c, uint32_t& ic_o, uint32_t& cnt_o, uint32_t& run_o) {
            uint32_t ic = min(cls_i, cls_n - 1u);
            uint32_t cnt = jq + (ic < jr ? 1u : 0u);
            if (cnt == 0u) { ic = 0u; cnt = 1u; }                        // (fewer pages than runs: an empty class asks for the range's first records, all masked)
            const uint32_t pg = first32 + ic, rc = pg / cls_n;             // the class's first page: record rc of run pg % D (wave-uniform)
            p_c = reinterpret_cast<const uint8_t*>(s_bases[pg - rc * cls_n]) + static_cast<uint64_t>(rc) * 2048u + lane_off;
            ic_o = ic; cnt_o = cnt; run_o = (pg - rc * cls_n) * run_cap + rc;
        };
        // the page scales of tile mm of class ic, ONE load per wave (this kernel is bound by its load instructions: 14 per tile and wave,
        // and four scalar loads per lane group in place of the linear form's one 16-byte load made it 20): lane l takes the scale of
        // slot l & 15 = the tile's page 2 kb' + r (r < 2) / 8 + 2 kb' + r - 2 -- table entries D pages apart (the table keeps every
        // aligned group of 16 pages of a region in slot order: attend_tile_slot) -- and the lane groups pick theirs across the wave
        auto cls_scale_raw = [&](uint32_t first32, uint32_t ic, uint32_t cnt, uint32_t mm, uint32_t run0) -> float {
            const uint32_t sl = lane & 15u, pgi = (sl & 2u) * 4u + (sl >> 2) * 2u + (sl & 1u);
            if (a.scale_run)                                                 // run order: the tile's 16 scales are consecutive entries
                return *reinterpret_cast<const float __attribute__((address_space(1)))*>(reinterpret_cast<uintptr_t>(a.scale_tab) - static_cast<uint64_t>(cls_n) * run_cap * 4u +
                                                                                          static_cast<uint64_t>(run0 + min(16u * mm + pgi, cnt - 1u)) * 4u);
            const uint32_t rel = ic + cls_n * min(16u * mm + pgi, cnt - 1u), j = rel & 15u;
#ifdef SPECKV_FP8_CLS_FAKE_SCALES                                          // (timing builds only, wrong results: the scales of a class tile from ONE line, as if the table were laid out by class)
            return *reinterpret_cast<const float __attribute__((address_space(1)))*>(reinterpret_cast<uintptr_t>(a.scale_tab + (first32 + 16u * (ic * cls_m + mm) % (a.n_pages & ~15u) + sl)));
#endif
            return *reinterpret_cast<const float __attribute__((address_space(1)))*>(reinterpret_cast<uintptr_t>(a.scale_tab + (first32 + rel - j + attend_tile_slot(j))));
        };
        auto cls_scales4 = [&](float raw) -> f32x4 {
            f32x4 r4;
#pragma unroll
            for (uint32_t r = 0; r < 4u; ++r) r4[r] = __shfl(raw, static_cast<int>(4u * kb + r));
            return r4;
        };
        float ks_raw = 0.0f, vs_raw = 0.0f;
        if (CLS) {
            cls_enter(qk_cls, kfirst32, head * 128u + kb * 16u + c * 1024u, kp_c, qk_ic, qk_cnt, qk_run);
            cls_enter(qv_cls, vfirst32, head * 128u + 8u * c + 4u * kb * 1024u, vp_c, qv_ic, qv_cnt, qv_run);
        }
        if (!STRIPED && !TABLE && !CLS) {
            kp = a.lin_base + (a.k_first + layer * a.layer_stride) * 2048ull + head * 128u + kb * 16u
                 + (static_cast<uint64_t>(t0) * 32u + c) * 1024u;            // block b: + b*16 KiB; second half of the row: + 64
            vp = a.lin_base + (a.v_first + layer * a.layer_stride) * 2048ull + head * 128u + 8u * c
                 + (static_cast<uint64_t>(t0) * 32u + 4u * kb) * 1024u;       // slot j: + (j&3) KiB + (j>>2)*16 KiB
        }
        const float* kt = a.scale_tab + (a.k_first + layer * a.layer_stride + static_cast<uint64_t>(t0) * 16u) + 4u * kb;
        const float* vt = a.scale_tab + (a.v_first + layer * a.layer_stride + static_cast<uint64_t>(t0) * 16u) + 4u * kb;
        // striped form: first page of the lane's K rows / V slots in tile 0 of the region, and the offsets inside a record
        const uint32_t kpage0 = static_cast<uint32_t>(a.k_first + layer * a.layer_stride) + (c >> 1);        // block b: + 8b
        const uint32_t vpage0 = static_cast<uint32_t>(a.v_first + layer * a.layer_stride) + 2u * kb;         // slot pair jj: + (jj & 1) + 8 (jj >> 1)
        const uint32_t koff = (c & 1u) * 1024u + head * 128u + kb * 16u, voff = head * 128u + 8u * c;
        uint32_t tile_k = t0, tile_v = t0;                                   // tile the next request is for
        auto rec = [&](uint32_t page) { return attend_stripe_rec(s_bases, page, a.stripe_n, a.stripe_magic, 2048u); };
        // table form: record base of an absolute page of the allocation (clamped to the end of the region; a page that
        // was never written reads zeros), and the bases of the tile the NEXT request is for
        const uint32_t k_end = static_cast<uint32_t>(a.k_first + layer * a.layer_stride) + a.n_pages - 1u;
        const uint32_t v_end = static_cast<uint32_t>(a.v_first + layer * a.layer_stride) + a.n_pages - 1u;
        auto lookup = [&](uint32_t page, uint32_t end) -> const uint8_t* {
            typedef const u32x4 __attribute__((address_space(1)))* gp;
            const u32x4 e = *(gp)(reinterpret_cast<uintptr_t>(a.entries + min(page, end)));      // {address lo, hi, record bytes, scale}
            const uint8_t* r = reinterpret_cast<const uint8_t*>(static_cast<uint64_t>(e.x) | (static_cast<uint64_t>(e.y) << 32));
            return e.z >= 2048u ? r : a.zero_page;
        };
        const uint8_t* kbase[2] = {nullptr, nullptr};
        const uint8_t* vbase[4] = {nullptr, nullptr, nullptr, nullptr};
        auto lookup_k = [&](uint32_t tile) {
#pragma unroll
            for (int b = 0; b < 2; ++b) kbase[b] = lookup(kpage0 + tile * 16u + 8u * b, k_end);
        };
        auto lookup_v = [&](uint32_t tile) {
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) vbase[jj] = lookup(vpage0 + tile * 16u + (jj & 1) + 8u * (jj >> 1), v_end);
        };
        if (TABLE) { lookup_k(t0); lookup_v(t0); }

        uint4 kx[2][2];
        uint2 vx[8];
        f32x4 ks4, vs4;
        auto issue_k = [&]() {
            const uint32_t kmm = CLS ? min(qk_m, (qk_cnt - 1u) >> 4) : 0u;   // (a tile past the class's end: its last one again, all masked)
            if (CLS) kp = kp_c + static_cast<uint64_t>(kmm) * 32768u;
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const uint8_t* src = TABLE ? kbase[b] + koff : STRIPED ? rec(kpage0 + tile_k * 16u + 8u * b) + koff : kp + 16384 * b;
                kx[b][0] = ldg16(src); kx[b][1] = ldg16(src + 64);
            }
            if (CLS) ks_raw = cls_scale_raw(kfirst32, qk_ic, qk_cnt, kmm, qk_run); else ks4 = ldg_f4(kt);
            if (TABLE) lookup_k(tile_k + 1u);
        };
        auto issue_v = [&]() {
            if (STRIPED || TABLE) {
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const uint8_t* src = (TABLE ? vbase[jj] : rec(vpage0 + tile_v * 16u + (jj & 1) + 8u * (jj >> 1))) + voff;
                    vx[(jj & 1) * 2 + (jj >> 1) * 4] = ldg8(src);               // slot j = 2 (jj & 1) + 4 (jj >> 1): position slot 0 of the page
                    vx[(jj & 1) * 2 + (jj >> 1) * 4 + 1] = ldg8(src + 1024);
                }
                if (TABLE) lookup_v(tile_v + 1u);
            } else {
                const uint32_t vmm = CLS ? min(qv_m, (qv_cnt - 1u) >> 4) : 0u;
                if (CLS) vp = vp_c + static_cast<uint64_t>(vmm) * 32768u;
#pragma unroll
                for (int j = 0; j < 8; ++j) vx[j] = ldg8(vp + 1024 * (j & 3) + 16384 * (j >> 2));
                if (CLS) { vs_raw = cls_scale_raw(vfirst32, qv_ic, qv_cnt, vmm, qv_run); return; }
            }
            vs4 = ldg_f4(vt);
        };
        // same request order as in the loop (K before V), pinned, so that the wait at the loop head is
        // "everything up to K" on both paths into it
        __builtin_amdgcn_sched_barrier(0);
        issue_k();
        __builtin_amdgcn_sched_barrier(0);
        issue_v();
        __builtin_amdgcn_sched_barrier(0);
        quantise_query();                                                // (its loads are the youngest: they and the tile arrive together)
        __builtin_amdgcn_sched_barrier(0);
        const bool ragged = (a.n_pages & 15u) != 0u;
#pragma unroll 1
        for (uint32_t tile = t0; tile < t1; ++tile) {
            // ONE register set, refilled as soon as its consumer has read it: K(t+1) is requested right after
            // the score MFMAs of tile t (in flight during the softmax and p.V), V(t+1) right after p.V (in flight
            // during the next scores).  The refills are unconditional -- the last iteration re-requests its own
            // tile -- so the loop body is one basic block and the compiler's s_waitcnt vmcnt(N) counts are exact.
            const uint32_t step = (tile + 1u < t1) ? 1u : 0u;            // scalar
            if (CLS) ks4 = cls_scales4(ks_raw);
            // ---- scores
            float sc[8];
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const uint32_t kd[8] = {kx[b][0].x, kx[b][0].y, kx[b][0].z, kx[b][0].w, kx[b][1].x, kx[b][1].y, kx[b][1].z, kx[b][1].w};
                f32x4 s = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                for (int st = 0; st < 4; ++st)
                    s = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(pack64(kd[2 * st], kd[2 * st + 1]), pack64(qd[2 * st], qd[2 * st + 1]), s, 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 4; ++i) sc[4 * b + i] = s[i] * (ks4[2 * b + (i >> 1)] * qscale);
            }
            if (!CLS && ((ragged && tile + 1u == n_tiles) || (a.skip_pages && tile == 0u))) {     // wave-uniform: positions beyond / in front of the range
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const uint32_t pg = tile * 16u + ((j >> 1) < 2 ? 2u * kb + (j >> 1) : 8u + 2u * kb + ((j >> 1) - 2));
                    if (pg >= a.n_pages || pg < a.skip_pages) sc[j] = -INFINITY;
                }
            }
            if (CLS) {                                                       // pages of this tile past the end of its class
                const uint32_t cnt = jq + (cc_cls < jr ? 1u : 0u);
                const uint32_t have = cnt > 16u * cc_m ? cnt - 16u * cc_m : 0u;          // wave-uniform
                if (have < 16u) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const uint32_t pg = (j >> 1) < 2 ? 2u * kb + (j >> 1) : 8u + 2u * kb + ((j >> 1) - 2);
                        if (pg >= have) sc[j] = -INFINITY;
                    }
                }
                if (++cc_m == cls_m) { cc_m = 0u; ++cc_cls; }
            }
            __builtin_amdgcn_sched_barrier(0);
            if (CLS) {
                if (step && ++qk_m == cls_m) { qk_m = 0u; ++qk_cls; cls_enter(qk_cls, kfirst32, head * 128u + kb * 16u + c * 1024u, kp_c, qk_ic, qk_cnt, qk_run); }
            } else { kp += step * 32768u; kt += step * 16u; tile_k += step; }
            issue_k();
            __builtin_amdgcn_sched_barrier(0);
            // ---- online softmax of query row c
            float mx = sc[0];
#pragma unroll
            for (int j = 1; j < 8; ++j) mx = fmaxf(mx, sc[j]);
            mx = max_over_kb(mx);
            const float m_new = fmaxf(m_run, mx);
            const float m_use = (m_new == -INFINITY) ? 0.0f : m_new;
            const float f = __builtin_amdgcn_exp2f(m_run - m_use);
            m_run = m_new;
            if (CLS) vs4 = cls_scales4(vs_raw);
            // the tile's V reference scale (its largest V page scale) and the weights in units of it
            const float vmx = max_over_kb(fmaxf(fmaxf(vs4[0], vs4[1]), fmaxf(vs4[2], vs4[3])));
            const float vref_t = vmx > 0.0f ? vmx : vref;
            const float rinv = __builtin_amdgcn_rcpf(vref_t);
            float psum = 0.0f;
            f16x8 P;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float p = __builtin_amdgcn_exp2f(sc[j] - m_use);
                psum += p;
                P[j] = static_cast<_Float16>(p * (vs4[j >> 1] * rinv));
            }
            l_run = l_run * f + psum;
            const float fa = f * (vref * rinv);                            // acc: old max -> new max, old V reference -> new
            vref = vref_t;
            // ---- out^T += V^T . P^T, accumulated in place
            uint32_t w[4][4];                                             // [row pair][byte pair of the 8 d]
#pragma unroll
            for (int jp = 0; jp < 4; ++jp) {
                w[jp][0] = __builtin_amdgcn_perm(vx[2 * jp + 1].x, vx[2 * jp].x, 0x05010400u);
                w[jp][1] = __builtin_amdgcn_perm(vx[2 * jp + 1].x, vx[2 * jp].x, 0x07030602u);
                w[jp][2] = __builtin_amdgcn_perm(vx[2 * jp + 1].y, vx[2 * jp].y, 0x05010400u);
                w[jp][3] = __builtin_amdgcn_perm(vx[2 * jp + 1].y, vx[2 * jp].y, 0x07030602u);
            }
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                f16x8 V;
#pragma unroll
                for (int jp = 0; jp < 4; ++jp) {
                    const f16x2 h = (t & 1) ? __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(w[jp][t >> 1], 1.0f, true)
                                            : __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(w[jp][t >> 1], 1.0f, false);
                    V[2 * jp] = h.x;
                    V[2 * jp + 1] = h.y;
                }
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(V, P, acc[t] * fa, 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (CLS) {
                if (step && ++qv_m == cls_m) { qv_m = 0u; ++qv_cls; cls_enter(qv_cls, vfirst32, head * 128u + 8u * c + 4u * kb * 1024u, vp_c, qv_ic, qv_cnt, qv_run); }
            } else { vp += step * 32768u; vt += step * 16u; tile_v += step; }
            issue_v();
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // ---- partial result of this split, back in true units (or, single split: the final result)
    const float l_tot = sum_over_kb(l_run);
    if (a.direct_out && (!a.direct_per_seq || my_splits == 1u)) {
        if (c < a.g) {
            const float w = l_tot > 0.0f ? vref / l_tot : 0.0f;
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
            *reinterpret_cast<f32x4*>(dst + 8 * i) = f32x4{acc[0][i], acc[1][i], acc[2][i], acc[3][i]} * vref;
            *reinterpret_cast<f32x4*>(dst + 8 * i + 4) = f32x4{acc[4][i], acc[5][i], acc[6][i], acc[7][i]} * vref;
        }
    }
}

// ---- linear form, LDS-DMA pipeline ------------------------------------------------------------------------------------
// Same arithmetic as k_attend_fp8_linear; the tile travels global -> LDS without passing through registers
// (global_load_lds_dwordx4, 1 KiB per wave instruction), two tiles deep per wave.  PMC on the register-staged kernel:
// texture addresser 80 % busy at 0.70 of HBM peak with 14 load instructions per tile and wave (V as 8-byte pieces), waves
// waiting 83 % of their cycles, VALU 23 % busy.  A head's row of one position is exactly one 128-byte line, so a wave
// fetches whole lines by itself: per tile 4 DMA instructions for K (8 rows each), 4 for V, 1 for the 32 page scales.
//   per wave two buffers of  [K 32 rows x 128 B | V 32 rows x 128 B | 16 K scales, 16 V scales]
//   iteration t:  K(t) landed -> K operand + scales to registers -> issue K(t+2) -> scores, softmax
//                 V(t) landed -> V pieces to registers           -> issue V(t+2) -> PV
// The destination of an LDS-DMA is lane-linear, the source per lane: row r is stored in row slot r ^ ((r >> 2) & 1) with
// its eight 16-byte pieces XOR-ed by (r >> 1) & 7 -- chosen on the source side -- which makes the K reads (ds_read_b128,
// lane (c, kb): pieces kb and 4 + kb of row c) and the V reads (ds_read_b64, rows 4 kb + j, bytes 8c..8c+7) conflict-free
// under the bank rules of MI355X_MICROARCH.md.  All vector-memory and LDS traffic of the loop is inline assembly with
// explicit counters (the tail re-requests the last tile, so the counts are the same in every iteration).
namespace {
constexpr uint32_t kFdBuf = 8320u, kFdV = 4096u, kFdS = 8192u;

// a pointer that is the same in every lane, pinned to scalar registers (the LDS-DMA statements below take their base
// address as an SGPR pair; a value loaded from a per-sequence descriptor is wave-uniform, but the compiler only proves
// that while no store of the kernel could alias the descriptor)
template <typename T> __device__ __forceinline__ T* uniform_ptr(T* p)
{
    const uint64_t v = reinterpret_cast<uint64_t>(p);
    const uint32_t lo = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(v)), hi = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(v >> 32));
    return reinterpret_cast<T*>((static_cast<uint64_t>(hi) << 32) | lo);
}
__device__ __forceinline__ void fd_dma16(uint32_t lds_dst, const uint8_t* base, uint32_t voff)
{
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(lds_dst), "v"(voff), "s"(base) : "memory");
}
// the same with a full address per lane (TABLE form: every page of a tile may lie anywhere)
__device__ __forceinline__ void fd_dma16v(uint32_t lds_dst, const uint8_t* addr)
{
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(lds_dst), "v"(addr) : "memory");
}
__device__ __forceinline__ void fd_dma4(uint32_t lds_dst, const uint8_t* base, uint32_t voff)      // active lanes only
{
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %2, %3\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(lds_dst), "v"(voff), "s"(base) : "memory");
}
// all but the 13 youngest DMAs have landed (V of this tile: 4, the next tile: 9): K operand of both blocks + the scales
// (N: the TABLE form has a page-table load per tile in its queue: 14)
template <int N = 13>
__device__ __forceinline__ void fd_take_k(uint32_t rd0, uint32_t rd1, uint32_t rs, u32x4 (&k)[4], f32x4& ks, f32x4& vs)
{
    static_assert(N == 13 || N == 14, "vmcnt immediate");
    asm volatile("s_waitcnt vmcnt(%9)\n\t"
                 "ds_read_b128 %0, %6\n\tds_read_b128 %1, %7\n\tds_read_b128 %2, %6 offset:2048\n\tds_read_b128 %3, %7 offset:2048\n\t"
                 "ds_read_b128 %4, %8\n\tds_read_b128 %5, %8 offset:64\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(k[0]), "=&v"(k[1]), "=&v"(k[2]), "=&v"(k[3]), "=&v"(ks), "=&v"(vs) : "v"(rd0), "v"(rd1), "v"(rs), "n"(N) : "memory");
}
// V of this tile has landed (younger: the next tile 9, K + scales of the one after 5)
typedef uint32_t u32x2v __attribute__((ext_vector_type(2)));
template <int N = 14>
__device__ __forceinline__ void fd_take_v(const uint32_t (&rd)[4], u32x2v (&v)[8])
{
    static_assert(N == 14 || N == 16, "vmcnt immediate");
    asm volatile("s_waitcnt vmcnt(%12)\n\t"
                 "ds_read_b64 %0, %8\n\tds_read_b64 %1, %9\n\tds_read_b64 %2, %10\n\tds_read_b64 %3, %11\n\t"
                 "ds_read_b64 %4, %8 offset:2048\n\tds_read_b64 %5, %9 offset:2048\n\tds_read_b64 %6, %10 offset:2048\n\tds_read_b64 %7, %11 offset:2048\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7])
                 : "v"(rd[0]), "v"(rd[1]), "v"(rd[2]), "v"(rd[3]), "n"(N) : "memory");
}
__device__ __forceinline__ uint32_t fd_row_slot(uint32_t r) { return r ^ ((r >> 2) & 1u); }
__device__ __forceinline__ uint32_t fd_piece_xor(uint32_t r) { return (r >> 1) & 7u; }
} // namespace

#ifndef SPECKV_FP8_WG_HEADS
#define SPECKV_FP8_WG_HEADS 4
#endif
constexpr uint32_t kFdHeads = SPECKV_FP8_WG_HEADS;          // kv heads (= waves) per workgroup
// TABLE (round 6): the same pipeline for allocations whose records do NOT lie in one run -- striped regularly over several pools
// (BASELINE configs[3]: page % 7) or moved page by page.  Every page's record address comes from its page-table entry: the 32
// entries of a tile (16 K pages, 16 V pages) are fetched FOUR tiles ahead with one load per wave (two register sets in rotation),
// parked in LDS when the tile's requests are due, and read back from there by the lanes that need them: a request is an LDS-DMA
// with a full address per lane (a lane's rows of one instruction belong to four pages).  Pages stay in page order, so the scale
// table's tile order holds as it is.  The queue of an iteration is K(t+2) 5, entries(t+4) 1, V(t+2) 4: "all but the 14 youngest"
// = K(t) and entries(t+2) have landed, "all but the 16 youngest" = V(t) has.  (The register-staged k_attend_fp8_linear<STRIPED / TABLE> computed or chased an address
// per page and request: 0.67-0.69 of the roofline over a pool striped x7 against 0.78 for this pipeline on one run.)
constexpr uint32_t kFdEnt = 512u;                           // the 32 page-table entries of the tile being requested
// FORM 2, CLS (round 6, after the table form): an allocation striped REGULARLY over 2..8 runs (page p = record p / D of run p % D) needs no
// page table at all if the range's pages are taken by residue class, as k_attend_mx4 and k_attend_int4_wg8 take them: the pages
// j = c, c + D, c + 2 D ... of the range are consecutive records of ONE run for K and of one for V, so a tile of 16 of them is
// 32 KiB in one piece and is fetched exactly like the linear form's (one SGPR base + a lane offset per request: the table form's
// address per lane is what costs it 5 points of the roofline on any layout).  Attention does not care in which order it meets
// positions; the scale table keeps its page order and is read with a stride of D pages; pages past a class's end are masked
// (every class gets the tile count of the largest; FP8 runs carry 15 records of slack so that a ragged last tile stays inside).
template <int FORM>
__global__ __launch_bounds__(64 * SPECKV_FP8_WG_HEADS) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_attend_fp8_dma(AttendArgs a)
{
    constexpr bool TABLE = FORM == 1, CLS = FORM == 2;
    __shared__ __attribute__((aligned(16))) uint8_t lds[kFdHeads][2 * kFdBuf + (TABLE ? kFdEnt : 0u)];
    __shared__ uint64_t s_bases[CLS ? 8 : 1];
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t c = lane & 15u, kb = lane >> 4;
    const uint32_t split = a.rows_first ? blockIdx.y : blockIdx.x, by = a.rows_first ? blockIdx.x : blockIdx.y;
    if (a.rows_first && by >= a.rows_real) return;                      // (padding row: see AttendArgs::rows_real)
    const uint32_t hq = a.heads / kFdHeads;
    uint32_t layer = by / hq;                                            // batch form: the sequence index
    if (a.seqs && a.order) layer = a.order[layer];                       // (workgroup-uniform)
    const uint32_t head = (by % hq) * kFdHeads + wave;
    const uint64_t row = static_cast<uint64_t>(layer) * a.heads + head;  // query / output row block
    uint64_t part = row * a.n_splits + split;
    uint32_t my_splits = a.n_splits;
    if (a.seqs) {                                                        // wave-uniform: per-sequence geometry
        const AttendSeq sq = a.seqs[layer];
        if (split >= sq.n_splits) {
            if (sq.n_splits == 0u && split == 0u && a.direct_out && a.direct_per_seq == 2u) attend_zero_rows(a.direct_out, a.direct_lse, a.g, row, lane);
            return;
        }
        a.lin_base = uniform_ptr(sq.lin_base);
        a.scale_tab = uniform_ptr(sq.scale_tab);
        a.k_first = sq.k_first;
        a.v_first = sq.v_first;
        a.n_pages = sq.n_pages;
        a.tiles_per_split = sq.tiles_per_split;
        a.k_first += static_cast<uint64_t>(a.batch_layer) * sq.layer_pages;
        a.v_first += static_cast<uint64_t>(a.batch_layer) * sq.layer_pages;
        my_splits = sq.n_splits;
        part = sq.part_base + static_cast<uint64_t>(head) * sq.n_splits + split;
        layer = 0;
        if (TABLE) a.entries = reinterpret_cast<const PageEntry*>(sq.lin_base);      // (table launches: the descriptor carries the page table)
        if (CLS) { a.stripe_bases = sq.stripe_bases; a.stripe_n = sq.stripe_n; a.scale_run = 0u; }      // (descriptors carry no run-order table: gather)
    }
    if (CLS) {                                                           // (workgroup-uniform up to here: every wave gets to the barrier)
        if (threadIdx.x < 8u) s_bases[threadIdx.x] = *reinterpret_cast<const uint64_t __attribute__((address_space(1)))*>(reinterpret_cast<uintptr_t>(a.stripe_bases + threadIdx.x));
        __syncthreads();
    }
    uint32_t qd[8];
    float qscale = 1.0f;

    const uint32_t n_tiles = CLS ? mx4_striped_tiles(a.n_pages, a.stripe_n) : (a.n_pages + 15u) / 16u;
    const uint32_t t0 = split * a.tiles_per_split;
    const uint32_t t1 = min(t0 + a.tiles_per_split, n_tiles);
    float m_run = -INFINITY, l_run = 0.0f, vref = 1.0f;
    f32x4 acc[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) acc[t] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};

    if (t0 < t1) {                                                       // wave-uniform
        // tile tt: rows at + tt * 32 KiB (32 positions x 1 KiB), scales at + tt * 16 floats
        const uint8_t* kreg = a.lin_base + (a.k_first + layer * a.layer_stride) * 2048ull + head * 128u;
        const uint8_t* vreg = a.lin_base + (a.v_first + layer * a.layer_stride) * 2048ull + head * 128u;
        // scale_tab byte offset of the lane's page scale in tile 0: lanes 0..15 K pages, 16..31 V pages (tile tt: + 64 tt)
        const uint32_t gsc = static_cast<uint32_t>(((lane < 16u ? a.k_first : a.v_first) + layer * a.layer_stride + (lane & 15u)) * 4u);
        // DMA instruction i (0..3) fills row slots 8i .. 8i+7: lane l -> row slot 8i + l/8, piece slot l%8
        uint32_t g[4];
#pragma unroll
        for (uint32_t i = 0; i < 4; ++i) {
            const uint32_t r = fd_row_slot(8u * i + (lane >> 3));        // the row that lives in this row slot (involution)
            g[i] = r * 1024u + (((lane & 7u) ^ fd_piece_xor(r)) * 16u);
        }
        const uint32_t lbase = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(reinterpret_cast<uintptr_t>(&lds[wave][0])));
        const uint32_t last = t1 - 1u;
        // CLS: the request cursor (class, tile of the class) and what it last resolved to -- the K / V bases of its tile (scalar) and
        // the lane's scale-table offset; class c holds jq + (c < jr) of the range's pages
        const uint32_t cls_n = CLS ? (a.stripe_n ? a.stripe_n : 1u) : 1u, cls_m = CLS ? mx4_class_tiles(a.n_pages, cls_n) : 1u;
        const uint32_t jq = a.n_pages / cls_n, jr = a.n_pages - jq * cls_n;
        const uint32_t kpage0 = static_cast<uint32_t>(a.k_first + layer * a.layer_stride), vpage0 = static_cast<uint32_t>(a.v_first + layer * a.layer_stride);
        uint32_t iq_cls = CLS ? t0 / cls_m : 0u, iq_m = CLS ? t0 - iq_cls * cls_m : 0u;
        const uint8_t* ksrc_c = nullptr; const uint8_t* vsrc_c = nullptr;
        uint32_t gsc_c = 0u;
        // (tried: every workgroup row starting on another class, so that the launch does not walk the runs in step -- no difference)
        // what the class the requests are in resolves to -- the K / V rows of its first page, its page count -- looked up when the
        // requests ENTER a class, not per tile (two divisions, two LDS reads and the wait behind them in front of every request cost
        // the class forms 3-4 points of the roofline: attend_mx4.hip)
        const uint8_t* cls_k0 = nullptr; const uint8_t* cls_v0 = nullptr;
        uint32_t cls_ic = 0u, cls_cnt = 1u, cls_krun = 0u, cls_vrun = 0u;
        // (the allocation's scales in run order, AttendArgs::scale_run: entry p / D of run p % D, in front of the page-order table)
        const uint32_t run_cap = CLS ? a.scale_run >> 4 : 0u;
        const uint8_t* sc_base = reinterpret_cast<const uint8_t*>(a.scale_tab) - (CLS && a.scale_run ? static_cast<uint64_t>(cls_n) * run_cap * 4u : 0ull);
        auto cls_enter = [&]() {
            uint32_t ic = min(iq_cls, cls_n - 1u);
            uint32_t cnt = jq + (ic < jr ? 1u : 0u);
            if (cnt == 0u) { ic = 0u; cnt = 1u; }                            // (fewer pages than runs: an empty class asks for the range's first records, all masked)
            const uint32_t pk = kpage0 + ic, pv = vpage0 + ic;
            const uint32_t rk = pk / cls_n, rv = pv / cls_n;                      // (wave-uniform)
            cls_k0 = uniform_ptr(reinterpret_cast<const uint8_t*>(s_bases[pk - rk * cls_n]) + static_cast<uint64_t>(rk) * 2048u + head * 128u);
            cls_v0 = uniform_ptr(reinterpret_cast<const uint8_t*>(s_bases[pv - rv * cls_n]) + static_cast<uint64_t>(rv) * 2048u + head * 128u);
            cls_ic = __builtin_amdgcn_readfirstlane(ic); cls_cnt = __builtin_amdgcn_readfirstlane(cnt);
            cls_krun = __builtin_amdgcn_readfirstlane((pk - rk * cls_n) * run_cap + rk); cls_vrun = __builtin_amdgcn_readfirstlane((pv - rv * cls_n) * run_cap + rv);
        };
        if (CLS) cls_enter();
        auto cls_next = [&]() {
            const uint32_t ic = cls_ic, cnt = cls_cnt;
            const uint32_t mm = min(iq_m, (cnt - 1u) >> 4);
            ksrc_c = uniform_ptr(cls_k0 + static_cast<uint64_t>(mm) * 32768u);
            vsrc_c = uniform_ptr(cls_v0 + static_cast<uint64_t>(mm) * 32768u);
            // the lane's scale: LDS slot (lane & 15) = [kb][r] holds the tile's page 2 kb + r (r < 2) or 8 + 2 kb + r - 2 (the order the
            // readers want: attend_tile_slot); the table keeps every aligned group of 16 pages of a region in that order too
            const uint32_t sl = lane & 15u, pgi = (sl & 2u) * 4u + (sl >> 2) * 2u + (sl & 1u);
            const uint32_t rel = ic + cls_n * min(16u * mm + pgi, cnt - 1u), j = rel & 15u;       // page of the range (its first page starts a group)
            gsc_c = ((lane < 16u ? kpage0 : vpage0) + rel - j + attend_tile_slot(j)) * 4u;
            if (a.scale_run) gsc_c = ((lane < 16u ? cls_krun : cls_vrun) + min(16u * mm + pgi, cnt - 1u)) * 4u;      // (run order: 16 consecutive entries)
            if (++iq_m == cls_m) { iq_m = 0u; ++iq_cls; cls_enter(); }
        };
        auto issue_k = [&](uint32_t tt, uint32_t buf) {
            const uint32_t tc = min(tt, last);
            if (CLS && tt <= last) cls_next();                               // (behind the split's last tile: that tile again)
            const uint8_t* src = CLS ? ksrc_c : kreg + static_cast<uint64_t>(tc) * 32768u;
            const uint32_t dst = lbase + buf * kFdBuf;
#pragma unroll
            for (uint32_t i = 0; i < 4; ++i) fd_dma16(dst + 1024u * i, src, g[i]);
            // one instruction for the 32 page scales of the tile: lanes 0..15 from the K table, 16..31 from the V table
            if (lane < 32u)
                fd_dma4(dst + kFdS, CLS ? sc_base : reinterpret_cast<const uint8_t*>(a.scale_tab), CLS ? gsc_c : gsc + tc * 64u);
        };
        auto issue_v = [&](uint32_t tt, uint32_t buf) {                     // (CLS: always the tile issue_k has just resolved)
            const uint8_t* src = CLS ? vsrc_c : vreg + static_cast<uint64_t>(min(tt, last)) * 32768u;
            const uint32_t dst = lbase + buf * kFdBuf + kFdV;
#pragma unroll
            for (uint32_t i = 0; i < 4; ++i) fd_dma16(dst + 1024u * i, src, g[i]);
        };
        // ---- TABLE: entries two tiles ahead (lane l < 16: K page l of the tile, 16 .. 31: V page l - 16), the tile's DMAs one ahead
        const uint32_t kfirst = static_cast<uint32_t>(a.k_first + layer * a.layer_stride), vfirst = static_cast<uint32_t>(a.v_first + layer * a.layer_stride);
        const uint32_t last_pg = a.n_pages - 1u;
        const uint32_t ent_lds = lbase + 2u * kFdBuf;
        auto ent_fetch = [&](uint32_t tt) -> u32x4 {
            const uint32_t idx = lane & 31u, pg = min(16u * min(tt, last) + (idx & 15u), last_pg);
            const PageEntry* ep = a.entries + ((idx < 16u ? kfirst : vfirst) + pg);
            u32x4 e;
            asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(e) : "v"(ep) : "memory");
            return e;
        };
        auto ent_store = [&](const u32x4 e) {                                             // (lanes 32 .. 63 repeat lanes 0 .. 31: same bytes)
            asm volatile("ds_write_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" :: "v"(ent_lds + (lane & 31u) * 16u), "v"(e) : "memory");
        };
        // the lane's eight request addresses of a tile (K rows of its four instructions, then V) from the entries parked in LDS: one
        // LDS round trip per tile; the V addresses wait in registers for their turn
        typedef const uint8_t* addr8[8];
        auto table_addrs = [&](addr8& ad) __attribute__((always_inline)) {
            const uint8_t* eb_ptr = &lds[wave][0] + 2u * kFdBuf;
            u32x4 e[8];
            uint32_t rr[4];
#pragma unroll
            for (uint32_t i = 0; i < 4u; ++i) {
                rr[i] = fd_row_slot(8u * i + (lane >> 3));                                                      // position row of the tile: page r / 2, position r % 2
                e[i] = *reinterpret_cast<const u32x4*>(eb_ptr + (rr[i] >> 1) * 16u);                            // {address lo, hi, record bytes, scale}
                e[4 + i] = *reinterpret_cast<const u32x4*>(eb_ptr + (16u + (rr[i] >> 1)) * 16u);
            }
#pragma unroll
            for (uint32_t j = 0; j < 8u; ++j) {
                const uint32_t r = rr[j & 3u];
                const uint8_t* rec = reinterpret_cast<const uint8_t*>(static_cast<uint64_t>(e[j].x) | (static_cast<uint64_t>(e[j].y) << 32));
                rec = e[j].z >= 2048u ? rec : a.zero_page;                                                      // never written: zeros (its table scale is 0)
                ad[j] = rec + (r & 1u) * 1024u + head * 128u + (((lane & 7u) ^ fd_piece_xor(r)) * 16u);
            }
        };
        // the K (rg = 0: + the 32 page scales) or V requests of tile tt into stage sbuf
        auto issue_table = [&](const addr8& ad, uint32_t tt, uint32_t sbuf, uint32_t rg) __attribute__((always_inline)) {
            const uint32_t dst = lbase + sbuf * kFdBuf + (rg ? kFdV : 0u);
#pragma unroll
            for (uint32_t i = 0; i < 4u; ++i) fd_dma16v(__builtin_amdgcn_readfirstlane(dst + 1024u * i), ad[4u * rg + i]);
            if (rg == 0u && lane < 32u) fd_dma4(__builtin_amdgcn_readfirstlane(lbase + sbuf * kFdBuf + kFdS), reinterpret_cast<const uint8_t*>(a.scale_tab), gsc + min(tt, last) * 64u);
        };
        // reader addresses inside buffer 0
        const uint32_t kslot = fd_row_slot(c), kx_ = fd_piece_xor(c);
        const uint32_t rdk0 = lbase + kslot * 128u + ((kb ^ kx_) * 16u);
        const uint32_t rdk1 = lbase + kslot * 128u + (((4u + kb) ^ kx_) * 16u);
        const uint32_t rsc = lbase + kFdS + kb * 16u;
        uint32_t rdv[4];
#pragma unroll
        for (uint32_t j = 0; j < 4; ++j) {
            const uint32_t r = 4u * kb + j;
            rdv[j] = lbase + kFdV + fd_row_slot(r) * 128u + (((c >> 1) ^ fd_piece_xor(r)) * 16u) + (c & 1u) * 8u;
        }
        // The query row is asked for BEHIND the first tile's K requests and in front of the rest (descriptor -> {tiles, query} ->
        // scores: two round trips, not three), by inline assembly like the DMAs: the compiler neither counts these loads nor waits
        // for them.  One wait with the row's registers as operands: all but the 13 youngest requests (V of tile 0, tile 1) have
        // landed -- the count fd_take_k uses at the head of every iteration.
        u32x4 eA, eB;                                                     // TABLE: entries of the tile two ahead of an even / odd iteration
        addr8 tad;                                                        // ... and the request addresses of the tile being asked for
        if (TABLE) {
            eA = ent_fetch(t0); eB = ent_fetch(t0 + 1u);
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(eA), "+v"(eB) :: "memory");
            ent_store(eA);
            table_addrs(tad);
            issue_table(tad, t0, 0u, 0u);
        } else {
            issue_k(t0, 0u);
        }
        u32x4 qw[4];
        {
            const uint16_t* qsrc = a.q16 + (row * a.g + min(c, a.g - 1u)) * 128u + kb * 16u;
            asm volatile("global_load_dwordx4 %0, %4, off\n\tglobal_load_dwordx4 %1, %4, off offset:16\n\t"
                         "global_load_dwordx4 %2, %4, off offset:128\n\tglobal_load_dwordx4 %3, %4, off offset:144"
                         : "=&v"(qw[0]), "=&v"(qw[1]), "=&v"(qw[2]), "=&v"(qw[3]) : "v"(qsrc) : "memory");
        }
        if (TABLE) {
            issue_table(tad, t0, 0u, 1u);
            ent_store(eB);
            table_addrs(tad);
            issue_table(tad, t0 + 1u, 1u, 0u);
            issue_table(tad, t0 + 1u, 1u, 1u);
            eA = ent_fetch(t0 + 2u); eB = ent_fetch(t0 + 3u);
            // (everything in flight, once per workgroup: the loop's counts hold from the third tile on, the first two find all landed)
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(qw[0]), "+v"(qw[1]), "+v"(qw[2]), "+v"(qw[3]), "+v"(eA), "+v"(eB) :: "memory");
        } else {
            issue_v(t0, 0u);
            issue_k(t0 + 1u, 1u);
            issue_v(t0 + 1u, 1u);
            asm volatile("s_waitcnt vmcnt(13)" : "+v"(qw[0]), "+v"(qw[1]), "+v"(qw[2]), "+v"(qw[3]) :: "memory");
        }
        quantize_query_rows(qw, c < a.g, qd, qscale);
        qscale *= a.scale_log2e;
        const bool ragged = (a.n_pages & 15u) != 0u;
        uint32_t cc_cls = CLS ? t0 / cls_m : 0u, cc_m = CLS ? t0 - cc_cls * cls_m : 0u;      // CLS: (class, tile of the class) the arithmetic is at
        auto tile_step = [&](uint32_t tile, uint32_t buf, u32x4& et) __attribute__((always_inline)) {
            const uint32_t bo = buf * kFdBuf;
            u32x4 kx[4];
            f32x4 ks4, vs4;
            if (TABLE) {
                fd_take_k<14>(rdk0 + bo, rdk1 + bo, rsc + bo, kx, ks4, vs4);       // K of this tile and the entries of tile + 2 have landed
                asm volatile("" : "+v"(et));
            } else {
                fd_take_k(rdk0 + bo, rdk1 + bo, rsc + bo, kx, ks4, vs4);
                issue_k(tile + 2u, buf);
            }
            // ---- scores
            float sc[8];
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const uint32_t kd[8] = {kx[2 * b].x, kx[2 * b].y, kx[2 * b].z, kx[2 * b].w, kx[2 * b + 1].x, kx[2 * b + 1].y, kx[2 * b + 1].z, kx[2 * b + 1].w};
                f32x4 s = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                for (int st = 0; st < 4; ++st)
                    s = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(pack64(kd[2 * st], kd[2 * st + 1]), pack64(qd[2 * st], qd[2 * st + 1]), s, 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 4; ++i) sc[4 * b + i] = s[i] * (ks4[2 * b + (i >> 1)] * qscale);
            }
            if (TABLE) {
                // behind the score MFMAs (they run while the entries make their way through LDS): the entries of tile + 2 are parked,
                // its eight request addresses formed, its K requests go out; the entries of tile + 4 are asked for
                ent_store(et);
                table_addrs(tad);
                issue_table(tad, tile + 2u, buf, 0u);
                et = ent_fetch(tile + 4u);
            }
            if (CLS) {                                                       // pages of this tile past the end of its class
                const uint32_t cnt = jq + (cc_cls < jr ? 1u : 0u);
                const uint32_t have = cnt > 16u * cc_m ? cnt - 16u * cc_m : 0u;          // wave-uniform
                if (have < 16u) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const uint32_t pg = (j >> 1) < 2 ? 2u * kb + (j >> 1) : 8u + 2u * kb + ((j >> 1) - 2);
                        if (pg >= have) sc[j] = -INFINITY;
                    }
                }
                if (++cc_m == cls_m) { cc_m = 0u; ++cc_cls; }
            } else
            if ((ragged && tile + 1u == n_tiles) || (a.skip_pages && tile == 0u)) {     // wave-uniform: positions beyond / in front of the range
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const uint32_t pg = tile * 16u + ((j >> 1) < 2 ? 2u * kb + (j >> 1) : 8u + 2u * kb + ((j >> 1) - 2));
                    if (pg >= a.n_pages || pg < a.skip_pages) sc[j] = -INFINITY;
                }
            }
            // ---- online softmax of query row c
            float mx = sc[0];
#pragma unroll
            for (int j = 1; j < 8; ++j) mx = fmaxf(mx, sc[j]);
            mx = max_over_kb(mx);
            const float m_new = fmaxf(m_run, mx);
            const float m_use = (m_new == -INFINITY) ? 0.0f : m_new;
            const float f = __builtin_amdgcn_exp2f(m_run - m_use);
            m_run = m_new;
            const float vmx = max_over_kb(fmaxf(fmaxf(vs4[0], vs4[1]), fmaxf(vs4[2], vs4[3])));
            const float vref_t = vmx > 0.0f ? vmx : vref;
            const float rinv = __builtin_amdgcn_rcpf(vref_t);
            float psum = 0.0f;
            f16x8 P;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float p = __builtin_amdgcn_exp2f(sc[j] - m_use);
                psum += p;
                P[j] = static_cast<_Float16>(p * (vs4[j >> 1] * rinv));
            }
            l_run = l_run * f + psum;
            const float fa = f * (vref * rinv);                            // acc: old max -> new max, old V reference -> new
            vref = vref_t;
            // ---- V pieces, then the next-but-one tile's V is requested into the buffer they came from
            u32x2v vx[8];
            const uint32_t rdvb[4] = {rdv[0] + bo, rdv[1] + bo, rdv[2] + bo, rdv[3] + bo};
            if (TABLE) { fd_take_v<16>(rdvb, vx); issue_table(tad, tile + 2u, buf, 1u); }
            else { fd_take_v(rdvb, vx); issue_v(tile + 2u, buf); }
            // ---- out^T += V^T . P^T, accumulated in place
            uint32_t w[4][4];                                             // [row pair][byte pair of the 8 d]
#pragma unroll
            for (int jp = 0; jp < 4; ++jp) {
                w[jp][0] = __builtin_amdgcn_perm(vx[2 * jp + 1].x, vx[2 * jp].x, 0x05010400u);
                w[jp][1] = __builtin_amdgcn_perm(vx[2 * jp + 1].x, vx[2 * jp].x, 0x07030602u);
                w[jp][2] = __builtin_amdgcn_perm(vx[2 * jp + 1].y, vx[2 * jp].y, 0x05010400u);
                w[jp][3] = __builtin_amdgcn_perm(vx[2 * jp + 1].y, vx[2 * jp].y, 0x07030602u);
            }
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                f16x8 V;
#pragma unroll
                for (int jp = 0; jp < 4; ++jp) {
                    const f16x2 h = (t & 1) ? __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(w[jp][t >> 1], 1.0f, true)
                                            : __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(w[jp][t >> 1], 1.0f, false);
                    V[2 * jp] = h.x;
                    V[2 * jp + 1] = h.y;
                }
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(V, P, acc[t] * fa, 0, 0, 0);
            }
        };
        if (TABLE) {
#pragma unroll 1
            for (uint32_t tile = t0; tile < t1; tile += 2u) {             // (two steps per trip: the entry registers of even and odd tiles are fixed)
                tile_step(tile, 0u, eA);
                if (tile + 1u >= t1) break;
                tile_step(tile + 1u, 1u, eB);
            }
        } else {
#pragma unroll 1
            for (uint32_t tile = t0; tile < t1; ++tile) tile_step(tile, (tile - t0) & 1u, eA);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                  // the re-requested tail tiles: nothing may land after the wave ends
    }
    // ---- partial result of this split, back in true units (or, single split: the final result)
    const float l_tot = sum_over_kb(l_run);
    if (a.direct_out && (!a.direct_per_seq || my_splits == 1u)) {
        if (c < a.g) {
            const float w = l_tot > 0.0f ? vref / l_tot : 0.0f;
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
            *reinterpret_cast<f32x4*>(dst + 8 * i) = f32x4{acc[0][i], acc[1][i], acc[2][i], acc[3][i]} * vref;
            *reinterpret_cast<f32x4*>(dst + 8 * i + 4) = f32x4{acc[4][i], acc[5][i], acc[6][i], acc[7][i]} * vref;
        }
    }
}

// Merge of the split partials: one workgroup of 512 threads per (layer, head, query row).  The per-layer call of a
// decode step has few rows and many splits, so the merge must not walk the splits serially (a 128-split merge with
// one dependent load chain per thread took 52 us, 3.5x the attention kernel itself): every thread first takes its
// own splits for the max and the weights (kept in LDS), then four thread groups share the splits of the weighted sum.
constexpr uint32_t kMaxSplits = 2048;
__global__ __launch_bounds__(512) void k_attend_combine(const float* __restrict__ part_acc, const float* __restrict__ part_ml,
                                                        uint32_t g, uint32_t n_splits, float* __restrict__ out,
                                                        float* __restrict__ lse, const AttendSeq* __restrict__ seqs,
                                                        uint32_t heads, uint32_t skip_single, AttendArgs::Stream sk, uint32_t n_tiles)
{
    __shared__ float w[kMaxSplits];
    __shared__ float red[8];
    __shared__ float osum[4][128];
    const uint32_t rowq = blockIdx.x / g, m = blockIdx.x % g, t = threadIdx.x;     // rowq = layer*heads + head
    const uint32_t lane = t & 63u, wv = t >> 6;
    uint64_t part0 = static_cast<uint64_t>(rowq) * n_splits;           // first partial of this (layer | sequence, head)
    if (seqs) {
        const AttendSeq sq = seqs[rowq / heads];
        if (skip_single && sq.n_splits == 1u) return;        // written by the attention kernel itself (workgroup-uniform)
        n_splits = sq.n_splits;
        part0 = sq.part_base + static_cast<uint64_t>(rowq % heads) * n_splits;
    }
    if (sk.n_wgs) {                                                    // stream form: the row's own count, slots max_slots apart
        part0 = static_cast<uint64_t>(rowq) * sk.max_slots;
        n_splits = attend_stream_count(rowq / heads, n_tiles, sk.len, sk.rem);
    }
    const float* ml = part_ml + part0 * 32u;
    auto block_reduce = [&](float v, bool is_max) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { const float u = __shfl_xor(v, o); v = is_max ? fmaxf(v, u) : v + u; }
        __syncthreads();
        if (lane == 0) red[wv] = v;
        __syncthreads();
        float r = red[0];
#pragma unroll
        for (int i = 1; i < 8; ++i) r = is_max ? fmaxf(r, red[i]) : r + red[i];
        return r;
    };
    float mloc = -INFINITY;
    for (uint32_t s = t; s < n_splits; s += 512u) mloc = fmaxf(mloc, ml[s * 32u + m]);
    const float M = block_reduce(mloc, true);
    const float Mu = (M == -INFINITY) ? 0.0f : M;
    float lloc = 0.0f;
    for (uint32_t s = t; s < n_splits; s += 512u) {
        const float ws = __builtin_amdgcn_exp2f(ml[s * 32u + m] - Mu);
        w[s] = ws;
        lloc += ws * ml[s * 32u + 16u + m];
    }
    const float L = block_reduce(lloc, false);                     // (its barriers also publish w[])
    const uint32_t grp = t >> 7, d = t & 127u;
    const float* acc = part_acc + (part0 * 16u + m) * 128u + d;
    float o0 = 0.0f, o1 = 0.0f, o2 = 0.0f, o3 = 0.0f;
    uint32_t s = grp;
    for (; s + 12u < n_splits; s += 16u) {                         // four independent loads in flight per thread
        o0 += w[s] * acc[static_cast<uint64_t>(s) * 2048u];
        o1 += w[s + 4u] * acc[static_cast<uint64_t>(s + 4u) * 2048u];
        o2 += w[s + 8u] * acc[static_cast<uint64_t>(s + 8u) * 2048u];
        o3 += w[s + 12u] * acc[static_cast<uint64_t>(s + 12u) * 2048u];
    }
    for (; s < n_splits; s += 4u) o0 += w[s] * acc[static_cast<uint64_t>(s) * 2048u];
    osum[grp][d] = (o0 + o1) + (o2 + o3);
    __syncthreads();
    if (grp == 0) {
        const float o = (osum[0][d] + osum[1][d]) + (osum[2][d] + osum[3][d]);
        out[(static_cast<uint64_t>(rowq) * g + m) * 128u + d] = L > 0.0f ? o / L : 0.0f;
        if (lse && d == 0) lse[static_cast<uint64_t>(rowq) * g + m] = L > 0.0f ? (M + log2f(L)) * 0.6931471805599453f : -INFINITY;
    }
}

// The merge for rows with at most kSmallCombineSplits splits (batch launches split two or three ways, the many-layers form with
// its 8): one WAVE per (layer | sequence, head),
// lane (c, kb) = query row c, dimensions 32 kb .. 32 kb + 31 -- the layout the partials are stored in, so every access is a
// 16-byte one.  (The general kernel above spends a 512-thread workgroup per query row: 16 384 workgroups for a batch of 256
// sequences, more than the merge is worth when there are two partials to add.)
// WPR = 4 (round 6; launches of few rows -- batches of some tens of sequences cut into pieces): the four waves of a workgroup share ONE row, wave w
// takes dimensions 8 w .. 8 w + 7 of every lane's 32 (two of its eight 16-byte pieces), all splits in the same order -- bit for bit the sums of
// WPR = 1, a quarter of the dependent loads per wave and four times the waves (32 sequences x 8 splits: 8.6 -> 5.6 us; one wave per CU was
// the whole launch).
constexpr uint32_t kSmallCombineSplits = 8;
template <int WPR>
__global__ __launch_bounds__(256) void k_attend_combine_small(const float* __restrict__ part_acc, const float* __restrict__ part_ml,
                                                              uint32_t g, uint32_t n_splits, float* __restrict__ out,
                                                              float* __restrict__ lse, const AttendSeq* __restrict__ seqs,
                                                              uint32_t heads, uint32_t skip_single, uint32_t n_rows, AttendArgs::Stream sk, uint32_t n_tiles)
{
    typedef float v4f __attribute__((ext_vector_type(4)));
    const uint32_t lane = threadIdx.x & 63u, c = lane & 15u, kb = lane >> 4;
    const uint32_t rowq = WPR == 1 ? blockIdx.x * 4u + (threadIdx.x >> 6) : blockIdx.x;
    constexpr int kI0 = 0, kIn = WPR == 1 ? 8 : 2;                     // this wave's 16-byte pieces of a lane's eight: [i0, i0 + kIn)
    const int i0 = WPR == 1 ? kI0 : 2 * static_cast<int>(threadIdx.x >> 6);
    if (rowq >= n_rows) return;
    uint64_t part0 = static_cast<uint64_t>(rowq) * n_splits;
    if (seqs) {
        const AttendSeq sq = seqs[rowq / heads];
        if (skip_single && sq.n_splits == 1u) return;
        n_splits = sq.n_splits;
        part0 = sq.part_base + static_cast<uint64_t>(rowq % heads) * n_splits;
    }
    if (sk.n_wgs) {                                                    // stream form (wave-uniform)
        part0 = static_cast<uint64_t>(rowq) * sk.max_slots;
        n_splits = attend_stream_count(rowq / heads, n_tiles, sk.len, sk.rem);
    }
    const float* ml = part_ml + part0 * 32u + c;
    float m[kSmallCombineSplits], l[kSmallCombineSplits];
    float M = -INFINITY;
#pragma unroll
    for (uint32_t s = 0; s < kSmallCombineSplits; ++s) {
        m[s] = s < n_splits ? ml[s * 32u] : -INFINITY;
        l[s] = s < n_splits ? ml[s * 32u + 16u] : 0.0f;
        M = fmaxf(M, m[s]);
    }
    const float Mu = (M == -INFINITY) ? 0.0f : M;
    float L = 0.0f;
    v4f o[kIn];
#pragma unroll
    for (int i = 0; i < kIn; ++i) o[i] = v4f{0.0f, 0.0f, 0.0f, 0.0f};
    const float* src = part_acc + (part0 * 16u + c) * 128u + 32u * kb + 4 * i0;
#pragma unroll
    for (uint32_t s = 0; s < kSmallCombineSplits; ++s) {
        if (s < n_splits) {                                          // wave-uniform
            const float w = __builtin_amdgcn_exp2f(m[s] - Mu);
            L += w * l[s];
            if (c < g) {
#pragma unroll
                for (int i = 0; i < kIn; ++i) o[i] += *reinterpret_cast<const v4f*>(src + static_cast<uint64_t>(s) * 2048u + 4 * i) * w;
            }
        }
    }
    if (c < g) {
        float* dst = out + (static_cast<uint64_t>(rowq) * g + c) * 128u + 32u * kb + 4 * i0;
#pragma unroll
        for (int i = 0; i < kIn; ++i) *reinterpret_cast<v4f*>(dst + 4 * i) = L > 0.0f ? o[i] / L : v4f{0.0f, 0.0f, 0.0f, 0.0f};
        if (lse && kb == 0u && i0 == 0) lse[static_cast<uint64_t>(rowq) * g + c] = L > 0.0f ? (M + log2f(L)) * 0.6931471805599453f : -INFINITY;
    }
}

// q.K^T scores only, linear form: the score half of k_attend_fp8_linear (same operand maps, same register refill
// pipeline for K) writing out[layer][head][row][position]; V is never touched.  tiles_per_split tiles per wave.
__global__ __launch_bounds__(256) void k_qk_scores_fp8_linear(AttendArgs a, float* __restrict__ out)
{
    // per wave: one tile of scores, [16 rows][32 positions] with a 36-dword row pitch (bank spread), so that the tile
    // leaves as whole 128-byte lines: the MFMA result has 4 positions of one row per lane, a row of the tile is 128 B
    __shared__ __attribute__((aligned(16))) float stage[4][16 * 36];
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t c = lane & 15u, kb = lane >> 4;
    const uint32_t hq = a.heads / 4u;
    const uint32_t layer = blockIdx.y / hq;
    const uint32_t head = (blockIdx.y % hq) * 4u + wave;
    const uint64_t row = static_cast<uint64_t>(layer) * a.heads + head;
    uint32_t qd[8];
    float qscale;
    quantize_query_operand(a.q16 + (row * a.g + min(c, a.g - 1u)) * 128u + kb * 16u, c < a.g, qd, qscale);
    const uint32_t n_tiles = (a.n_pages + 15u) / 16u, n_pos = 2u * a.n_pages;
    const uint32_t t0 = blockIdx.x * a.tiles_per_split;
    const uint32_t t1 = min(t0 + a.tiles_per_split, n_tiles);
    if (t0 >= t1) return;
    const uint8_t* kp = a.lin_base + (a.k_first + layer * a.layer_stride) * 2048ull + head * 128u + kb * 16u
                        + (static_cast<uint64_t>(t0) * 32u + c) * 1024u;
    const float* kt = a.scale_tab + (a.k_first + layer * a.layer_stride + static_cast<uint64_t>(t0) * 16u) + 4u * kb;
    float* st = stage[wave];
    // store side: lane -> (row lane/8 [+8], 16-byte piece lane%8) of the staged tile
    float* dst = out + (row * a.g + (lane >> 3)) * n_pos + static_cast<uint64_t>(t0) * 32u + 4u * (lane & 7u);
    // two K register sets: the requests of tiles t+1 and t+2 are in flight while tile t is scored
    struct KTile { uint4 kx[2][2]; f32x4 ks4; };
    auto issue_k = [&](KTile& T) {
#pragma unroll
        for (int b = 0; b < 2; ++b) { T.kx[b][0] = ldg16(kp + 16384 * b); T.kx[b][1] = ldg16(kp + 16384 * b + 64); }
        T.ks4 = ldg_f4(kt);
    };
    auto score_store = [&](uint32_t tile, const KTile& T) {
        f32x4 sc[2];
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const uint32_t kd[8] = {T.kx[b][0].x, T.kx[b][0].y, T.kx[b][0].z, T.kx[b][0].w, T.kx[b][1].x, T.kx[b][1].y, T.kx[b][1].z, T.kx[b][1].w};
            f32x4 s = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int st = 0; st < 4; ++st)
                s = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(pack64(kd[2 * st], kd[2 * st + 1]), pack64(qd[2 * st], qd[2 * st + 1]), s, 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 4; ++i) sc[b][i] = s[i] * T.ks4[2 * b + (i >> 1)] * qscale;    // (acc * k scale) * q scale, as the page-table form
        }
#pragma unroll
        for (int b = 0; b < 2; ++b) *reinterpret_cast<f32x4*>(st + c * 36u + 16u * b + 4u * kb) = sc[b];
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const uint32_t r = (lane >> 3) + 8u * half;                                        // query row of this lane's piece
            if (r < a.g) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(st + r * 36u + 4u * (lane & 7u));
                const uint32_t p = tile * 32u + 4u * (lane & 7u);
                float* d = dst + static_cast<uint64_t>(8u * half) * n_pos;
                if (p + 3u < n_pos) __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(d));
                else
                    for (int i = 0; i < 4; ++i)
                        if (p + i < n_pos) d[i] = v[i];
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
        dst += 32;
    };
    auto advance = [&](uint32_t next_tile) { const uint32_t st = next_tile < t1 ? 1u : 0u; kp += st * 32768u; kt += st * 16u; };
    KTile A, B;
    __builtin_amdgcn_sched_barrier(0);
    issue_k(A);
    __builtin_amdgcn_sched_barrier(0);
    advance(t0 + 1u);
    issue_k(B);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll 1
    for (uint32_t tile = t0; tile < t1; tile += 2) {
        score_store(tile, A);
        __builtin_amdgcn_sched_barrier(0);
        advance(tile + 2u);
        issue_k(A);                                   // (past the end: re-requests the last tile, never used)
        __builtin_amdgcn_sched_barrier(0);
        if (tile + 1u >= t1) break;
        score_store(tile + 1u, B);
        __builtin_amdgcn_sched_barrier(0);
        advance(tile + 3u);
        issue_k(B);
        __builtin_amdgcn_sched_barrier(0);
    }
}

hipError_t launch_qk_scores_fp8_linear(const AttendArgs& a, uint32_t n_layers, float* d_out, hipStream_t s)
{
    if (a.n_pages == 0 || n_layers == 0) return hipSuccess;
    const uint32_t n_tiles = (a.n_pages + 15u) / 16u;
    hipLaunchKernelGGL(k_qk_scores_fp8_linear, dim3((n_tiles + a.tiles_per_split - 1u) / a.tiles_per_split, n_layers * (a.heads / 4u)), dim3(256), 0, s, a, d_out);
    return hipGetLastError();
}

constexpr uint32_t kSmallCombineRowsShared = 8192;     // rows up to which the small merge spends a workgroup per row (k_attend_combine_small<4>)
hipError_t launch_attend_combine(const AttendArgs& a, uint32_t n_layers, float* d_out, float* d_lse, hipStream_t s)
{
    if (n_layers == 0) return hipSuccess;
    const uint32_t splits = a.stream.n_wgs ? a.stream.max_slots : a.n_splits;      // most partials a row can have
    const uint32_t n_tiles = a.stream.tiles ? a.stream.tiles : (a.n_pages + 15u) / 16u;
    if (splits > kMaxSplits) return hipErrorInvalidValue;
    if (splits <= kSmallCombineSplits) {
        const uint32_t n_rows = n_layers * a.heads;
        if (n_rows <= kSmallCombineRowsShared)
            hipLaunchKernelGGL(k_attend_combine_small<4>, dim3(n_rows), dim3(256), 0, s, a.part_acc, a.part_ml, a.g,
                               a.n_splits, d_out, d_lse, a.seqs, a.heads, (a.direct_out && a.direct_per_seq) ? 1u : 0u, n_rows, a.stream, n_tiles);
        else
            hipLaunchKernelGGL(k_attend_combine_small<1>, dim3((n_rows + 3u) / 4u), dim3(256), 0, s, a.part_acc, a.part_ml, a.g,
                               a.n_splits, d_out, d_lse, a.seqs, a.heads, (a.direct_out && a.direct_per_seq) ? 1u : 0u, n_rows, a.stream, n_tiles);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(k_attend_combine, dim3(n_layers * a.heads * a.g), dim3(512), 0, s, a.part_acc, a.part_ml, a.g,
                       a.n_splits, d_out, d_lse, a.seqs, a.heads, (a.direct_out && a.direct_per_seq) ? 1u : 0u, a.stream, n_tiles);
    return hipGetLastError();
}

// One more position for rows that already hold softmax(q.K^T).V and its log-sum-exp: the fp16 K / V row a decode step
// has produced but not yet stored (a connector keeps the odd position of a pair until its partner arrives).
//   s = q.k * sm_scale;  new = logaddexp(lse, s);  out = out * exp(lse - new) + v * exp(s - new);  lse = new
// One wave per (row, kv head, query row), a lane holds two of the 128 dimensions.  A row without stored positions comes
// in as out = 0, lse = -inf and leaves as out = v, lse = s.
// Several layers in one launch (round 6; a planned step whose attention runs layer by layer folds all its layers at the end):
// waves_per_layer != 0 -> wave w belongs to layer w / waves_per_layer, whose q / out / lse rows start layer_rows row blocks further on
// and whose tail rows lie heads x 128 elements further on ([tail][layers][heads][128]).
__global__ __launch_bounds__(256) void k_attend_fold_tail(uint32_t n_waves, const uint32_t* __restrict__ rows, uint32_t heads, uint32_t g,
                                                          const f16x2* __restrict__ q, const f16x2* __restrict__ k_tail,
                                                          const f16x2* __restrict__ v_tail, uint64_t tail_stride_h2, float sm_scale,
                                                          float2* __restrict__ out, float* __restrict__ lse, uint32_t waves_per_layer, uint32_t layer_rows)
{
    uint32_t w = blockIdx.x * 4u + (threadIdx.x >> 6);
    const uint32_t lane = threadIdx.x & 63u;
    if (w >= n_waves) return;
    uint32_t l = 0;
    if (waves_per_layer) { l = w / waves_per_layer; w -= l * waves_per_layer; }
    const uint32_t m = w % g, head = (w / g) % heads, i = w / (g * heads);
    const uint32_t b = (rows ? rows[i] : i) + l * layer_rows;
    const uint64_t row = (static_cast<uint64_t>(b) * heads + head) * g + m;
    const uint64_t t_at = i * tail_stride_h2 + (static_cast<uint64_t>(l) * heads + head) * 64u + lane;
    const f16x2 qh = q[row * 64u + lane], kh = k_tail[t_at], vh = v_tail[t_at];
    const float2 qv = make_float2(static_cast<float>(qh.x), static_cast<float>(qh.y));
    const float2 kv = make_float2(static_cast<float>(kh.x), static_cast<float>(kh.y));
    const float2 vv = make_float2(static_cast<float>(vh.x), static_cast<float>(vh.y));
    float dot = qv.x * kv.x + qv.y * kv.y;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) dot += __shfl_xor(dot, o);
    const float sc = dot * sm_scale, old = lse[row];
    const float hi = fmaxf(old, sc);
    const float e_old = __expf(old - hi), e_new = __expf(sc - hi);          // old = -inf: e_old = 0
    const float inv = 1.0f / (e_old + e_new);
    float2 o = out[row * 64u + lane];
    o.x = (o.x * e_old + vv.x * e_new) * inv;
    o.y = (o.y * e_old + vv.y * e_new) * inv;
    out[row * 64u + lane] = o;
    if (lane == 0) lse[row] = hi + __logf(e_old + e_new);
}

hipError_t launch_attend_fold_tail(uint32_t n_rows, const uint32_t* d_rows, uint32_t heads, uint32_t g, const void* d_q_f16,
                                   const void* d_k_tail, const void* d_v_tail, uint64_t tail_stride_elems, float sm_scale,
                                   float* d_out, float* d_lse, hipStream_t s, uint32_t n_layers, uint32_t layer_rows)
{
    const uint64_t per_layer = static_cast<uint64_t>(n_rows) * heads * g, n_waves = per_layer * n_layers;
    if (n_waves == 0) return hipSuccess;
    if (n_waves > 0xFFFFFFFFull) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_attend_fold_tail, dim3(static_cast<uint32_t>((n_waves + 3u) / 4u)), dim3(256), 0, s, static_cast<uint32_t>(n_waves),
                       d_rows, heads, g, static_cast<const f16x2*>(d_q_f16), static_cast<const f16x2*>(d_k_tail),
                       static_cast<const f16x2*>(d_v_tail), tail_stride_elems / 2u, sm_scale, reinterpret_cast<float2*>(d_out), d_lse,
                       n_layers > 1u ? static_cast<uint32_t>(per_layer) : 0u, layer_rows);
    return hipGetLastError();
}

hipError_t launch_attend_fp8_batch(const AttendArgs& a, uint32_t n_seq, float* d_out, float* d_lse, hipStream_t s)
{
    if (n_seq == 0 || a.n_splits == 0) return hipSuccess;
    // (rows_first: the sequences' first pieces side by side, then their second ones ...: a batch in which few members have several pieces would otherwise
    //  put every real workgroup on the same XCDs -- linear ids x + splits * y with most x > 0 empty)
    AttendArgs ar = a;
    auto grid_of = [&](uint32_t hq) { ar.rows_real = n_seq * hq; return a.rows_first ? dim3((n_seq * hq) | 1u, a.n_splits) : dim3(a.n_splits, n_seq * hq); };
    // the batch form keeps the register-staged kernel: 256 sequences x 8k context, one layer: 0.69 of HBM peak against 0.67
    // with the LDS-DMA kernel (many short splits: 4 waves/SIMD hide more than two tiles per wave at 2 waves/SIMD); the
    // single-sequence form below is the other way round (8k x 80 layers: 0.61 against 0.55; 32k x 80: 0.70 both)
    if (a.fp8_cls && a.stripe_bases && !a.table_form)                   // every member placed regularly: pages by residue class
        { const dim3 gr = grid_of(a.heads / 4u); hipLaunchKernelGGL((k_attend_fp8_linear<false, false, true>), gr, dim3(256), 0, s, ar); }
    else if (a.table_form && tuning().attend_fp8_table_regs == 0)       // striped / moved placements: the DMA pipeline with addresses from the page tables
        { const dim3 gr = grid_of(a.heads / kFdHeads); hipLaunchKernelGGL(k_attend_fp8_dma<1>, gr, dim3(64 * kFdHeads), 0, s, ar); }
    else if (a.table_form) { const dim3 gr = grid_of(a.heads / 4u); hipLaunchKernelGGL((k_attend_fp8_linear<false, true>), gr, dim3(256), 0, s, ar); }
    else if (a.stripe_bases) { const dim3 gr = grid_of(a.heads / 4u); hipLaunchKernelGGL(k_attend_fp8_linear<true>, gr, dim3(256), 0, s, ar); }
    else     { const dim3 gr = grid_of(a.heads / 4u); hipLaunchKernelGGL(k_attend_fp8_linear<false>, gr, dim3(256), 0, s, ar); }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess || (a.direct_out && a.direct_per_seq != 1u)) return e;      // every row final: no merge
    return launch_attend_combine(a, n_seq, d_out, d_lse, s);
}

hipError_t launch_build_scale_tab(const PageEntry* d_entries, uint64_t n_pages, uint32_t region_pages, float* d_scale_tab, hipStream_t s, uint32_t scale_run)
{
    if (n_pages == 0) return hipSuccess;
    hipLaunchKernelGGL(k_build_scale_tab, dim3(static_cast<uint32_t>((n_pages + 255u) / 256u)), dim3(256), 0, s, d_entries, n_pages,
                       region_pages, d_scale_tab, scale_run);
    return hipGetLastError();
}

hipError_t launch_attend_fp8(const AttendArgs& a, uint32_t n_layers, float* d_out, float* d_lse, hipStream_t s)
{
    if (a.n_pages == 0 || n_layers == 0) return hipSuccess;
    // one long split per row: the LDS-DMA kernel (two tiles deep per wave, 8 waves per CU); several shorter splits: the
    // register-staged one (16 waves per CU hide more) -- 80 layers x 32k: 8 splits 0.756 (register-staged) / 0.72 (DMA),
    // 1 split 0.58 / 0.73; 8k: 1 split DMA 0.69, 8 splits 0.63 / 0.62
    // (launches with few workgroup columns -- the per-layer calls of one sequence -- are latency-bound either way: DMA kernel,
    // 13.8 against 15.5 us at 8k context)
    const bool dma_table = !a.lin_base && (a.stripe_bases || a.table_form) && a.scale_tab && a.entries && !a.skip_pages && tuning().attend_fp8_table_regs == 0;
    if (a.fp8_cls && a.stripe_bases && a.scale_tab && !a.skip_pages) {   // striped regularly: the linear kernels over residue classes, by the linear forms' rule
        if (a.n_splits == 1u || n_layers * (a.heads / 4u) < 128u)
            hipLaunchKernelGGL(k_attend_fp8_dma<2>, dim3(a.n_splits, n_layers * (a.heads / kFdHeads)), dim3(64 * kFdHeads), 0, s, a);
        else
            hipLaunchKernelGGL((k_attend_fp8_linear<false, false, true>), dim3(a.n_splits, n_layers * (a.heads / 4u)), dim3(256), 0, s, a);
    }
    else if (dma_table)                                                  // moved page by page (or striped, on request): the DMA pipeline with addresses from the page table
        hipLaunchKernelGGL(k_attend_fp8_dma<1>, dim3(a.n_splits, n_layers * (a.heads / kFdHeads)), dim3(64 * kFdHeads), 0, s, a);
    else if (a.lin_base && tuning().attend_fp8_dma >= 0 && (a.n_splits == 1u || n_layers * (a.heads / 4u) < 128u || tuning().attend_fp8_dma > 0))
        hipLaunchKernelGGL(k_attend_fp8_dma<0>, dim3(a.n_splits, n_layers * (a.heads / kFdHeads)), dim3(64 * kFdHeads), 0, s, a);
    else if (a.lin_base)
        hipLaunchKernelGGL(k_attend_fp8_linear<false>, dim3(a.n_splits, n_layers * (a.heads / 4u)), dim3(256), 0, s, a);
    else if (a.stripe_bases)
        hipLaunchKernelGGL(k_attend_fp8_linear<true>, dim3(a.n_splits, n_layers * (a.heads / 4u)), dim3(256), 0, s, a);
    else if (a.table_form)
        hipLaunchKernelGGL((k_attend_fp8_linear<false, true>), dim3(a.n_splits, n_layers * (a.heads / 4u)), dim3(256), 0, s, a);
    else            hipLaunchKernelGGL(k_attend_fp8, dim3(a.n_splits, n_layers * (a.heads / 4u)), dim3(256), 0, s, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess || (a.direct_out && (a.lin_base || a.stripe_bases || a.table_form))) return e;      // (the page-table kernel always writes partials)
    return launch_attend_combine(a, n_layers, d_out, d_lse, s);
}

} // namespace speckv
