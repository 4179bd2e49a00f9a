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
__device__ __forceinline__ uint4 ldg16(const uint8_t* p)
{
    const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p));
    return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ uint32_t ldg4(const uint8_t* p)
{
    return __builtin_nontemporal_load(reinterpret_cast<const uint32_t*>(p));
}
// max over the four lanes {c, c+16, c+32, c+48}
__device__ __forceinline__ float max_over_kb(float v)
{
    v = fmaxf(v, __shfl_xor(v, 16));
    return fmaxf(v, __shfl_xor(v, 32));
}
__device__ __forceinline__ float sum_over_kb(float v)
{
    v += __shfl_xor(v, 16);
    return v + __shfl_xor(v, 32);
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

// scores -> online softmax -> out^T += V^T . P^T for one tile.  inr[r]: slot page r lies inside the range.
__device__ __forceinline__ void attend_tile(const AttTile& T, const bool (&inr)[4], const uint32_t (&qd)[8], float qscale,
                                            AttState& S)
{
    // ---- scores of the lane's 8 position slots (log2 domain)
    float sc[8];
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const uint32_t kd[8] = {T.kx[b][0].x, T.kx[b][0].y, T.kx[b][0].z, T.kx[b][0].w,
                                T.kx[b][1].x, T.kx[b][1].y, T.kx[b][1].z, T.kx[b][1].w};
        f32x4 s = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int st = 0; st < 4; ++st)
            s = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(pack64(kd[2 * st], kd[2 * st + 1]), pack64(qd[2 * st], qd[2 * st + 1]), s, 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int j = 4 * b + i;
            sc[j] = inr[j >> 1] ? s[i] * T.ks[j >> 1] * qscale : -INFINITY;
        }
    }
    // ---- online softmax of query row c (its 32 positions sit in lanes c, c+16, c+32, c+48)
    float mx = sc[0];
#pragma unroll
    for (int j = 1; j < 8; ++j) mx = fmaxf(mx, sc[j]);
    mx = max_over_kb(mx);
    const float m_new = fmaxf(S.m_run, mx);
    const float m_use = (m_new == -INFINITY) ? 0.0f : m_new;              // a fully masked row stays at weight 0
    const float alpha = __builtin_amdgcn_exp2f(S.m_run - m_use);
    S.m_run = m_new;
    float vmx = fmaxf(fmaxf(T.vs[0], T.vs[1]), fmaxf(T.vs[2], T.vs[3]));
    vmx = max_over_kb(vmx);                                               // the tile's largest V page scale
    const float vinv = vmx > 0.0f ? 1.0f / vmx : 0.0f;
    float psum = 0.0f;
    f16x8 P;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float p = __builtin_amdgcn_exp2f(sc[j] - m_use);
        psum += p;
        P[j] = static_cast<_Float16>(p * (T.vs[j >> 1] * vinv));          // V page scale rides on the weight, <= 1
    }
    S.l_run = S.l_run * alpha + psum;
    // ---- out^T += V^T . P^T
#pragma unroll
    for (int blk = 0; blk < 2; ++blk) {
        uint32_t w01[4], w23[4];
#pragma unroll
        for (int jp = 0; jp < 4; ++jp) {
            // [lo.b0, hi.b0, lo.b1, hi.b1] / [lo.b2, hi.b2, lo.b3, hi.b3]: two positions of one d column per word
            w01[jp] = __builtin_amdgcn_perm(T.vx[blk][2 * jp + 1], T.vx[blk][2 * jp], 0x05010400u);
            w23[jp] = __builtin_amdgcn_perm(T.vx[blk][2 * jp + 1], T.vx[blk][2 * jp], 0x07030602u);
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

} // namespace

// General form: every page goes through its page-table entry (pool address, validity, scale), so
// records may sit anywhere (striped over pool GPUs, migrated, fragmented).  Two dependent
// memory round trips per tile; the linear form below is the fast one.
__global__ __launch_bounds__(256) void k_attend_fp8(AttendArgs a)
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
    const uint8_t* qrow = a.q8 + (row * 16u + c) * 128u + kb * 32u;
    const uint4 qa0 = *reinterpret_cast<const uint4*>(qrow), qa1 = *reinterpret_cast<const uint4*>(qrow + 16);
    const uint32_t qd[8] = {qa0.x, qa0.y, qa0.z, qa0.w, qa1.x, qa1.y, qa1.z, qa1.w};
    const float qscale = a.qs[row * 16u + c] * a.scale_log2e;

    const uint32_t n_tiles = (a.n_pages + 15u) / 16u;
    const uint32_t t0 = split * a.tiles_per_split;
    const uint32_t t1 = min(t0 + a.tiles_per_split, n_tiles);
    AttState S;
    att_init(S);
#pragma unroll 1
    for (uint32_t tile = t0; tile < t1; ++tile) {
        const uint32_t pg0 = tile * 16u;
        // ---- page descriptors: the two K rows this lane feeds, and its four position-slot pages
        const uint8_t* kaddr[2];
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const uint32_t pg = pg0 + 8u * b + (c >> 1);
            kaddr[b] = a.zero_page + kb * 32u;
            if (pg < a.n_pages) {
                const PageEntry e = kent[pg];
                if (e.rec_bytes >= kBlockElems)
                    kaddr[b] = reinterpret_cast<const uint8_t*>(e.pool_addr) + (c & 1u) * 1024u + head * 128u + kb * 32u;
            }
        }
        AttTile T;
        const uint8_t* vaddr[4];
        bool inr[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const uint32_t pg = pg0 + (r < 2 ? 2u * kb + r : 8u + 2u * kb + (r - 2));
            inr[r] = pg < a.n_pages;
            vaddr[r] = a.zero_page + 4u * c;
            T.ks[r] = 0.0f;
            T.vs[r] = 0.0f;
            if (inr[r]) {
                const PageEntry ke = kent[pg], ve = vent[pg];
                if (ke.rec_bytes >= kBlockElems) T.ks[r] = ke.scale;
                if (ve.rec_bytes >= kBlockElems) {
                    T.vs[r] = ve.scale;
                    vaddr[r] = reinterpret_cast<const uint8_t*>(ve.pool_addr) + head * 128u + 4u * c;
                }
            }
        }
        // ---- data: K 2 x 32 B per lane, V 16 dwords per lane
#pragma unroll
        for (int b = 0; b < 2; ++b) { T.kx[b][0] = ldg16(kaddr[b]); T.kx[b][1] = ldg16(kaddr[b] + 16); }
#pragma unroll
        for (int blk = 0; blk < 2; ++blk)
#pragma unroll
            for (int j = 0; j < 8; ++j) T.vx[blk][j] = ldg4(vaddr[j >> 1] + (j & 1) * 1024 + 64 * blk);
        attend_tile(T, inr, qd, qscale, S);
    }
    att_store(a, S, row * a.n_splits + split, c, kb);
}

// Linear form: the allocation's records lie in one run (record p at lin_base + p*2048, the engine's
// default placement) and never-written records are zero bytes (the engine zero-fills FP8 pools), so
// every data address is arithmetic and only the page scales come from the page table -- nothing
// gates the data loads.  Tiles are double-buffered in registers: the loads of tile t+1 are in
// flight while tile t is computed.
__global__ __launch_bounds__(256) void k_attend_fp8_linear(AttendArgs a)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t c = lane & 15u, kb = lane >> 4;
    const uint32_t split = blockIdx.x;
    const uint32_t hq = a.heads / 4u;
    const uint32_t layer = blockIdx.y / hq;
    const uint32_t head = (blockIdx.y % hq) * 4u + wave;
    const uint64_t kpage0 = a.k_first + layer * a.layer_stride, vpage0 = a.v_first + layer * a.layer_stride;
    const PageEntry* kent = a.entries + kpage0;
    const PageEntry* vent = a.entries + vpage0;
    const uint8_t* kbase = a.lin_base + kpage0 * 2048ull + head * 128u + kb * 32u;     // + position * 1024
    const uint8_t* vbase = a.lin_base + vpage0 * 2048ull + head * 128u + 4u * c;
    const uint64_t row = static_cast<uint64_t>(layer) * a.heads + head;

    const uint8_t* qrow = a.q8 + (row * 16u + c) * 128u + kb * 32u;
    const uint4 qa0 = *reinterpret_cast<const uint4*>(qrow), qa1 = *reinterpret_cast<const uint4*>(qrow + 16);
    const uint32_t qd[8] = {qa0.x, qa0.y, qa0.z, qa0.w, qa1.x, qa1.y, qa1.z, qa1.w};
    const float qscale = a.qs[row * 16u + c] * a.scale_log2e;

    const uint32_t n_tiles = (a.n_pages + 15u) / 16u;
    const uint32_t t0 = split * a.tiles_per_split;
    const uint32_t t1 = min(t0 + a.tiles_per_split, n_tiles);
    const uint32_t last_pos = 2u * a.n_pages - 1u, last_page = a.n_pages - 1u;

    // positions and pages beyond the range are clamped to the last valid one for the loads and masked by inr
    auto issue = [&](uint32_t tile, AttTile& T) {
        const uint32_t p0 = tile * 32u, pg0 = tile * 16u;
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const uint8_t* r = kbase + static_cast<uint64_t>(min(p0 + 16u * b + c, last_pos)) * 1024u;
            T.kx[b][0] = ldg16(r);
            T.kx[b][1] = ldg16(r + 16);
        }
#pragma unroll
        for (int blk = 0; blk < 2; ++blk)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const uint32_t pos = p0 + (j < 4 ? 4u * kb + j : 16u + 4u * kb + (j - 4));
                T.vx[blk][j] = ldg4(vbase + static_cast<uint64_t>(min(pos, last_pos)) * 1024u + 64 * blk);
            }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const uint32_t pg = min(pg0 + (r < 2 ? 2u * kb + r : 8u + 2u * kb + (r - 2)), last_page);
            T.ks[r] = kent[pg].scale;
            T.vs[r] = vent[pg].scale;
        }
    };
    auto compute = [&](uint32_t tile, const AttTile& T, AttState& S) {
        bool inr[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) inr[r] = tile * 16u + (r < 2 ? 2u * kb + r : 8u + 2u * kb + (r - 2)) < a.n_pages;
        attend_tile(T, inr, qd, qscale, S);
    };

    AttState S;
    att_init(S);
    AttTile A, B;
    if (t0 < t1) issue(t0, A);
#pragma unroll 1
    for (uint32_t tile = t0; tile < t1; tile += 2) {
        if (tile + 1 < t1) issue(tile + 1, B);
        compute(tile, A, S);
        if (tile + 1 >= t1) break;
        if (tile + 2 < t1) issue(tile + 2, A);
        compute(tile + 1, B, S);
    }
    att_store(a, S, row * a.n_splits + split, c, kb);
}

// one workgroup (128 threads = the 128 d) per (layer, head, query row): merge the splits
__global__ __launch_bounds__(128) void k_attend_combine(const float* __restrict__ part_acc, const float* __restrict__ part_ml,
                                                        uint32_t g, uint32_t n_splits, float* __restrict__ out,
                                                        float* __restrict__ lse)
{
    const uint32_t rowq = blockIdx.x / g, m = blockIdx.x % g, d = threadIdx.x;     // rowq = layer*heads + head
    float M = -INFINITY;
    for (uint32_t s = 0; s < n_splits; ++s) M = fmaxf(M, part_ml[(static_cast<uint64_t>(rowq) * n_splits + s) * 32u + m]);
    const float Mu = (M == -INFINITY) ? 0.0f : M;
    float L = 0.0f, o = 0.0f;
    for (uint32_t s = 0; s < n_splits; ++s) {
        const uint64_t part = static_cast<uint64_t>(rowq) * n_splits + s;
        const float w = __builtin_amdgcn_exp2f(part_ml[part * 32u + m] - Mu);
        L += w * part_ml[part * 32u + 16u + m];
        o += w * part_acc[(part * 16u + m) * 128u + d];
    }
    out[(static_cast<uint64_t>(rowq) * g + m) * 128u + d] = L > 0.0f ? o / L : 0.0f;
    if (lse && d == 0) lse[static_cast<uint64_t>(rowq) * g + m] = L > 0.0f ? (M + log2f(L)) * 0.6931471805599453f : -INFINITY;
}

hipError_t launch_attend_fp8(const AttendArgs& a, uint32_t n_layers, float* d_out, float* d_lse, hipStream_t s)
{
    if (a.n_pages == 0 || n_layers == 0) return hipSuccess;
    if (a.lin_base) hipLaunchKernelGGL(k_attend_fp8_linear, dim3(a.n_splits, n_layers * (a.heads / 4u)), dim3(256), 0, s, a);
    else            hipLaunchKernelGGL(k_attend_fp8, dim3(a.n_splits, n_layers * (a.heads / 4u)), dim3(256), 0, s, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_attend_combine, dim3(n_layers * a.heads * a.g), dim3(128), 0, s, a.part_acc, a.part_ml, a.g,
                       a.n_splits, d_out, d_lse);
    return hipGetLastError();
}

} // namespace speckv
