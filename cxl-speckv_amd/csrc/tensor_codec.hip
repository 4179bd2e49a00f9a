// cxl-speckv_amd/csrc/tensor_codec.hip -- FPGACacheEngine::compress / ::decompress over a tensor of ANY length, as the
// reference defines them (src/fpga_engine/cache_engine.cpp:40-116): ONE scale over the n elements (compute_scale_factor,
// :172-184), ONE int8 delta chain (:198-211, :260-273) and ONE run-length stream (:213-258) -- runs, 255-element splits and
// the delta chain all cross the 2048-element tiles the work is cut into.  (The pool itself stores KV per 4 KiB block,
// kernels.hip: each compress() call of the engine is one block.  This file is the reference's own call shape: n = 11 in the
// survey's known-answer test, the RTL tile of 1024 x 128 = 131 072 elements, ...)
//
// The serial recurrences become scans at two levels: inside a tile they are the wave scans of the block codec (DPP add / max
// scans with carries); across tiles, since round 4, a decoupled look-back over one 8-byte status word per WORKGROUP of 16 tiles:
//   compress:   k_tc_absmax -> k_tc_fused    (one pass over the source behind the abs-max pass; pairs written where they belong)
//   decompress: k_td_fused                   (one pass over the stream; a byte scattered per run, as the block decoder does)
// The multi-launch forms of rounds 2-3 stay as SPECKV_TC_MULTIPASS=1 (tests run both; A/B):
//   compress:   k_tc_absmax -> k_tc_tiles<summary> -> k_tc_scan_a/_b/_c -> k_tc_tiles<emit> -> k_tc_pack
//   decompress: k_td_summary -> k_td_scan_local/_apply -> k_td_expand
// Positions p = 0..n-1; d[p] = q[p] - q[p-1] (q[-1] = 0); a STRETCH starts at p == 0 or d[p] != d[p-1]; a RUN starts every
// 255 elements of a stretch (cache_engine.cpp:224: `count < 255`); pair = (d[p], distance to the next run start).
// The multi-launch forms write whole 128-byte lines per wave (tile-local pair buffers, then an output-centric pack pass); the
// one-pass kernels store whole aligned 16-byte pieces and single bytes / elements only at the two ragged ends of a tile's stretch.
#include "kernels.hpp"
#include "tuning.hpp"
#include "codec_device.hpp"
#include "encode_device.hpp"

#include <algorithm>
#include <cstdlib>
#include <cstring>

namespace speckv {
namespace {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr uint32_t kTile = 2048;            // elements per tile (one wave)
constexpr uint32_t kTcWaves = 4;
#ifdef SPECKV_TC_NO_FAST
constexpr bool kTcNoFast = true;             // (A/B builds: the element-wise tile loop only)
#else
constexpr bool kTcNoFast = false;
#endif
#ifdef SPECKV_TC_PRE_RELOAD
constexpr bool kTcPreRegs = false;           // (A/B builds: fp16 tiles read again behind the rendezvous, like fp32 ones)
#else
constexpr bool kTcPreRegs = true;
#endif
constexpr uint32_t kTcLead = 16;            // bytes in front of a wave's pair buffer: "count of the pair before the first" lands here

struct TcSummary {                          // what a tile knows without its left neighbours (positions tile-relative, +1; 0 = none)
    uint32_t first_ss, last_ss;             // first / last stretch start
    uint32_t cnt_b, last_b;                 // run starts at or behind the first stretch start: how many, the last one
};
struct TcCarry {                            // what the scan hands back to a tile
    uint32_t lead_phase;                    // (tile start - stretch start entering the tile) % 255
    uint32_t runs;                          // run starts in the tile
    uint64_t run_base;                      // run starts before the tile
    uint64_t next_run;                      // position of the first run start behind the tile (n if none)
};

template <bool F32>
__device__ __forceinline__ float tc_load(const void* src, uint64_t p)
{
    return F32 ? static_cast<const float*>(src)[p] : half_bits_to_float(static_cast<const uint16_t*>(src)[p]);
}

// max|x| over the tensor as fp32 bits (non-negative floats order like their bit patterns); a NaN never wins
// (cache_engine.cpp:176-180: `if (abs > max_val)`).  16 bytes per lane and step where the source allows it (the elements in
// front of the first 16-byte boundary and behind the last one are taken one by one).
// one thread's share (thread `tid` of `nthr`) of that maximum: fp32 sources |x| bits, fp16 sources |x| half bits (converted by the
// caller at the end).  NT: non-temporal loads (a pass that nobody re-reads); the batched kernel below keeps the default policy --
// its second pass over the same 512 KiB comes out of the L2 / Infinity Cache.
template <bool F32, bool NT>
__device__ __forceinline__ uint32_t tc_absmax_thread(const void* __restrict__ src, uint64_t n, uint64_t tid, uint64_t nthr)
{
    constexpr uint32_t kPer = F32 ? 4u : 8u, kEsz = F32 ? 4u : 2u;
    const uintptr_t addr = reinterpret_cast<uintptr_t>(src);
    uint64_t head = ((16u - (addr & 15u)) & 15u) / kEsz;
    if ((addr & (kEsz - 1u)) != 0u || head > n) head = n;            // (a source that is not even element-aligned: all scalar)
    const uint64_t nvec = (n - head) / kPer;
    uint32_t m = 0;
    auto one = [&](uint64_t p) {
        if (F32) { const uint32_t a = static_cast<const uint32_t*>(src)[p] & 0x7FFFFFFFu; if (a <= 0x7F800000u) m = umax(m, a); }
        else     { const uint32_t a = static_cast<const uint16_t*>(src)[p] & 0x7FFFu;     if (a <= 0x7C00u) m = umax(m, a); }
    };
    for (uint64_t p = tid; p < head; p += nthr) one(p);
    const u32x4* vsrc = reinterpret_cast<const u32x4*>(static_cast<const uint8_t*>(src) + head * kEsz);
    auto take = [&](const u32x4 x) {
        const uint32_t w[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (F32) { const uint32_t a = w[i] & 0x7FFFFFFFu; if (a <= 0x7F800000u) m = umax(m, a); }
            else {
                const uint32_t a = w[i] & 0x7FFFu, b = (w[i] >> 16) & 0x7FFFu;
                if (a <= 0x7C00u) m = umax(m, a);
                if (b <= 0x7C00u) m = umax(m, b);
            }
        }
    };
    auto ld = [&](const u32x4* p) { return NT ? __builtin_nontemporal_load(p) : *p; };
    // four 16-byte loads in flight per lane (the pass is a pure read: it ran at 5.7 TB/s with one, the chip reads at 7)
    uint64_t v = tid;
    for (; v + 3u * nthr < nvec; v += 4u * nthr) {
        const u32x4 x0 = ld(vsrc + v), x1 = ld(vsrc + v + nthr);
        const u32x4 x2 = ld(vsrc + v + 2u * nthr), x3 = ld(vsrc + v + 3u * nthr);
        take(x0); take(x1); take(x2); take(x3);
    }
    for (; v < nvec; v += nthr) take(ld(vsrc + v));
    for (uint64_t p = head + nvec * kPer + tid; p < n; p += nthr) one(p);
    return m;
}
template <bool F32>
__global__ __launch_bounds__(1024) void k_tc_absmax(const void* __restrict__ src, uint64_t n, uint32_t* __restrict__ out_bits)
{
    const uint64_t tid = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x, nthr = static_cast<uint64_t>(gridDim.x) * blockDim.x;
    uint32_t m = tc_absmax_thread<F32, true>(src, n, tid, nthr);
    // one atomic per workgroup of 1024 threads, one workgroup per CU: same-address atomics take ~12 ns each one after the other
    // (16 384 of them, one per wave of a 4096-block grid, 190 us by themselves; 1024, one per 256-thread workgroup, still 12 of
    // the kernel's 22 us)
    __shared__ uint32_t s_m[16];
    m = lane63(wave_incl_max(m));
    if ((threadIdx.x & 63u) == 0u) s_m[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0u) {
        m = 0;
        for (uint32_t w = 0; w < (blockDim.x >> 6); ++w) m = umax(m, s_m[w]);
        if (!F32) m = __float_as_uint(half_bits_to_float(m));
        if (m) atomicMax(out_bits, m);
    }
}

__device__ __forceinline__ float tc_scale(uint32_t absmax_bits)
{
    const float mx = __uint_as_float(absmax_bits);
    return (mx > 0.0f) ? (mx / 127.0f) : 1.0f;                      // cache_engine.cpp:183
}

// A whole tile of fp16 elements by the block encoder's FAST path (kernels.hip: encode_rle_fast -- 8 elements per lane and step,
// packed-fp32 quantisation, SDWA deltas, run-start predicates in SGPR pairs, EXEC-predicated pair scatter), started from the
// two elements in front of the tile instead of from zero.  Valid while every change of the delta starts a run and no run has
// to be split: it gives up (false; nothing of its LDS output is used) when the tile holds inf / NaN, when a stretch between
// two starts of the tile, or behind its last start, may reach 255 elements -- the caller then walks the tile element by
// element.  Stretches that ENTER the tile are the scan's business (TcCarry::lead_phase); the caller checks that they start no
// run inside the tile before it trusts the pairs of the emit pass.
//   first_ss / last_ss: first / last run start of the tile (tile-relative + 1, 0 = none), n_starts: how many.
// fp32 sources (SRC32; the reference's own input type, cache_engine.cpp:40 `const float* data`): the same path on eight floats
// per lane and step -- two 16-byte loads -- with the IEEE divide of quantize<MODE> (the reciprocal short cut of quantize8 is
// exact for fp16-valued operands only).  Such tiles took the element-wise loop before: a 256 Mi-element fp32 tensor compressed
// in 1.54 ms against 0.47 ms for the same values as fp16.
template <int MODE, bool FASTDIV>
__device__ __forceinline__ void quantize8_f32(const u32x4 a, const u32x4 b, float scale, float rcp, uint32_t (&q)[8])
{
    const uint32_t w[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        f32x2 y;
        if (FASTDIV) {                                               // (the scale is inside 2^-60 .. 2^60: codec_device.hpp, div_f32_by_scale)
            y.x = div_f32_by_scale(__uint_as_float(w[2 * t]), scale, rcp);
            y.y = div_f32_by_scale(__uint_as_float(w[2 * t + 1]), scale, rcp);
        } else {
            y.x = __uint_as_float(w[2 * t]) / scale;                 // (correctly rounded: no fast-math in this build)
            y.y = __uint_as_float(w[2 * t + 1]) / scale;
        }
        if (MODE == kRefExact) { const f32x2 k127 = {127.0f, 127.0f}; y = y * k127; }      // cache_engine.cpp:190-191
        f32x2 h;
        h.x = __builtin_copysignf(0x1.fffffep-2f, y.x);              // (round_to_int_f32: pred(0.5), exact for every fp32 y)
        h.y = __builtin_copysignf(0x1.fffffep-2f, y.y);
        const f32x2 r = y + h;                                       // round half away from zero = truncate(y + copysign(pred(0.5), y))
        int i0 = static_cast<int>(r.x), i1 = static_cast<int>(r.y);
        if (MODE != kRefExact) { i0 = min(max(i0, -127), 127); i1 = min(max(i1, -127), 127); }
        q[2 * t] = static_cast<uint32_t>(i0);
        q[2 * t + 1] = static_cast<uint32_t>(i1);
    }
}
//   PRE (k_tcm_fused took the tile's max|x| itself): fp16 -- the tile's 4 KiB are in registers already and known to be finite; fp32 -- known
//   to be finite, loaded again (out of the L2: holding 32 registers across the rendezvous spilled 20 of the kernel's 64).
template <int MODE, bool EMIT, bool SRC32 = false, bool PRE = false>
__device__ __forceinline__ bool tc_tile_fast(const uint8_t* tsrc, float scale, float rcp, uint32_t qtail, uint32_t dtail, uint32_t pair_m1,
                                             uint32_t lane, uint32_t& first_ss, uint32_t& last_ss, uint32_t& n_starts, const u32x4* pre = nullptr)
{
    uint4 raw[4];
    u32x4 rf[SRC32 ? 8 : 1];
    if (PRE && !SRC32 && kTcPreRegs) {
#pragma unroll
        for (int j = 0; j < 4; ++j) raw[j] = make_uint4(pre[j].x, pre[j].y, pre[j].z, pre[j].w);
    } else if (SRC32) {
        uint32_t m = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            rf[j] = __builtin_nontemporal_load((const u32x4 __attribute__((address_space(1)))*)(reinterpret_cast<uintptr_t>(tsrc + 4ull * (512u * (j >> 1) + 8u * lane) + 16u * (j & 1))));
            if (!PRE) m = umax(m, umax(umax(rf[j].x & 0x7FFFFFFFu, rf[j].y & 0x7FFFFFFFu), umax(rf[j].z & 0x7FFFFFFFu, rf[j].w & 0x7FFFFFFFu)));
        }
        if (!PRE && lane63(wave_incl_max(m)) >= 0x7F800000u) return false;  // wave-uniform: inf / NaN in the tile
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const u32x4 v = __builtin_nontemporal_load((const u32x4 __attribute__((address_space(1)))*)(reinterpret_cast<uintptr_t>(tsrc + 2ull * (512u * j + 8u * lane))));
            raw[j] = make_uint4(v.x, v.y, v.z, v.w);
        }
        if (!PRE && absmax_bits(raw) >= 0x7C00u) return false;      // wave-uniform
    }
    uint32_t mcarry = 0, icarry = 0, first = 0;
    bool prev_sparse = false, failed = false;
    const bool fast_div = SRC32 && __builtin_amdgcn_readfirstlane(scale_in_fast_div_range(scale) ? 1u : 0u) != 0u;      // (one scale per tensor)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const uint32_t p0 = 512u * j + 8u * lane;
        uint32_t q[8];
        if (SRC32) {
            if (fast_div) quantize8_f32<MODE, true>(rf[SRC32 ? 2 * j : 0], rf[SRC32 ? 2 * j + 1 : 0], scale, rcp, q);
            else          quantize8_f32<MODE, false>(rf[SRC32 ? 2 * j : 0], rf[SRC32 ? 2 * j + 1 : 0], scale, rcp, q);
        }
        else       quantize8<MODE>(raw[j], scale, rcp, q);
        const uint32_t prevq = wave_shr1(q[7], qtail);
        qtail = lane63(q[7]);
        uint32_t d[8];
        d[0] = sub_bytes(q[0], prevq);
#pragma unroll
        for (int k = 1; k < 8; ++k) d[k] = sub_bytes(q[k], q[k - 1]);
        const uint32_t prevd = wave_shr1(d[7], dtail);
        dtail = lane63(d[7]);
        bool st[8];
        uint32_t mask = 0;                                          // bit 7-k = element k starts a run
        unsigned long long sm[8];
        st[0] = d[0] != prevd;
        sm[0] = __builtin_amdgcn_ballot_w64(st[0]);
        shift_in(mask, sm[0]);
#pragma unroll
        for (int k = 1; k < 8; ++k) { st[k] = d[k] != d[k - 1]; sm[k] = __builtin_amdgcn_ballot_w64(st[k]); shift_in(mask, sm[k]); }
        const uint32_t cnt = static_cast<uint32_t>(__builtin_popcount(mask));
        const uint32_t lm = mask ? p0 + 8u - static_cast<uint32_t>(__builtin_ctz(mask)) : 0u;          // last start of the lane, position + 1
        const unsigned long long have = __ballot(mask != 0u);
        if (!first && have) {                                       // wave-uniform: the tile's first start
            const uint32_t fl = static_cast<uint32_t>(__builtin_ctzll(have));
            const uint32_t lane_first = mask ? p0 + 8u - (31u - static_cast<uint32_t>(__builtin_clz(mask))) : 0u;     // first start of the lane, position + 1
            first = static_cast<uint32_t>(__shfl(static_cast<int>(lane_first), static_cast<int>(fl)));
        }
        const bool sparse = __popcll(~have) >= 14;                  // wave-uniform (see encode_rle_fast)
        const bool suspicious = sparse || prev_sparse;
        prev_sparse = sparse;
        const uint32_t ic = wave_incl_add(cnt);
        uint32_t idx = icarry + ic - cnt;
        icarry += lane63(ic);
        const uint32_t im = wave_incl_max(lm);
        const uint32_t m = umax(wave_shr1(im, 0u), mcarry);         // last start before this lane, position + 1 (0: none in the tile yet)
        mcarry = umax(mcarry, lane63(im));
        if (suspicious && __ballot(mask != 0u && m != 0u && p0 + 8u - m > 255u) != 0ull) failed = true;
        if (EMIT) {
            uint32_t rel = m - p0 - 1u;
            uint32_t addr[8], cntv[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                addr[k] = lshl1_add(idx, pair_m1);
                cntv[k] = static_cast<uint32_t>(k) - rel;
                rel = st[k] ? static_cast<uint32_t>(k) : rel;
                add_pred(idx, sm[k]);
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) store_pair_if(sm[k], addr[k], cntv[k], d[k]);
        }
    }
    if (failed || (mcarry != 0u && kTile + 1u - mcarry > 255u)) return false;     // a run inside the tile may need splitting
    first_ss = first; last_ss = mcarry; n_starts = icarry;
    return true;
}

// The same for tiles with stretches of ANY length (round 4): the SPLIT form of the block encoder (kernels.hip, encode_rle_fast<.., true>
// -- a run also starts at every 255th element of a stretch; at most one such element falls into the part of a lane in front of its
// first change of the delta) with the stretch that ENTERS the tile: `lead_phase` = offset of the tile's element 0 in it (mod 255),
// known once chain 1 has answered.  Used twice on a tile the plain form declines: EMIT = false, has_lead = false for the summary
// (first / last change of the delta, run starts at or behind the first change -- the head in front of it belongs to the entering
// stretch and is counted by arithmetic), then EMIT = true with the phase for the pairs (the tile is read again, from L2).  Tiles
// inside long stretches took the element-wise loop for both: a tensor of long runs compressed 4 x slower than noise.
//   first_change / last_change: tile-relative + 1, 0 = none; n_starts: run starts found; last_start: the last of them (+ 1).
// Out of line: inlined three times into k_tc_fused it made the kernel a quarter slower on tensors that never get here.
template <int MODE, bool EMIT>
__device__ __noinline__ bool tc_tile_split(const uint8_t* tsrc, float scale, float rcp, uint32_t qtail, uint32_t dtail, uint32_t pair_m1,
                                              uint32_t lane, bool has_lead, uint32_t lead_phase, uint32_t& first_change, uint32_t& last_change,
                                              uint32_t& n_starts, uint32_t& last_start)
{
    // (registers: the kernel this is called from must stay at 64 VGPRs -- two 16-wave workgroups per CU -- and a callee's
    // count is the kernel's: the tile is read chunk by chunk, a first time for the finiteness test alone)
    auto load_chunk = [&](int j) {
        const u32x4 v = __builtin_nontemporal_load((const u32x4 __attribute__((address_space(1)))*)(reinterpret_cast<uintptr_t>(tsrc + 2ull * (512u * j + 8u * lane))));
        return make_uint4(v.x, v.y, v.z, v.w);
    };
    {
        u16x2 m2 = {0, 0};
#pragma unroll 1
        for (int j = 0; j < 4; ++j) {
            const uint4 r = load_chunk(j);
            const uint32_t w4[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
            for (int t = 0; t < 4; ++t) m2 = __builtin_elementwise_max(m2, __builtin_bit_cast(u16x2, w4[t] & 0x7FFF7FFFu));
        }
        const uint32_t lanemax = m2.x > m2.y ? m2.x : m2.y;
        if (lane63(wave_incl_max(lanemax)) >= 0x7C00u) return false;    // wave-uniform
    }
    constexpr uint32_t kBias = 256u;                                 // change positions are kept as position + 1 + kBias: the entering
    uint32_t ccarry = has_lead ? 1u + kBias - lead_phase : 0u;       // stretch starts at 1 - lead_phase (<= 1); 0 = no stretch yet
    uint32_t mcarry = 0, icarry = 0, first = 0, lcarry = 0;        // lcarry: the tile's own last change (biased; 0: none)
#pragma unroll 1
    for (int j = 0; j < 4; ++j) {
        const uint32_t p0 = 512u * j + 8u * lane;
        uint32_t q[8];
        quantize8<MODE>(load_chunk(j), scale, rcp, q);
        const uint32_t prevq = wave_shr1(q[7], qtail);
        qtail = lane63(q[7]);
        uint32_t d[8];
        d[0] = sub_bytes(q[0], prevq);
#pragma unroll
        for (int k = 1; k < 8; ++k) d[k] = sub_bytes(q[k], q[k - 1]);
        const uint32_t prevd = wave_shr1(d[7], dtail);
        dtail = lane63(d[7]);
        bool st[8];
        st[0] = d[0] != prevd;
#pragma unroll
        for (int k = 1; k < 8; ++k) st[k] = d[k] != d[k - 1];
        uint32_t cmask = 0;                                          // the changes alone (bit 7-k = element k)
#pragma unroll
        for (int k = 0; k < 8; ++k) shift_in(cmask, __builtin_amdgcn_ballot_w64(st[k]));
        const unsigned long long have_c = __ballot(cmask != 0u);
        if (!first && have_c) {                                      // wave-uniform: the tile's first change
            const uint32_t fl = static_cast<uint32_t>(__builtin_ctzll(have_c));
            const uint32_t lane_first = cmask ? p0 + 8u - (31u - static_cast<uint32_t>(__builtin_clz(cmask))) : 0u;
            first = static_cast<uint32_t>(__shfl(static_cast<int>(lane_first), static_cast<int>(fl)));
        }
        const uint32_t lmc = cmask ? p0 + 8u - static_cast<uint32_t>(__builtin_ctz(cmask)) + kBias : 0u;
        const uint32_t imc = wave_incl_max(lmc);
        const uint32_t mc = umax(wave_shr1(imc, 0u), ccarry);        // start of the stretch entering the lane (biased; 0: none)
        ccarry = umax(ccarry, lane63(imc));
        lcarry = umax(lcarry, lane63(imc));
        const uint32_t o0 = p0 + 1u + kBias - mc;                    // offset of the lane's element 0 in that stretch (< 4096)
        const uint32_t r = o0 - 255u * ((o0 * 0x8081u) >> 23);
        uint32_t ks = r ? 255u - r : 0u;
        const uint32_t kfirst = cmask ? static_cast<uint32_t>(__builtin_clz(cmask)) - 24u : 8u;
        if (mc == 0u || ks >= kfirst) ks = 8u;
#pragma unroll
        for (int k = 0; k < 8; ++k) st[k] = st[k] || ks == static_cast<uint32_t>(k);
        uint32_t mask = 0;
        unsigned long long sm[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) { sm[k] = __builtin_amdgcn_ballot_w64(st[k]); shift_in(mask, sm[k]); }
        const uint32_t cnt = static_cast<uint32_t>(__builtin_popcount(mask));
        const uint32_t lm = mask ? p0 + 8u - static_cast<uint32_t>(__builtin_ctz(mask)) : 0u;
        const uint32_t ic = wave_incl_add(cnt);
        uint32_t idx = icarry + ic - cnt;
        icarry += lane63(ic);
        const uint32_t im = wave_incl_max(lm);
        const uint32_t m = umax(wave_shr1(im, 0u), mcarry);         // last start before this lane, position + 1 (0: none in the tile yet)
        mcarry = umax(mcarry, lane63(im));
        if (EMIT) {
            uint32_t rel = m - p0 - 1u;
            uint32_t addr[8], cntv[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                addr[k] = lshl1_add(idx, pair_m1);
                cntv[k] = static_cast<uint32_t>(k) - rel;
                rel = st[k] ? static_cast<uint32_t>(k) : rel;
                add_pred(idx, sm[k]);
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) store_pair_if(sm[k], addr[k], cntv[k], d[k]);
        }
    }
    first_change = first;
    last_change = lcarry ? lcarry - kBias : 0u;
    n_starts = icarry;
    last_start = mcarry;
    return true;
}

// One tile, one wave: the loop of the block encoder's general path with carries that may come from other tiles.
// EMIT = false: the tile's summary.  EMIT = true: the tile's pairs into its 4 KiB slot of the pair scratch.
// PRE (summary pass of fp16 sources): a tile the fast path takes is EMITTED already here -- its pairs depend on nothing outside
// the tile and the two elements in front of it, except the count of its last pair (it ends at the next run start, which the
// scan finds) -- and flagged in pre_flags; the emit pass then only stores that one byte for it, unless a stretch entering the
// tile starts a run inside it (TcCarry::runs != the tile's own starts), which is the emit pass's business as before.  For data
// that does not compress the second pass over the source disappears: summary 19 + emit 28 us -> 29 + 4 for 32 Mi elements.
template <int MODE, bool F32, bool EMIT, bool PRE = false>
__global__ __launch_bounds__(64 * kTcWaves) void k_tc_tiles(const void* __restrict__ src, uint64_t n, const uint32_t* __restrict__ absmax_bits,
                                                            TcSummary* __restrict__ summ, const TcCarry* __restrict__ carry,
                                                            uint8_t* __restrict__ pair_scratch, uint8_t* __restrict__ pre_flags)
{
    static_assert(!(EMIT && PRE) && !(PRE && F32), "PRE is the summary pass of fp16 sources");
    __shared__ __attribute__((aligned(16))) uint8_t lds[(EMIT || PRE) ? kTcWaves * (kTcLead + 2 * kTile) : 16];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint64_t tile = static_cast<uint64_t>(blockIdx.x) * kTcWaves + wave;
    const uint64_t t0 = tile * kTile;
    if (t0 >= n) return;
    const uint32_t len = static_cast<uint32_t>((n - t0 < kTile) ? (n - t0) : kTile);
    if (EMIT && pre_flags && pre_flags[tile]) {                      // wave-uniform: emitted by the summary pass
        const TcCarry c0 = carry[tile];
        const TcSummary s0 = summ[tile];
        if (c0.runs == s0.cnt_b) {
            if (s0.cnt_b && lane == 0u)                             // the last pair ends where the next run starts
                pair_scratch[tile * (2ull * kTile) + 2ull * s0.cnt_b - 1u] = static_cast<uint8_t>(c0.next_run - (t0 + s0.last_b - 1u));
            return;
        }
    }
    const float scale = tc_scale(*absmax_bits);
    // the two elements in front of the tile give q[t0-1] and d[t0-1]
    uint32_t qtail = 0, dtail = 0;
    if (t0 >= 1) {
        qtail = quantize<MODE>(tc_load<F32>(src, t0 - 1), scale);
        const uint32_t q2 = (t0 >= 2) ? quantize<MODE>(tc_load<F32>(src, t0 - 2), scale) : 0u;
        dtail = (qtail - q2) & 0xFFu;
    }
    TcCarry cy{};
    uint8_t* wl = nullptr;
    uint32_t pair_addr = 0;
    if (EMIT) cy = carry[tile];
    if (EMIT || PRE) {
        wl = lds + wave * (kTcLead + 2 * kTile);
        pair_addr = static_cast<uint32_t>(reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) uint8_t*)(wl + kTcLead)));
    }
    uint32_t scarry = 0;        // last stretch start inside the tile so far (rel + 1)
    uint32_t mcarry = 0;        // last run start inside the tile so far (rel + 1)
    uint32_t icarry = 0;        // run starts so far
    uint32_t first_ss = 0;
    bool fast = false;
    if (!F32 && len == kTile && ((reinterpret_cast<uintptr_t>(src) + 2ull * t0) & 15u) == 0u && !kTcNoFast) {
        // whole fp16 tiles: eight elements per lane and step (tc_tile_fast); everything else, and whatever it declines, element-wise below
        uint32_t f_first = 0, f_last = 0, f_n = 0;
        const uint32_t pair_m1 = (EMIT || PRE) ? pair_addr - 1u : 0u;
        if (tc_tile_fast<MODE, EMIT || PRE>(static_cast<const uint8_t*>(src) + 2ull * t0, scale, 1.0f / scale, qtail, t0 == 0u ? 0x100u : dtail, pair_m1, lane,
                                            f_first, f_last, f_n) &&
            (!EMIT || cy.runs == f_n)) {                            // (emit: no run start of an entering stretch inside the tile)
            first_ss = f_first; scarry = f_last; mcarry = f_last; icarry = f_n;
            fast = true;
        }
    }
    if (PRE) {
        if (fast) {                                                 // the tile's pairs, all but the count of the last one
            wave_lds_fence();
            uint8_t* dst = pair_scratch + tile * (2ull * kTile);
            const uint32_t bytes = 2u * icarry;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t b = 1024u * j + 16u * lane;
                if (b < bytes) *reinterpret_cast<uint4*>(dst + b) = *reinterpret_cast<const uint4*>(wl + kTcLead + b);
            }
        }
        if (lane == 0u) pre_flags[tile] = fast ? 1u : 0u;
    }
#pragma unroll 1
    for (uint32_t step = 0; step < kTile / 64u && !fast; ++step) {
        const uint32_t rel = 64u * step + lane;
        if (64u * step >= len) break;                               // wave-uniform
        const bool live = rel < len;
        const uint32_t qv = live ? quantize<MODE>(tc_load<F32>(src, t0 + rel), scale) : 0u;
        const uint32_t prevq = wave_shr1(qv, qtail);
        qtail = lane63(qv);
        const uint32_t d = (qv - prevq) & 0xFFu;
        const uint32_t prevd = wave_shr1(d, dtail);
        dtail = lane63(d);
        const bool neq = live && ((t0 + rel == 0u) || (d != prevd));
        const unsigned long long nb = __ballot(neq);
        if (!first_ss && nb) first_ss = 64u * step + static_cast<uint32_t>(__builtin_ctzll(nb)) + 1u;
        const uint32_t is = wave_incl_max(neq ? rel + 1u : 0u);
        const uint32_t ss = umax(is, scarry);                       // this element's stretch start, rel + 1 (0: before the tile)
        scarry = umax(scarry, lane63(is));
        bool isrun;
        if (ss) isrun = live && ((rel + 1u - ss) % 255u) == 0u;
        else    isrun = EMIT && live && ((cy.lead_phase + rel) % 255u) == 0u;      // continuation of a stretch of earlier tiles
        const uint32_t ic = wave_incl_add(isrun ? 1u : 0u);
        const uint32_t idx = icarry + ic - (isrun ? 1u : 0u);
        icarry += lane63(ic);
        const uint32_t im = wave_incl_max(isrun ? rel + 1u : 0u);
        const uint32_t prev = umax(wave_shr1(im, 0u), mcarry);      // previous run start inside the tile (rel + 1, 0 = none)
        mcarry = umax(mcarry, lane63(im));
        if (EMIT && isrun) {
            // value byte of this pair, count byte of the previous one (the first pair of the tile writes into the lead bytes)
            *reinterpret_cast<__attribute__((address_space(3))) uint8_t*>(static_cast<uintptr_t>(pair_addr + 2u * idx - 1u)) = static_cast<uint8_t>(rel + 1u - prev);
            *reinterpret_cast<__attribute__((address_space(3))) uint8_t*>(static_cast<uintptr_t>(pair_addr + 2u * idx)) = static_cast<uint8_t>(d);
        }
    }
    if (!EMIT) {
        if (lane == 0u) summ[tile] = TcSummary{first_ss, scarry, icarry, mcarry};
        return;
    }
    // close the tile's last pair: it ends where the next run starts (in a later tile, or at n)
    if (icarry && lane == 0u) {
        const uint64_t last_abs = t0 + mcarry - 1u;
        *reinterpret_cast<__attribute__((address_space(3))) uint8_t*>(static_cast<uintptr_t>(pair_addr + 2u * icarry - 1u)) =
            static_cast<uint8_t>(cy.next_run - last_abs);
    }
    wave_lds_fence();
    uint8_t* dst = pair_scratch + tile * (2ull * kTile);
    const uint32_t bytes = 2u * icarry;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const uint32_t b = 1024u * j + 16u * lane;
        if (b < bytes) *reinterpret_cast<uint4*>(dst + b) = *reinterpret_cast<const uint4*>(wl + kTcLead + b);
    }
}

// One wave over all tiles, 64 per step: what enters each tile from the left (stretch start, run count) and from the
// right (next run start).  Also the tensor's scale and its stream length.
__global__ __launch_bounds__(64) void k_tc_scan(const TcSummary* __restrict__ summ, TcCarry* __restrict__ carry, uint64_t n_tiles, uint64_t n,
                                               const uint32_t* __restrict__ absmax_bits, float* __restrict__ out_scale,
                                               uint64_t* __restrict__ out_bytes, uint64_t* __restrict__ first_run_tmp)
{
    const uint32_t lane = threadIdx.x;
    uint64_t ss_carry = 0;              // last stretch start before the step (absolute position + 1)
    uint64_t run_carry = 0;
    for (uint64_t base = 0; base < n_tiles; base += 64u) {
        const uint64_t t = base + lane;
        const bool live = t < n_tiles;
        TcSummary s{0u, 0u, 0u, 0u};
        if (live) s = summ[t];
        const uint64_t t0 = t * kTile;
        const uint32_t len = live ? static_cast<uint32_t>((n - t0 < kTile) ? (n - t0) : kTile) : 0u;
        // stretch start entering the tile: the last one of an earlier tile of this step, else of earlier steps
        const uint32_t enc = s.last_ss ? lane * kTile + s.last_ss : 0u;
        const uint32_t inc = wave_incl_max(enc);
        const uint32_t exc = wave_shr1(inc, 0u);
        const uint64_t ss_in = exc ? base * kTile + exc : ss_carry;             // absolute position + 1 (0 only for tile 0)
        const uint32_t tot = lane63(inc);
        if (tot) ss_carry = base * kTile + tot;
        // run starts in front of the tile's first stretch start: every 255 elements of the entering stretch
        uint32_t lead_phase = 0, cnt_a = 0, first_a = 0;
        const uint32_t f_end = s.first_ss ? s.first_ss - 1u : len;
        if (live && ss_in && f_end) {
            lead_phase = static_cast<uint32_t>((t0 - (ss_in - 1u)) % 255u);
            first_a = (255u - lead_phase) % 255u;
            if (first_a < f_end) cnt_a = (f_end - 1u - first_a) / 255u + 1u;
        }
        const uint32_t runs = cnt_a + s.cnt_b;
        const uint32_t rinc = wave_incl_add(runs);
        if (live) {
            TcCarry c;
            c.lead_phase = lead_phase;
            c.runs = runs;
            c.run_base = run_carry + rinc - runs;
            c.next_run = n;
            carry[t] = c;
            first_run_tmp[t] = cnt_a ? t0 + first_a + 1u : (s.cnt_b ? t0 + s.first_ss : 0u);   // absolute position + 1
        }
        run_carry += lane63(rinc);
    }
    if (lane == 0u) {
        carry[n_tiles].run_base = run_carry;                         // sentinel for the pack pass
        carry[n_tiles].runs = 0; carry[n_tiles].lead_phase = 0; carry[n_tiles].next_run = n;
        *out_bytes = 2ull * run_carry;
        *out_scale = tc_scale(*absmax_bits);
    }
    __threadfence();
    // right to left: the first run start behind each tile.  Lane l takes tile base + 63 - l, so "nearest later tile" is
    // "nearest earlier lane" and the forward max-scan applies (encoding: larger = nearer).
    uint64_t next_carry = n;
    const uint64_t steps = (n_tiles + 63u) / 64u;
    for (uint64_t sidx = steps; sidx-- > 0;) {
        const uint64_t base = sidx * 64u;
        const uint64_t t = base + 63u - lane;
        const bool live = t < n_tiles;
        const uint64_t fr = live ? first_run_tmp[t] : 0u;           // absolute position + 1
        const uint32_t span = 64u * kTile;
        const uint32_t enc = fr ? span - static_cast<uint32_t>(fr - 1u - base * kTile) : 0u;    // position relative to the step, reversed
        const uint32_t inc = wave_incl_max(enc);
        const uint32_t exc = wave_shr1(inc, 0u);
        if (live) carry[t].next_run = exc ? base * kTile + (span - exc) : next_carry;
        const uint32_t tot = lane63(inc);
        if (tot) next_carry = base * kTile + (span - tot);
    }
}

// The same scan by ONE workgroup of 16 waves (the single wave above walks 16 384 tiles -- 32 Mi elements -- in 2 x 256
// dependent steps: 227 us, a third of the whole compress).  A step (64 tiles) is scanned locally by whichever wave it falls
// to; what crosses steps -- the stretch start entering a step, the run count before it, the first run start behind it --
// is a scan over per-step totals in LDS, done by wave 0 between the phases.  Up to kScanMaxSteps steps (256 Mi elements);
// longer tensors take the single-wave kernel.
// What bounds it now (27 us for 16 384 tiles) is the instruction issue of the ONE CU it runs on: phase 3 is ~250 instructions
// per step (six wave scans, four 64-bit shuffles), 16 steps per wave, four waves per SIMD at 4 clocks per instruction = 64 k
// clocks.  Its three per-tile phases are independent across steps: k_tc_scan_a / _b / _c below run them as grids of their own
// (three launches, ~8 us together); this kernel stays as the SPECKV_TC_SCAN=wg form (tests run all three).
constexpr uint32_t kScanWaves = 16, kScanMaxSteps = 2048;
__device__ __forceinline__ uint64_t shfl64(uint64_t v, uint32_t src)
{
    const uint32_t lo = __shfl(static_cast<uint32_t>(v), static_cast<int>(src)), hi = __shfl(static_cast<uint32_t>(v >> 32), static_cast<int>(src));
    return (static_cast<uint64_t>(hi) << 32) | lo;
}
__device__ __forceinline__ unsigned long long lanes_above(uint32_t lane) { return lane == 63u ? 0ull : ~((2ull << lane) - 1ull); }

__global__ __launch_bounds__(64 * kScanWaves) void k_tc_scan_wg(const TcSummary* __restrict__ summ, TcCarry* __restrict__ carry, uint64_t n_tiles, uint64_t n,
                                                                const uint32_t* __restrict__ absmax_bits, float* __restrict__ out_scale,
                                                                uint64_t* __restrict__ out_bytes)
{
    __shared__ uint64_t s_ss[kScanMaxSteps];        // last stretch start of the step (absolute + 1) -> stretch start entering the step
    __shared__ uint64_t s_runs[kScanMaxSteps];      // run starts in the step -> run starts before the step
    __shared__ uint64_t s_first[kScanMaxSteps];     // first run start of the step (absolute + 1, 0 none) -> first run start behind the step (absolute, n if none)
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t n_steps = static_cast<uint32_t>((n_tiles + 63u) / 64u);
    constexpr uint64_t kOpen = ~0ull;
    // phase 1: the last stretch start of every step
    // (in every phase a wave takes its steps eight at a time and issues their loads together: step after step the kernel was a
    // chain of load latencies)
    for (uint32_t st0 = wave; st0 < n_steps; st0 += 8u * kScanWaves) {
        uint32_t lb[8];
#pragma unroll
        for (uint32_t u = 0; u < 8u; ++u) {
            const uint64_t t = static_cast<uint64_t>(st0 + u * kScanWaves) * 64u + lane;
            lb[u] = (st0 + u * kScanWaves < n_steps && t < n_tiles) ? summ[t].last_ss : 0u;
        }
#pragma unroll
        for (uint32_t u = 0; u < 8u; ++u) {
            const uint32_t st = st0 + u * kScanWaves;
            if (st >= n_steps) break;
            const uint32_t tot = lane63(wave_incl_max(lb[u] ? lane * kTile + lb[u] : 0u));
            if (lane == 0u) s_ss[st] = tot ? static_cast<uint64_t>(st) * 64u * kTile + tot : 0ull;
        }
    }
    __syncthreads();
    if (wave == 0u) {                               // phase 2: "last non-zero before" over the steps
        uint64_t cy = 0;
        for (uint32_t j = 0; j < n_steps; j += 64u) {
            const uint32_t i = j + lane;
            const uint64_t v = i < n_steps ? s_ss[i] : 0ull;
            const uint32_t idx = wave_incl_max(v ? lane + 1u : 0u);
            const uint32_t ex = wave_shr1(idx, 0u), last = lane63(idx);
            const uint64_t got = shfl64(v, ex ? ex - 1u : 0u), top = shfl64(v, last ? last - 1u : 0u);
            if (i < n_steps) s_ss[i] = ex ? got : cy;
            if (last) cy = top;
        }
    }
    __syncthreads();
    // phase 3: per tile what depends on the entering stretch start; local scans of the run counts and of the first run starts
    for (uint32_t st0 = wave; st0 < n_steps; st0 += 8u * kScanWaves) {
      TcSummary sbuf[8];
#pragma unroll
      for (uint32_t u = 0; u < 8u; ++u) {
          const uint64_t t = static_cast<uint64_t>(st0 + u * kScanWaves) * 64u + lane;
          sbuf[u] = TcSummary{0u, 0u, 0u, 0u};
          if (st0 + u * kScanWaves < n_steps && t < n_tiles) sbuf[u] = summ[t];
      }
#pragma unroll
      for (uint32_t u = 0; u < 8u; ++u) {
        const uint32_t st = st0 + u * kScanWaves;
        if (st >= n_steps) break;
        const uint64_t base = static_cast<uint64_t>(st) * 64u, t = base + lane;
        const bool live = t < n_tiles;
        const TcSummary s = sbuf[u];
        const uint64_t t0 = t * kTile;
        const uint32_t len = live ? static_cast<uint32_t>((n - t0 < kTile) ? (n - t0) : kTile) : 0u;
        const uint32_t inc = wave_incl_max(s.last_ss ? lane * kTile + s.last_ss : 0u);
        const uint32_t exc = wave_shr1(inc, 0u);
        const uint64_t ss_in = exc ? base * kTile + exc : s_ss[st];
        uint32_t lead_phase = 0, cnt_a = 0, first_a = 0;
        const uint32_t f_end = s.first_ss ? s.first_ss - 1u : len;
        if (live && ss_in && f_end) {
            lead_phase = static_cast<uint32_t>((t0 - (ss_in - 1u)) % 255u);
            first_a = (255u - lead_phase) % 255u;
            if (first_a < f_end) cnt_a = (f_end - 1u - first_a) / 255u + 1u;
        }
        const uint32_t runs = cnt_a + s.cnt_b;
        const uint32_t rinc = wave_incl_add(runs);
        const uint64_t fr = !live ? 0ull : (cnt_a ? t0 + first_a + 1u : (s.cnt_b ? t0 + s.first_ss : 0ull));     // absolute + 1
        const unsigned long long have = __ballot(fr != 0ull);
        const unsigned long long later = have & lanes_above(lane);
        const uint64_t nxt = shfl64(fr, later ? static_cast<uint32_t>(__builtin_ctzll(later)) : 0u);
        if (live) {
            TcCarry c;
            c.lead_phase = lead_phase;
            c.runs = runs;
            c.run_base = rinc - runs;                               // + the step's base in phase 5
            c.next_run = later ? nxt - 1u : kOpen;                  // open: the first run start behind the step (phase 5)
            carry[t] = c;
        }
        const uint64_t first = shfl64(fr, have ? static_cast<uint32_t>(__builtin_ctzll(have)) : 0u);
        if (lane == 0u) { s_runs[st] = lane63(rinc); s_first[st] = have ? first : 0ull; }
      }
    }
    __syncthreads();
    if (wave == 0u) {                               // phase 4: run counts before every step, first run start behind every step
        uint64_t cy = 0;
        for (uint32_t j = 0; j < n_steps; j += 64u) {
            const uint32_t i = j + lane;
            const uint32_t v = i < n_steps ? static_cast<uint32_t>(s_runs[i]) : 0u;      // <= 64 * 2048
            const uint32_t inc = wave_incl_add(v);
            if (i < n_steps) s_runs[i] = cy + inc - v;
            cy += lane63(inc);
        }
        uint64_t behind = n;
        for (uint32_t j = (n_steps + 63u) / 64u; j-- > 0u;) {
            const uint32_t i = j * 64u + lane;
            const uint64_t v = i < n_steps ? s_first[i] : 0ull;
            const unsigned long long have = __ballot(v != 0ull);
            const unsigned long long later = have & lanes_above(lane);
            const uint64_t nxt = shfl64(v, later ? static_cast<uint32_t>(__builtin_ctzll(later)) : 0u);
            const uint64_t first = shfl64(v, have ? static_cast<uint32_t>(__builtin_ctzll(have)) : 0u);
            if (i < n_steps) s_first[i] = later ? nxt - 1u : behind;
            if (have) behind = first - 1u;
        }
        if (lane == 0u) {
            carry[n_tiles].run_base = cy;                            // sentinel for the pack pass
            carry[n_tiles].runs = 0; carry[n_tiles].lead_phase = 0; carry[n_tiles].next_run = n;
            *out_bytes = 2ull * cy;
            *out_scale = tc_scale(*absmax_bits);
        }
    }
    __syncthreads();
    // phase 5: every lane finishes the entry it wrote in phase 3
    for (uint32_t st0 = wave; st0 < n_steps; st0 += 8u * kScanWaves) {
        uint64_t rb[8], nr[8];
#pragma unroll
        for (uint32_t u = 0; u < 8u; ++u) {
            const uint64_t t = static_cast<uint64_t>(st0 + u * kScanWaves) * 64u + lane;
            rb[u] = 0; nr[u] = 0;
            if (st0 + u * kScanWaves < n_steps && t < n_tiles) { rb[u] = carry[t].run_base; nr[u] = carry[t].next_run; }
        }
#pragma unroll
        for (uint32_t u = 0; u < 8u; ++u) {
            const uint32_t st = st0 + u * kScanWaves;
            const uint64_t t = static_cast<uint64_t>(st) * 64u + lane;
            if (st < n_steps && t < n_tiles) {
                carry[t].run_base = rb[u] + s_runs[st];
                if (nr[u] == kOpen) carry[t].next_run = s_first[st];
            }
        }
    }
}

// The same scan as three grids, a wave per step of 64 tiles (the one-workgroup form above is bound by the instruction issue of
// the single CU it runs on: 27 us for 16 384 tiles): what crosses steps is short enough for every wave to look up itself --
// the last stretch start in front of its step (A -> B), the run starts in front of it and the first run start behind it
// (B -> C).  Step arrays in global memory: step_ss / step_runs / step_first, n_steps entries each.
__global__ __launch_bounds__(256) void k_tc_scan_a(const TcSummary* __restrict__ summ, uint64_t n_tiles, uint64_t* __restrict__ step_ss)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t st = static_cast<uint64_t>(blockIdx.x) * 4u + (threadIdx.x >> 6);
    if (st * 64u >= n_tiles) return;
    const uint64_t t = st * 64u + lane;
    const uint32_t last_ss = t < n_tiles ? summ[t].last_ss : 0u;
    const uint32_t tot = lane63(wave_incl_max(last_ss ? lane * kTile + last_ss : 0u));
    if (lane == 0u) step_ss[st] = tot ? st * 64u * kTile + tot : 0ull;      // last stretch start of the step (absolute + 1), 0: none
}
__global__ __launch_bounds__(256) void k_tc_scan_b(const TcSummary* __restrict__ summ, TcCarry* __restrict__ carry, uint64_t n_tiles, uint64_t n,
                                                  const uint64_t* __restrict__ step_ss, uint64_t* __restrict__ step_runs, uint64_t* __restrict__ step_first)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t st = static_cast<uint64_t>(blockIdx.x) * 4u + (threadIdx.x >> 6);
    if (st * 64u >= n_tiles) return;
    constexpr uint64_t kOpen = ~0ull;
    // the stretch start entering the step: the last non-zero step_ss in front of it (as a rule the step before)
    uint64_t ss_step = 0;
    for (uint64_t j1 = st; j1 > 0u && ss_step == 0u;) {
        const uint64_t j0 = j1 > 64u ? j1 - 64u : 0u;                 // steps [j0, j1)
        const uint64_t j = j0 + lane;
        const uint64_t v = j < j1 ? step_ss[j] : 0ull;
        const unsigned long long have = __ballot(v != 0ull);
        if (have) ss_step = shfl64(v, 63u - static_cast<uint32_t>(__builtin_clzll(have)));
        j1 = j0;
    }
    const uint64_t base = st * 64u, t = base + lane;
    const bool live = t < n_tiles;
    TcSummary s{0u, 0u, 0u, 0u};
    if (live) s = summ[t];
    const uint64_t t0 = t * kTile;
    const uint32_t len = live ? static_cast<uint32_t>((n - t0 < kTile) ? (n - t0) : kTile) : 0u;
    const uint32_t inc = wave_incl_max(s.last_ss ? lane * kTile + s.last_ss : 0u);
    const uint32_t exc = wave_shr1(inc, 0u);
    const uint64_t ss_in = exc ? base * kTile + exc : ss_step;
    uint32_t lead_phase = 0, cnt_a = 0, first_a = 0;
    const uint32_t f_end = s.first_ss ? s.first_ss - 1u : len;
    if (live && ss_in && f_end) {
        lead_phase = static_cast<uint32_t>((t0 - (ss_in - 1u)) % 255u);
        first_a = (255u - lead_phase) % 255u;
        if (first_a < f_end) cnt_a = (f_end - 1u - first_a) / 255u + 1u;
    }
    const uint32_t runs = cnt_a + s.cnt_b;
    const uint32_t rinc = wave_incl_add(runs);
    const uint64_t fr = !live ? 0ull : (cnt_a ? t0 + first_a + 1u : (s.cnt_b ? t0 + s.first_ss : 0ull));     // absolute + 1
    const unsigned long long have = __ballot(fr != 0ull);
    const unsigned long long later = have & lanes_above(lane);
    const uint64_t nxt = shfl64(fr, later ? static_cast<uint32_t>(__builtin_ctzll(later)) : 0u);
    if (live) {
        TcCarry c;
        c.lead_phase = lead_phase;
        c.runs = runs;
        c.run_base = rinc - runs;                                   // + the step's base in k_tc_scan_c
        c.next_run = later ? nxt - 1u : kOpen;                      // open: the first run start behind the step (k_tc_scan_c)
        carry[t] = c;
    }
    const uint64_t first = shfl64(fr, have ? static_cast<uint32_t>(__builtin_ctzll(have)) : 0u);
    if (lane == 0u) { step_runs[st] = lane63(rinc); step_first[st] = have ? first : 0ull; }
}
__global__ __launch_bounds__(256) void k_tc_scan_c(TcCarry* __restrict__ carry, uint64_t n_tiles, uint64_t n, const uint64_t* __restrict__ step_runs,
                                                  const uint64_t* __restrict__ step_first, const uint32_t* __restrict__ absmax_bits,
                                                  float* __restrict__ out_scale, uint64_t* __restrict__ out_bytes)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t st = static_cast<uint64_t>(blockIdx.x) * 4u + (threadIdx.x >> 6);
    const uint64_t n_steps = (n_tiles + 63u) / 64u;
    if (st >= n_steps) return;
    constexpr uint64_t kOpen = ~0ull;
    uint32_t part = 0;                                               // run starts in front of the step (each step <= 64 * 2048; a lane sums <= 32 steps)
    uint64_t wide = 0;
    for (uint64_t j0 = 0; j0 < st; j0 += 64u) {
        const uint64_t j = j0 + lane;
        if (j < st) wide += step_runs[j];
    }
    part = static_cast<uint32_t>(wide & 0xFFFFu);
    const uint64_t before = static_cast<uint64_t>(lane63(wave_incl_add(part))) + (static_cast<uint64_t>(lane63(wave_incl_add(static_cast<uint32_t>(wide >> 16)))) << 16);
    // the first run start behind the step: the first non-zero step_first after it (as a rule the step after), n if none
    uint64_t behind = n;
    bool found = false;
    for (uint64_t j0 = st + 1u; j0 < n_steps && !found; j0 += 64u) {
        const uint64_t j = j0 + lane;
        const uint64_t v = j < n_steps ? step_first[j] : 0ull;
        const unsigned long long have = __ballot(v != 0ull);
        if (have) { behind = shfl64(v, static_cast<uint32_t>(__builtin_ctzll(have))) - 1u; found = true; }
    }
    const uint64_t t = st * 64u + lane;
    if (t < n_tiles) {
        carry[t].run_base += before;
        if (carry[t].next_run == kOpen) carry[t].next_run = behind;
    }
    if (st + 1u == n_steps && lane == 0u) {
        const uint64_t total = before + step_runs[st];
        carry[n_tiles].run_base = total;                             // sentinel for the pack pass
        carry[n_tiles].runs = 0; carry[n_tiles].lead_phase = 0; carry[n_tiles].next_run = n;
        *out_bytes = 2ull * total;
        *out_scale = tc_scale(*absmax_bits);
    }
}

// Output-centric pack: a wave owns 2048 consecutive pairs (4 KiB, line-aligned) of the stream and gathers them from the
// tiles' slots.  The tiles that hold the wave's first and last pair are found by the whole wave at once (64 probes per step of
// a 64-ary search over run_base: 3 steps for 16 384 tiles, where a binary search per lane took 14 dependent loads, four
// times per lane); a lane piece of 8 pairs then searches only between those two tiles -- they are neighbours unless the data
// is sparse -- and walks.
__device__ __forceinline__ uint64_t tc_wave_find_tile(const TcCarry* __restrict__ carry, uint64_t n_tiles, uint64_t target, uint32_t lane)
{
    // last tile whose run_base <= target (carry[n_tiles].run_base = total > target); run_base is non-decreasing
    uint64_t lo = 0, hi = n_tiles;
    while (hi - lo > 1u) {
        const uint64_t span = hi - lo, step = (span + 63u) / 64u;
        const uint64_t c = lo + static_cast<uint64_t>(lane + 1u) * step;
        const bool le = c < hi && carry[c].run_base <= target;
        const uint32_t k = static_cast<uint32_t>(__popcll(__ballot(le)));
        const uint64_t nlo = lo + static_cast<uint64_t>(k) * step, nhi = lo + static_cast<uint64_t>(k + 1u) * step;
        lo = nlo;
        hi = nhi < hi ? nhi : hi;
    }
    return lo;
}
__global__ __launch_bounds__(256) void k_tc_pack(const TcCarry* __restrict__ carry, uint64_t n_tiles, const uint8_t* __restrict__ pair_scratch,
                                                uint8_t* __restrict__ rle)
{
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint64_t total = carry[n_tiles].run_base;
    const uint64_t chunk = (static_cast<uint64_t>(blockIdx.x) * 4u + wave) * kTile;
    if (chunk >= total) return;
    const uint64_t last_pair = (chunk + kTile - 1u < total) ? chunk + kTile - 1u : total - 1u;
    const uint64_t t_first = tc_wave_find_tile(carry, n_tiles, chunk, lane);
    const uint64_t t_last = tc_wave_find_tile(carry, n_tiles, last_pair, lane);
    if (t_last - t_first <= 3u) {
        // the chunk lies in at most four tiles (data that does not compress: three): their bases once per wave, then every
        // lane piece places itself by comparisons -- the four pieces of a lane are independent loads instead of four chains
        // of dependent ones
        uint64_t rb[5];
#pragma unroll
        for (uint32_t q = 0; q < 5u; ++q) rb[q] = carry[(t_first + q <= n_tiles) ? t_first + q : n_tiles].run_base;
        struct __attribute__((packed, aligned(2))) Unaligned16 { u32x4 v; };
        bool slow[4];
        u32x4 x[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint64_t r0 = chunk + 512u * j + 8u * lane;
            slow[j] = false;
            x[j] = u32x4{0u, 0u, 0u, 0u};
            if (r0 >= total) continue;
            uint32_t q = 0;                                           // last of the four tiles whose base is <= r0 (tiles without runs share their successor's base and are skipped)
#pragma unroll
            for (uint32_t k = 1; k < 4u; ++k) if (t_first + k <= t_last && rb[k] <= r0) q = k;
            if (r0 + 8u <= rb[q + 1u] && r0 + 8u <= total)
                x[j] = reinterpret_cast<const Unaligned16*>(pair_scratch + (t_first + q) * (2ull * kTile) + 2ull * (r0 - rb[q]))->v;
            else slow[j] = true;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint64_t r0 = chunk + 512u * j + 8u * lane;
            if (r0 >= total) continue;
            if (slow[j]) {                                            // the piece straddles tiles (or the stream's end): pair by pair
                uint64_t t = t_first;
                while (t < t_last && carry[t + 1].run_base <= r0) ++t;
                uint64_t tb = carry[t].run_base, te = carry[t + 1].run_base;
                uint32_t w[4] = {0u, 0u, 0u, 0u};
                for (int k = 0; k < 8; ++k) {
                    const uint64_t r = r0 + k;
                    if (r < total) {
                        while (r >= te) { ++t; tb = te; te = carry[t + 1].run_base; }
                        const uint32_t pr = *reinterpret_cast<const uint16_t*>(pair_scratch + t * (2ull * kTile) + 2ull * (r - tb));
                        w[k >> 1] |= pr << ((k & 1) * 16);
                    }
                }
                x[j] = u32x4{w[0], w[1], w[2], w[3]};
            }
            *reinterpret_cast<uint4*>(rle + 2ull * r0) = make_uint4(x[j].x, x[j].y, x[j].z, x[j].w);
        }
        return;
    }
#pragma unroll 1
    for (int j = 0; j < 4; ++j) {
        const uint64_t r0 = chunk + 512u * j + 8u * lane;
        if (r0 >= total) continue;
        // last tile in [t_first, t_last] whose run_base <= r0 (tiles without runs share their successor's base and are skipped)
        uint64_t lo = t_first, hi = t_last + 1u;                    // carry[hi].run_base > last_pair >= r0
        while (hi - lo > 1u) {
            const uint64_t mid = (lo + hi) >> 1;
            if (carry[mid].run_base <= r0) lo = mid; else hi = mid;
        }
        uint64_t t = lo;
        uint64_t tb = carry[t].run_base, te = carry[t + 1].run_base;
        if (r0 + 8u <= te && r0 + 8u <= total) {
            // all eight pairs in one tile's slot (all but the piece that straddles a tile boundary, in data that does not
            // compress): one 16-byte load at a 2-byte-aligned address -- the eight 2-byte loads below were most of this pass
            struct __attribute__((packed, aligned(2))) Unaligned16 { u32x4 v; };
            const u32x4 x = reinterpret_cast<const Unaligned16*>(pair_scratch + t * (2ull * kTile) + 2ull * (r0 - tb))->v;
            *reinterpret_cast<uint4*>(rle + 2ull * r0) = make_uint4(x.x, x.y, x.z, x.w);
            continue;
        }
        uint32_t w[4] = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const uint64_t r = r0 + k;
            if (r < total) {
                while (r >= te) { ++t; tb = te; te = carry[t + 1].run_base; }
                const uint32_t pr = *reinterpret_cast<const uint16_t*>(pair_scratch + t * (2ull * kTile) + 2ull * (r - tb));
                w[k >> 1] |= pr << ((k & 1) * 16);
            }
        }
        *reinterpret_cast<uint4*>(rle + 2ull * r0) = make_uint4(w[0], w[1], w[2], w[3]);
    }
}

// ---------------------------------------------------------------- decompress
struct TdSummary { uint32_t sum_c, sum_v; };           // of one chunk of 2048 pairs: elements it emits, sum of value*count mod 256
struct TdCarry { uint64_t start; uint32_t q_pre, pad; }; // elements / int8 prefix in front of the chunk

__global__ __launch_bounds__(256) void k_td_summary(const uint8_t* __restrict__ rle, uint64_t n_pairs, TdSummary* __restrict__ summ)
{
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint64_t chunk = static_cast<uint64_t>(blockIdx.x) * 4u + wave;
    const uint64_t p0 = chunk * kTile;
    if (p0 >= n_pairs) return;
    uint32_t sc = 0, sv = 0;
    // 8 pairs (16 bytes) per lane and step where the stream allows it (the chunk starts 4 KiB into the stream: aligned with it)
    const bool wide = (reinterpret_cast<uintptr_t>(rle) & 15u) == 0u;
#pragma unroll 1
    for (uint32_t step = 0; step < kTile / 512u; ++step) {
        const uint64_t i = p0 + 512u * step + 8u * lane;
        if (wide && i + 8u <= n_pairs) {
            const u32x4 v = *reinterpret_cast<const u32x4*>(rle + 2ull * i);      // (temporal: the expand pass reads the stream again)
            const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int t = 0; t < 4; ++t) {                               // dword = v0 | c0 << 8 | v1 << 16 | c1 << 24
                const uint32_t counts = (w[t] >> 8) & 0x00FF00FFu;
                sc = __builtin_amdgcn_udot4(counts, 0x00010001u, sc, false);
                sv = __builtin_amdgcn_udot4(w[t], counts, sv, false);  // v0 c0 + v1 c1
            }
        } else {
            for (uint32_t k = 0; k < 8u; ++k) {
                uint32_t bits = 0;
                if (i + k < n_pairs) bits = *reinterpret_cast<const uint16_t*>(rle + 2ull * (i + k));
                sc += bits >> 8;
                sv += (bits & 0xFFu) * (bits >> 8);
            }
        }
    }
    sc = lane63(wave_incl_add(sc));
    sv = lane63(wave_incl_add(sv & 0xFFu));
    if (lane == 0u) summ[chunk] = TdSummary{sc, sv & 0xFFu};
}

__global__ __launch_bounds__(64) void k_td_scan(const TdSummary* __restrict__ summ, TdCarry* __restrict__ carry, uint64_t n_chunks,
                                               uint64_t cap, uint64_t* __restrict__ out_n)
{
    const uint32_t lane = threadIdx.x;
    uint64_t start = 0;
    uint32_t qpre = 0;
    for (uint64_t base = 0; base < n_chunks; base += 64u) {
        const uint64_t c = base + lane;
        TdSummary s{0u, 0u};
        if (c < n_chunks) s = summ[c];
        // 64 x 2048 x 255 elements per step do not fit 32 bits: scan the two halves of the count separately
        const uint32_t lo = wave_incl_add(s.sum_c & 0xFFFFu), hi = wave_incl_add(s.sum_c >> 16);
        const uint64_t inc = static_cast<uint64_t>(lo) + (static_cast<uint64_t>(hi) << 16);
        const uint32_t vinc = wave_incl_add(s.sum_v);
        if (c < n_chunks) carry[c] = TdCarry{start + inc - s.sum_c, (qpre + vinc - s.sum_v) & 0xFFu, 0u};
        start += static_cast<uint64_t>(lane63(lo)) + (static_cast<uint64_t>(lane63(hi)) << 16);
        qpre = (qpre + lane63(vinc)) & 0xFFu;
    }
    if (lane == 0u) {
        carry[n_chunks] = TdCarry{start, qpre, 0u};
        *out_n = start < cap ? start : cap;                          // elements written (the stream's total, clipped at the buffer)
    }
}

// The decode scan by one workgroup of 16 waves (see k_tc_scan_wg; up to kScanMaxSteps x 64 chunks of 2048 pairs).
__global__ __launch_bounds__(64 * kScanWaves) void k_td_scan_wg(const TdSummary* __restrict__ summ, TdCarry* __restrict__ carry, uint64_t n_chunks,
                                                                uint64_t cap, uint64_t* __restrict__ out_n)
{
    __shared__ uint64_t s_cnt[kScanMaxSteps];
    __shared__ uint32_t s_val[kScanMaxSteps];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t n_steps = static_cast<uint32_t>((n_chunks + 63u) / 64u);
    // (a wave's steps are taken eight at a time, their loads issued together: one step after the other the kernel was a chain
    // of load latencies -- 21 us for 16 384 chunks)
    for (uint32_t st0 = wave; st0 < n_steps; st0 += 8u * kScanWaves) {
        TdSummary sb[8];
#pragma unroll
        for (uint32_t u = 0; u < 8u; ++u) {
            const uint64_t c = static_cast<uint64_t>(st0 + u * kScanWaves) * 64u + lane;
            sb[u] = TdSummary{0u, 0u};
            if (st0 + u * kScanWaves < n_steps && c < n_chunks) sb[u] = summ[c];
        }
#pragma unroll
        for (uint32_t u = 0; u < 8u; ++u) {
            const uint32_t st = st0 + u * kScanWaves;
            if (st >= n_steps) break;
            const uint64_t c = static_cast<uint64_t>(st) * 64u + lane;
            const TdSummary s = sb[u];
            const uint32_t lo = wave_incl_add(s.sum_c & 0xFFFFu), hi = wave_incl_add(s.sum_c >> 16);
            const uint64_t inc = static_cast<uint64_t>(lo) + (static_cast<uint64_t>(hi) << 16);
            const uint32_t vinc = wave_incl_add(s.sum_v);
            if (c < n_chunks) carry[c] = TdCarry{inc - s.sum_c, (vinc - s.sum_v) & 0xFFu, 0u};
            if (lane == 0u) {
                s_cnt[st] = static_cast<uint64_t>(lane63(lo)) + (static_cast<uint64_t>(lane63(hi)) << 16);
                s_val[st] = lane63(vinc) & 0xFFu;
            }
        }
    }
    __syncthreads();
    if (wave == 0u) {
        uint64_t start = 0;
        uint32_t qpre = 0;
        for (uint32_t j = 0; j < n_steps; j += 64u) {
            const uint32_t i = j + lane;
            const uint64_t v = i < n_steps ? s_cnt[i] : 0ull;          // <= 64 * 2048 * 255 < 2^25
            const uint32_t q = i < n_steps ? s_val[i] : 0u;
            const uint32_t lo = wave_incl_add(static_cast<uint32_t>(v) & 0xFFFFu), hi = wave_incl_add(static_cast<uint32_t>(v >> 16));
            const uint64_t inc = static_cast<uint64_t>(lo) + (static_cast<uint64_t>(hi) << 16);
            const uint32_t qinc = wave_incl_add(q);
            if (i < n_steps) { s_cnt[i] = start + inc - v; s_val[i] = (qpre + qinc - q) & 0xFFu; }
            start += static_cast<uint64_t>(lane63(lo)) + (static_cast<uint64_t>(lane63(hi)) << 16);
            qpre = (qpre + lane63(qinc)) & 0xFFu;
        }
        if (lane == 0u) {
            carry[n_chunks] = TdCarry{start, qpre, 0u};
            *out_n = start < cap ? start : cap;
        }
    }
    __syncthreads();
    for (uint32_t st0 = wave; st0 < n_steps; st0 += 8u * kScanWaves) {
        TdCarry cb[8];
#pragma unroll
        for (uint32_t u = 0; u < 8u; ++u) {
            const uint64_t c = static_cast<uint64_t>(st0 + u * kScanWaves) * 64u + lane;
            cb[u] = TdCarry{0ull, 0u, 0u};
            if (st0 + u * kScanWaves < n_steps && c < n_chunks) cb[u] = carry[c];
        }
#pragma unroll
        for (uint32_t u = 0; u < 8u; ++u) {
            const uint32_t st = st0 + u * kScanWaves;
            const uint64_t c = static_cast<uint64_t>(st) * 64u + lane;
            if (st < n_steps && c < n_chunks) carry[c] = TdCarry{cb[u].start + s_cnt[st], (cb[u].q_pre + s_val[st]) & 0xFFu, 0u};
        }
    }
}

// The decode scan as two grids (a wave per step of 64 chunks): the one-workgroup form is bound by the instruction issue of
// the single CU it runs on (14 us for 16 384 chunks); the scan over the step totals is short enough for every wave to redo.
struct TdStepTotal { uint64_t cnt; uint32_t val, pad; };
__global__ __launch_bounds__(256) void k_td_scan_local(const TdSummary* __restrict__ summ, TdCarry* __restrict__ carry, uint64_t n_chunks,
                                                      TdStepTotal* __restrict__ step_tot)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t st = static_cast<uint64_t>(blockIdx.x) * 4u + (threadIdx.x >> 6);
    const uint64_t c = st * 64u + lane;
    if (st * 64u >= n_chunks) return;
    TdSummary s{0u, 0u};
    if (c < n_chunks) s = summ[c];
    const uint32_t lo = wave_incl_add(s.sum_c & 0xFFFFu), hi = wave_incl_add(s.sum_c >> 16);
    const uint64_t inc = static_cast<uint64_t>(lo) + (static_cast<uint64_t>(hi) << 16);
    const uint32_t vinc = wave_incl_add(s.sum_v);
    if (c < n_chunks) carry[c] = TdCarry{inc - s.sum_c, (vinc - s.sum_v) & 0xFFu, 0u};
    if (lane == 0u) step_tot[st] = TdStepTotal{static_cast<uint64_t>(lane63(lo)) + (static_cast<uint64_t>(lane63(hi)) << 16), lane63(vinc) & 0xFFu, 0u};
}
__global__ __launch_bounds__(256) void k_td_scan_apply(TdCarry* __restrict__ carry, uint64_t n_chunks, const TdStepTotal* __restrict__ step_tot,
                                                      uint64_t cap, uint64_t* __restrict__ out_n)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t st = static_cast<uint64_t>(blockIdx.x) * 4u + (threadIdx.x >> 6);
    const uint64_t n_steps = (n_chunks + 63u) / 64u;
    if (st >= n_steps) return;
    uint64_t cnt = 0;                                               // totals of the steps in front of this one (< 2^25 each)
    uint32_t val = 0;
    for (uint64_t j0 = 0; j0 < st; j0 += 64u) {
        const uint64_t j = j0 + lane;
        if (j < st) { const TdStepTotal t = step_tot[j]; cnt += t.cnt; val += t.val; }
    }
    const uint32_t lo = lane63(wave_incl_add(static_cast<uint32_t>(cnt) & 0xFFFFu)), mid = lane63(wave_incl_add(static_cast<uint32_t>(cnt >> 16) & 0xFFFFu));
    const uint32_t hi = lane63(wave_incl_add(static_cast<uint32_t>(cnt >> 32)));
    const uint64_t before = static_cast<uint64_t>(lo) + (static_cast<uint64_t>(mid) << 16) + (static_cast<uint64_t>(hi) << 32);
    const uint32_t qbefore = lane63(wave_incl_add(val)) & 0xFFu;
    const uint64_t c = st * 64u + lane;
    if (c < n_chunks) {
        const TdCarry mine = carry[c];
        carry[c] = TdCarry{mine.start + before, (mine.q_pre + qbefore) & 0xFFu, 0u};
    }
    if (st + 1u == n_steps && lane == 0u) {
        const TdStepTotal t = step_tot[st];
        const uint64_t total = before + t.cnt;
        carry[n_chunks] = TdCarry{total, (qbefore + t.val) & 0xFFu, 0u};
        *out_n = total < cap ? total : cap;                          // elements written (the stream's total, clipped at the buffer)
    }
}

// Output-centric expand: a wave owns 2048 consecutive output elements.  It finds the chunk of pairs its first element
// lies in, walks the pairs 64 per step (add-scan of (value*count mod 256) << 24 | count gives each run its start and the
// int8 prefix of all earlier deltas), and every lane writes the part of its run that falls into the tile:
//     q[start + m] = prefix + (m + 1) * value  (mod 256)        (cache_engine.cpp:241-273)
// into a 2 KiB byte table; dequantisation and the stores are then one coalesced pass.
template <int MODE, bool F32>
__global__ __launch_bounds__(256) void k_td_expand(const uint8_t* __restrict__ rle, uint64_t n_pairs, const TdCarry* __restrict__ carry,
                                                  uint64_t n_chunks, const uint64_t* __restrict__ n_out_p, float scale, uint8_t* __restrict__ dst)
{
    __shared__ __attribute__((aligned(16))) uint8_t tabs[4][kTile];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint64_t n_out = *n_out_p;
    const uint64_t o0 = (static_cast<uint64_t>(blockIdx.x) * 4u + wave) * kTile;
    if (o0 >= n_out) return;
    const uint64_t o1 = (n_out - o0 < kTile) ? n_out : o0 + kTile;
    uint8_t* tab = tabs[wave];
    // last chunk whose first element is at or before o0 (chunks that emit nothing share their successor's start), found by the
    // whole wave at once: 64 probes per step of a 64-ary search (3 steps for 16 384 chunks; a binary search was 14 dependent
    // loads in front of everything else the wave does)
    uint64_t lo = 0, hi = n_chunks;                                 // carry[n_chunks].start = total > o0; start is non-decreasing
    while (hi - lo > 1u) {
        const uint64_t span = hi - lo, step = (span + 63u) / 64u;
        const uint64_t c = lo + static_cast<uint64_t>(lane + 1u) * step;
        const bool le = c < hi && carry[c].start <= o0;
        const uint32_t kk = static_cast<uint32_t>(__popcll(__ballot(le)));
        const uint64_t nlo = lo + static_cast<uint64_t>(kk) * step, nhi = lo + static_cast<uint64_t>(kk + 1u) * step;
        lo = nlo;
        hi = nhi < hi ? nhi : hi;
    }
    // positions relative to the tile from here on (32-bit): the chunk starts at or before o0, by less than 2048 x 255 elements
    const int32_t valid_i = static_cast<int32_t>(o1 - o0);
    int32_t tot = -static_cast<int32_t>(o0 - carry[lo].start);       // elements in front of pair `i0`, relative to o0
    uint32_t qp = carry[lo].q_pre;
    // eight pairs per lane and step (one 16-byte load where the stream is aligned): a lane sums its own eight, one add-scan over
    // the lane totals places them
    const bool wide = (reinterpret_cast<uintptr_t>(rle) & 15u) == 0u;
#pragma unroll 1
    for (uint64_t i0 = lo * kTile; i0 < n_pairs && tot < valid_i; i0 += 512u) {
        const uint64_t i = i0 + 8u * lane;
        uint32_t w[4] = {0u, 0u, 0u, 0u};                              // pairs i, i+1 | i+2, i+3 | ...
        if (wide && i + 8u <= n_pairs) {
            const u32x4 x = *reinterpret_cast<const u32x4*>(rle + 2ull * i);
            w[0] = x.x; w[1] = x.y; w[2] = x.z; w[3] = x.w;
        } else {
#pragma unroll
            for (uint32_t k = 0; k < 8u; ++k)
                if (i + k < n_pairs) w[k >> 1] |= static_cast<uint32_t>(*reinterpret_cast<const uint16_t*>(rle + 2ull * (i + k))) << (16u * (k & 1u));
        }
        uint32_t sc = 0, sv = 0;                                       // the lane's counts and value x count sums (dword = v0 | c0 << 8 | v1 << 16 | c1 << 24)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const uint32_t counts = (w[t] >> 8) & 0x00FF00FFu;
            sc = __builtin_amdgcn_udot4(counts, 0x00010001u, sc, false);
            sv = __builtin_amdgcn_udot4(w[t], counts, sv, false);
        }
        const uint32_t lane_total = (sv << 24) | sc;                   // counts: < 2^24 per step, the value sums wrap mod 256
        const uint32_t incl = wave_incl_add(lane_total);
        const uint32_t e = incl - lane_total;
        int32_t start = tot + static_cast<int32_t>(e & 0xFFFFFFu);
        uint32_t q = qp + (e >> 24);
#pragma unroll
        for (uint32_t k = 0; k < 8u; ++k) {
            const uint32_t bits = (w[k >> 1] >> (16u * (k & 1u))) & 0xFFFFu;
            const uint32_t v = bits & 0xFFu;
            const int32_t c = static_cast<int32_t>(bits >> 8);
            // the part of [start, start + c) inside [0, valid): the run's first element by itself (data that does not compress
            // is runs of one: no loop, no bounds arithmetic), the rest of a longer run in a loop
            if (c != 0) {
                if (static_cast<uint32_t>(start) < static_cast<uint32_t>(valid_i)) tab[start] = static_cast<uint8_t>(q + v);
                if (c > 1) {
                    const int32_t a0 = start + 1 > 0 ? start + 1 : 0, bnd = (start + c < valid_i) ? start + c : valid_i;
                    uint32_t qq = q + static_cast<uint32_t>(a0 - start) * v;
#pragma unroll 1
                    for (int32_t p = a0; p < bnd; ++p) { qq += v; tab[p] = static_cast<uint8_t>(qq); }
                }
            }
            start += c;
            q += v * static_cast<uint32_t>(c);
        }
        const uint32_t step_tot = lane63(incl);
        tot += static_cast<int32_t>(step_tot & 0xFFFFFFu);
        qp = (qp + (step_tot >> 24)) & 0xFFu;
    }
    wave_lds_fence();
    const uint32_t valid = static_cast<uint32_t>(o1 - o0);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const uint32_t p0 = 512u * j + 8u * lane;
        if (p0 >= valid) continue;
        const uint2 x = *reinterpret_cast<const uint2*>(tab + p0);
        float y[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const uint32_t b = ((k < 4 ? x.x : x.y) >> ((k & 3) * 8)) & 0xFFu;
            y[k] = dequant<MODE>(static_cast<int>(static_cast<int8_t>(b)), scale);
        }
        if (p0 + 8u <= valid) {
            if (F32) {
                float* o = reinterpret_cast<float*>(dst) + o0 + p0;
                *reinterpret_cast<float4*>(o) = make_float4(y[0], y[1], y[2], y[3]);
                *reinterpret_cast<float4*>(o + 4) = make_float4(y[4], y[5], y[6], y[7]);
            } else {
                *reinterpret_cast<uint4*>(reinterpret_cast<uint16_t*>(dst) + o0 + p0) =
                    make_uint4(pack_half2(y[0], y[1]), pack_half2(y[2], y[3]), pack_half2(y[4], y[5]), pack_half2(y[6], y[7]));
            }
        } else {
            for (uint32_t k = 0; k < 8u && p0 + k < valid; ++k) {
                if (F32) reinterpret_cast<float*>(dst)[o0 + p0 + k] = y[k];
                else { float a = y[k], z = 0.0f; reinterpret_cast<uint16_t*>(dst)[o0 + p0 + k] = static_cast<uint16_t>(pack_half2(a, z) & 0xFFFFu); }
            }
        }
    }
}

// ---------------------------------------------------------------- compress in ONE pass over the source (after the abs-max pass)
// The multi-launch form above reads the source for the tile summaries, scans them in three grids, reads the source again for
// the tiles its fast path declined, and then GATHERS the tiles' pair slots into the stream (k_tc_pack): 92 us for 32 Mi
// elements of which the gather alone is 29 and the scans 15.  Here a tile learns what it needs from its left neighbours while
// it still holds its pairs in LDS, and writes them where they belong.
//   A workgroup takes kTfWaves consecutive tiles (one per wave; ticket counter: ascending order, so a workgroup's predecessors are
//   finished or resident).  What crosses tiles INSIDE the workgroup goes through LDS; what crosses workgroups through status
//   words, one 8-byte agent-scope word per workgroup and chain (state << 62 | value: 0 not yet, 1 own aggregate, 2 inclusive
//   prefix; the value IS the word, nothing to fence), looked back over 64 at a time by wave 0 alone:
//     chain 1  last stretch start at or before the workgroup's last tile (absolute position + 1): a workgroup that has a stretch
//              start of its own publishes its inclusive prefix at once -- it is its own last start; only a workgroup inside one
//              long stretch publishes "nothing here" and copies the prefix it finds on its left.  From the stretch start
//              entering a tile follow the phase of the 255-splits in its head and the last run start in front of it.
//     chain 2  run starts up to the workgroup's end (a plain sum): aggregate = runs of its tiles, known once chain 1 has answered.
//   Every resident workgroup publishes its words without waiting for anyone (the chain-2 aggregate waits for chain 1 of workgroups
//   that are resident), so no look-back waits for a workgroup that has not started.  (Per-TILE words were the first form: 8192
//   waves polling 64 words each slowed the whole chip and the first tiles walked back through 128 windows -- 97 us for the pass
//   at 32 Mi elements against 60 for the launches it replaces.  profiles/r04_tensor_codec.txt)
//   A pair's count byte belongs to the run that ENDS it: the tile whose first run start follows writes it (one byte in front
//   of its own pairs); the last tile closes the stream.  A tile's bytes go out as whole aligned 16-byte pieces, re-aligned
//   from LDS with v_alignbyte (the stream position of a tile is even, not aligned), single bytes at the two ragged ends.
#ifndef SPECKV_TF_WAVES
#define SPECKV_TF_WAVES 16
#endif
constexpr uint32_t kTfWaves = SPECKV_TF_WAVES;
typedef unsigned long long __attribute__((address_space(1))) tc_gu64;
__device__ __forceinline__ void tc_lb_store(uint64_t* w, uint64_t state, uint64_t value)
{
    __hip_atomic_store(reinterpret_cast<tc_gu64*>(reinterpret_cast<uintptr_t>(w)), (state << 62) | value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ uint64_t tc_lb_load(const uint64_t* w)
{
    return __hip_atomic_load(reinterpret_cast<const tc_gu64*>(reinterpret_cast<uintptr_t>(w)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
constexpr uint64_t kTcLbMask = (1ull << 62) - 1ull;
// words [.., t) of `status`, 64 per step from the right (one wave): SUM = true adds the aggregates in front of the nearest
// inclusive prefix to it (an aggregate is < 2^16, a prefix < 2^62); SUM = false returns that prefix alone (chain 1: the words
// in between say "nothing here").  Waits only for words of workgroups that are running (see above); backs off while it waits.
template <bool SUM>
__device__ __forceinline__ uint64_t tc_look_back(const uint64_t* status, uint64_t t, uint32_t lane)
{
    uint64_t acc = 0, end = t;
    for (;;) {
        const bool have = lane < end;
        uint64_t w;
        unsigned long long ready, incl;
        uint32_t first_incl;
        for (;;) {
            w = have ? tc_lb_load(status + (end - 1u - lane)) : (2ull << 62);        // (in front of workgroup 0: a prefix of nothing)
            ready = __ballot((w >> 62) != 0ull);
            incl = __ballot((w >> 62) == 2ull);
            first_incl = incl ? static_cast<uint32_t>(__builtin_ctzll(incl)) : 64u;
            const unsigned long long need = first_incl >= 63u ? ~0ull : ((2ull << first_incl) - 1ull);
            if ((ready & need) == need) break;
            __builtin_amdgcn_s_sleep(8);
        }
        const uint64_t v = w & kTcLbMask;
        if (SUM) acc += lane63(wave_incl_add(lane < first_incl ? static_cast<uint32_t>(v) : 0u));
        if (first_incl < 64u) {
            const uint32_t pl = static_cast<uint32_t>(__shfl(static_cast<int>(v & 0xFFFFFFFFull), static_cast<int>(first_incl)));
            const uint32_t ph = static_cast<uint32_t>(__shfl(static_cast<int>(v >> 32), static_cast<int>(first_incl)));
            return acc + ((static_cast<uint64_t>(ph) << 32) | pl);
        }
        end -= 64u;
    }
}

// (amdgpu_waves_per_eu: with the out-of-line SPLIT tile path in it the kernel ran a quarter slower on tensors that never call it
//  -- same instructions, other registers -- until the compiler was told to aim at 8 waves per SIMD: two workgroups per CU)
// BATCH (speckv_ext_codec_compress_tensors: many tensors per launch, ONE workgroup per tensor): the workgroup walks its tensor in
// rounds of kTfWaves tiles, and what enters a round from the left is what the previous round left in LDS -- no status words, no
// look-back, no ticket: thousands of independent chains of a few rounds each instead of one chain over the whole launch.
template <int MODE, bool F32, bool BATCH, bool PRE = false>
__device__ __forceinline__ void tc_fused_body(const void* __restrict__ src, uint64_t n, float scale, uint64_t n_tiles, uint64_t wg0,
                                              uint64_t* __restrict__ w1, uint64_t* __restrict__ w2, uint8_t* __restrict__ out,
                                              float* __restrict__ out_scale, uint64_t* __restrict__ out_bytes, uint32_t no_split,
                                              const u32x4* pre = nullptr, bool pre_ok = false)
{
    constexpr uint32_t kSlot = kTcLead + 2 * kTile + 16;
    __shared__ __attribute__((aligned(16))) uint8_t lds[kTfWaves * kSlot];
    __shared__ uint64_t s_ss[kTfWaves];                                  // last stretch start of the tile (absolute + 1, 0 = none)
    __shared__ uint32_t s_runs[kTfWaves];
    __shared__ uint64_t s_ss_in, s_run_base;
    __shared__ uint64_t s_carry_ss, s_carry_runs;                       // BATCH: the two chains across the rounds of the workgroup
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    if (BATCH && threadIdx.x == 0u) { s_carry_ss = 0ull; s_carry_runs = 0ull; }      // (published by the first barrier of round 0)
    const uint64_t n_rounds = BATCH ? (n_tiles + kTfWaves - 1u) / kTfWaves : 1u;
#pragma unroll 1
    for (uint64_t round = 0; round < n_rounds; ++round) {
    const uint64_t wg = BATCH ? round : wg0;
    const uint64_t tile = wg * kTfWaves + wave;
    const bool valid = tile < n_tiles;                                  // (waves behind the last tile only keep the barriers company)
    const uint64_t t0 = tile * kTile;
    const uint32_t len = valid ? static_cast<uint32_t>((n - t0 < kTile) ? (n - t0) : kTile) : 0u;
    if (tile == 0u && lane == 0u) *out_scale = scale;
    // the two elements in front of the tile give q[t0-1] and d[t0-1]
    uint32_t qtail0 = 0, dtail0 = 0;
    if (valid && t0 >= 1) {
        qtail0 = quantize<MODE>(tc_load<F32>(src, t0 - 1), scale);
        const uint32_t q2 = (t0 >= 2) ? quantize<MODE>(tc_load<F32>(src, t0 - 2), scale) : 0u;
        dtail0 = (qtail0 - q2) & 0xFFu;
    }
    uint8_t* wl = lds + wave * kSlot;
    const uint32_t pair_addr = static_cast<uint32_t>(reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) uint8_t*)(wl + kTcLead)));
    // ---- the element-wise tile loop (k_tc_tiles' general path): summary, or pairs into LDS once the entering phase is known
    uint32_t first_ss = 0, scarry = 0, mcarry = 0, icarry = 0;
    auto general = [&](bool emit, uint32_t lead_phase) {
        uint32_t qtail = qtail0, dtail = dtail0;
        first_ss = 0; scarry = 0; mcarry = 0; icarry = 0;
#pragma unroll 1
        for (uint32_t step = 0; step < kTile / 64u; ++step) {
            const uint32_t rel = 64u * step + lane;
            if (64u * step >= len) break;                               // wave-uniform
            const bool live = rel < len;
            const uint32_t qv = live ? quantize<MODE>(tc_load<F32>(src, t0 + rel), scale) : 0u;
            const uint32_t prevq = wave_shr1(qv, qtail);
            qtail = lane63(qv);
            const uint32_t d = (qv - prevq) & 0xFFu;
            const uint32_t prevd = wave_shr1(d, dtail);
            dtail = lane63(d);
            const bool neq = live && ((t0 + rel == 0u) || (d != prevd));
            const unsigned long long nb = __ballot(neq);
            if (!first_ss && nb) first_ss = 64u * step + static_cast<uint32_t>(__builtin_ctzll(nb)) + 1u;
            const uint32_t is = wave_incl_max(neq ? rel + 1u : 0u);
            const uint32_t ss = umax(is, scarry);                       // this element's stretch start, rel + 1 (0: before the tile)
            scarry = umax(scarry, lane63(is));
            bool isrun;
            if (ss) isrun = live && ((rel + 1u - ss) % 255u) == 0u;
            else    isrun = emit && live && ((lead_phase + rel) % 255u) == 0u;       // continuation of a stretch of earlier tiles
            const uint32_t ic = wave_incl_add(isrun ? 1u : 0u);
            const uint32_t idx = icarry + ic - (isrun ? 1u : 0u);
            icarry += lane63(ic);
            const uint32_t im = wave_incl_max(isrun ? rel + 1u : 0u);
            const uint32_t prev = umax(wave_shr1(im, 0u), mcarry);      // previous run start inside the tile (rel + 1, 0 = none)
            mcarry = umax(mcarry, lane63(im));
            if (emit && isrun) {
                // value byte of this pair, count byte of the previous one (the first pair of the tile writes into the lead bytes)
                *reinterpret_cast<__attribute__((address_space(3))) uint8_t*>(static_cast<uintptr_t>(pair_addr + 2u * idx - 1u)) = static_cast<uint8_t>(rel + 1u - prev);
                *reinterpret_cast<__attribute__((address_space(3))) uint8_t*>(static_cast<uintptr_t>(pair_addr + 2u * idx)) = static_cast<uint8_t>(d);
            }
        }
    };
    // ---- local pass: whole aligned fp16 tiles by the 8-elements-per-lane path (its pairs land in LDS), the rest element-wise
    bool fast = false, split = false;
    uint32_t own_runs = 0;                                              // run starts at or behind the tile's first stretch start
    const bool tile_fast_ok = len == kTile && ((reinterpret_cast<uintptr_t>(src) + (F32 ? 4ull : 2ull) * t0) & 15u) == 0u && !kTcNoFast;
    const bool getenv_no_split = no_split != 0u;                        // (tuning.hpp tc_no_split_tiles: the element-wise loop for long stretches, tests)
    if (valid) {
        if (PRE ? pre_ok : tile_fast_ok) {                              // (PRE: the wave's tile is whole, aligned, finite and in registers)
            uint32_t f_first = 0, f_last = 0, f_n = 0;
            if (tc_tile_fast<MODE, true, F32, PRE>(static_cast<const uint8_t*>(src) + (F32 ? 4ull : 2ull) * t0, scale, 1.0f / scale, qtail0, t0 == 0u ? 0x100u : dtail0, pair_addr - 1u, lane,
                                                   f_first, f_last, f_n, pre)) {
                first_ss = f_first; scarry = f_last; mcarry = f_last; own_runs = f_n;
                fast = true;
            }
        }
        // long stretches: the SPLIT form's summary (the head in front of the first change is counted by arithmetic below)
        if (!F32 && !fast && tile_fast_ok && !getenv_no_split) {         // (fp32 sources: the element-wise loop for those)
            uint32_t f_first = 0, f_last = 0, f_n = 0, f_m = 0;
            if (tc_tile_split<MODE, false>(static_cast<const uint8_t*>(src) + 2ull * t0, scale, 1.0f / scale, qtail0, t0 == 0u ? 0x100u : dtail0, pair_addr - 1u, lane,
                                           false, 0u, f_first, f_last, f_n, f_m)) {
                first_ss = f_first; scarry = f_last; mcarry = f_m; own_runs = f_n;
                split = true;
            }
        }
        if (!fast && !split) { general(false, 0u); own_runs = icarry; }
    }
    const uint32_t last_ss = scarry;                                    // last stretch start of the tile (rel + 1, 0 = none)
    // ---- chain 1: the stretch start entering each tile
    if (lane == 0u) s_ss[wave] = (valid && last_ss) ? t0 + last_ss : 0ull;
    __syncthreads();
    if (wave == 0u) {
        uint64_t mine = 0;
        for (uint32_t i = 0; i < kTfWaves; ++i) mine = s_ss[i] ? s_ss[i] : mine;       // (ascending positions: the last one that has any)
        if (!BATCH && lane == 0u && wg != 0u) tc_lb_store(w1 + wg, mine ? 2ull : 1ull, mine);
        uint64_t in = 0;
        if (BATCH) in = s_carry_ss;
        else if (wg != 0u) in = tc_look_back<false>(w1, wg, lane);
        if (lane == 0u) {
            if (BATCH) s_carry_ss = mine ? mine : in;
            else if (wg == 0u) tc_lb_store(w1, 2ull, mine);              // (tile 0 always has position 0)
            else if (!mine) tc_lb_store(w1 + wg, 2ull, in);
            s_ss_in = in;
        }
    }
    __syncthreads();
    uint64_t ss_in = s_ss_in;                                           // absolute position + 1 of the stretch start entering this tile
    for (uint32_t i = 0; i < wave; ++i) ss_in = s_ss[i] ? s_ss[i] : ss_in;
    // phase of the entering stretch at t0, its run starts in the head of the tile (in front of the first stretch start)
    const uint32_t lead_phase = (valid && tile) ? static_cast<uint32_t>((t0 - (ss_in - 1u)) % 255u) : 0u;
    const uint32_t head_len = first_ss ? first_ss - 1u : len;
    uint32_t head_runs = 0;
    if (valid && tile && head_len) head_runs = (lead_phase + head_len - 1u) / 255u - (lead_phase ? (lead_phase - 1u) / 255u : 0u) + (lead_phase ? 0u : 1u);
    uint32_t n_runs = head_runs + own_runs;
    if (valid && (!fast || head_runs)) {                                // the pairs of this tile once more, now that the entering phase is known
        bool done = false;
        if (!F32 && tile_fast_ok && (fast || split) && !getenv_no_split) {       // ... by the SPLIT form (head and own runs in one pass)
            uint32_t f_first = 0, f_last = 0, f_n = 0, f_m = 0;
            done = tc_tile_split<MODE, true>(static_cast<const uint8_t*>(src) + 2ull * t0, scale, 1.0f / scale, qtail0, t0 == 0u ? 0x100u : dtail0, pair_addr - 1u, lane,
                                             tile != 0u, lead_phase, f_first, f_last, f_n, f_m);
            if (done) { n_runs = f_n; mcarry = f_m; }
        }
        if (!done) {                                                     // ... or element by element
            general(true, lead_phase);
            n_runs = icarry;
        }
    }
    // ---- chain 2: run starts in front of each tile
    if (lane == 0u) s_runs[wave] = valid ? n_runs : 0u;
    __syncthreads();
    if (wave == 0u) {
        uint32_t total = 0;
        for (uint32_t i = 0; i < kTfWaves; ++i) total += s_runs[i];
        uint64_t base = 0;
        if (BATCH) { base = s_carry_runs; if (lane == 0u) s_carry_runs = base + total; }
        else if (wg == 0u) { if (lane == 0u) tc_lb_store(w2, 2ull, total); }
        else {
            if (lane == 0u) tc_lb_store(w2 + wg, 1ull, total);
            base = tc_look_back<true>(w2, wg, lane);
            if (lane == 0u) tc_lb_store(w2 + wg, 2ull, base + total);
        }
        if (lane == 0u) s_run_base = base;
    }
    __syncthreads();
    if (!valid) continue;                                               // (no barrier behind this point in the round)
    uint64_t run_base = s_run_base;
    for (uint32_t i = 0; i < wave; ++i) run_base += s_runs[i];
    // ---- count bytes at the seams.  The byte in front of this tile's pairs closes the last pair of its left neighbours: it ends
    // at this tile's first run start (or, in the last tile without a run of its own, at n); the run it belongs to started at
    // ss_in + 255 m, so the elements in front of t0 that still belong to it are ((lead_phase + 254) % 255) + 1.
    const bool is_last = tile + 1u == n_tiles;
    const uint32_t back = tile ? ((lead_phase + 254u) % 255u) + 1u : 0u;
    wave_lds_fence();
    if (lane == 0u) {
        uint8_t* lead = wl + kTcLead - 1u;
        if (n_runs) {
            const uint32_t first_rel_p1 = *lead;                        // both loops store (position of the first run start + 1) here
            *lead = static_cast<uint8_t>(first_rel_p1 - 1u + back);
            if (is_last) wl[kTcLead + 2u * n_runs - 1u] = static_cast<uint8_t>(len - (mcarry - 1u));      // the stream's last pair ends at n
        } else {
            *lead = static_cast<uint8_t>(back + (is_last ? len : 0u));   // (only a short last tile has no run start: it closes the stream)
        }
    }
    wave_lds_fence();
    if (is_last && lane == 0u) *out_bytes = 2u * (run_base + n_runs);
    // ---- this tile's bytes: [2 run_base - 1, 2 (run_base + n_runs) - 1), plus the stream's last byte in the last tile; the
    // byte in front only if there is a pair in front (run_base > 0) that this tile has to close (it has a run, or is the last)
    const bool lead_out = run_base != 0u && (n_runs != 0u || is_last);
    const uint64_t g_begin = 2u * run_base - (lead_out ? 1u : 0u);
    const uint64_t g_end = 2u * (run_base + n_runs) - ((is_last || n_runs == 0u) ? 0u : 1u);
    if (g_end <= g_begin) continue;
    const uint32_t l_begin = kTcLead - (lead_out ? 1u : 0u);             // LDS offset (in wl) of the byte that goes to g_begin
    const uintptr_t gaddr = reinterpret_cast<uintptr_t>(out) + g_begin, gend = reinterpret_cast<uintptr_t>(out) + g_end;
    const uintptr_t a0v = gaddr & ~static_cast<uintptr_t>(15);
    // Everything about the stretch is the same for the 64 lanes: said so (SGPRs), and as 32-bit offsets from a0 -- the loop below
    // was 53 vector instructions per KiB of stream on 64-bit addresses (a fifth of the kernel's: PMC, round 6).
    // (the builtin returns a signed int: through uint32_t, or a low half with bit 31 set sign-extends over the high one)
    const uintptr_t a0 = static_cast<uintptr_t>(static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<uint32_t>(a0v)))) |
                         (static_cast<uintptr_t>(static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<uint32_t>(a0v >> 32)))) << 32);
    const uint32_t first = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(gaddr - a0v));       // bytes of a0's 16 that are the left neighbour's
    const uint32_t total = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(gend - a0v));         // the stretch ends `total` bytes behind a0
    // LDS offset (in wl) of global address a0 + x: delta + x (delta >= 0: the lead is 16 bytes).  For x a multiple of 16 the offset
    // modulo 16 is one value: sh = byte shift inside a dword, dsel = which five of eight dwords
    const uint32_t delta = __builtin_amdgcn_readfirstlane(l_begin - first);
    const uint32_t wl_addr = pair_addr - kTcLead;
    const uint32_t sh = delta & 3u, dsel = (delta >> 2) & 3u;
    const uint32_t full_begin = first ? 16u : 0u, full_end = total & ~15u;      // whole 16-byte pieces: [full_begin, full_end)
    typedef u32x4 __attribute__((address_space(1)))* g_u32x4_p;        // (global address space spelled out: a pointer made from an
    typedef uint8_t __attribute__((address_space(1)))* g_u8_p;          //  integer is a FLAT pointer to the compiler, and flat stores count in lgkmcnt beside the LDS reads of the next piece)
    typedef const u32x4 __attribute__((address_space(3)))* l_u32x4_p;
    // the 20 bytes around a piece as two ALIGNED 16-byte reads (lanes 16 bytes apart: conflict-free); which five of the eight
    // dwords are wanted is wave-uniform: one loop per value
    auto pieces = [&](auto ds_c) {
        constexpr uint32_t ds = decltype(ds_c)::value;
#pragma unroll 1
        for (uint32_t x = full_begin + 16u * lane; x < full_end; x += 1024u) {
            const uint32_t base = wl_addr + ((delta + x) & ~15u);
            const u32x4 q0 = *(l_u32x4_p)(static_cast<uintptr_t>(base)), q1 = *(l_u32x4_p)(static_cast<uintptr_t>(base + 16u));
            const uint32_t e8[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
            u32x4 v;
            v.x = __builtin_amdgcn_alignbyte(e8[ds + 1], e8[ds], sh); v.y = __builtin_amdgcn_alignbyte(e8[ds + 2], e8[ds + 1], sh);
            v.z = __builtin_amdgcn_alignbyte(e8[ds + 3], e8[ds + 2], sh); v.w = __builtin_amdgcn_alignbyte(e8[ds + 4], e8[ds + 3], sh);
            __builtin_nontemporal_store(v, (g_u32x4_p)(a0 + x));
        }
    };
    if (dsel == 0u)      pieces(std::integral_constant<uint32_t, 0u>{});
    else if (dsel == 1u) pieces(std::integral_constant<uint32_t, 1u>{});
    else if (dsel == 2u) pieces(std::integral_constant<uint32_t, 2u>{});
    else                 pieces(std::integral_constant<uint32_t, 3u>{});
    // the ragged pieces at the two ends of the stretch, a byte per lane: lanes 0-15 the first piece, 16-31 the last
    if (lane < 32u) {
        const uint32_t piece = lane >> 4, x = (piece ? full_end : 0u) + (lane & 15u);
        const bool want = piece ? (full_end < total && full_end >= full_begin) : (first != 0u);
        if (want && x >= first && x < total) *(g_u8_p)(a0 + x) = wl[delta + x];
    }
    }   // rounds
}

// (amdgpu_waves_per_eu: see above.)  One tensor: ticket order, chains through status words.
template <int MODE, bool F32>
__global__ __launch_bounds__(64 * kTfWaves) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_tc_fused(const void* __restrict__ src, uint64_t n, const uint32_t* __restrict__ absmax_bits,
                                                            uint64_t n_tiles, uint64_t* __restrict__ w1, uint64_t* __restrict__ w2,
                                                            uint32_t* __restrict__ ticket, uint8_t* __restrict__ out,
                                                            float* __restrict__ out_scale, uint64_t* __restrict__ out_bytes, uint32_t no_split)
{
    __shared__ uint32_t s_ticket;
    if (threadIdx.x == 0u) s_ticket = atomicAdd(ticket, 1u);
    __syncthreads();
    tc_fused_body<MODE, F32, false>(src, n, tc_scale(*absmax_bits), n_tiles, s_ticket, w1, w2, out, out_scale, out_bytes, no_split);
}

// Many tensors, one workgroup each (speckv_ext_codec_compress_tensors; the reference's compress(data, n) is called per KV tile --
// cache_engine.cpp:40-82, RTL tile 1024 x 128 = 131 072 elements, hardware/rtl/kv_compress.v:5-11 -- and at that size the
// single-tensor launch pair is all fixed cost: 64 tiles = 4 workgroups and two launches).  The workgroup finds its tensor's
// max|x| itself (first pass over the source, default cache policy), then encodes it in rounds (second pass: the same bytes out of
// the L2 / Infinity Cache, so the HBM sees the source once).  Any length works; a tensor is walked by ONE workgroup, so tensors of
// many millions of elements belong to speckv_ext_codec_compress_tensor.
struct TcbDesc { void* data; uint64_t n; uint8_t* rle; uint64_t rle_cap; };        // = speckv_ext_tensor_t (include/speckv_ext.h)
template <int MODE, bool F32>
__global__ __launch_bounds__(64 * kTfWaves) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_tcb_fused(const TcbDesc* __restrict__ desc, float* __restrict__ out_scale,
                                                            uint64_t* __restrict__ out_bytes, uint32_t no_split)
{
    __shared__ uint32_t s_am[kTfWaves];
    const TcbDesc d = desc[blockIdx.x];
    out_scale += blockIdx.x; out_bytes += blockIdx.x;
    if (d.n == 0u) {                                                     // compress(data, 0): an empty stream, scale 1 (cache_engine.cpp:172-183)
        if (threadIdx.x == 0u) { *out_bytes = 0ull; *out_scale = 1.0f; }
        return;
    }
    uint32_t m = lane63(wave_incl_max(tc_absmax_thread<F32, false>(d.data, d.n, threadIdx.x, 64u * kTfWaves)));
    if ((threadIdx.x & 63u) == 0u) s_am[threadIdx.x >> 6] = m;
    __syncthreads();
    m = 0;
    for (uint32_t w = 0; w < kTfWaves; ++w) m = umax(m, s_am[w]);
    if (!F32) m = __float_as_uint(half_bits_to_float(m));
    tc_fused_body<MODE, F32, true>(d.data, d.n, tc_scale(m), (d.n + kTile - 1u) / kTile, 0ull, nullptr, nullptr, d.rle, out_scale, out_bytes, no_split);
}

// The same launch shape with SEVERAL workgroups per tensor (`wpt` = workgroups of the largest tensor; tensor t = ticket / wpt,
// its workgroup ticket % wpt takes 16 tiles as in k_tc_fused).  One workgroup per tensor (above) leaves a tensor's 64 tiles to
// four strictly serial rounds between barriers: 0.31 of the roofline at 4096 x 131 072.  Here the tiles of a tensor are encoded
// side by side, the chains are k_tc_fused's look-back over the tensor's OWN status words (at most wpt - 1 predecessors: one
// window, mostly answered at once), and the max|x| comes from a rendezvous of the tensor's workgroups: each publishes the maximum
// of its 16 tiles as a tagged word and reads the others'.  Tickets make that safe: the workgroups of a tensor hold consecutive
// tickets, every earlier ticket has started, and at most wpt - 1 workgroups ever wait for a ticket that has not (they are
// resident and the rest of the chip is not waiting for them).  A workgroup re-reads the 128 KiB it has just taken the maximum
// of: about 500 workgroups = 64 MiB are in flight at a time, so that second read is served by the L2 / Infinity Cache.
template <int MODE, bool F32>
__global__ __launch_bounds__(64 * kTfWaves) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_tcm_fused(const TcbDesc* __restrict__ desc, uint32_t wpt, uint32_t* __restrict__ ticket,
                                                            uint64_t* __restrict__ amax_words, uint64_t* __restrict__ w1, uint64_t* __restrict__ w2,
                                                            float* __restrict__ out_scale, uint64_t* __restrict__ out_bytes, uint32_t no_split)
{
    __shared__ uint32_t s_ticket, s_bits;
    __shared__ uint32_t s_am[kTfWaves];
    if (threadIdx.x == 0u) s_ticket = atomicAdd(ticket, 1u);
    __syncthreads();
    const uint32_t t = s_ticket / wpt, l = s_ticket - t * wpt;
    const TcbDesc d = desc[t];
    const uint64_t n_tiles = (d.n + kTile - 1u) / kTile;
    const uint32_t my_wgs = static_cast<uint32_t>((n_tiles + kTfWaves - 1u) / kTfWaves);
    if (l >= my_wgs) {                                                   // (a shorter tensor than the largest; an empty one: scale 1, no stream)
        if (l == 0u && threadIdx.x == 0u) { out_bytes[t] = 0ull; out_scale[t] = 1.0f; }
        return;
    }
    // max|x| of this wave's OWN tile, in the layout tc_tile_fast encodes from (round 6; before, the workgroup took the maximum of
    // its 16 tiles thread by thread with a compare and a select per element, and tc_tile_fast tested its tile for inf / NaN once
    // more: 160 + 64 vector instructions of the wave's 1300, here 48).  fp32: one shift per word puts |x| in front (order kept),
    // inf / NaN are the values >= 0xFF000000 of that; fp16: the block encoder's packed maximum, and the tile stays in registers
    // (16 per lane; fp32's 32 did not fit beside the rendezvous and are read again, out of the L2).  A tile that holds an inf or a
    // NaN takes the element rule (a NaN never wins, cache_engine.cpp:176-180) and the element-wise encoder, as before.
    const uint32_t wv = threadIdx.x >> 6, ln = threadIdx.x & 63u;
    const uint64_t my_tile = static_cast<uint64_t>(l) * kTfWaves + wv, my_t0 = my_tile * kTile;
    const bool tile_valid = my_tile < n_tiles;
    const uint32_t my_len = tile_valid ? static_cast<uint32_t>((d.n - my_t0 < kTile) ? (d.n - my_t0) : kTile) : 0u;
    const uint8_t* tsrc = static_cast<const uint8_t*>(d.data) + my_t0 * (F32 ? 4u : 2u);
    bool pre_ok = tile_valid && my_len == kTile && (reinterpret_cast<uintptr_t>(tsrc) & 15u) == 0u && !kTcNoFast;       // wave-uniform
    u32x4 pre[4];                                                       // (fp16 sources: the tile; fp32 sources read theirs again, out of the L2)
    uint32_t m = 0;
    if (pre_ok) {
        typedef const u32x4 __attribute__((address_space(1)))* g_u32x4_cp;
        if (F32) {
            uint32_t mm = 0;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const u32x4 x = *(g_u32x4_cp)(reinterpret_cast<uintptr_t>(tsrc + 4ull * (512u * (j >> 1) + 8u * ln) + 16u * (j & 1)));
                mm = umax(umax(mm, x.x << 1), x.y << 1);                 // (v_max3_u32)
                mm = umax(umax(mm, x.z << 1), x.w << 1);
            }
            mm = lane63(wave_incl_max(mm));
            if (mm >= 0xFF000000u) pre_ok = false; else m = mm >> 1;
        } else {
            uint4 raw[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                pre[j] = __builtin_nontemporal_load((g_u32x4_cp)(reinterpret_cast<uintptr_t>(tsrc + 2ull * (512u * j + 8u * ln))));
                raw[j] = make_uint4(pre[j].x, pre[j].y, pre[j].z, pre[j].w);
            }
            const uint32_t mm = absmax_bits(raw);
            if (mm >= 0x7C00u) pre_ok = false; else m = mm;
        }
    }
    if (!pre_ok && tile_valid) m = lane63(wave_incl_max(tc_absmax_thread<F32, false>(tsrc, my_len, ln, 64u)));
    if ((threadIdx.x & 63u) == 0u) s_am[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x < 64u) {
        const uint32_t lane = threadIdx.x;
        m = 0;
        for (uint32_t w = 0; w < kTfWaves; ++w) m = umax(m, s_am[w]);
        if (!F32) m = __float_as_uint(half_bits_to_float(m));
        uint64_t* words = amax_words + static_cast<uint64_t>(t) * wpt;
        if (lane == 0u) tc_lb_store(words + l, 1ull, m);                 // the value IS the word: nothing to fence
        uint32_t all = 0;
        for (uint32_t b = 0; b < my_wgs; b += 64u) {                     // the others' (and mine), 64 at a time
            const bool have = b + lane < my_wgs;
            uint64_t w;
            for (;;) {
                w = have ? tc_lb_load(words + b + lane) : (1ull << 62);
                if (__ballot((w >> 62) == 0ull) == 0ull) break;
                __builtin_amdgcn_s_sleep(4);
            }
            all = umax(all, lane63(wave_incl_max(have ? static_cast<uint32_t>(w) : 0u)));
        }
        if (lane == 0u) s_bits = all;
    }
    __syncthreads();
    tc_fused_body<MODE, F32, false, true>(d.data, d.n, tc_scale(s_bits), n_tiles, l, w1 + static_cast<uint64_t>(t) * wpt, w2 + static_cast<uint64_t>(t) * wpt, d.rle,
                                          out_scale + t, out_bytes + t, no_split, pre, pre_ok);
}

// ---------------------------------------------------------------- decompress in ONE pass over the stream
// The three launches above read the stream twice (summary, expand), and their expand loop takes a lane's 8 pairs one after the
// other with a byte store per ELEMENT.  This form reads the stream once and expands the way the block decoder of the pool does
// (kernels.hip: decode_rle_fast) -- one byte scattered per RUN, everything else constant work:
//   the expanded deltas are piecewise constant, so the decoded int8 sequence is the SECOND prefix sum (mod 256) of
//       E[start of run r] = value_r - value_{r-1},   0 elsewhere        (cache_engine.cpp:241-273)
//   a wave owns a chunk of 2048 pairs (4 KiB of the stream: four 16-byte loads per lane, kept in registers) and sums its counts and
//   value x count products; a workgroup of kTdfWaves waves takes that many consecutive chunks (ticket counter: ascending order, so
//   a workgroup's predecessors are finished or resident) and publishes ONE 8-byte status word (state << 62 | int8 prefix << 54 |
//   elements; the value is the word -- agent-scope relaxed store / load, nothing to fence); wave 0 looks back over the
//   predecessors' words 64 at a time; every wave then knows where its stretch of the output starts and what q was in front of it.
//   The stretch goes out window by window (4096 elements, aligned to 8 elements of the OUTPUT so that whole groups are 16-byte
//   stores; a chunk of data that does not compress is one window): clear the window's byte table, scatter E for the pairs that
//   start in it, then 512 elements per step: two packed wave scans + eight byte recurrences per lane, dequantise, store.  The
//   groups at the two ragged ends of the stretch are stored element by element (their neighbours belong to the next chunk's wave).
// A chunk that holds a ZERO count (nothing our encoders or the reference's emit; cache_engine.cpp:262 just skips it) would put two
// runs on one table byte: such a chunk is expanded pair by pair, straight to memory (td_chunk_general).
#ifndef SPECKV_TDF_WAVES
#define SPECKV_TDF_WAVES 16
#endif
constexpr uint32_t kTdfWaves = SPECKV_TDF_WAVES;
#ifndef SPECKV_TDF_LB_WAVE
#define SPECKV_TDF_LB_WAVE 0
#endif
constexpr uint32_t kTdfLbWave = SPECKV_TDF_LB_WAVE;   // 1: wave 0 of a workgroup takes no chunk, it only looks back (kTdfWaves - 1 chunks per workgroup)
constexpr uint32_t kTdfChunks = kTdfWaves - kTdfLbWave;
// The many-streams kernels (k_tdm_fused / k_tdb_fused) take 8-wave workgroups: four of them per CU wait less at their barriers than
// two of 16 waves (4096 x 131 072 to fp32 0.63 -> 0.65, to fp16 0.56 -> 0.59); ONE long stream wants the 16 -- half as many status
// words to look back over (32 Mi elements: 0.37 against 0.30 with 8).
#ifndef SPECKV_TDM_WAVES
#define SPECKV_TDM_WAVES 8
#endif
constexpr uint32_t kTdmWaves = SPECKV_TDM_WAVES, kTdmChunks = kTdmWaves - kTdfLbWave;
constexpr uint32_t kTdfWin = 4096;           // elements per window
#define SPECKV_TD_SDWA2(NAME, OP, S0, S1)                                                       \
    __device__ __forceinline__ uint32_t NAME(uint32_t a, uint32_t b)                            \
    {                                                                                           \
        uint32_t r;                                                                             \
        asm(OP " %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:" S0 " src1_sel:" S1   \
            : "=v"(r) : "v"(a), "v"(b));                                                        \
        return r;                                                                               \
    }
SPECKV_TD_SDWA2(td_add_b0, "v_add_u32_sdwa", "DWORD", "BYTE_0")
SPECKV_TD_SDWA2(td_add_b1, "v_add_u32_sdwa", "DWORD", "BYTE_1")
SPECKV_TD_SDWA2(td_add_b2, "v_add_u32_sdwa", "DWORD", "BYTE_2")
SPECKV_TD_SDWA2(td_add_b3, "v_add_u32_sdwa", "DWORD", "BYTE_3")
SPECKV_TD_SDWA2(td_sub_b0_b2, "v_sub_u32_sdwa", "BYTE_0", "BYTE_2")   // a.b0 - b.b2
SPECKV_TD_SDWA2(td_sub_b2_b0, "v_sub_u32_sdwa", "BYTE_2", "BYTE_0")   // a.b2 - b.b0
#undef SPECKV_TD_SDWA2
template <int K> __device__ __forceinline__ uint32_t td_add_byte(uint32_t a, uint32_t lo, uint32_t hi)
{
    return K == 0 ? td_add_b0(a, lo) : K == 1 ? td_add_b1(a, lo) : K == 2 ? td_add_b2(a, lo) : K == 3 ? td_add_b3(a, lo)
         : K == 4 ? td_add_b0(a, hi) : K == 5 ? td_add_b1(a, hi) : K == 6 ? td_add_b2(a, hi) : td_add_b3(a, hi);
}
__device__ __forceinline__ void td_lds_b8(uint32_t addr, uint32_t v) { *reinterpret_cast<lds_u8*>(static_cast<uintptr_t>(addr)) = static_cast<uint8_t>(v); }

// Exclusive prefix in front of workgroup `t` (t > 0): elements (low 54 bits of the words) and the int8 prefix (the 8 bits above,
// summed modulo 256 by the caller).  An aggregate's element count is < 2^24 x kTdfWaves.
__device__ __forceinline__ void td_look_back(const uint64_t* status, uint64_t t, uint32_t lane, uint64_t& elems, uint32_t& qsum)
{
    constexpr uint64_t kLo = (1ull << 54) - 1ull;
    elems = 0; qsum = 0;
    uint64_t end = t;                                                    // words [.., end) are still to be looked at
    for (;;) {
        const bool have = lane < end;
        uint64_t w;
        unsigned long long ready, incl;
        uint32_t first_incl;
        for (;;) {                                                       // this window: every word in front of the nearest prefix must be there
            w = have ? tc_lb_load(status + (end - 1u - lane)) : (2ull << 62);        // (in front of workgroup 0: a prefix of nothing)
            ready = __ballot((w >> 62) != 0ull);
            incl = __ballot((w >> 62) == 2ull);
            first_incl = incl ? static_cast<uint32_t>(__builtin_ctzll(incl)) : 64u;
            const unsigned long long need = first_incl >= 63u ? ~0ull : ((2ull << first_incl) - 1ull);
            if ((ready & need) == need) break;
            __builtin_amdgcn_s_sleep(8);
        }
        const uint64_t v = w & kTcLbMask;
        const bool mine = lane < first_incl;                            // aggregates in front of the prefix
        const uint32_t lo_lo = lane63(wave_incl_add(mine ? static_cast<uint32_t>(v & 0xFFFFu) : 0u));
        const uint32_t lo_hi = lane63(wave_incl_add(mine ? static_cast<uint32_t>((v & kLo) >> 16) : 0u));
        elems += (static_cast<uint64_t>(lo_hi) << 16) + lo_lo;
        qsum += lane63(wave_incl_add(mine ? static_cast<uint32_t>(v >> 54) : 0u));
        if (first_incl < 64u) {
            const uint32_t pl = static_cast<uint32_t>(__shfl(static_cast<int>(v & 0xFFFFFFFFull), static_cast<int>(first_incl)));
            const uint32_t ph = static_cast<uint32_t>(__shfl(static_cast<int>(v >> 32), static_cast<int>(first_incl)));
            const uint64_t pv = (static_cast<uint64_t>(ph) << 32) | pl;
            elems += pv & kLo; qsum += static_cast<uint32_t>(pv >> 54);
            return;
        }
        end -= 64u;
    }
}

// A chunk with zero counts in it: pair by pair, 64 pairs per step (an add-scan of (value x count mod 256) << 24 | count gives each
// run its start and the int8 prefix of all earlier deltas; the lane writes its run element by element), clipped at `end`.
template <int MODE, bool F32>
__device__ __noinline__ void td_chunk_general(const uint8_t* __restrict__ rle, uint64_t p0, uint64_t n_pairs, uint64_t start, uint64_t end,
                                              uint32_t qp, float scale, uint8_t* __restrict__ dst, uint32_t lane)
{
    uint64_t pos = start;
    uint32_t q0 = qp;
#pragma unroll 1
    for (uint32_t b = 0; b < kTile && pos < end; b += 64u) {
        const uint64_t i = p0 + b + lane;
        uint32_t bits = 0;
        if (i < n_pairs) bits = *reinterpret_cast<const uint16_t*>(rle + 2ull * i);
        const uint32_t v = bits & 0xFFu, c = bits >> 8;
        const uint32_t packed = ((v * c) << 24) | c;
        const uint32_t incl = wave_incl_add(packed);
        const uint32_t e = incl - packed;
        uint64_t p = pos + (e & 0xFFFFFFu);
        uint32_t q = q0 + (e >> 24);
#pragma unroll 1
        for (uint32_t m = 0; m < c && p < end; ++m, ++p) {
            q += v;
            const float y = dequant<MODE>(static_cast<int>(static_cast<int8_t>(q & 0xFFu)), scale);
            if (F32) reinterpret_cast<float*>(dst)[p] = y;
            else { float a = y, z = 0.0f; reinterpret_cast<uint16_t*>(dst)[p] = static_cast<uint16_t>(pack_half2(a, z) & 0xFFFFu); }
        }
        const uint32_t tot = lane63(incl);
        pos += tot & 0xFFFFFFu;
        q0 = (q0 + (tot >> 24)) & 0xFFu;
    }
}

// The pairs of chunk `p0 / 2048`: dword = v0 | c0 << 8 | v1 << 16 | c1 << 24, pairs p0 + 512 j + 8 lane .. + 7 in w[j]; pairs behind
// the stream's end read as (0, 0).  mn: the smallest count among the lane's pairs that exist.
__device__ __forceinline__ void td_load_pairs(const uint8_t* __restrict__ rle, uint64_t p0, uint64_t n_pairs, bool live, uint32_t lane,
                                              uint32_t (&w)[4][4], uint32_t& mn)
{
    const bool wide = (reinterpret_cast<uintptr_t>(rle) & 15u) == 0u;
#pragma unroll
    for (uint32_t j = 0; j < 4u; ++j) {
        const uint64_t i = p0 + 512u * j + 8u * lane;
        w[j][0] = w[j][1] = w[j][2] = w[j][3] = 0u;
        if (live && i < n_pairs) {
            if (wide && i + 8u <= n_pairs) {
                const u32x4 x = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(rle + 2ull * i));
                w[j][0] = x.x; w[j][1] = x.y; w[j][2] = x.z; w[j][3] = x.w;
#pragma unroll
                for (int t = 0; t < 4; ++t) mn = umin(mn, umin((w[j][t] >> 8) & 0xFFu, w[j][t] >> 24));
            } else {
#pragma unroll
                for (uint32_t e = 0; e < 8u; ++e)
                    if (i + e < n_pairs) {
                        const uint32_t bits = *reinterpret_cast<const uint16_t*>(rle + 2ull * (i + e));
                        w[j][e >> 1] |= bits << (16u * (e & 1u));
                        mn = umin(mn, bits >> 8);
                    }
            }
        }
    }
}
// Per step of 512 pairs: the lane's count and value x count sums scanned across the wave.  ex: exclusive prefix of the lane in its
// step, st: the step's total; both (sum v c mod 256) << 24 | sum c (counts < 2^24 per step).
__device__ __forceinline__ void td_scan_pairs(const uint32_t (&w)[4][4], uint32_t (&ex)[4], uint32_t (&st)[4])
{
#pragma unroll
    for (uint32_t j = 0; j < 4u; ++j) {
        uint32_t sc = 0, sv = 0;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const uint32_t counts = (w[j][t] >> 8) & 0x00FF00FFu;
            sc = __builtin_amdgcn_udot4(counts, 0x00010001u, sc, false);
            sv = __builtin_amdgcn_udot4(w[j][t], counts, sv, false);
        }
        const uint32_t lt = (sv << 24) | sc;
        const uint32_t incl = wave_incl_add(lt);
        ex[j] = incl - lt;
        st[j] = lane63(incl);
    }
}
// One window of a chunk: elements [wo, wo + 4096) counted from the chunk's OWN first element, as int8 values relative to the q in
// front of the chunk, into the wave's table (byte p = element wo + p).  Needs nothing from outside the chunk.
// c1 / c2: S1 (the delta) / S2 (q - q in front of the chunk) at the end of the previous window.
__device__ __forceinline__ void td_window_values(const uint32_t (&w)[4][4], const uint32_t (&ex)[4], const uint32_t (&st)[4], uint32_t wo,
                                                 uint32_t tot_c, uint8_t* tab, uint32_t lane, uint32_t& c1, uint32_t& c2)
{
    const uint32_t tab_addr = lds_addr_of(tab);
    const uint32_t cnt = tot_c - wo < kTdfWin ? tot_c - wo : kTdfWin;      // elements of the chunk in this window
    {
        u32x4* t4 = reinterpret_cast<u32x4*>(tab);
        const u32x4 z = {0u, 0u, 0u, 0u};
#pragma unroll
        for (uint32_t k = 0; k < kTdfWin / 1024u; ++k) t4[64u * k + lane] = z;
    }
    int32_t tot = -static_cast<int32_t>(wo);                             // first element of step 0, relative to the window (> -2^24)
#pragma unroll
    for (uint32_t j = 0; j < 4u; ++j) {
        const int32_t step_c = static_cast<int32_t>(st[j] & 0xFFFFFFu);
        if (tot + step_c > 0 && tot < static_cast<int32_t>(kTdfWin)) {      // (wave-uniform) the step has pairs that start in the window
            const int32_t base = tot + static_cast<int32_t>(ex[j] & 0xFFFFFFu);
            // value of the pair in front of the lane's first (0 in front of the chunk: E of the first run is its value)
            const uint32_t pwj = wave_shr1(w[j][3], j == 0u ? 0u : lane63(w[j == 0u ? 0u : j - 1u][3]));
            const uint32_t d0 = td_sub_b0_b2(w[j][0], pwj), d1 = td_sub_b2_b0(w[j][0], w[j][0]);
            const uint32_t d2 = td_sub_b0_b2(w[j][1], w[j][0]), d3 = td_sub_b2_b0(w[j][1], w[j][1]);
            const uint32_t d4 = td_sub_b0_b2(w[j][2], w[j][1]), d5 = td_sub_b2_b0(w[j][2], w[j][2]);
            const uint32_t d6 = td_sub_b0_b2(w[j][3], w[j][2]), d7 = td_sub_b2_b0(w[j][3], w[j][3]);
            if (tot >= 0 && tot + step_c <= static_cast<int32_t>(kTdfWin)) {    // all of them do: plain addresses
                uint32_t a = tab_addr + static_cast<uint32_t>(base);
                td_lds_b8(a, d0); a = td_add_b1(a, w[j][0]);
                td_lds_b8(a, d1); a = td_add_b3(a, w[j][0]);
                td_lds_b8(a, d2); a = td_add_b1(a, w[j][1]);
                td_lds_b8(a, d3); a = td_add_b3(a, w[j][1]);
                td_lds_b8(a, d4); a = td_add_b1(a, w[j][2]);
                td_lds_b8(a, d5); a = td_add_b3(a, w[j][2]);
                td_lds_b8(a, d6); a = td_add_b1(a, w[j][3]);
                td_lds_b8(a, d7);                                       // (pairs behind the stream's end have count 0: they land on the byte behind
                                                                        //  the chunk's last element -- nobody's, or the spare byte at the window's end)
            } else {                                                    // a run may start in front of the window or behind it: those go to the spare bytes
                int32_t a = base;
                const uint32_t dd[8] = {d0, d1, d2, d3, d4, d5, d6, d7};
#pragma unroll
                for (uint32_t e = 0; e < 8u; ++e) {
                    const uint32_t off = static_cast<uint32_t>(a) < kTdfWin ? static_cast<uint32_t>(a) : kTdfWin;
                    td_lds_b8(tab_addr + off, dd[e]);
                    a += static_cast<int32_t>((w[j][e >> 1] >> (8u + 16u * (e & 1u))) & 0xFFu);
                }
            }
        }
        tot += step_c;
    }
    wave_lds_fence();
    // E -> q (relative), in place: 512 elements per step, two packed wave scans + eight byte recurrences per lane
#pragma unroll 1
    for (uint32_t g0 = 0; g0 < kTdfWin; g0 += 512u) {
        if (g0 >= cnt) break;                                            // (wave-uniform) nothing of this chunk behind here
        uint2* cell = reinterpret_cast<uint2*>(tab + g0 + 8u * lane);
        const uint2 x = *cell;
        const uint32_t t1 = __builtin_amdgcn_sad_u8(x.x, 0u, __builtin_amdgcn_sad_u8(x.y, 0u, 0u));
        const uint32_t t2 = __builtin_amdgcn_udot4(x.x, 0x05060708u, __builtin_amdgcn_udot4(x.y, 0x01020304u, 0u, false), false);
        const uint32_t i1 = wave_incl_add(t1);
        const uint32_t x1 = c1 + i1 - t1;                                // S1 entering this lane
        const uint32_t u = t2 + 8u * x1;                                 // this lane's S2 increment
        const uint32_t i2 = wave_incl_add(u);
        const uint32_t x2 = c2 + i2 - u;                                 // S2 entering this lane
        c1 += lane63(i1);
        c2 += lane63(i2);
        uint32_t s1 = x1, s2 = x2;
        uint32_t q[8];
        s1 = td_add_byte<0>(s1, x.x, x.y); s2 += s1; q[0] = s2;
        s1 = td_add_byte<1>(s1, x.x, x.y); s2 += s1; q[1] = s2;
        s1 = td_add_byte<2>(s1, x.x, x.y); s2 += s1; q[2] = s2;
        s1 = td_add_byte<3>(s1, x.x, x.y); s2 += s1; q[3] = s2;
        s1 = td_add_byte<4>(s1, x.x, x.y); s2 += s1; q[4] = s2;
        s1 = td_add_byte<5>(s1, x.x, x.y); s2 += s1; q[5] = s2;
        s1 = td_add_byte<6>(s1, x.x, x.y); s2 += s1; q[6] = s2;
        s1 = td_add_byte<7>(s1, x.x, x.y); s2 += s1; q[7] = s2;
        uint2 o;                                                         // byte k = q[k] & 0xFF (v_perm: two low bytes, then two halves)
        o.x = __builtin_amdgcn_perm(__builtin_amdgcn_perm(q[3], q[2], 0x0c0c0400u), __builtin_amdgcn_perm(q[1], q[0], 0x0c0c0400u), 0x05040100u);
        o.y = __builtin_amdgcn_perm(__builtin_amdgcn_perm(q[7], q[6], 0x0c0c0400u), __builtin_amdgcn_perm(q[5], q[4], 0x0c0c0400u), 0x05040100u);
        *cell = o;
    }
    wave_lds_fence();
}
// ... and out (round 6): groups of one 16-byte store -- 4 fp32 / 8 fp16 elements -- aligned in the OUTPUT, from a table that is
// aligned to the chunk: the group's bytes re-aligned with v_alignbyte from aligned dwords (group G of the window covers table
// bytes [g G - al, g G - al + g), al = start mod g), and every byte turned into its output value by ONE LDS read: `lut_q` = LDS
// address of entry qp of the workgroup's 512-entry table, entry i = the output bits of int8(i & 0xFF) under the tensor's scale
// (dequant<MODE>, converted once; the table byte is q - qp mod 256, so byte + qp < 512 needs no wrap).  Before: byte extract, add,
// sign extension, conversion, the exact division by 127 and the scale per element, and for fp32 a lane's 8 elements exchanged
// inside the quad so that a store instruction writes whole sectors -- 92 vector instructions per 8 elements, a third of the
// kernel's; now a lane takes 4 consecutive fp32 elements per store (lanes 16 bytes apart: a contiguous KiB per instruction).
// The groups at the window's two ends that hold fewer of its elements than a whole group go element by element (the others belong
// to the neighbouring chunk's wave, or to the window before / behind).
template <bool F32>
__device__ __forceinline__ uint32_t td_lut_entry(uint32_t lut_q, uint32_t bytes, int k)
{
    typedef const uint32_t __attribute__((address_space(3)))* l_u32_p;
    return *(l_u32_p)(static_cast<uintptr_t>(lut_q + (((bytes >> (8 * k)) & 0xFFu) << 2)));
}
template <int MODE, bool F32>
__device__ __forceinline__ void td_window_store(const uint8_t* tab, uint32_t wo, uint64_t start, uint64_t end, uint32_t lut_q,
                                                uint8_t* __restrict__ dst, uint32_t lane)
{
    constexpr uint32_t kG = F32 ? 4u : 8u;                              // elements per 16-byte store
    const uint32_t al = static_cast<uint32_t>(start) & (kG - 1u);
    const uint64_t obase = (start & ~static_cast<uint64_t>(kG - 1u)) + wo;      // output position of group 0's first element
    const int32_t lo_i = static_cast<int32_t>(al);                      // group-relative bounds of what this window stores
    const uint64_t wend = start + wo + kTdfWin < end ? start + wo + kTdfWin : end;
    const int32_t hi_i = static_cast<int32_t>(wend - obase);
    const uint32_t tab_addr = lds_addr_of(tab);
    typedef const uint32_t __attribute__((address_space(3)))* l_u32_p;
    typedef u32x4 __attribute__((address_space(1)))* g_u32x4_p;
    const uint32_t sh = (0u - al) & 3u;                                 // (table byte of a group's first element) mod 4: the same for every group
#pragma unroll 1
    for (uint32_t g0 = 0; static_cast<int32_t>(g0) < hi_i; g0 += 512u) {
#pragma unroll
        for (uint32_t part = 0; part < (F32 ? 2u : 1u); ++part) {
            const int32_t q0 = static_cast<int32_t>(g0 + (F32 ? 256u * part + 4u * lane : 8u * lane));     // group-relative element index of the lane's group
            if (q0 + static_cast<int32_t>(kG) <= lo_i || q0 >= hi_i) continue;
            const int32_t tb = q0 - static_cast<int32_t>(al);            // table byte of its first element (>= 1 - kG)
            const uint32_t da = tab_addr + static_cast<uint32_t>((tb >> 2) << 2);      // (arithmetic shift: -1 -> the dword in front)
            const uint32_t a0 = *(l_u32_p)(static_cast<uintptr_t>(da)), a1 = *(l_u32_p)(static_cast<uintptr_t>(da + 4u));
            const uint32_t lo4 = __builtin_amdgcn_alignbyte(a1, a0, sh);
            uint32_t hi4 = 0u;
            if (!F32) { const uint32_t a2 = *(l_u32_p)(static_cast<uintptr_t>(da + 8u)); hi4 = __builtin_amdgcn_alignbyte(a2, a1, sh); }
            uint32_t y[8];
#pragma unroll
            for (int e = 0; e < static_cast<int>(kG); ++e) y[e] = td_lut_entry<F32>(lut_q, e < 4 ? lo4 : hi4, e & 3);
            const uint64_t o = obase + static_cast<uint64_t>(q0);
            if (q0 >= lo_i && q0 + static_cast<int32_t>(kG) <= hi_i) {
                u32x4 v;
                if (F32) { v.x = y[0]; v.y = y[1]; v.z = y[2]; v.w = y[3]; }
                else { v.x = y[0] | (y[1] << 16); v.y = y[2] | (y[3] << 16); v.z = y[4] | (y[5] << 16); v.w = y[6] | (y[7] << 16); }
                __builtin_nontemporal_store(v, (g_u32x4_p)(reinterpret_cast<uintptr_t>(dst) + (F32 ? 4ull : 2ull) * o));
            } else {
                for (int e = 0; e < static_cast<int>(kG); ++e) {
                    if (q0 + e < lo_i || q0 + e >= hi_i) continue;
                    if (F32) reinterpret_cast<uint32_t*>(dst)[o + e] = y[e];
                    else reinterpret_cast<uint16_t*>(dst)[o + e] = static_cast<uint16_t>(y[e]);
                }
            }
        }
    }
    wave_lds_fence();
}
// A chunk of long runs: its windows behind the first, with the pairs read again (kept in registers over the first store pass
// they would cost the kernel a workgroup per CU).  Out of line: data that does not compress never gets here.
template <int MODE, bool F32>
__device__ __noinline__ void td_windows_behind_the_first(const uint8_t* __restrict__ rle, uint64_t p0, uint64_t n_pairs, uint64_t start,
                                                         uint64_t end, uint32_t lut_q, uint32_t tot_c, uint32_t c1, uint32_t c2, uint8_t* tab,
                                                         uint8_t* __restrict__ dst, uint32_t lane)
{
    uint32_t w[4][4], ex[4], st[4], mn = 255u;
    td_load_pairs(rle, p0, n_pairs, true, lane, w, mn);
    td_scan_pairs(w, ex, st);
#pragma unroll 1
    for (uint32_t wo = kTdfWin; start + wo < end; wo += kTdfWin) {
        td_window_values(w, ex, st, wo, tot_c, tab, lane, c1, c2);
        td_window_store<MODE, F32>(tab, wo, start, end, lut_q, dst, lane);
    }
}

// BATCH (speckv_ext_codec_decompress_tensors): one workgroup per stream, rounds of kTdfChunks chunks, the prefix in front of a
// round carried in LDS (see tc_fused_body).
template <int MODE, bool F32, bool BATCH, uint32_t WAVES = kTdfWaves>
__device__ __forceinline__ void td_fused_body(const uint8_t* __restrict__ rle, uint64_t n_pairs, uint64_t n_chunks, uint64_t wg0, uint64_t* __restrict__ status,
                                              uint64_t cap, uint64_t* __restrict__ out_n, float scale, uint8_t* __restrict__ dst)
{
    constexpr uint32_t kFront = 16;           // bytes in front of a wave's table (the store pass may read up to 7 of them: never used)
    __shared__ __attribute__((aligned(16))) uint8_t tabs[WAVES][kFront + kTdfWin + 16];      // (+16 behind: written, never used)
    __shared__ uint32_t s_tot[WAVES];
    __shared__ uint64_t s_start[WAVES];
    __shared__ uint32_t s_qp[WAVES];
    __shared__ uint64_t s_carry_before;                                 // BATCH: elements / int8 prefix in front of the round
    __shared__ uint32_t s_carry_q;
    __shared__ uint32_t s_lut[512];                                     // output bits of int8(i & 0xFF) under this tensor's scale (td_window_store)
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    for (uint32_t i = threadIdx.x; i < 512u; i += 64u * WAVES) {    // (published by the first barrier of round 0)
        const float y = dequant<MODE>(static_cast<int>(static_cast<int8_t>(i & 0xFFu)), scale);
        float a = y, z = 0.0f;
        s_lut[i] = F32 ? __float_as_uint(y) : (pack_half2(a, z) & 0xFFFFu);
    }
    if (BATCH && threadIdx.x == 0u) { s_carry_before = 0ull; s_carry_q = 0u; }       // (published by the first barrier of round 0)
    const uint64_t n_rounds = BATCH ? (n_chunks + (WAVES - kTdfLbWave) - 1u) / (WAVES - kTdfLbWave) : 1u;
#pragma unroll 1
    for (uint64_t round = 0; round < n_rounds; ++round) {
    const uint64_t wg = BATCH ? round : wg0;
    const uint64_t chunk = wg * (WAVES - kTdfLbWave) + wave - kTdfLbWave;
    const bool live = wave >= kTdfLbWave && chunk < n_chunks;           // (waves behind the last chunk only keep the barriers company)
    const uint64_t p0 = chunk * kTile;
    uint32_t w[4][4], ex[4], st[4], mn = 255u;
    td_load_pairs(rle, p0, n_pairs, live, lane, w, mn);
    td_scan_pairs(w, ex, st);
    const uint32_t tot_c = (st[0] & 0xFFFFFFu) + (st[1] & 0xFFFFFFu) + (st[2] & 0xFFFFFFu) + (st[3] & 0xFFFFFFu);     // < 2^24 (2048 x 255)
    const uint32_t tot_v = ((st[0] >> 24) + (st[1] >> 24) + (st[2] >> 24) + (st[3] >> 24)) & 0xFFu;
    const bool has_zero = __ballot(mn == 0u) != 0ull;
    if (lane == 0u) s_tot[wave] = (tot_v << 24) | tot_c;
    __syncthreads();
    // ---- wave 0: the workgroup's aggregate goes out at once (nobody in front of it is needed for that)
    if (wave == 0u) {
        const uint32_t wt = lane < WAVES ? s_tot[lane] : 0u;
        const uint32_t agg_c = lane63(wave_incl_add(wt & 0xFFFFFFu)), agg_v = lane63(wave_incl_add(wt >> 24)) & 0xFFu;      // < 2^24 x 16
        if (!BATCH && lane == 0u) tc_lb_store(status + wg, wg == 0u ? 2ull : 1ull, (static_cast<uint64_t>(agg_v) << 54) | agg_c);
    }
    // ---- the first window's values: nothing in them needs the look-back, which they therefore hide
    uint8_t* tab = tabs[wave] + kFront;
    uint32_t c1 = 0u, c2 = 0u;
    if (live && tot_c != 0u && !has_zero) td_window_values(w, ex, st, 0u, tot_c, tab, lane, c1, c2);
    // ---- wave 0: the prefix in front of the workgroup by look-back, every wave's own start; then everybody knows where it writes
    if (wave == 0u) {
        const uint32_t wt = lane < WAVES ? s_tot[lane] : 0u;         // (again: cheaper than keeping them over the window)
        const uint32_t wic = wave_incl_add(wt & 0xFFFFFFu), wiv = wave_incl_add(wt >> 24);
        const uint32_t agg_c = lane63(wic), agg_v = lane63(wiv) & 0xFFu;
        uint64_t before = 0;
        uint32_t qb = 0;
        if (BATCH) {
            before = s_carry_before; qb = s_carry_q;
            if (lane == 0u) { s_carry_before = before + agg_c; s_carry_q = (qb + agg_v) & 0xFFu; }
        } else if (wg != 0u) {
#ifdef SPECKV_TD_NO_LB
            before = wg * (WAVES - kTdfLbWave) * 2048ull;                         // (timing builds only: wrong output)
#else
            td_look_back(status, wg, lane, before, qb);
#endif
            qb &= 0xFFu;
            if (lane == 0u) tc_lb_store(status + wg, 2ull, (static_cast<uint64_t>((qb + agg_v) & 0xFFu) << 54) | (before + agg_c));
        }
        if (lane < WAVES) {
            s_start[lane] = before + (wic - (wt & 0xFFFFFFu));
            s_qp[lane] = (qb + wiv - (wt >> 24)) & 0xFFu;
        }
        if (lane == 0u && (wg + 1u) * (WAVES - kTdfLbWave) >= n_chunks) { const uint64_t total = before + agg_c; *out_n = total < cap ? total : cap; }
    }
    __syncthreads();
    if (!live) continue;
    const uint64_t start = s_start[wave];
    const uint32_t qp = s_qp[wave];
    const uint64_t end = (start + tot_c < cap) ? start + tot_c : cap;
    if (start >= end) continue;
    if (has_zero) { td_chunk_general<MODE, F32>(rle, p0, n_pairs, start, end, qp, scale, dst, lane); continue; }
    const uint32_t lut_q = lds_addr_of(reinterpret_cast<const uint8_t*>(s_lut)) + 4u * qp;
    td_window_store<MODE, F32>(tab, 0u, start, end, lut_q, dst, lane);
    if (start + kTdfWin < end) td_windows_behind_the_first<MODE, F32>(rle, p0, n_pairs, start, end, lut_q, tot_c, c1, c2, tab, dst, lane);
    }   // rounds
}

template <int MODE, bool F32>
__global__ __launch_bounds__(64 * kTdfWaves) __attribute__((amdgpu_waves_per_eu(8, 8)))
void k_td_fused(const uint8_t* __restrict__ rle, uint64_t n_pairs, uint64_t n_chunks, uint64_t* __restrict__ status, uint32_t* __restrict__ ticket,
                uint64_t cap, uint64_t* __restrict__ out_n, float scale, uint8_t* __restrict__ dst)
{
    __shared__ uint32_t s_ticket;
    if (threadIdx.x == 0u) s_ticket = atomicAdd(ticket, 1u);
    __syncthreads();
    td_fused_body<MODE, F32, false>(rle, n_pairs, n_chunks, s_ticket, status, cap, out_n, scale, dst);
}

// Many streams, one workgroup each (speckv_ext_codec_decompress_tensors): stream i = desc[i].rle, rle_bytes[i] bytes, scale
// scales[i], decoded into desc[i].data (room for desc[i].n elements); n_out[i] = elements the stream holds, clipped to the room.
template <int MODE, bool F32>
__global__ __launch_bounds__(64 * kTdmWaves) __attribute__((amdgpu_waves_per_eu(8, 8)))
void k_tdb_fused(const TcbDesc* __restrict__ desc, const uint64_t* __restrict__ rle_bytes, const float* __restrict__ scales, uint64_t* __restrict__ n_out)
{
    __shared__ uint64_t s_none;
    const TcbDesc d = desc[blockIdx.x];
    const uint64_t n_pairs = rle_bytes[blockIdx.x] >> 1;                 // an odd trailing byte is dropped (cache_engine.cpp:245)
    const uint64_t chunks = (n_pairs + kTile - 1u) / kTile;
    uint64_t* on = n_out ? n_out + blockIdx.x : &s_none;
    if (chunks == 0u || d.n == 0u) { if (threadIdx.x == 0u) *on = 0ull; return; }
    td_fused_body<MODE, F32, true, kTdmWaves>(d.rle, n_pairs, chunks, 0ull, nullptr, d.n, on, scales[blockIdx.x], static_cast<uint8_t*>(d.data));
}

// ... and with several workgroups per stream (see k_tcm_fused): stream t = ticket / wpt, its workgroup ticket % wpt takes 16 chunks of
// 2048 pairs, the prefix in front of it by look-back over the stream's own status words.
template <int MODE, bool F32>
__global__ __launch_bounds__(64 * kTdmWaves) __attribute__((amdgpu_waves_per_eu(8, 8)))
void k_tdm_fused(const TcbDesc* __restrict__ desc, uint32_t wpt, uint32_t* __restrict__ ticket, uint64_t* __restrict__ status,
                 const uint64_t* __restrict__ rle_bytes, const float* __restrict__ scales, uint64_t* __restrict__ n_out)
{
    __shared__ uint32_t s_ticket;
    __shared__ uint64_t s_none;
    if (threadIdx.x == 0u) s_ticket = atomicAdd(ticket, 1u);
    __syncthreads();
    const uint32_t t = s_ticket / wpt, l = s_ticket - t * wpt;
    const TcbDesc d = desc[t];
    const uint64_t n_pairs = rle_bytes[t] >> 1;                           // an odd trailing byte is dropped (cache_engine.cpp:245)
    const uint64_t chunks = (n_pairs + kTile - 1u) / kTile;
    const uint64_t my_wgs = (chunks + kTdmChunks - 1u) / kTdmChunks;
    uint64_t* on = n_out ? n_out + t : &s_none;
    if (l >= my_wgs || d.n == 0u) {
        if (l == 0u && threadIdx.x == 0u) *on = 0ull;
        return;
    }
    td_fused_body<MODE, F32, false, kTdmWaves>(d.rle, n_pairs, chunks, l, status + static_cast<uint64_t>(t) * wpt, d.n, on, scales[t], static_cast<uint8_t*>(d.data));
}

// ---------------------------------------------------------------- output-centric expand with the run scatter (multi-launch form)
// k_td_expand above lets a lane write the part of its run that falls into the tile element by element: a tile covered by three
// runs of 700 elements is three lanes storing 700 bytes each.  On tensors that compress (long runs, few pairs) BOTH decoders
// collapse -- the one-pass kernel because its work is cut by PAIRS (a chunk of 2048 pairs is a megabyte of output walked by one
// wave), this one because of that loop: 128 Mi elements at 155 : 1 took 0.67-0.71 ms, 5 % of the roofline.  Here a wave owns 2048
// OUTPUT elements as before, but fills its table the block decoder's way: E[start of run] = value - previous value for the runs
// that start inside the tile (one byte per run), the state in front of the tile (q and the delta of element -1) from the packed
// sums of the pairs in front of it, then two wave scans + eight byte recurrences per 512 elements.  Constant work per tile
// whatever the run lengths.  A tile that meets a zero count (hand-made streams) takes td_tile_general.
template <int MODE, bool F32>
__device__ __noinline__ void td_tile_general(const uint8_t* __restrict__ rle, uint64_t n_pairs, uint64_t i0, uint64_t pos0, uint32_t q0,
                                             uint64_t o0, uint64_t o1, float scale, uint8_t* __restrict__ dst, uint32_t lane)
{
    uint64_t pos = pos0;                                                 // element index of pair i0's first element
    uint32_t qa = q0;
#pragma unroll 1
    for (uint64_t b = i0; b < n_pairs && pos < o1; b += 64u) {
        const uint64_t i = b + lane;
        uint32_t bits = 0;
        if (i < n_pairs) bits = *reinterpret_cast<const uint16_t*>(rle + 2ull * i);
        const uint32_t v = bits & 0xFFu, c = bits >> 8;
        const uint32_t packed = ((v * c) << 24) | c;
        const uint32_t incl = wave_incl_add(packed);
        const uint32_t e = incl - packed;
        uint64_t p = pos + (e & 0xFFFFFFu);
        uint32_t q = qa + (e >> 24);
#pragma unroll 1
        for (uint32_t m = 0; m < c && p < o1; ++m, ++p) {
            q += v;
            if (p < o0) continue;
            const float y = dequant<MODE>(static_cast<int>(static_cast<int8_t>(q & 0xFFu)), scale);
            if (F32) reinterpret_cast<float*>(dst)[p] = y;
            else { float a = y, z = 0.0f; reinterpret_cast<uint16_t*>(dst)[p] = static_cast<uint16_t>(pack_half2(a, z) & 0xFFFFu); }
        }
        const uint32_t tot = lane63(incl);
        pos += tot & 0xFFFFFFu;
        qa = (qa + (tot >> 24)) & 0xFFu;
    }
}
template <int MODE, bool F32>
__global__ __launch_bounds__(256) void k_td_expand_scatter(const uint8_t* __restrict__ rle, uint64_t n_pairs, const TdCarry* __restrict__ carry,
                                                          uint64_t n_chunks, const uint64_t* __restrict__ n_out_p, float scale, uint8_t* __restrict__ dst)
{
    __shared__ __attribute__((aligned(16))) uint8_t tabs[4][kTile + 16];      // (+16: bytes that may be written and are never read)
    __shared__ __attribute__((aligned(16))) uint8_t stage[4][F32 ? 2048 : 16];   // fp32 output: the step's values on their way to whole-KiB stores
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint64_t n_out = *n_out_p;
    const uint64_t o0 = (static_cast<uint64_t>(blockIdx.x) * 4u + wave) * kTile;
    if (o0 >= n_out) return;
    const uint64_t o1 = (n_out - o0 < kTile) ? n_out : o0 + kTile;
    uint8_t* tab = tabs[wave];
    const uint32_t tab_addr = lds_addr_of(tab);
    uint64_t lo = 0, hi = n_chunks;                                      // last chunk whose first element is at or before o0 (64-ary search)
    while (hi - lo > 1u) {
        const uint64_t span = hi - lo, step = (span + 63u) / 64u;
        const uint64_t c = lo + static_cast<uint64_t>(lane + 1u) * step;
        const bool le = c < hi && carry[c].start <= o0;
        const uint32_t kk = static_cast<uint32_t>(__popcll(__ballot(le)));
        const uint64_t nlo = lo + static_cast<uint64_t>(kk) * step, nhi = lo + static_cast<uint64_t>(kk + 1u) * step;
        lo = nlo;
        hi = nhi < hi ? nhi : hi;
    }
    const uint32_t valid = static_cast<uint32_t>(o1 - o0);
    const uint64_t chunk_start = carry[lo].start;
    const uint32_t chunk_q = carry[lo].q_pre;
    int32_t tot = -static_cast<int32_t>(o0 - chunk_start);               // first element of the step's first pair, relative to o0
    uint32_t qm1 = chunk_q;                                               // becomes q[-1]: everything in front of the tile
    uint32_t dm1 = 0u;                                                   // the delta of element -1 (value of the pair that covers it)
    uint32_t tail = 0u;                                                  // the dword of the pair in front of the step's first (value in byte 2)
    if (lo != 0u) {                                                      // (the last pair of the chunk before: it exists and has been read by the summary pass)
        const uint32_t b = *reinterpret_cast<const uint16_t*>(rle + 2ull * (lo * kTile - 1u));
        tail = (b & 0xFFu) << 16;
        dm1 = b & 0xFFu;
        if ((b >> 8) == 0u) { td_tile_general<MODE, F32>(rle, n_pairs, lo * kTile, chunk_start, chunk_q, o0, o1, scale, dst, lane); return; }
    }
    {
        u32x4* t4 = reinterpret_cast<u32x4*>(tab);
        const u32x4 z = {0u, 0u, 0u, 0u};
        t4[lane] = z; t4[64u + lane] = z;
    }
    const bool wide = (reinterpret_cast<uintptr_t>(rle) & 15u) == 0u;
    bool zero_seen = false;
#pragma unroll 1
    for (uint64_t i0 = lo * kTile; i0 < n_pairs && tot < static_cast<int32_t>(valid); i0 += 512u) {
        const uint64_t i = i0 + 8u * lane;
        uint32_t w[4] = {0u, 0u, 0u, 0u};                                // pairs i, i+1 | i+2, i+3 | ... (v | c << 8 | v << 16 | c << 24)
        uint32_t mn = 255u;
        if (wide && i + 8u <= n_pairs) {
            const u32x4 x = *reinterpret_cast<const u32x4*>(rle + 2ull * i);
            w[0] = x.x; w[1] = x.y; w[2] = x.z; w[3] = x.w;
#pragma unroll
            for (int t = 0; t < 4; ++t) mn = umin(mn, umin((w[t] >> 8) & 0xFFu, w[t] >> 24));
        } else {
#pragma unroll
            for (uint32_t k = 0; k < 8u; ++k)
                if (i + k < n_pairs) {
                    const uint32_t bits = *reinterpret_cast<const uint16_t*>(rle + 2ull * (i + k));
                    w[k >> 1] |= bits << (16u * (k & 1u));
                    mn = umin(mn, bits >> 8);
                }
        }
        if (__ballot(mn == 0u) != 0ull) { zero_seen = true; break; }
        uint32_t sc = 0, sv = 0;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const uint32_t counts = (w[t] >> 8) & 0x00FF00FFu;
            sc = __builtin_amdgcn_udot4(counts, 0x00010001u, sc, false);
            sv = __builtin_amdgcn_udot4(w[t], counts, sv, false);
        }
        const uint32_t lt = (sv << 24) | sc;
        const uint32_t incl = wave_incl_add(lt);
        const uint32_t stp = lane63(incl);
        const int32_t step_c = static_cast<int32_t>(stp & 0xFFFFFFu);
        const uint32_t last_dw = lane63(w[3]);
        if (tot + step_c <= 0) {                                         // (wave-uniform) every element of the step lies in front of the tile
            qm1 += stp >> 24;
            if (step_c != 0) dm1 = (last_dw >> 16) & 0xFFu;              // (a full step: its last pair exists and covers the last element so far)
            tot += step_c;
            tail = last_dw;
            continue;
        }
        const uint32_t ex = incl - lt;
        int32_t a = tot + static_cast<int32_t>(ex & 0xFFFFFFu);           // first element of the lane's first pair
        const uint32_t pwj = wave_shr1(w[3], tail);
        const uint32_t dd[8] = {td_sub_b0_b2(w[0], pwj), td_sub_b2_b0(w[0], w[0]), td_sub_b0_b2(w[1], w[0]), td_sub_b2_b0(w[1], w[1]),
                                td_sub_b0_b2(w[2], w[1]), td_sub_b2_b0(w[2], w[2]), td_sub_b0_b2(w[3], w[2]), td_sub_b2_b0(w[3], w[3])};
        if (tot < 0) {
            // the step straddles the tile's first element: what its pairs put IN FRONT of the tile goes into q[-1], and the last
            // pair that starts in front of the tile is the one that covers element -1
            uint32_t qb = 0u, cov = 0u, has = 0u;
            int32_t st_e = a;
#pragma unroll
            for (uint32_t e = 0; e < 8u; ++e) {
                const uint32_t v = (w[e >> 1] >> (16u * (e & 1u))) & 0xFFu, c = (w[e >> 1] >> (8u + 16u * (e & 1u))) & 0xFFu;
                const int32_t nb = st_e < 0 ? ((-st_e) < static_cast<int32_t>(c) ? -st_e : static_cast<int32_t>(c)) : 0;
                qb += v * static_cast<uint32_t>(nb);
                if (st_e < 0 && c != 0u) { cov = v; has = 1u; }
                st_e += static_cast<int32_t>(c);
            }
            qm1 += lane63(wave_incl_add(qb & 0xFFu));
            const unsigned long long hm = __ballot(has != 0u);
            if (hm) dm1 = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(cov), 63 - __builtin_clzll(hm)));
        }
        if (tot >= 0 && tot + step_c <= static_cast<int32_t>(valid)) {      // (wave-uniform) every pair of the step starts inside the tile: plain addresses
            uint32_t ad = tab_addr + static_cast<uint32_t>(a);
            td_lds_b8(ad, dd[0]); ad = td_add_b1(ad, w[0]);
            td_lds_b8(ad, dd[1]); ad = td_add_b3(ad, w[0]);
            td_lds_b8(ad, dd[2]); ad = td_add_b1(ad, w[1]);
            td_lds_b8(ad, dd[3]); ad = td_add_b3(ad, w[1]);
            td_lds_b8(ad, dd[4]); ad = td_add_b1(ad, w[2]);
            td_lds_b8(ad, dd[5]); ad = td_add_b3(ad, w[2]);
            td_lds_b8(ad, dd[6]); ad = td_add_b1(ad, w[3]);
            td_lds_b8(ad, dd[7]);                                         // (pairs behind the stream's end: count 0, they land on the byte behind the last element)
        } else {
#pragma unroll
            for (uint32_t e = 0; e < 8u; ++e) {
                const uint32_t off = static_cast<uint32_t>(a) < valid ? static_cast<uint32_t>(a) : kTile;  // in front of / behind the tile: the spare byte
                td_lds_b8(tab_addr + off, dd[e]);
                a += static_cast<int32_t>((w[e >> 1] >> (8u + 16u * (e & 1u))) & 0xFFu);
            }
        }
        tot += step_c;
        tail = last_dw;
    }
    if (zero_seen) { td_tile_general<MODE, F32>(rle, n_pairs, lo * kTile, chunk_start, chunk_q, o0, o1, scale, dst, lane); return; }
    wave_lds_fence();
    uint32_t c1 = dm1, c2 = qm1;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const uint32_t p0 = 512u * j + 8u * lane;
        if (512u * j >= valid) break;                                    // (wave-uniform)
        const uint2 x = *reinterpret_cast<const uint2*>(tab + p0);
        const uint32_t t1 = __builtin_amdgcn_sad_u8(x.x, 0u, __builtin_amdgcn_sad_u8(x.y, 0u, 0u));
        const uint32_t t2 = __builtin_amdgcn_udot4(x.x, 0x05060708u, __builtin_amdgcn_udot4(x.y, 0x01020304u, 0u, false), false);
        const uint32_t i1 = wave_incl_add(t1);
        const uint32_t x1 = c1 + i1 - t1;
        const uint32_t u = t2 + 8u * x1;
        const uint32_t i2 = wave_incl_add(u);
        const uint32_t x2 = c2 + i2 - u;
        c1 += lane63(i1);
        c2 += lane63(i2);
        if (!F32 && p0 >= valid) continue;                               // (fp32: the lane may have to store a neighbour's values, below)
        uint32_t s1 = x1, s2 = x2;
        uint32_t q[8];
        s1 = td_add_byte<0>(s1, x.x, x.y); s2 += s1; q[0] = s2;
        s1 = td_add_byte<1>(s1, x.x, x.y); s2 += s1; q[1] = s2;
        s1 = td_add_byte<2>(s1, x.x, x.y); s2 += s1; q[2] = s2;
        s1 = td_add_byte<3>(s1, x.x, x.y); s2 += s1; q[3] = s2;
        s1 = td_add_byte<4>(s1, x.x, x.y); s2 += s1; q[4] = s2;
        s1 = td_add_byte<5>(s1, x.x, x.y); s2 += s1; q[5] = s2;
        s1 = td_add_byte<6>(s1, x.x, x.y); s2 += s1; q[6] = s2;
        s1 = td_add_byte<7>(s1, x.x, x.y); s2 += s1; q[7] = s2;
        float y[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) y[e] = dequant<MODE>(static_cast<int>(static_cast<int8_t>(q[e] & 0xFFu)), scale);
        if (F32) {
            // fp32: a lane's eight values are 32 bytes; stored from the lane, every instruction of the wave would write half of each
            // 64-byte sector (the block decoders lost two thirds of their rate that way: kernels.hip store8_f32).  Through 2 KiB of
            // LDS instead: written per lane, read back 16 bytes x 64 lanes in a row -- one contiguous KiB per store instruction.
            // Piece t of the step (16 bytes) holds half of lane t / 2's group and goes out only if that group is whole.
            typedef float f32x4v __attribute__((ext_vector_type(4)));
            f32x4v* stg = reinterpret_cast<f32x4v*>(stage[wave]);
            wave_lds_fence();
            stg[2u * lane] = f32x4v{y[0], y[1], y[2], y[3]};
            stg[2u * lane + 1u] = f32x4v{y[4], y[5], y[6], y[7]};
            wave_lds_fence();
            const f32x4v a4 = stg[lane], b4 = stg[64u + lane];
            float* ob = reinterpret_cast<float*>(dst) + o0 + 512u * j;
            const uint32_t g0 = 512u * j + 8u * (lane >> 1), g1 = g0 + 256u;       // first element of the groups the two pieces belong to
            if (g0 + 8u <= valid) __builtin_nontemporal_store(a4, reinterpret_cast<f32x4v*>(ob + 4u * lane));
            if (g1 + 8u <= valid) __builtin_nontemporal_store(b4, reinterpret_cast<f32x4v*>(ob + 256u + 4u * lane));
            if (p0 < valid && p0 + 8u > valid)
                for (uint32_t e = 0; e < 8u && p0 + e < valid; ++e) reinterpret_cast<float*>(dst)[o0 + p0 + e] = y[e];
            continue;
        }
        if (p0 + 8u <= valid) {
            {
                const u32x4 pk = {pack_half2(y[0], y[1]), pack_half2(y[2], y[3]), pack_half2(y[4], y[5]), pack_half2(y[6], y[7])};
                __builtin_nontemporal_store(pk, reinterpret_cast<u32x4*>(reinterpret_cast<uint16_t*>(dst) + o0 + p0));
            }
        } else {
            for (uint32_t e = 0; e < 8u && p0 + e < valid; ++e) {
                if (F32) reinterpret_cast<float*>(dst)[o0 + p0 + e] = y[e];
                else { float a2 = y[e], z = 0.0f; reinterpret_cast<uint16_t*>(dst)[o0 + p0 + e] = static_cast<uint16_t>(pack_half2(a2, z) & 0xFFFFu); }
            }
        }
    }
}

size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

} // namespace

// workspace layout (compress): absmax word (256 B) | TcSummary[tiles] | TcCarry[tiles + 1] | first_run u64[tiles] | pair scratch
size_t tensor_compress_workspace_bytes(uint64_t n)
{
    const uint64_t tiles = (n + kTile - 1) / kTile;
    return 256 + align_up(tiles * sizeof(TcSummary), 256) + align_up((tiles + 1) * sizeof(TcCarry), 256) + align_up(tiles * 8, 256) +
           tiles * (2ull * kTile);
}
size_t tensor_decompress_workspace_bytes(uint64_t rle_bytes)
{
    const uint64_t chunks = ((rle_bytes >> 1) + kTile - 1) / kTile;
    return align_up(chunks * sizeof(TdSummary), 256) + align_up((chunks + 1) * sizeof(TdCarry), 256) + 256 +
           align_up(((chunks + 63) / 64) * 16, 256);                 // + the step totals of the two-grid scan
}

hipError_t launch_tensor_compress(const void* d_src, uint64_t n, bool src_f32, uint8_t* d_rle, uint64_t* d_rle_bytes, float* d_scale,
                                  void* d_ws, size_t ws_bytes, int quant_mode, hipStream_t s)
{
    if (ws_bytes < tensor_compress_workspace_bytes(n) || (reinterpret_cast<uintptr_t>(d_ws) & 255u)) return hipErrorInvalidValue;
    const uint64_t tiles = (n + kTile - 1) / kTile;
    uint8_t* w = static_cast<uint8_t*>(d_ws);
    uint32_t* absmax = reinterpret_cast<uint32_t*>(w); w += 256;
    TcSummary* summ = reinterpret_cast<TcSummary*>(w); w += align_up(tiles * sizeof(TcSummary), 256);
    TcCarry* carry = reinterpret_cast<TcCarry*>(w); w += align_up((tiles + 1) * sizeof(TcCarry), 256);
    uint64_t* first_run = reinterpret_cast<uint64_t*>(w); w += align_up(tiles * 8, 256);
    uint8_t* scratch = w;
    // single pass (k_tc_fused) unless a multi-launch form is asked for (SPECKV_TC_MULTIPASS, or one of the scan / pre-emit
    // switches of the tests): its status words and ticket sit where the summaries of the multi-launch form would be
    const Tuning& tn = tuning();
    const bool fused = n != 0 && !tn.tc_multipass && !tn.tc_scan && !tn.tc_no_pre;
    const uint64_t tf_wgs = (tiles + kTfWaves - 1) / kTfWaves;
    hipError_t e = hipMemsetAsync(absmax, 0, fused ? 512 + 16 * tf_wgs : 256, s);
    if (e != hipSuccess) return e;
    if (n) {
        const uint32_t g = static_cast<uint32_t>(std::min<uint64_t>((n + 8191) / 8192, 256));     // 16 bytes per lane and step, one workgroup of 16 waves per CU
        if (src_f32) hipLaunchKernelGGL(k_tc_absmax<true>, dim3(g), dim3(1024), 0, s, d_src, n, absmax);
        else         hipLaunchKernelGGL(k_tc_absmax<false>, dim3(g), dim3(1024), 0, s, d_src, n, absmax);
    }
    if (fused) {
        uint32_t* ticket = reinterpret_cast<uint32_t*>(static_cast<uint8_t*>(d_ws) + 256);
        uint64_t* w1 = reinterpret_cast<uint64_t*>(static_cast<uint8_t*>(d_ws) + 512);
        uint64_t* w2 = w1 + tf_wgs;
        const uint32_t g = static_cast<uint32_t>(tf_wgs);
        const uint32_t no_split = tn.tc_no_split_tiles ? 1u : 0u;
#define SPECKV_TF(MODE, F32) hipLaunchKernelGGL((k_tc_fused<MODE, F32>), dim3(g), dim3(64 * kTfWaves), 0, s, d_src, n, absmax, tiles, w1, w2, ticket, d_rle, d_scale, d_rle_bytes, no_split)
        if (quant_mode == kIntent) { if (src_f32) SPECKV_TF(kIntent, true); else SPECKV_TF(kIntent, false); }
        else                       { if (src_f32) SPECKV_TF(kRefExact, true); else SPECKV_TF(kRefExact, false); }
#undef SPECKV_TF
        return hipGetLastError();
    }
    const uint32_t tg = static_cast<uint32_t>((tiles + kTcWaves - 1) / kTcWaves);
    // fp16 sources: the summary pass emits the tiles its fast path takes (k_tc_tiles, PRE); the flags live behind the step
    // arrays of the three-grid scan, so only with that scan (SPECKV_TC_NO_PRE=1: the two plain passes, A/B and test switch)
    const bool grids = tiles >= 64 && !tn.tc_scan;
    const bool pre = grids && !src_f32 && !kTcNoFast && !tn.tc_no_pre;
    uint8_t* pre_flags = pre ? reinterpret_cast<uint8_t*>(first_run + 3 * ((tiles + 63) / 64)) : nullptr;
#define SPECKV_TC(MODE, F32, EMIT) hipLaunchKernelGGL((k_tc_tiles<MODE, F32, EMIT>), dim3(tg), dim3(64 * kTcWaves), 0, s, d_src, n, absmax, summ, carry, scratch, pre_flags)
#define SPECKV_TC2(EMIT) do { if (quant_mode == kIntent) { if (src_f32) SPECKV_TC(kIntent, true, EMIT); else SPECKV_TC(kIntent, false, EMIT); } \
                              else { if (src_f32) SPECKV_TC(kRefExact, true, EMIT); else SPECKV_TC(kRefExact, false, EMIT); } } while (0)
    if (tiles && pre) {
        if (quant_mode == kIntent) hipLaunchKernelGGL((k_tc_tiles<kIntent, false, false, true>), dim3(tg), dim3(64 * kTcWaves), 0, s, d_src, n, absmax, summ, carry, scratch, pre_flags);
        else                       hipLaunchKernelGGL((k_tc_tiles<kRefExact, false, false, true>), dim3(tg), dim3(64 * kTcWaves), 0, s, d_src, n, absmax, summ, carry, scratch, pre_flags);
    } else if (tiles) SPECKV_TC2(false);
    {
        const uint64_t n_steps = (tiles + 63) / 64;
        if (tiles < 64 || tn.tc_scan == 2)            // (tuning.hpp tc_scan: 1 = one workgroup, 2 = one wave)
            hipLaunchKernelGGL(k_tc_scan, dim3(1), dim3(64), 0, s, summ, carry, tiles, n, absmax, d_scale, d_rle_bytes, first_run);
        else if (tn.tc_scan == 1 && n_steps <= kScanMaxSteps)
            hipLaunchKernelGGL(k_tc_scan_wg, dim3(1), dim3(64 * kScanWaves), 0, s, summ, carry, tiles, n, absmax, d_scale, d_rle_bytes);
        else {
            // the step arrays live where the one-wave scan keeps its first-run starts (tiles x 8 bytes >= 3 x n_steps x 8)
            uint64_t* step_ss = first_run;
            uint64_t* step_runs = first_run + n_steps;
            uint64_t* step_first = first_run + 2 * n_steps;
            const uint32_t g = static_cast<uint32_t>((n_steps + 3) / 4);
            hipLaunchKernelGGL(k_tc_scan_a, dim3(g), dim3(256), 0, s, summ, tiles, step_ss);
            hipLaunchKernelGGL(k_tc_scan_b, dim3(g), dim3(256), 0, s, summ, carry, tiles, n, step_ss, step_runs, step_first);
            hipLaunchKernelGGL(k_tc_scan_c, dim3(g), dim3(256), 0, s, carry, tiles, n, step_runs, step_first, absmax, d_scale, d_rle_bytes);
        }
    }
    if (tiles) {
        SPECKV_TC2(true);
        // the pack grid covers the worst case (one pair per element); waves beyond the stream's end return at once
        const uint64_t chunks = (n + kTile - 1) / kTile;
        hipLaunchKernelGGL(k_tc_pack, dim3(static_cast<uint32_t>((chunks + 3) / 4)), dim3(256), 0, s, carry, tiles, scratch, d_rle);
    }
#undef SPECKV_TC2
#undef SPECKV_TC
    return hipGetLastError();
}

// Many tensors per launch.  d_desc[i] = {source, elements, stream, room} (device array, kernels.hpp TensorDesc); max_elems = the
// largest tensor (the host's bound: it sizes the grid, wpt workgroups per tensor).  Workspace: ticket (256 B) | max|x| words | chain 1 |
// chain 2, one 8-byte word per (tensor, workgroup) each; cleared here.  wpt == 1 (tensors of at most 16 tiles) and SPECKV_TC_BATCH_ONE_WG:
// the one-workgroup-per-tensor kernels (chains in LDS, no workspace words).
// (the two directions cut a tensor into workgroups of their own size: kTfWaves tiles to compress, kTdmChunks chunks to decompress)
static uint64_t tensors_wpt(uint64_t max_elems, uint32_t per_wg)
{
    return std::max<uint64_t>(1, (max_elems + static_cast<uint64_t>(per_wg) * kTile - 1) / (static_cast<uint64_t>(per_wg) * kTile));
}
size_t tensors_workspace_bytes(uint32_t n_tensors, uint64_t max_elems)
{
    return 256 + 8ull * n_tensors * std::max<uint64_t>(3ull * tensors_wpt(max_elems, kTfWaves), tensors_wpt(max_elems, kTdmChunks));
}
hipError_t launch_tensors_compress(uint32_t n_tensors, const TensorDesc* d_desc, uint64_t max_elems, bool src_f32, uint64_t* d_rle_bytes, float* d_scales,
                                   void* d_ws, size_t ws_bytes, int quant_mode, hipStream_t s)
{
    static_assert(sizeof(TcbDesc) == sizeof(TensorDesc) && sizeof(TensorDesc) == 32, "descriptor layout");
    if (n_tensors == 0) return hipSuccess;
    const uint32_t no_split = tuning().tc_no_split_tiles ? 1u : 0u;
    const TcbDesc* dd = reinterpret_cast<const TcbDesc*>(d_desc);
    const uint64_t wpt = tensors_wpt(max_elems, kTfWaves);
    if (wpt == 1 || tuning().tc_batch_one_wg) {
#define SPECKV_TCB(MODE, F32) hipLaunchKernelGGL((k_tcb_fused<MODE, F32>), dim3(n_tensors), dim3(64 * kTfWaves), 0, s, dd, d_scales, d_rle_bytes, no_split)
        if (quant_mode == kIntent) { if (src_f32) SPECKV_TCB(kIntent, true); else SPECKV_TCB(kIntent, false); }
        else                       { if (src_f32) SPECKV_TCB(kRefExact, true); else SPECKV_TCB(kRefExact, false); }
#undef SPECKV_TCB
        return hipGetLastError();
    }
    if (!d_ws || ws_bytes < tensors_workspace_bytes(n_tensors, max_elems) || (reinterpret_cast<uintptr_t>(d_ws) & 255u) || n_tensors * wpt > 0x7FFFFFFFull) return hipErrorInvalidValue;
    const hipError_t e = hipMemsetAsync(d_ws, 0, 256 + 3ull * n_tensors * wpt * 8ull, s);
    if (e != hipSuccess) return e;
    uint32_t* ticket = static_cast<uint32_t*>(d_ws);
    uint64_t* amax = reinterpret_cast<uint64_t*>(static_cast<uint8_t*>(d_ws) + 256);
    uint64_t* w1 = amax + n_tensors * wpt;
    uint64_t* w2 = w1 + n_tensors * wpt;
    const uint32_t g = static_cast<uint32_t>(n_tensors * wpt), w = static_cast<uint32_t>(wpt);
#define SPECKV_TCM(MODE, F32) hipLaunchKernelGGL((k_tcm_fused<MODE, F32>), dim3(g), dim3(64 * kTfWaves), 0, s, dd, w, ticket, amax, w1, w2, d_scales, d_rle_bytes, no_split)
    if (quant_mode == kIntent) { if (src_f32) SPECKV_TCM(kIntent, true); else SPECKV_TCM(kIntent, false); }
    else                       { if (src_f32) SPECKV_TCM(kRefExact, true); else SPECKV_TCM(kRefExact, false); }
#undef SPECKV_TCM
    return hipGetLastError();
}

hipError_t launch_tensors_decompress(uint32_t n_tensors, const TensorDesc* d_desc, uint64_t max_elems, const uint64_t* d_rle_bytes, const float* d_scales, bool out_f32,
                                     uint64_t* d_n_out, void* d_ws, size_t ws_bytes, int quant_mode, hipStream_t s)
{
    if (n_tensors == 0) return hipSuccess;
    const TcbDesc* dd = reinterpret_cast<const TcbDesc*>(d_desc);
    const uint64_t wpt = tensors_wpt(max_elems, kTdmChunks);
    if (wpt == 1 || tuning().tc_batch_one_wg) {
#define SPECKV_TDB(MODE, F32) hipLaunchKernelGGL((k_tdb_fused<MODE, F32>), dim3(n_tensors), dim3(64 * kTdmWaves), 0, s, dd, d_rle_bytes, d_scales, d_n_out)
        if (quant_mode == kIntent) { if (out_f32) SPECKV_TDB(kIntent, true); else SPECKV_TDB(kIntent, false); }
        else                       { if (out_f32) SPECKV_TDB(kRefExact, true); else SPECKV_TDB(kRefExact, false); }
#undef SPECKV_TDB
        return hipGetLastError();
    }
    if (!d_ws || ws_bytes < tensors_workspace_bytes(n_tensors, max_elems) || (reinterpret_cast<uintptr_t>(d_ws) & 255u) || n_tensors * wpt > 0x7FFFFFFFull) return hipErrorInvalidValue;
    const hipError_t e = hipMemsetAsync(d_ws, 0, 256 + n_tensors * wpt * 8ull, s);
    if (e != hipSuccess) return e;
    uint32_t* ticket = static_cast<uint32_t*>(d_ws);
    uint64_t* status = reinterpret_cast<uint64_t*>(static_cast<uint8_t*>(d_ws) + 256);
    const uint32_t g = static_cast<uint32_t>(n_tensors * wpt), w = static_cast<uint32_t>(wpt);
#define SPECKV_TDM(MODE, F32) hipLaunchKernelGGL((k_tdm_fused<MODE, F32>), dim3(g), dim3(64 * kTdmWaves), 0, s, dd, w, ticket, status, d_rle_bytes, d_scales, d_n_out)
    if (quant_mode == kIntent) { if (out_f32) SPECKV_TDM(kIntent, true); else SPECKV_TDM(kIntent, false); }
    else                       { if (out_f32) SPECKV_TDM(kRefExact, true); else SPECKV_TDM(kRefExact, false); }
#undef SPECKV_TDM
    return hipGetLastError();
}

hipError_t launch_tensor_decompress(const uint8_t* d_rle, uint64_t rle_bytes, float scale, void* d_dst, uint64_t dst_cap, bool out_f32,
                                    uint64_t* d_n_out, void* d_ws, size_t ws_bytes, int quant_mode, hipStream_t s)
{
    if (ws_bytes < tensor_decompress_workspace_bytes(rle_bytes) || (reinterpret_cast<uintptr_t>(d_ws) & 255u)) return hipErrorInvalidValue;
    const uint64_t n_pairs = rle_bytes >> 1;                         // an odd trailing byte is dropped (cache_engine.cpp:245)
    const uint64_t chunks = (n_pairs + kTile - 1) / kTile;
    uint8_t* w = static_cast<uint8_t*>(d_ws);
    TdSummary* summ = reinterpret_cast<TdSummary*>(w); w += align_up(chunks * sizeof(TdSummary), 256);
    TdCarry* carry = reinterpret_cast<TdCarry*>(w); w += align_up((chunks + 1) * sizeof(TdCarry), 256);
    uint64_t* n_out = d_n_out ? d_n_out : reinterpret_cast<uint64_t*>(w);
    // Which form: the one-pass kernel cuts the work by PAIRS (a chunk of 2048 pairs per wave) -- right for data that does not
    // compress, but ONE wave per megabyte of output wherever the data does: a stream that is noise with a tail of zero padding
    // leaves a few dozen waves walking hundreds of windows each.  The multi-launch form cuts the work by OUTPUT (2048 elements per
    // wave, constant work per tile whatever the run lengths: k_td_expand_scatter) and pays a summary pass over the stream for it:
    // 20 % behind the one-pass kernel on runs of 4, five times ahead of it on a tensor that is one third noise and two thirds long
    // runs (profiles/r04_long_runs.txt).  So: one pass only for streams that are practically incompressible (at most 1.02 elements
    // per pair), by output otherwise.  SPECKV_TC_MULTIPASS=1 / SPECKV_TD_ONE_PASS=1 force a form (tests, A/B).
    const bool few_pairs = n_pairs * 51u < std::min<uint64_t>(dst_cap, n_pairs * 255u) * 50u;
    const Tuning& tn = tuning();
    if (chunks && dst_cap && !tn.tc_multipass && (!few_pairs || tn.td_one_pass)) {
        // single pass: a memset node (ticket, element count, one status word per workgroup), then ONE kernel
        const uint64_t wgs = (chunks + kTdfChunks - 1) / kTdfChunks;
        uint32_t* ticket = reinterpret_cast<uint32_t*>(d_ws);
        uint64_t* n_out1 = d_n_out ? d_n_out : reinterpret_cast<uint64_t*>(static_cast<uint8_t*>(d_ws) + 128);
        uint64_t* status = reinterpret_cast<uint64_t*>(static_cast<uint8_t*>(d_ws) + 256);
        const hipError_t e = hipMemsetAsync(d_ws, 0, 256 + wgs * 8, s);
        if (e != hipSuccess) return e;
#define SPECKV_TDF(MODE, F32) hipLaunchKernelGGL((k_td_fused<MODE, F32>), dim3(static_cast<uint32_t>(wgs)), dim3(64 * kTdfWaves), 0, s, d_rle, n_pairs, chunks, status, ticket, dst_cap, n_out1, scale, static_cast<uint8_t*>(d_dst))
        if (quant_mode == kIntent) { if (out_f32) SPECKV_TDF(kIntent, true); else SPECKV_TDF(kIntent, false); }
        else                       { if (out_f32) SPECKV_TDF(kRefExact, true); else SPECKV_TDF(kRefExact, false); }
#undef SPECKV_TDF
        return hipGetLastError();
    }
    if (chunks) hipLaunchKernelGGL(k_td_summary, dim3(static_cast<uint32_t>((chunks + 3) / 4)), dim3(256), 0, s, d_rle, n_pairs, summ);
    const uint64_t n_steps = (chunks + 63) / 64;
    if (chunks == 0 || tn.tc_scan == 2)
        hipLaunchKernelGGL(k_td_scan, dim3(1), dim3(64), 0, s, summ, carry, chunks, dst_cap, n_out);
    else if (tn.tc_scan == 1 && n_steps <= kScanMaxSteps)
        hipLaunchKernelGGL(k_td_scan_wg, dim3(1), dim3(64 * kScanWaves), 0, s, summ, carry, chunks, dst_cap, n_out);
    else {
        TdStepTotal* step_tot = reinterpret_cast<TdStepTotal*>(w + 256);
        const uint32_t g = static_cast<uint32_t>((n_steps + 3) / 4);
        hipLaunchKernelGGL(k_td_scan_local, dim3(g), dim3(256), 0, s, summ, carry, chunks, step_tot);
        hipLaunchKernelGGL(k_td_scan_apply, dim3(g), dim3(256), 0, s, carry, chunks, step_tot, dst_cap, n_out);
    }
    if (chunks && dst_cap) {
        // grid: the output can hold at most min(dst_cap, 255 * n_pairs) elements; waves behind the stream's total return at once
        const uint64_t max_out = std::min<uint64_t>(dst_cap, n_pairs * 255u);
        const uint32_t g = static_cast<uint32_t>(((max_out + kTile - 1) / kTile + 3) / 4);
        const bool old_expand = tn.td_expand_per_element != 0;             // (tests: the expand loop of rounds 2-3)
#define SPECKV_TD(MODE, F32) do { if (old_expand) hipLaunchKernelGGL((k_td_expand<MODE, F32>), dim3(g), dim3(256), 0, s, d_rle, n_pairs, carry, chunks, n_out, scale, static_cast<uint8_t*>(d_dst)); \
                                  else hipLaunchKernelGGL((k_td_expand_scatter<MODE, F32>), dim3(g), dim3(256), 0, s, d_rle, n_pairs, carry, chunks, n_out, scale, static_cast<uint8_t*>(d_dst)); } while (0)
        if (quant_mode == kIntent) { if (out_f32) SPECKV_TD(kIntent, true); else SPECKV_TD(kIntent, false); }
        else                       { if (out_f32) SPECKV_TD(kRefExact, true); else SPECKV_TD(kRefExact, false); }
#undef SPECKV_TD
    }
    return hipGetLastError();
}

} // namespace speckv
