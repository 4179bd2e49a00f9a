// cxl-speckv_amd/csrc/kernels.hip -- hand-written CDNA4 (gfx950) kernels of the
// KV hot path: block compress, fetch + decompress, prefetch lookup, verify.
//
// What they replace in the reference (paths under /root/reference):
//   FPGACacheEngine::compress / ::decompress   src/fpga_engine/cache_engine.cpp:40-116,172-284
//   (RTL twins hardware/rtl/kv_compress.v, kv_decompress.v)
//   prefetch_core lookup loop                  hardware/rtl/prefetch_core.v:150-241
//   SpeculativePrefetcher::handle_misprediction src/prefetcher/speculative_prefetcher.cpp:84-96
//
// Execution model.  These are HBM-bound byte kernels, not GEMMs: no MFMA.
// One wavefront (64 lanes) owns one 2048-element block; lane l holds elements
// [8l+512j, 8l+512j+8), j=0..3, so each global access is a coalesced 1 KiB
// instruction (16 B per lane).  The serial recurrences of the reference
// (run-length expand, int8 delta chain) become wave-level scans:
//   * run start positions   = exclusive add-scan of the run counts
//   * delta prefix at a run = exclusive add-scan (mod 256) of value*count,
//     carried in the top byte of the same 32-bit scan word
//   * "which run covers output p" = max-scan over a 2048-entry head table in
//     LDS into which every run scatters one word at its start position
// Scans are DPP (row_shr / row_bcast) inside the wave; there is no workgroup
// barrier anywhere, so wavefronts never wait for each other.
#include "kernels.hpp"
#include "tuning.hpp"
#include "ring_rule.hpp"
#include "codec_device.hpp"
#include "encode_device.hpp"

#include <hip/hip_fp16.h>

#include <cstdlib>
#include <cstring>

namespace speckv {
namespace {

#ifndef SPECKV_WAVES
#define SPECKV_WAVES 4
#endif
constexpr int kWaves = SPECKV_WAVES;      // wavefronts per workgroup (independent of each other)
constexpr int kThreads = 64 * kWaves;
constexpr int kDecLdsWords = 528;         // per wave: 2 KiB byte table + 64 B of write-only dummies
constexpr int kEncLdsHalves = 2064;        // per wave: 4128 B (see kEncWaveBytes)

// Streaming accesses: every record byte is read once and every output byte
// written once per launch, so both are marked non-temporal (measured on MI355X,
// 131072 blocks: nt stores +4..8 %, nt loads+stores +6..10 % over plain accesses;
// -DSPECKV_PLAIN_LOAD / -DSPECKV_PLAIN_STORE restore the plain forms for A/B).
#if !defined(SPECKV_PLAIN_LOAD)
#define SPECKV_NT_LOAD 1
#endif
#if !defined(SPECKV_PLAIN_STORE)
#define SPECKV_NT_STORE 1
#endif
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
// Records and block images are reached through pointers read from page-table entries / pointer lists, which the compiler cannot
// trace back to a kernel argument: plain C++ dereferences of them become FLAT loads and stores (an address-space check per
// access, and every one counts in lgkmcnt as well as vmcnt, so the waits in front of LDS and scalar reads wait for them too).
// The codec's accesses therefore go through explicit global-address-space pointers (attend.hip: the same hazard cost the
// linear FP8 attention kernel 15 %).
typedef u32x4 __attribute__((address_space(1))) g_u32x4;
template <typename T> __device__ __forceinline__ T gload(const void* p)
{
    typedef T __attribute__((address_space(1))) G;
    return *(const G*)(reinterpret_cast<uintptr_t>(p));
}
template <typename T> __device__ __forceinline__ void gstore(void* p, T v)
{
    typedef T __attribute__((address_space(1))) G;
    *(G*)(reinterpret_cast<uintptr_t>(p)) = v;
}
__device__ __forceinline__ uint2 gload_u2(const void* p)
{
    typedef uint32_t v2 __attribute__((ext_vector_type(2)));
    const v2 v = gload<v2>(p);
    return make_uint2(v.x, v.y);
}
__device__ __forceinline__ uint4 ld16(const uint8_t* p)
{
    const g_u32x4* gp = (const g_u32x4*)(reinterpret_cast<uintptr_t>(p));
#if defined(SPECKV_NT_LOAD)
    const u32x4 v = __builtin_nontemporal_load(gp);
#else
    const u32x4 v = *gp;
#endif
    return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void st16(uint8_t* p, uint4 v)
{
    g_u32x4* gp = (g_u32x4*)(reinterpret_cast<uintptr_t>(p));
    const u32x4 t = {v.x, v.y, v.z, v.w};
#if defined(SPECKV_NT_STORE)
    __builtin_nontemporal_store(t, gp);
#else
    *gp = t;
#endif
}
// The compress direction streams too: every source byte is read once, every record byte written once (the fp16 "compress"
// is a plain copy and ran at 0.65-0.67 of HBM peak with temporal accesses against 0.77 for the same copy in the decode
// direction; -DSPECKV_ENC_PLAIN restores them for the A/B).
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint4 enc_ld16(const uint8_t* p)
{
#if defined(SPECKV_ENC_PLAIN)
    const u32x4 v = *(const g_u32x4*)(reinterpret_cast<uintptr_t>(p));
    return make_uint4(v.x, v.y, v.z, v.w);
#else
    return ld16(p);
#endif
}
__device__ __forceinline__ void enc_st16(uint8_t* p, uint4 v)
{
#if defined(SPECKV_ENC_PLAIN)
    *(g_u32x4*)(reinterpret_cast<uintptr_t>(p)) = u32x4{v.x, v.y, v.z, v.w};
#else
    st16(p, v);
#endif
}
__device__ __forceinline__ void enc_st8(uint8_t* p, uint2 v)
{
    typedef u32x2 __attribute__((address_space(1))) g_u32x2;
    g_u32x2* gp = (g_u32x2*)(reinterpret_cast<uintptr_t>(p));
    const u32x2 t = {v.x, v.y};
#if defined(SPECKV_ENC_PLAIN)
    *gp = t;
#else
    __builtin_nontemporal_store(t, gp);
#endif
}
// fp32 output: a lane's eight values are 32 bytes, so two 16-byte stores per lane at a 32-byte stride -- every store instruction
// of the wave then writes a 16-byte piece of every other 32, half of each 64-byte sector of memory, and the sector is written again by
// the second instruction: 512 MiB of fp32 output took 558 us where the fp16 form of the same blocks takes 171 (0.36 of the HBM
// roofline against 0.78).  The values go through 2 KiB of the wave's LDS instead (written per lane, read back 16 bytes x 64 lanes in
// a row), so that each instruction writes one contiguous KiB, as the fp16 form does.  p0 = 512 j + 8 lane (every caller's layout).
__device__ __forceinline__ void store8_f32(uint8_t* dst, uint32_t p0, const float (&y)[8])
{
    __shared__ __attribute__((aligned(16))) uint4 stage[kWaves][128];
    const uint32_t lane = threadIdx.x & 63u;
    uint4* st = stage[(threadIdx.x >> 6) % kWaves];
    wave_lds_fence();                                       // (the reads of the chunk before are done: LDS is in order per wave)
    st[2u * lane] = make_uint4(__float_as_uint(y[0]), __float_as_uint(y[1]), __float_as_uint(y[2]), __float_as_uint(y[3]));
    st[2u * lane + 1u] = make_uint4(__float_as_uint(y[4]), __float_as_uint(y[5]), __float_as_uint(y[6]), __float_as_uint(y[7]));
    wave_lds_fence();
    const uint4 a = st[lane], b = st[64u + lane];
    uint8_t* base = dst + 4ull * (p0 - 8u * lane) + 16u * lane;
    st16(base, a);
    st16(base + 1024, b);
}
template <bool F32>
__device__ __forceinline__ void store8(uint8_t* dst, uint32_t p0, const float (&y)[8])
{
    if (F32) {
        store8_f32(dst, p0, y);
    } else {
        st16(dst + 2ull * p0, make_uint4(pack_half2(y[0], y[1]), pack_half2(y[2], y[3]),
                                         pack_half2(y[4], y[5]), pack_half2(y[6], y[7])));
    }
}

// ===================================================================
// decode: INT8_DELTA_RLE  (cache_engine.cpp:241-284)
// ===================================================================
// float(q)/127.0f without a divide, 2 ops: fma(q, hi, q*lo) with hi = fl(1/127),
// lo = fl(1/127 - hi) is correctly rounded for every int8 q (exhaustive check in
// tests/test_cabi_boundary.py::test_div127_identity).
template <int MODE>
__device__ __forceinline__ float dequant_q8(uint32_t q8, float scale)
{
    const float fq = static_cast<float>(static_cast<int>(static_cast<int8_t>(q8 & 0xFFu)));
    if (MODE == kRefExact) {
        const float hi = 0x1.020408p-7f, lo = 0x1.020408p-35f;
        const float r = __builtin_fmaf(fq, hi, fq * lo);
        return r * scale;                                  // cache_engine.cpp:279-280
    }
    return fq * scale;
}

// ---- sub-dword VALU forms (SDWA): one instruction extracts a byte AND uses it ----
// (pure register ops, no memory, no hazards beyond what the hardware interlocks)
#define SPECKV_SDWA2(NAME, OP, S0, S1)                                                          \
    __device__ __forceinline__ uint32_t NAME(uint32_t a, uint32_t b)                            \
    {                                                                                           \
        uint32_t r;                                                                             \
        asm(OP " %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:" S0 " src1_sel:" S1   \
            : "=v"(r) : "v"(a), "v"(b));                                                        \
        return r;                                                                               \
    }
SPECKV_SDWA2(add_b0, "v_add_u32_sdwa", "DWORD", "BYTE_0")
SPECKV_SDWA2(add_b1, "v_add_u32_sdwa", "DWORD", "BYTE_1")
SPECKV_SDWA2(add_b2, "v_add_u32_sdwa", "DWORD", "BYTE_2")
SPECKV_SDWA2(add_b3, "v_add_u32_sdwa", "DWORD", "BYTE_3")
SPECKV_SDWA2(min_b1, "v_min_u32_sdwa", "DWORD", "BYTE_1")
SPECKV_SDWA2(min_b3, "v_min_u32_sdwa", "DWORD", "BYTE_3")
SPECKV_SDWA2(sub_b0_b2, "v_sub_u32_sdwa", "BYTE_0", "BYTE_2")   // a.b0 - b.b2
SPECKV_SDWA2(sub_b2_b0, "v_sub_u32_sdwa", "BYTE_2", "BYTE_0")   // a.b2 - b.b0
#undef SPECKV_SDWA2
template <int K> __device__ __forceinline__ uint32_t add_byte(uint32_t a, uint32_t lo, uint32_t hi)
{
    return K == 0 ? add_b0(a, lo) : K == 1 ? add_b1(a, lo) : K == 2 ? add_b2(a, lo) : K == 3 ? add_b3(a, lo)
         : K == 4 ? add_b0(a, hi) : K == 5 ? add_b1(a, hi) : K == 6 ? add_b2(a, hi) : add_b3(a, hi);
}
__device__ __forceinline__ void lds_store_b8(uint32_t addr, uint32_t v)
{
    *reinterpret_cast<lds_u8*>(static_cast<uintptr_t>(addr)) = static_cast<uint8_t>(v);
}

// ---- fast path: well-formed block (every count >= 1, counts sum to 2048) ----
// The decoded int8 sequence is the SECOND-order prefix sum (mod 256) of
//   E[start_r] = value_r - value_{r-1},  0 elsewhere,
// because the expanded deltas are piecewise constant (first prefix sum of E)
// and q is their running sum (second prefix sum).  So each run scatters ONE
// byte into a 2 KiB table at its start position, and every lane then walks its
// 8 table bytes with two adds per element; lane carries come from two DPP
// add-scans per 512-element chunk (v_sad_u8 / v_dot4_u32_u8 give the lane totals).
// Per pair the VALU work is three SDWA instructions: next address, E, min count.
//
// One 512-pair chunk of the pair phase.  A pair dword is  v0 | c0<<8 | v1<<16 | c1<<24.
// Returns false when the running count passes 2048 (caller falls back).
template <bool FULL, bool CHECK>
__device__ __forceinline__ bool rle_pair_chunk(const uint4 wv, uint32_t pair0, uint32_t npairs,
                                               uint32_t tab_addr, uint32_t& ccarry, uint32_t& wtail,
                                               uint32_t& mn)
{
    uint32_t w[4] = {wv.x, wv.y, wv.z, wv.w};
    uint32_t wm[4] = {wv.x, wv.y, wv.z, wv.w};              // count bytes as seen by the zero-count test
    if (!FULL) {
        // pairs at or beyond npairs are not pairs: count 0 for addressing, 0xFF for the min test
        const int nv = static_cast<int>(npairs) - static_cast<int>(pair0);     // valid pairs of this lane
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const uint32_t keep = 0x00FF00FFu | (nv > 2 * t ? 0x0000FF00u : 0u) | (nv > 2 * t + 1 ? 0xFF000000u : 0u);
            w[t] &= keep;
            wm[t] |= ~keep;
        }
    }
    // lane total of the 8 counts (bytes 1 and 3 of each dword)
    uint32_t run = __builtin_amdgcn_udot4(w[0], 0x01000100u, 0u, false);
    run = __builtin_amdgcn_udot4(w[1], 0x01000100u, run, false);
    run = __builtin_amdgcn_udot4(w[2], 0x01000100u, run, false);
    run = __builtin_amdgcn_udot4(w[3], 0x01000100u, run, false);
    const uint32_t incl = wave_incl_add(run);
    const uint32_t base = ccarry + incl - run;              // start position of this lane's first pair
    ccarry += lane63(incl);
    if (ccarry > kBlockElems) return false;                 // wave-uniform; nothing of this chunk is scattered
    // every start below is <= 2048; 2048 itself is a dummy byte behind the table
    const uint32_t prevw = wave_shr1(w[3], wtail);          // previous pair's dword (value in byte 2)
    wtail = lane63(w[3]);
    uint32_t a = tab_addr + base;
    lds_store_b8(a, sub_b0_b2(w[0], prevw)); a = add_b1(a, w[0]);
    lds_store_b8(a, sub_b2_b0(w[0], w[0]));  a = add_b3(a, w[0]);
    lds_store_b8(a, sub_b0_b2(w[1], w[0]));  a = add_b1(a, w[1]);
    lds_store_b8(a, sub_b2_b0(w[1], w[1]));  a = add_b3(a, w[1]);
    lds_store_b8(a, sub_b0_b2(w[2], w[1]));  a = add_b1(a, w[2]);
    lds_store_b8(a, sub_b2_b0(w[2], w[2]));  a = add_b3(a, w[2]);
    lds_store_b8(a, sub_b0_b2(w[3], w[2]));  a = add_b1(a, w[3]);
    lds_store_b8(a, sub_b2_b0(w[3], w[3]));
    if (CHECK) {
#pragma unroll
        for (int t = 0; t < 4; ++t) { mn = min_b1(mn, wm[t]); mn = min_b3(mn, wm[t]); }
    }
    return true;
}

// Returns false (nothing stored) when the block is not well-formed.
// `trusted` (wave-uniform): the stream comes from k_compress -- no zero counts, and the
// bytes between len and the next 16-byte boundary are zero pairs, which scatter into
// the dummy byte; every chunk then takes the unmasked, unchecked form.
// FLAT (the kernel of its own that launches hinted "structured" run, k_fetch_decompress_flat: CodecArgs::structured_hint; the
// kernels the headline fetch runs are instantiated without it): blocks that are piecewise constant on 8-element boundaries
// skip the scans, recurrences and conversions of the general loop -- see the whole-block path in front of it.
template <int MODE, bool F32, bool FLAT = false>
__device__ __forceinline__ bool decode_rle_fast(const uint8_t* __restrict__ rec, uint32_t len,
                                                float scale, uint8_t* __restrict__ dst,
                                                uint8_t* tab, uint32_t lane, bool trusted)
{
    const uint32_t npairs = len >> 1;                       // odd trailing byte dropped
    uint4 w[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const uint32_t pair0 = 512u * j + 8u * lane;
        w[j] = make_uint4(0u, 0u, 0u, 0u);
        if (pair0 < npairs) w[j] = ld16(rec + 2ull * pair0);
    }
    // A stream whose value bytes are all zero (a block of zeros, a never-written page: len 0) decodes to +0 whatever its
    // counts are -- 0/127 * scale with a finite scale of positive sign -- and needs neither table nor scans: plain vector stores.
    // (Otherwise such blocks take the full constant-work path below, or, with len 0, the element-wise general path.)
    {
        uint32_t anyv = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) anyv |= (w[j].x | w[j].y | w[j].z | w[j].w) & 0x00FF00FFu;
        if (__builtin_amdgcn_ballot_w64(anyv != 0u) == 0ull && __float_as_uint(scale) < 0x7F800000u) {    // sign clear (not -0 either), finite
            const float z[8] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int j = 0; j < 4; ++j) store8<F32>(dst, 512u * j + 8u * lane, z);
            return true;
        }
    }
    // clear the byte table (2 KiB); byte 2048 is a write-only dummy
    uint4* t4 = reinterpret_cast<uint4*>(tab);
    t4[lane] = make_uint4(0u, 0u, 0u, 0u);
    t4[64 + lane] = make_uint4(0u, 0u, 0u, 0u);
    const uint32_t tab_addr = static_cast<uint32_t>(reinterpret_cast<uintptr_t>((lds_u8*)tab));

    uint32_t ccarry = 0;        // running count total
    uint32_t wtail = 0;         // last pair dword of the previous chunk (value 0 before the first run)
    uint32_t mn = 255u;         // min count over valid pairs
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (512u * j >= npairs) break;                      // wave-uniform
        const uint32_t pair0 = 512u * j + 8u * lane;
        bool go;
        if (trusted)                       go = rle_pair_chunk<true, false>(w[j], pair0, npairs, tab_addr, ccarry, wtail, mn);
        else if (512u * (j + 1) <= npairs) go = rle_pair_chunk<true, true>(w[j], pair0, npairs, tab_addr, ccarry, wtail, mn);
        else                               go = rle_pair_chunk<false, true>(w[j], pair0, npairs, tab_addr, ccarry, wtail, mn);
        if (!go) return false;
    }
    const bool ok = (ccarry == kBlockElems) && (__ballot(mn == 0u) == 0ull);
    if (!ok) return false;
    wave_lds_fence();

    if (FLAT && npairs <= 256u) {                            // (wave-uniform)
        // The whole block piecewise constant on 8-element boundaries: every lane's table bytes are (+d, -d, 0 x 6) in all four
        // chunks.  Then no slope enters any lane (each lane's bytes sum to 0), a lane's eight elements are ONE value, and that
        // value is the inclusive prefix sum of the +d bytes: one packed scan per two chunks (fields of 16 bits: 64 x 255 fits),
        // one conversion and plain 16-byte stores per chunk -- instead of two scans, eight recurrences and eight conversions.
        // A block that breaks the pattern in any lane takes the general loop below.
        uint2 xs[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) xs[j] = *reinterpret_cast<const uint2*>(tab + 512u * j + 8u * lane);
        uint32_t bad = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) bad |= (xs[j].x & 0xFFFF0000u) | xs[j].y | ((xs[j].x + (xs[j].x >> 8)) & 0xFFu);
        if (__builtin_amdgcn_ballot_w64(bad != 0u) == 0ull) {
            const uint32_t ia = wave_incl_add(__builtin_amdgcn_perm(xs[1].x, xs[0].x, 0x0C040C00u));     // chunk 0 | chunk 1 << 16
            const uint32_t ib = wave_incl_add(__builtin_amdgcn_perm(xs[3].x, xs[2].x, 0x0C040C00u));     // chunk 2 | chunk 3 << 16
            const uint32_t ta = lane63(ia), tb = lane63(ib);
            const uint32_t carry1 = ta & 0xFFFFu, carry2 = carry1 + (ta >> 16), carry3 = carry2 + (tb & 0xFFFFu);
            const uint32_t qv[4] = {ia, (ia >> 16) + carry1, ib + carry2, (ib >> 16) + carry3};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float v = dequant<MODE>(static_cast<int>(static_cast<int8_t>(qv[j] & 0xFFu)), scale);
                const float y8[8] = {v, v, v, v, v, v, v, v};
                store8<F32>(dst, 512u * j + 8u * lane, y8);
            }
            wave_lds_fence();
            return true;
        }
    }
    uint32_t c1 = 0, c2 = 0;    // S1 / S2 at the end of the previous chunk
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const uint32_t p0 = 512u * j + 8u * lane;
        const uint2 x = *reinterpret_cast<const uint2*>(tab + p0);
        const uint32_t t1 = __builtin_amdgcn_sad_u8(x.x, 0u, __builtin_amdgcn_sad_u8(x.y, 0u, 0u));
        const uint32_t t2 = __builtin_amdgcn_udot4(x.x, 0x05060708u,
                                                   __builtin_amdgcn_udot4(x.y, 0x01020304u, 0u, false), false);
        const uint32_t i1 = wave_incl_add(t1);
        const uint32_t x1 = c1 + i1 - t1;                   // S1 entering this lane
        const uint32_t u = t2 + 8u * x1;                    // this lane's S2 increment
        const uint32_t i2 = wave_incl_add(u);
        const uint32_t x2 = c2 + i2 - u;                    // S2 entering this lane
        c1 += lane63(i1);
        c2 += lane63(i2);
        uint32_t s1 = x1, s2 = x2;
        uint32_t q[8];
        s1 = add_byte<0>(s1, x.x, x.y); s2 += s1; q[0] = s2;
        s1 = add_byte<1>(s1, x.x, x.y); s2 += s1; q[1] = s2;
        s1 = add_byte<2>(s1, x.x, x.y); s2 += s1; q[2] = s2;
        s1 = add_byte<3>(s1, x.x, x.y); s2 += s1; q[3] = s2;
        s1 = add_byte<4>(s1, x.x, x.y); s2 += s1; q[4] = s2;
        s1 = add_byte<5>(s1, x.x, x.y); s2 += s1; q[5] = s2;
        s1 = add_byte<6>(s1, x.x, x.y); s2 += s1; q[6] = s2;
        s1 = add_byte<7>(s1, x.x, x.y); s2 += s1; q[7] = s2;
        float y[8];
#pragma unroll
        for (int k = 0; k < 8; k += 2) {
            f32x2 fq;
            fq.x = static_cast<float>(static_cast<int>(static_cast<int8_t>(q[k] & 0xFFu)));
            fq.y = static_cast<float>(static_cast<int>(static_cast<int8_t>(q[k + 1] & 0xFFu)));
            f32x2 r;
            if (MODE == kRefExact) {
                const f32x2 hi = {0x1.020408p-7f, 0x1.020408p-7f}, lo = {0x1.020408p-35f, 0x1.020408p-35f};
                r = __builtin_elementwise_fma(fq, hi, fq * lo);        // float(q)/127.0f, correctly rounded
            } else {
                r = fq;
            }
            const f32x2 sc = {scale, scale};
            r = r * sc;
            y[k] = r.x;
            y[k + 1] = r.y;
        }
        store8<F32>(dst, p0, y);
    }
    wave_lds_fence();
    return true;
}

// ---- general path: any byte stream (zero counts, short or overlong streams) ----
// Rare, so it is written for size, not speed, and needs no LDS at all: one pair
// per lane per step; an add-scan of (value*count mod 256)<<24 | count gives each
// run its start position and the int8 prefix of all earlier deltas, and the lane
// then writes the run's elements itself:  q[start+m] = prefix + (m+1)*value (mod 256).
// Zero counts emit nothing, an odd trailing byte is dropped, output is clipped at
// the block and zero-filled behind the last run (cache_engine.cpp:241-284).
template <bool F32>
__device__ __forceinline__ void store_elem(uint8_t* dst, uint32_t p, float y)
{
    if (F32) {
        gstore<float>(dst + 4ull * p, y);
    } else {
        float a = y, z = 0.0f;
        gstore<uint16_t>(dst + 2ull * p, static_cast<uint16_t>(pack_half2(a, z) & 0xFFFFu));
    }
}
template <int MODE, bool F32>
__device__ __noinline__ void decode_rle_general(const uint8_t* __restrict__ rec, uint32_t len,
                                                float scale, uint8_t* __restrict__ dst, uint32_t lane)
{
    const uint32_t npairs = len >> 1;
    uint32_t carry = 0;                                     // (sum v*c mod 256)<<24 | sum c
#pragma unroll 1
    for (uint32_t b = 0; b < npairs; b += 64u) {
        const uint32_t i = b + lane;
        uint32_t bits = 0;
        if (i < npairs) bits = gload<uint16_t>(rec + 2ull * i);
        const uint32_t v = bits & 0xFFu, c = bits >> 8;
        const uint32_t packed = ((v * c) << 24) | c;
        const uint32_t incl = wave_incl_add(packed);
        const uint32_t e = carry + incl - packed;
        carry += lane63(incl);
        const uint32_t start = e & 0xFFFFFFu;
        uint32_t q = e >> 24;
#pragma unroll 1
        for (uint32_t m = 0; m < c; ++m) {
            const uint32_t p = start + m;
            if (p >= kBlockElems) break;
            q += v;
            store_elem<F32>(dst, p, dequant_q8<MODE>(q, scale));
        }
        if ((carry & 0xFFFFFFu) >= kBlockElems) break;      // wave-uniform: the block is full
    }
    const uint32_t total = carry & 0xFFFFFFu;
#pragma unroll 1
    for (uint32_t p = total + lane; p < kBlockElems; p += 64u) store_elem<F32>(dst, p, 0.0f);
}

// decode: INT8 (quantise only)
template <int MODE, bool F32>
__device__ __forceinline__ void decode_int8(const uint8_t* __restrict__ rec, uint32_t len,
                                            float scale, uint8_t* __restrict__ dst, uint32_t lane)
{
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const uint32_t p0 = 512u * j + 8u * lane;
        uint2 w = make_uint2(0u, 0u);
        if (p0 < len) w = gload_u2(rec + p0);
        float y[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const uint32_t b = ((k < 4 ? w.x : w.y) >> ((k & 3) * 8)) & 0xFFu;
            const int q = static_cast<int>(static_cast<int8_t>(b));
            y[k] = (p0 + k < len) ? dequant<MODE>(q, scale) : 0.0f;
        }
        store8<F32>(dst, p0, y);
    }
}

// decode: FP16 (raw copy, optional widening)
template <bool F32>
__device__ __forceinline__ void decode_fp16(const uint8_t* __restrict__ rec, uint32_t len,
                                            uint8_t* __restrict__ dst, uint32_t lane)
{
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const uint32_t p0 = 512u * j + 8u * lane;
        uint4 w = make_uint4(0u, 0u, 0u, 0u);
        if (2u * p0 < len) w = ld16(rec + 2ull * p0);
        if (!F32) {
            // mask a ragged tail at element granularity
            const uint32_t words[4] = {w.x, w.y, w.z, w.w};
            uint32_t o[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                uint32_t lo = (2u * (p0 + 2 * t) + 1u < len) ? (words[t] & 0xFFFFu) : 0u;
                uint32_t hi = (2u * (p0 + 2 * t + 1) + 1u < len) ? (words[t] & 0xFFFF0000u) : 0u;
                o[t] = lo | hi;
            }
            st16(dst + 2ull * p0, make_uint4(o[0], o[1], o[2], o[3]));
        } else {
            const uint32_t words[4] = {w.x, w.y, w.z, w.w};
            float y[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const uint32_t h = (words[k >> 1] >> ((k & 1) * 16)) & 0xFFFFu;
                y[k] = (2u * (p0 + k) + 1u < len) ? half_bits_to_float(h) : 0.0f;
            }
            store8<true>(dst, p0, y);
        }
    }
}

// decode: INT4_G32 (config 5 extension; record = 64 fp16 scales + 1024 B nibbles)
template <bool F32>
__device__ __forceinline__ void decode_int4(const uint8_t* __restrict__ rec, uint32_t len,
                                            uint8_t* __restrict__ dst, uint32_t lane)
{
    const bool ok = len >= kInt4RecBytes;                    // short record decodes to zeros
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const uint32_t p0 = 512u * j + 8u * lane;
        uint32_t nib = 0;
        float s = 0.0f;
        if (ok) {
            nib = gload<uint32_t>(rec + 128u + (p0 >> 1));
            s = half_bits_to_float(gload<uint16_t>(rec + 2u * (p0 >> 5)));
        }
        float y[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int q = (static_cast<int>(nib << (28 - 4 * k))) >> 28;       // sign-extended nibble
            y[k] = ok ? static_cast<float>(q) * s : 0.0f;
        }
        store8<F32>(dst, p0, y);
    }
}

// decode: MXFP4 (OCP MX v1.0 elements and scales; record = 1024 nibble bytes + 64 E8M0 codes; oracle: compress_mxfp4 / ORC_COMP_MXFP4)
// The two halves of the block are interleaved element by element: byte i = element i (low nibble) and element 1024 + i (high),
// code j scales bytes 16j .. 16j+15.  y = e2m1(nibble) * 2^(code - 127): the nibble pairs widen through
// v_cvt_scalef32_pk_f32_fp4 (exact), the scale is a float whose exponent field is the code (a subnormal for code 0, NaN for
// 255) -- the product is exact in fp32.  A lane takes bytes 8l .. 8l+7 of both 512-byte halves of the nibble area: elements
// [512 j + 8 l, + 8) for j = 0, 1 and their partners 1024 further on (j = 2, 3 of every other decoder's lane map).
typedef float f32x2v __attribute__((ext_vector_type(2)));
template <bool F32>
__device__ __forceinline__ void decode_mx4(const uint8_t* __restrict__ rec, const uint8_t* __restrict__ codes, uint32_t len,
                                           uint8_t* __restrict__ dst, uint32_t lane)
{
    const bool ok = len >= kMx4RecBytes;                     // short record decodes to zeros
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const uint32_t p0 = 512u * j + 8u * lane;
        uint2 nib = make_uint2(0u, 0u);
        uint32_t code = 127u;
        if (ok) {
            nib = gload_u2(rec + p0);
            code = gload<uint8_t>(codes + (p0 >> 4));
        }
        const float s = __uint_as_float(code == 0u ? 0x00400000u : code == 255u ? 0x7FC00000u : code << 23);
        float y0[8], y1[8];
#define MX_DEC(K, W, SEL) { const f32x2v f = __builtin_amdgcn_cvt_scalef32_pk_f32_fp4(W, 1.0f, SEL); y0[K] = f.x * s; y1[K] = f.y * s; }
        MX_DEC(0, nib.x, 0) MX_DEC(1, nib.x, 1) MX_DEC(2, nib.x, 2) MX_DEC(3, nib.x, 3)
        MX_DEC(4, nib.y, 0) MX_DEC(5, nib.y, 1) MX_DEC(6, nib.y, 2) MX_DEC(7, nib.y, 3)
#undef MX_DEC
        store8<F32>(dst, p0, y0);
        store8<F32>(dst, p0 + 1024u, y1);
    }
}

// decode: FP8_E4M3 (config 5 extension; per-block scale)
template <bool F32>
__device__ __forceinline__ void decode_fp8(const uint8_t* __restrict__ rec, uint32_t len,
                                           float scale, uint8_t* __restrict__ dst, uint32_t lane)
{
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const uint32_t p0 = 512u * j + 8u * lane;
        uint2 w = make_uint2(0u, 0u);
        if (p0 < len) w = gload_u2(rec + p0);
        float y[8];
        y[0] = __builtin_amdgcn_cvt_f32_fp8(static_cast<int>(w.x), 0);
        y[1] = __builtin_amdgcn_cvt_f32_fp8(static_cast<int>(w.x), 1);
        y[2] = __builtin_amdgcn_cvt_f32_fp8(static_cast<int>(w.x), 2);
        y[3] = __builtin_amdgcn_cvt_f32_fp8(static_cast<int>(w.x), 3);
        y[4] = __builtin_amdgcn_cvt_f32_fp8(static_cast<int>(w.y), 0);
        y[5] = __builtin_amdgcn_cvt_f32_fp8(static_cast<int>(w.y), 1);
        y[6] = __builtin_amdgcn_cvt_f32_fp8(static_cast<int>(w.y), 2);
        y[7] = __builtin_amdgcn_cvt_f32_fp8(static_cast<int>(w.y), 3);
#pragma unroll
        for (int k = 0; k < 8; ++k) y[k] = (p0 + k < len) ? y[k] * scale : 0.0f;
        store8<F32>(dst, p0, y);
    }
}

// One block's source/destination, all wave-uniform (lives in SGPRs: the block
// index is made provably uniform, so these are scalar loads through the K$).
struct BlockDesc {
    const uint8_t* rec;
    uint8_t* dst;
    uint64_t page;
    uint32_t len;
    float scale;
    uint32_t row;       // table row of the block's allocation (EXT 2: ring bookkeeping / multi-allocation launches)
};
// EXT selects what a launch needs beyond the plain forms, so that the bulk path keeps its register budget
// (62 VGPRs, no SGPR spills, 8 waves per SIMD):
//   0  page table / raw records, range or list            (speckv_ext_fetch_range / _list, codec operators)
//   1  + records staged by the copy engines               (stripe_delta)
//   2  + blocks of several allocations, L2-ring bookkeeping (synchronous misses, device-side flush)
//   3  form 0 + one host-visible word per block (the fetch launch of a device-side flush)
template <int EXT>
__device__ __forceinline__ BlockDesc load_desc(const CodecArgs& a, uint64_t i, uint32_t slot0)
{
    BlockDesc d;
    d.page = a.page_list ? a.page_list[i] : a.first + i;
    d.row = 0;
    const PageEntry* entries = a.entries;
    if (EXT == 2) {
        d.row = a.alloc_list ? a.alloc_list[i] : a.alloc_idx;
        if (a.alloc_list) entries = a.tab[d.row].entries;
    }
    if (EXT == 2 || entries) {
        PageEntry e{0, 0, 1.0f};
        if (entries) e = entries[d.page];                  // a row freed meanwhile decodes as a never-written page
        d.rec = reinterpret_cast<const uint8_t*>(e.pool_addr);
        d.len = e.rec_bytes;
        d.scale = e.scale;
        if (EXT == 1) {                                    // staged copy of the record (copy-engine fetch)
            const uint64_t q = (d.page * a.stripe_magic) >> 35;        // page / stripe_n without a divide (page < 2^28)
            d.rec += a.stripe_delta[d.page - q * a.stripe_n];
        }
    } else {
        d.rec = a.recs + d.page * a.rec_stride;
        d.len = a.rec_bytes[d.page];
        d.scale = a.scales ? a.scales[d.page] : 1.0f;
    }
    if (EXT == 2 && a.ring_owner) d.dst = a.ring_base + (static_cast<uint64_t>(slot0) + i) * kPageSize;
    else d.dst = a.data_list ? reinterpret_cast<uint8_t*>(a.data_list[i]) : a.data + i * a.data_stride;
    return d;
}

// Ring bookkeeping of one fetched block (one lane): see CodecArgs::ring_owner.  Device side: the slot's previous owner
// loses its L2 bit, the new owner is recorded.  Host side: ONE store, the page's ring sequence number -- the host derives
// "still in the ring" from it (Engine::l2_live), so evictions need no host-visible store.  (Every 4-byte store to pinned
// memory is a PCIe transaction of its own: with slot, flags and the evicted page's flags stored per block, a flush of
// 19 000 blocks ran at the link's transaction rate -- 70 us against 42 us without the stores, 47 us with one.)
//
// In two halves, because a miss of one page is nothing but dependent round trips (profiles/r05_access_miss.txt: the kernel
// was 6.3 us of a 12.5 us miss -- page entry, record, then owner word, owner's table row, owner's slot word, own table row,
// one after another behind the decode).  ring_look reads what the bookkeeping needs BEFORE the block is decoded, in three
// steps of wave-uniform loads: {page entry, the slot's previous owner}, {that owner's table row, this block's row}, {the
// owner's slot word}; the last one is an ordinary load that the record's own loads follow without a wait in between.
// ring_note (one lane, after the decode) only stores.  The first two steps are scalar loads written out by hand, each
// statement ending in its own s_waitcnt: left to the compiler the chain becomes vector loads behind branches, every one
// waited for before the next and before the record (the words are written by kernels, so it will not pick scalar loads
// itself; the scalar cache is invalidated when a kernel starts, and within a launch a wave only reads words that no
// other wave of the launch writes, the previous owner's slot word aside -- which is read by a coherent vector load).
struct RingLook {
    uint32_t* evict_flags;      // previous owner's flag word, or nullptr: nothing to evict
    uint32_t  their_slot;       // previous owner's slot word (compared in ring_note: the load stays in flight over the decode)
    uint32_t* d_flags;          // this block's allocation (nullptr: row freed meanwhile)
    uint32_t* d_slot;
    uint32_t* h_slot;
};
typedef uint32_t sgpr4 __attribute__((ext_vector_type(4)));
typedef uint64_t sgpr8 __attribute__((ext_vector_type(4)));     // four pointers: entries, d_flags, d_slot, h_slot (DevAlloc)
// load_desc_ring reads these structs with hand-written scalar loads: one s_load_dwordx4 = a PageEntry {address, bytes, scale}, one
// s_load_dwordx8 = the first four pointers of a DevAlloc row, one s_load_dwordx2 = Layout::alloc_pages.  A reordered or resized
// field would silently defeat the `pp < their_pages` bound and send the atomicAnd on d_flags out of bounds: the layout
// the assembly assumes is pinned here.  (-DSPECKV_RING_VECTOR_LOADS builds the same function from plain loads, for debugging.)
static_assert(sizeof(PageEntry) == 16 && offsetof(PageEntry, pool_addr) == 0 && offsetof(PageEntry, rec_bytes) == 8 && offsetof(PageEntry, scale) == 12,
              "load_desc_ring: s_load_dwordx4 of a PageEntry");
static_assert(offsetof(DevAlloc, entries) == 0 && offsetof(DevAlloc, d_flags) == 8 && offsetof(DevAlloc, d_slot) == 16 && offsetof(DevAlloc, h_slot) == 24,
              "load_desc_ring: s_load_dwordx8 of a DevAlloc row = {entries, d_flags, d_slot, h_slot}");
static_assert(sizeof(Layout::alloc_pages) == 8 && (offsetof(DevAlloc, layout) + offsetof(Layout, alloc_pages)) % 8 == 0,
              "load_desc_ring: s_load_dwordx2 of Layout::alloc_pages");
__device__ __forceinline__ BlockDesc load_desc_ring(const CodecArgs& a, uint64_t i, uint32_t slot0, RingLook& r)
{
    BlockDesc d;
    d.page = a.page_list ? a.page_list[i] : a.first + i;
    d.row = a.alloc_idx;
    const uint32_t slot = slot0 + static_cast<uint32_t>(i);
    d.dst = a.ring_base + static_cast<uint64_t>(slot) * kPageSize;
    const uint64_t me = (static_cast<uint64_t>(d.row) << 32) | d.page;
    const uint64_t* owner_word = a.ring_owner + slot;
    const DevAlloc* mine = a.tab + d.row;
    uint64_t prev;
    sgpr4 e = {0u, 0u, 0u, 0x3f800000u};
#ifdef SPECKV_RING_VECTOR_LOADS
    if (a.entries) { const PageEntry pe = a.entries[d.page]; e = sgpr4{static_cast<uint32_t>(pe.pool_addr), static_cast<uint32_t>(pe.pool_addr >> 32), pe.rec_bytes, __float_as_uint(pe.scale)}; }
    prev = __hip_atomic_load(owner_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
    if (a.entries) {
        const PageEntry* ep = a.entries + d.page;
        asm volatile("s_load_dwordx4 %0, %2, 0x0\n\ts_load_dwordx2 %1, %3, 0x0\n\ts_waitcnt lgkmcnt(0)"
                     : "=&s"(e), "=&s"(prev) : "s"(ep), "s"(owner_word) : "memory");
    } else {                                                 // a row freed meanwhile decodes as a never-written page
        asm volatile("s_load_dwordx2 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=&s"(prev) : "s"(owner_word) : "memory");
    }
#endif
    d.rec = reinterpret_cast<const uint8_t*>((static_cast<uint64_t>(e.y) << 32) | e.x);
    d.len = e.z;
    d.scale = __uint_as_float(e.w);
    const bool other = prev != kNoOwner && prev != me;
    const DevAlloc* theirs = other ? a.tab + (prev >> 32) : mine;
    sgpr8 tp, mp;
    uint64_t their_pages;
#ifdef SPECKV_RING_VECTOR_LOADS
    tp = sgpr8{reinterpret_cast<uint64_t>(theirs->entries), reinterpret_cast<uint64_t>(theirs->d_flags), reinterpret_cast<uint64_t>(theirs->d_slot), reinterpret_cast<uint64_t>(theirs->h_slot)};
    mp = sgpr8{reinterpret_cast<uint64_t>(mine->entries), reinterpret_cast<uint64_t>(mine->d_flags), reinterpret_cast<uint64_t>(mine->d_slot), reinterpret_cast<uint64_t>(mine->h_slot)};
    their_pages = theirs->layout.alloc_pages;
#else
    asm volatile("s_load_dwordx8 %0, %3, 0x0\n\ts_load_dwordx2 %1, %3, %5\n\ts_load_dwordx8 %2, %4, 0x0\n\ts_waitcnt lgkmcnt(0)"
                 : "=&s"(tp), "=&s"(their_pages), "=&s"(mp) : "s"(theirs), "s"(mine),
                   "i"(offsetof(DevAlloc, layout) + offsetof(Layout, alloc_pages)) : "memory");
#endif
    const uint32_t pp = static_cast<uint32_t>(prev);
    // the row may have been recycled for a smaller allocation since the slot was filled
    const bool in_row = other && tp.x != 0 && pp < their_pages;
    const uint32_t* word = in_row ? reinterpret_cast<const uint32_t*>(tp.z) + pp : reinterpret_cast<const uint32_t*>(owner_word);
    r.their_slot = __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    r.evict_flags = in_row ? reinterpret_cast<uint32_t*>(tp.y) + pp : nullptr;
    r.d_flags = mp.x != 0 ? reinterpret_cast<uint32_t*>(mp.y) : nullptr;
    r.d_slot = reinterpret_cast<uint32_t*>(mp.z);
    r.h_slot = reinterpret_cast<uint32_t*>(mp.w);
    return d;
}
__device__ __forceinline__ void ring_note(const CodecArgs& a, const BlockDesc& d, const RingLook& r, uint32_t slot, uint32_t seq)
{
    if (r.evict_flags && r.their_slot == slot) atomicAnd(r.evict_flags, ~2u);      // still pointing here: the page leaves L2
    a.ring_owner[slot] = (static_cast<uint64_t>(d.row) << 32) | d.page;
    if (!r.d_flags) return;
    r.d_slot[d.page] = slot;
    r.h_slot[d.page] = seq;
    atomicOr(&r.d_flags[d.page], 2u);
}

template <int SCHEME, int MODE, bool F32, int EXT, bool FLAT>
__device__ __forceinline__ void fetch_decompress_body(const CodecArgs& a)
{
    __shared__ __attribute__((aligned(16))) uint32_t lds[SCHEME == kInt8DeltaRle ? kWaves * kDecLdsWords : 4];
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    uint64_t n = a.n;
    if (a.n_dev) { const uint64_t nd = *a.n_dev; n = nd < n ? nd : n; }
    uint32_t slot0 = 0, seq0 = 0;
    if (EXT == 2) {
        slot0 = a.slot0_dev ? *a.slot0_dev : a.slot0;
        seq0 = a.seq0_dev ? *a.seq0_dev : a.seq0;
        if (a.hand_ptr && blockIdx.x == 0 && threadIdx.x == 0) *a.hand_ptr = a.new_hand;
    }
    if (EXT == 3) seq0 = *a.seq0_dev;
    // a wave's blocks: one, or (launches beyond the grid cap) `per_wave` of them, one grid apart (wave_step) or
    // consecutive (SPECKV_ROUNDS=consecutive); see codec_grid / round_strided
    const uint64_t per_wave = a.per_wave ? a.per_wave : 1;
    const uint64_t gw = static_cast<uint64_t>(blockIdx.x) * kWaves + wave;
    const uint64_t step = a.wave_step ? a.wave_step : 1;                  // 1: consecutive blocks, else one per round
    uint64_t i = a.wave_step ? gw : gw * per_wave;
    uint64_t end = a.wave_step ? n : ((i + per_wave < n) ? i + per_wave : n);
    if (i >= n) return;
    const bool ring = EXT == 2 && a.ring_owner;                          // (launch-uniform)
    RingLook look{};
    BlockDesc cur{};
    if (!ring) cur = load_desc<EXT>(a, i, slot0);
    for (;;) {
        // the next block's descriptor is fetched while this block is decoded (ring form: in front of its own decode,
        // these launches are short and wait on one block's chain of loads, not on throughput -- and inside the loop:
        // a load issued in front of it is waited for there, whatever uses it)
        if (ring) cur = load_desc_ring(a, i, slot0, look);
        const uint64_t nx = i + step;
        BlockDesc nxt = cur;
        if (nx < end && !ring) nxt = load_desc<EXT>(a, nx, slot0);
        uint32_t len = cur.len;
        if (SCHEME == kInt8DeltaRle) {
            if (len > 2u * kBlockElems) len = 2u * kBlockElems;
            uint32_t* region = lds + wave * kDecLdsWords;
            if (!decode_rle_fast<MODE, F32, FLAT>(cur.rec, len, cur.scale, cur.dst, reinterpret_cast<uint8_t*>(region), lane,
                                                  a.trusted != 0))
                decode_rle_general<MODE, F32>(cur.rec, len, cur.scale, cur.dst, lane);
        } else if (SCHEME == kInt8) {
            if (len > kBlockElems) len = kBlockElems;
            decode_int8<MODE, F32>(cur.rec, len, cur.scale, cur.dst, lane);
        } else if (SCHEME == kInt4G32) {
            decode_int4<F32>(cur.rec, len, cur.dst, lane);
        } else if (SCHEME == kMxFp4) {
            // pool records are tile-planar: the codes lie mx4_code_delta behind the nibbles (the entry's scale word); raw records: 1024
            decode_mx4<F32>(cur.rec, cur.rec + (a.recs ? 1024u : __float_as_uint(cur.scale)), len, cur.dst, lane);
        } else if (SCHEME == kFp8E4m3) {
            if (len > kBlockElems) len = kBlockElems;
            decode_fp8<F32>(cur.rec, len, cur.scale, cur.dst, lane);
        } else {
            if (len > 2u * kBlockElems) len = 2u * kBlockElems;
            decode_fp16<F32>(cur.rec, len, cur.dst, lane);
        }
        if (lane == 0u) {
            if (ring) ring_note(a, cur, look, slot0 + static_cast<uint32_t>(i), seq0 + static_cast<uint32_t>(i));
            else if (EXT == 3) *a.host_words[i] = seq0 + static_cast<uint32_t>(i);
            else if (a.flags) atomicOr(&a.flags[cur.page], a.set_flags);   // neighbours belong to other waves / XCDs
        }
        if (nx >= end) break;
        cur = nxt;
        i = nx;
    }
    if (EXT == 2 && a.done_flag) {                                      // (launch-uniform; one block per wave: n waves had work)
        __threadfence_system();                                          // the page, its residency words and the host-visible word are out
        if (n == 1u) {                                                   // the lone wave has nobody to count
            if (lane == 0u) __hip_atomic_store(a.done_flag, a.done_token, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        } else if (lane == 0u && atomicAdd(a.done_count, 1u) + 1u == static_cast<uint32_t>(n)) {
            *a.done_count = 0u;
            __hip_atomic_store(a.done_flag, a.done_token, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}
template <int SCHEME, int MODE, bool F32, int EXT>
__global__ __launch_bounds__(kThreads) void k_fetch_decompress(CodecArgs a) { fetch_decompress_body<SCHEME, MODE, F32, EXT, false>(a); }
// the FLAT instantiation (data known to compress) as a kernel of its own: these launches are store-bound with short waves, and
// a wave slot more or less per SIMD shows (all-zero blocks 86 us at 8 waves per SIMD, 93 us at 7: the compiler is told to stay at 8)
template <int MODE>
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_fetch_decompress_flat(CodecArgs a)
{
    fetch_decompress_body<kInt8DeltaRle, MODE, false, 0, true>(a);
}
template <int MODE>
__global__ __launch_bounds__(kThreads) void k_fetch_decompress_flat_f32(CodecArgs a) { fetch_decompress_body<kInt8DeltaRle, MODE, true, 0, true>(a); }
// ... and for records the copy engines staged (EXT 1: a remote pool's allocation that compresses)
template <int MODE>
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_fetch_decompress_flat_staged(CodecArgs a)
{
    fetch_decompress_body<kInt8DeltaRle, MODE, false, 1, true>(a);
}
template <int MODE>
__global__ __launch_bounds__(kThreads) void k_fetch_decompress_flat_f32_staged(CodecArgs a) { fetch_decompress_body<kInt8DeltaRle, MODE, true, 1, true>(a); }

// ===================================================================
// RLE encode of the delta stream (cache_engine.cpp:198-239)
// ===================================================================
// Wave LDS layout (bytes): [0,16) lead (the count byte "before the first pair" lands here), [16, 16+4096) pair
// buffer.  Both paths scatter, per RUN START at position p, the value byte of its own pair and the count byte of the
// PREVIOUS pair (p - previous start); the last pair is closed after the loop.  No read-back from LDS.
// (kEncPairOff / kEncFail, quantize8 and the SDWA / EXEC-predicated helpers of this path: encode_device.hpp, shared with
// tensor_codec.hip)

// Fast path: no stretch of equal deltas reaches 255 elements, so every change of the delta starts a run and no run
// has to be split (count < 255 rule).  Returns the number of runs, or kEncFail when a long stretch may exist (>= 14
// lanes of a chunk without any run start: a 255-stretch needs 30 such lanes in two chunks) -- the caller then runs
// the SPLIT form, which starts over.  The block must be finite (quantize8).
// SPLIT (round 4; its own instantiation, the plain one is untouched): stretches of any length.  A run also starts at every
// 255th element of a stretch (cache_engine.cpp:224).  A lane holds 8 consecutive elements, so at most one such element falls
// into the part of the lane in front of its first change of the delta -- the part that belongs to the stretch ENTERING the lane,
// whose start is the last change in front of the lane (a second max-scan, over change positions only): if the entering
// stretch has offset o0 at the lane's element 0, the split start is element (255 - o0 % 255) % 255 when that is one of the
// lane's.  Everything behind that -- run indices, "count = distance to the previous start" -- is the plain path's, with the
// split start as one more start.  (Blocks of long runs took the element-wise path before: 745 us per 131 072 blocks, four
// times a block of noise, for the data that compresses 100 : 1.  profiles/r04_long_runs.txt)
template <int MODE, bool SPLIT = false>
__device__ __forceinline__ uint32_t encode_rle_fast(const uint4 (&raw)[4], float scale, float rcp, uint8_t* wl, uint32_t lane)
{
    const uint32_t pair_m1 = lds_addr_of(wl + kEncPairOff) - 1u;
    uint32_t qtail = 0, dtail = 0x100u, mcarry = 0, icarry = 0;  // dtail 0x100: no delta precedes element 0 (it starts a run)
    bool prev_sparse = false;                                    // mcarry: position+1 of the last run start
    uint32_t ccarry = 0;                                         // SPLIT: position+1 of the last CHANGE of the delta
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const uint32_t p0 = 512u * j + 8u * lane;
        uint32_t q[8];
        quantize8<MODE>(raw[j], scale, rcp, q);
        const uint32_t prevq = wave_shr1(q[7], qtail);
        qtail = lane63(q[7]);
        uint32_t d[8];
        d[0] = sub_bytes(q[0], prevq);
#pragma unroll
        for (int k = 1; k < 8; ++k) d[k] = sub_bytes(q[k], q[k - 1]);
        const uint32_t prevd = wave_shr1(d[7], dtail);
        dtail = lane63(d[7]);
        bool st[8];                                              // element k starts a run (a v_cmp result: an SGPR pair)
        uint32_t mask = 0;                                       // bit 7-k = element k starts a run
        unsigned long long sm[8];                                // the same as lane masks
        st[0] = d[0] != prevd;
#pragma unroll
        for (int k = 1; k < 8; ++k) st[k] = d[k] != d[k - 1];
        if (SPLIT) {
            uint32_t cmask = 0;                                  // the changes alone
#pragma unroll
            for (int k = 0; k < 8; ++k) shift_in(cmask, __builtin_amdgcn_ballot_w64(st[k]));
            const uint32_t lmc = cmask ? p0 + 8u - static_cast<uint32_t>(__builtin_ctz(cmask)) : 0u;
            const uint32_t imc = wave_incl_max(lmc);
            const uint32_t mc = umax(wave_shr1(imc, 0u), ccarry);                // last change in front of the lane, position+1 (0: none)
            ccarry = umax(ccarry, lane63(imc));
            const uint32_t o0 = p0 + 1u - mc;                                    // offset of element 0 in the entering stretch (>= 1)
            const uint32_t r = o0 - 255u * ((o0 * 0x8081u) >> 23);                // o0 % 255 (exact below 4096)
            uint32_t ks = r ? 255u - r : 0u;                                      // the lane's element at a multiple of 255, if < 8
            const uint32_t kfirst = cmask ? static_cast<uint32_t>(__builtin_clz(cmask)) - 24u : 8u;     // the lane's first change
            if (mc == 0u || ks >= kfirst) ks = 8u;                               // (behind a change the offsets are < 8: never a split)
#pragma unroll
            for (int k = 0; k < 8; ++k) st[k] = st[k] || ks == static_cast<uint32_t>(k);
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) { sm[k] = __builtin_amdgcn_ballot_w64(st[k]); shift_in(mask, sm[k]); }
        const uint32_t cnt = static_cast<uint32_t>(__builtin_popcount(mask));
        // last start of the lane, as position+1 (0 = none): element 7 - ctz(mask)
        const uint32_t lm = mask ? p0 + 8u - static_cast<uint32_t>(__builtin_ctz(mask)) : 0u;
        // cheap pre-filter: a run longer than 248 elements needs >= 28 start-free lanes in this chunk and the previous
        // one together, i.e. >= 14 in one of them
        const bool sparse = !SPLIT && __popcll(__ballot(mask == 0u)) >= 14;            // wave-uniform
        const bool suspicious = sparse || prev_sparse;
        prev_sparse = sparse;
        const uint32_t ic = wave_incl_add(cnt);
        uint32_t idx = icarry + ic - cnt;
        icarry += lane63(ic);
        const uint32_t im = wave_incl_max(lm);
        const uint32_t m = umax(wave_shr1(im, 0u), mcarry);      // last start before this lane, position+1
        mcarry = umax(mcarry, lane63(im));
        // A count is "position of this run start - previous run start"; only a lane's FIRST start of the chunk can
        // close a long run, and that run is shorter than (end of the lane's 8 elements - previous start).  If that
        // bound passes 255 anywhere the run may need splitting (cache_engine.cpp:224), which the SPLIT form does.
        // (wave-uniform: the SPLIT form starts over, nothing of this attempt is used)
        if (!SPLIT && suspicious) {
            if (__ballot(mask != 0u && p0 + 8u - m > 255u) != 0ull) return kEncFail;
            // ... or the run that is open at the end of the chunk is too long already (then its last 31 lanes hold no start: sparse)
            if (512u * static_cast<uint32_t>(j + 1) + 1u - mcarry > 255u) return kEncFail;
        }
        // rel = (last start, position+1) - (p0 + 1): count of the run closed by a start at element k is k - rel
        uint32_t rel = m - p0 - 1u;
        uint32_t addr[8], cntv[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            addr[k] = lshl1_add(idx, pair_m1);                   // count byte of the previous pair; +1 = value byte of this one
            cntv[k] = static_cast<uint32_t>(k) - rel;
            rel = st[k] ? static_cast<uint32_t>(k) : rel;
            add_pred(idx, sm[k]);
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) store_pair_if(sm[k], addr[k], cntv[k], d[k]);
    }
    lds_store_b8(pair_m1 + 2u * icarry, kBlockElems + 1u - mcarry);         // close the last run
    return icarry;
}

// General path (long stretches: zeros, constants; blocks with inf / NaN): one element per lane per step, rolled,
// quantised with the exact scalar form straight from the source; stretch starts by max-scan, a run starts every 255
// elements of a stretch.
template <int MODE>
__device__ __noinline__ uint32_t encode_rle_general(const uint8_t* __restrict__ src, float scale, uint8_t* wl, uint32_t lane)
{
    const uint32_t pair_addr = lds_addr_of(wl + kEncPairOff);
    uint32_t qtail = 0, dtail = 0, scarry = 0, mcarry = 0, icarry = 0;
#pragma unroll 1
    for (uint32_t step = 0; step < kBlockElems / 64u; ++step) {
        const uint32_t p = 64u * step + lane;
        const uint32_t qv = quantize<MODE>(half_bits_to_float(gload<uint16_t>(src + 2ull * p)), scale);
        const uint32_t prevq = wave_shr1(qv, qtail);
        qtail = lane63(qv);
        const uint32_t d = (qv - prevq) & 0xFFu;
        const uint32_t prevd = wave_shr1(d, dtail);
        dtail = lane63(d);
        const bool neq = (p == 0u) || (d != prevd);
        const uint32_t is = wave_incl_max(neq ? p + 1u : 0u);
        const uint32_t ss = umax(is, scarry);                      // stretch start, position+1
        scarry = umax(scarry, lane63(is));
        const bool isrun = ((p + 1u - ss) % 255u) == 0u;           // count < 255 rule (cache_engine.cpp:224)
        const uint32_t ic = wave_incl_add(isrun ? 1u : 0u);
        const uint32_t idx = icarry + ic - (isrun ? 1u : 0u);
        icarry += lane63(ic);
        const uint32_t im = wave_incl_max(isrun ? p + 1u : 0u);
        const uint32_t prev = umax(wave_shr1(im, 0u), mcarry);
        mcarry = umax(mcarry, lane63(im));
        if (isrun) {
            lds_store_b8(pair_addr + 2u * idx - 1u, p + 1u - prev);
            lds_store_b8(pair_addr + 2u * idx, d);
        }
    }
    lds_store_b8(pair_addr + 2u * icarry - 1u, kBlockElems + 1u - mcarry);
    return icarry;
}

// the reference's own rule for blocks that hold inf / NaN: a NaN never wins the '>' compare (cache_engine.cpp:176-180).
// Rare, out of line, and fed from memory again (a register array passed by reference would move to scratch).
__device__ __noinline__ float absmax_with_nonfinite(const uint8_t* __restrict__ src, uint32_t lane)
{
    float mx = 0.0f;
#pragma unroll 1
    for (uint32_t p = lane; p < kBlockElems; p += 64u)
        mx = __builtin_fmaxf(mx, fabsf(half_bits_to_float(gload<uint16_t>(src + 2ull * p))));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float other = __shfl_xor(mx, o);
        mx = (other > mx) ? other : mx;
    }
    return mx;
}

// ===================================================================
// encode  (cache_engine.cpp:40-82,172-239)
// ===================================================================
template <int SCHEME, int MODE>
__global__ __launch_bounds__(kThreads) void k_compress(CodecArgs a)
{
    __shared__ __attribute__((aligned(16))) uint16_t lds[SCHEME == kInt8DeltaRle ? kWaves * kEncLdsHalves : SCHEME == kInt4G32 ? kWaves * (kInt4RecBytes / 2)
                                                         : SCHEME == kMxFp4 ? kWaves * (kMx4RecBytes / 2) : 8];
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = threadIdx.x >> 6;
    const uint64_t n = a.n;
    const uint64_t per_wave = a.per_wave ? a.per_wave : 1;       // blocks per wave: see k_fetch_decompress
    const uint64_t gw = static_cast<uint64_t>(blockIdx.x) * kWaves + wave;
    const uint64_t step = a.wave_step ? a.wave_step : 1;
    const uint64_t i0 = a.wave_step ? gw : gw * per_wave;
    const uint64_t i1 = a.wave_step ? n : ((i0 + per_wave < n) ? i0 + per_wave : n);
    for (uint64_t i = i0; i < i1; i += step) {
        uint64_t page = a.page_list ? a.page_list[i] : a.first + i * (a.page_step ? a.page_step : 1);
        const uint8_t* src = a.data_list ? reinterpret_cast<const uint8_t*>(a.data_list[i])
                                         : a.data + i * a.data_stride;
        PageEntry* entries = a.entries;
        float* scale_tab = a.scale_tab;
        uint32_t region_pages = a.region_pages, scale_run = a.scale_run;
        if (a.groups) {                                              // wave-uniform: this block's allocation
            const uint64_t gi = i / a.group_n, j = i - gi * a.group_n;
            const CompressGroup g = a.groups[gi];
            entries = g.entries; scale_tab = g.scale_tab; region_pages = g.region_pages; scale_run = g.scale_run;
            page = g.first + j * (a.page_step ? a.page_step : 1);
            src = g.data + j * a.data_stride;
        }
        uint8_t* rec = entries ? reinterpret_cast<uint8_t*>(gload<uint64_t>(&entries[page].pool_addr))
                               : a.recs + page * a.rec_stride;
        // MXFP4 in the pool is tile-planar (kernels.hpp): the codes go mx4_code_delta behind the nibbles -- the entry's scale word
        uint8_t* rec_codes = rec + 1024u;
        if (SCHEME == kMxFp4 && entries) rec_codes = rec + gload<uint32_t>(reinterpret_cast<const uint32_t*>(&entries[page].scale));
        uint32_t out_len;
        float scale = 1.0f;

        uint4 raw[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
            raw[j] = enc_ld16(src + 2ull * (512u * j + 8u * lane));

        if (SCHEME == kFp16) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                enc_st16(rec + 2ull * (512u * j + 8u * lane), raw[j]);
            out_len = 2u * kBlockElems;
        } else if (SCHEME == kInt4G32) {
            // per group of 32 elements (4 lanes x 8): s = fp16(max|x|/7), q = clamp(round(x/s), -7, 7).  The group's max|x| and
            // its "all finite" test come from the fp16 bit patterns (v_pk_max_u16, two elements per instruction), the
            // quantisation runs in packed fp32 (the same multiply / fma / fma / add / truncate per element as the scalar form,
            // so the same bits).  Round 4, instruction count (the kernel ran 18 % over what the chip moves at its read : write
            // mix, profiles/r04_store_shape.txt): max|x| / 7 and 1 / s without the IEEE divide (codec_device.hpp: two and three
            // operations, bit-identical for fp16-valued operands); no clamp when every scale of the chunk is a normal fp16
            // (then no quotient reaches 7.5); the nibbles summed as signed i << 4k and un-biased once per dword.
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t words[4] = {raw[j].x, raw[j].y, raw[j].z, raw[j].w};
                u16x2 m2 = {0, 0};
#pragma unroll
                for (int t = 0; t < 4; ++t) m2 = __builtin_elementwise_max(m2, __builtin_bit_cast(u16x2, words[t] & 0x7FFF7FFFu));
                uint32_t mbits = m2.x > m2.y ? m2.x : m2.y;                 // largest |bits| of the lane's 8 elements
                mbits = umax(mbits, dpp<0xB1>(0u, mbits));                  // quad_perm [1,0,3,2]: lane ^ 1
                mbits = umax(mbits, dpp<0x4E>(0u, mbits));                  // quad_perm [2,3,0,1]: lane ^ 2 -> the group of 32
                const bool finite = __ballot(mbits >= 0x7C00u) == 0ull;     // wave-uniform: no inf / NaN in any group of this chunk
                uint32_t nib = 0;
                _Float16 s16;
                if (finite) {
                    float sdiv = div7_of_f16_value(half_bits_to_float(mbits));
                    asm volatile("" : "+v"(sdiv));                      // keep the fp32 rounding of the divide
                    s16 = static_cast<_Float16>(sdiv);
                    const float sc = static_cast<float>(s16);
                    // |x| <= 7.5*sc in a finite group: the reciprocal divide is exact (test_fast_division_is_exact); a scale that
                    // rounds to zero (subnormal groups) gives rcp = 0 and every quotient 0, as the definition says
                    const float rcp = sc != 0.0f ? rcp_of_f16_value(sc) : 0.0f;
                    const f32x2 ss = {sc, sc}, rr = {rcp, rcp};
                    // a subnormal scale may be off by more than half a step: only then can a quotient leave [-7.5, 7.5]
                    const uint32_t sbits = __builtin_bit_cast(uint16_t, s16);
                    const bool clamp = __ballot(sbits - 1u < 0x3FFu) != 0ull;                   // (wave-uniform)
                    int iq[8];
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        f32x2 x;
                        x.x = half_bits_to_float(words[t] & 0xFFFFu);
                        x.y = half_bits_to_float(words[t] >> 16);
                        const f32x2 q0 = x * rr;
                        const f32x2 e = __builtin_elementwise_fma(-q0, ss, x);
                        const f32x2 y = __builtin_elementwise_fma(e, rr, q0);
                        f32x2 h;
                        h.x = __builtin_copysignf(0.5f, y.x);
                        h.y = __builtin_copysignf(0.5f, y.y);
                        const f32x2 r = y + h;                          // round half away from zero = truncate(y + copysign(0.5, y))
                        iq[2 * t] = static_cast<int>(r.x);
                        iq[2 * t + 1] = static_cast<int>(r.y);
                    }
                    if (clamp) {
#pragma unroll
                        for (int k = 0; k < 8; ++k) iq[k] = min(max(iq[k], -7), 7);
                    }
                    // nibble k = iq[k] & 0xF:  sum of iq[k] << 4k  =  sum of (iq[k] + 8) << 4k  -  0x88888888, and (i + 8) ^ 8 = i & 0xF
                    uint32_t acc = 0x88888888u;
#pragma unroll
                    for (int k = 0; k < 8; ++k) acc += static_cast<uint32_t>(iq[k]) << (4 * k);
                    nib = acc ^ 0x88888888u;
                } else {
                    float xv[8];
                    float mx = 0.0f, nanacc = 0.0f;
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        xv[k] = half_bits_to_float((words[k >> 1] >> ((k & 1) * 16)) & 0xFFFFu);
                        absmax_finite(xv[k], mx, nanacc);
                    }
                    float o = __shfl_xor(mx, 1); mx = (o > mx) ? o : mx;
                    o = __shfl_xor(mx, 2);       mx = (o > mx) ? o : mx;
                    float sdiv = mx / 7.0f;
                    asm volatile("" : "+v"(sdiv));
                    s16 = static_cast<_Float16>(sdiv);
                    const float sc = static_cast<float>(s16);
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        float r = 0.0f;
                        if (sc != 0.0f && sc == sc) {
                            r = roundf(xv[k] / sc);
                            if (!(r == r)) r = 0.0f;
                            r = fminf(fmaxf(r, -7.0f), 7.0f);
                        }
                        nib |= (static_cast<uint32_t>(static_cast<int>(r)) & 0xFu) << (4 * k);
                    }
                }
                // The record is assembled in LDS and leaves as whole 16-byte pieces per lane: written straight from the
                // registers it took eight store instructions of 4 / 2 bytes per lane (0.63-0.65 of HBM peak).
                const uint32_t p0 = 512u * j + 8u * lane;
                uint8_t* wl = reinterpret_cast<uint8_t*>(lds) + wave * kInt4RecBytes;
                *reinterpret_cast<uint32_t*>(wl + 128u + (p0 >> 1)) = nib;
                if ((lane & 3u) == 0u)
                    *reinterpret_cast<uint16_t*>(wl + 2u * (p0 >> 5)) = __builtin_bit_cast(uint16_t, s16);
            }
            {
                uint8_t* wl = reinterpret_cast<uint8_t*>(lds) + wave * kInt4RecBytes;
                wave_lds_fence();
                enc_st16(rec + 128u + 16u * lane, *reinterpret_cast<const uint4*>(wl + 128u + 16u * lane));
                if (lane < 8u) enc_st16(rec + 16u * lane, *reinterpret_cast<const uint4*>(wl + 16u * lane));
                wave_lds_fence();
            }
            out_len = kInt4RecBytes;
        } else if (SCHEME == kMxFp4) {
            // OCP MX v1.0 conversion (oracle: compress_mxfp4) over the block's two halves interleaved element by element: byte i of the
            // record = element i (low nibble) and element 1024 + i (high), one E8M0 code per 16 bytes.  This lane's raw[j] and
            // raw[j + 2] (j = 0, 1) are exactly such partners -- elements [512 j + 8 l, + 8) and the same 1024 further on -- so the
            // interleave never leaves the lane; a block of 32 = this lane's and lane ^ 1's 8 + 8 elements of both halves.
            //   code = floor(log2 max|x|) - 2 + 127 = the exponent field of max|x| as a float, minus 2
            //   q    = E2M1 of x / 2^(code-127), nearest even, saturating: one v_cvt_scalef32_pk_fp4_f16 per element pair (it divides
            //          by the power of two its scale operand's exponent names: profiles/probes/mxprobe.hip)
            typedef _Float16 f16x2v __attribute__((ext_vector_type(2)));
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const uint32_t w0[4] = {raw[j].x, raw[j].y, raw[j].z, raw[j].w};                    // first half: elements 2t, 2t+1 per word
                const uint32_t w1[4] = {raw[j + 2].x, raw[j + 2].y, raw[j + 2].z, raw[j + 2].w};    // their partners in the second half
                u16x2 m2 = {0, 0};
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    m2 = __builtin_elementwise_max(m2, __builtin_bit_cast(u16x2, w0[t] & 0x7FFF7FFFu));
                    m2 = __builtin_elementwise_max(m2, __builtin_bit_cast(u16x2, w1[t] & 0x7FFF7FFFu));
                }
                uint32_t mbits = m2.x > m2.y ? m2.x : m2.y;                 // largest |bits| of the lane's 16 elements
                mbits = umax(mbits, dpp<0xB1>(0u, mbits));                  // lane ^ 1 -> the block of 32
                const bool finite = __ballot(mbits >= 0x7C00u) == 0ull;     // wave-uniform: no inf / NaN in any block of this chunk
                uint32_t nib[2] = {0u, 0u}, code = 0;
                if (finite) {
                    code = mbits ? (__float_as_uint(half_bits_to_float(mbits)) >> 23) - 2u : 0u;
                    const float sc = mbits ? __uint_as_float(code << 23) : 1.0f;      // (code >= 101 for any non-zero fp16: a normal float)
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        // (element 2t of both halves), (element 2t+1 of both halves) -> bytes 2t, 2t+1
                        const f16x2v ea = __builtin_bit_cast(f16x2v, __builtin_amdgcn_perm(w1[t], w0[t], 0x05040100u));
                        const f16x2v eb = __builtin_bit_cast(f16x2v, __builtin_amdgcn_perm(w1[t], w0[t], 0x07060302u));
                        uint32_t& o = nib[t >> 1];
                        if (t & 1) { o = __builtin_amdgcn_cvt_scalef32_pk_fp4_f16(o, ea, sc, 2); o = __builtin_amdgcn_cvt_scalef32_pk_fp4_f16(o, eb, sc, 3); }
                        else       { o = __builtin_amdgcn_cvt_scalef32_pk_fp4_f16(o, ea, sc, 0); o = __builtin_amdgcn_cvt_scalef32_pk_fp4_f16(o, eb, sc, 1); }
                    }
                } else {
                    // a chunk with inf / NaN somewhere: NaN elements are skipped in the maximum and store +0, inf counts as 65504
                    float x0[8], x1[8];
                    float mx = 0.0f;
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        float a0 = half_bits_to_float((w0[k >> 1] >> ((k & 1) * 16)) & 0xFFFFu), a1 = half_bits_to_float((w1[k >> 1] >> ((k & 1) * 16)) & 0xFFFFu);
                        a0 = (a0 == a0) ? fminf(fmaxf(a0, -65504.0f), 65504.0f) : 0.0f;
                        a1 = (a1 == a1) ? fminf(fmaxf(a1, -65504.0f), 65504.0f) : 0.0f;
                        x0[k] = a0; x1[k] = a1;
                        mx = fmaxf(mx, fmaxf(fabsf(a0), fabsf(a1)));
                    }
                    const float o = __shfl_xor(mx, 1);
                    mx = (o > mx) ? o : mx;
                    code = mx > 0.0f ? (__float_as_uint(mx) >> 23) - 2u : 0u;
                    const float sc = mx > 0.0f ? __uint_as_float(code << 23) : 1.0f;
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        uint32_t& ob = nib[k >> 2];
                        switch (k & 3) {
                        case 0: ob = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(ob, x0[k], x1[k], sc, 0); break;
                        case 1: ob = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(ob, x0[k], x1[k], sc, 1); break;
                        case 2: ob = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(ob, x0[k], x1[k], sc, 2); break;
                        default: ob = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(ob, x0[k], x1[k], sc, 3); break;
                        }
                        // a NaN element keeps no sign: +0
                        if (((w0[k >> 1] >> ((k & 1) * 16)) & 0x7FFFu) > 0x7C00u) ob &= ~(0x0Fu << (8 * (k & 3)));
                        if (((w1[k >> 1] >> ((k & 1) * 16)) & 0x7FFFu) > 0x7C00u) ob &= ~(0xF0u << (8 * (k & 3)));
                    }
                }
                // the record is assembled in LDS and leaves as whole 16-byte pieces per lane (as the INT4 record does)
                const uint32_t p0 = 512u * j + 8u * lane;
                uint8_t* wl = reinterpret_cast<uint8_t*>(lds) + wave * kMx4RecBytes;
                *reinterpret_cast<uint2*>(wl + p0) = make_uint2(nib[0], nib[1]);
                if ((lane & 1u) == 0u) wl[1024u + (p0 >> 4)] = static_cast<uint8_t>(code);
            }
            {
                uint8_t* wl = reinterpret_cast<uint8_t*>(lds) + wave * kMx4RecBytes;
                wave_lds_fence();
                enc_st16(rec + 16u * lane, *reinterpret_cast<const uint4*>(wl + 16u * lane));
                if (lane < 4u) enc_st16(rec_codes + 16u * lane, *reinterpret_cast<const uint4*>(wl + 1024u + 16u * lane));
                wave_lds_fence();
            }
            out_len = kMx4RecBytes;
        } else if (SCHEME == kFp8E4m3) {
            float x[4][8];
            float mx = 0.0f, nanacc = 0.0f;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t words[4] = {raw[j].x, raw[j].y, raw[j].z, raw[j].w};
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    x[j][k] = half_bits_to_float((words[k >> 1] >> ((k & 1) * 16)) & 0xFFFFu);
                    absmax_finite(x[j][k], mx, nanacc);
                }
            }
            // (mx is never a NaN and never negative: its bit pattern orders like its value -- the DPP max scan of the encoders
            // instead of six trips through the LDS crossbar)
            mx = __uint_as_float(lane63(wave_incl_max(__float_as_uint(mx))));
            const bool finite = __ballot(!(nanacc == 0.0f)) == 0ull;       // wave-uniform
            // a finite block's max|x| is an fp16 value: its scale and the reciprocal without the IEEE divide, bit-identical
            // (codec_device.hpp; exhaustive: test_fast_division_is_exact)
            scale = (mx > 0.0f) ? (finite ? div448_of_f16_value(mx) : mx / 448.0f) : 1.0f;
            const float rcp = finite ? rcp_of_scale(scale) : 1.0f / scale;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float v[8];
                if (finite) {
#pragma unroll
                    for (int k = 0; k < 8; ++k)
                        v[k] = fminf(fmaxf(__builtin_copysignf(div_by_scale(x[j][k], scale, rcp), x[j][k]), -448.0f), 448.0f);
                } else {
#pragma unroll
                    for (int k = 0; k < 8; ++k) v[k] = fminf(fmaxf(x[j][k] / scale, -448.0f), 448.0f);
                }
                int lo = __builtin_amdgcn_cvt_pk_fp8_f32(v[0], v[1], 0, false);
                lo = __builtin_amdgcn_cvt_pk_fp8_f32(v[2], v[3], lo, true);
                int hi = __builtin_amdgcn_cvt_pk_fp8_f32(v[4], v[5], 0, false);
                hi = __builtin_amdgcn_cvt_pk_fp8_f32(v[6], v[7], hi, true);
                enc_st8(rec + 512u * j + 8u * lane, make_uint2(static_cast<uint32_t>(lo), static_cast<uint32_t>(hi)));
            }
            out_len = kBlockElems;
        } else {
            const uint32_t mbits = absmax_bits(raw);                       // wave-uniform
            const bool finite = mbits < 0x7C00u;
            const float mx = finite ? half_bits_to_float(mbits) : absmax_with_nonfinite(src, lane);
            scale = (mx > 0.0f) ? (finite ? div127_of_f16_value(mx) : mx / 127.0f) : 1.0f;     // cache_engine.cpp:172-183 (finite: an fp16 value
            const float rcp = finite ? rcp_of_scale(scale) : 1.0f / scale;                      //  over 127 without the IEEE divide, same bits)

            if (SCHEME == kInt8) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    uint32_t q[8];
                    if (finite) {
                        quantize8<MODE>(raw[j], scale, rcp, q);
                    } else {
                        const uint32_t words[4] = {raw[j].x, raw[j].y, raw[j].z, raw[j].w};
#pragma unroll
                        for (int k = 0; k < 8; ++k)
                            q[k] = quantize<MODE>(half_bits_to_float((words[k >> 1] >> ((k & 1) * 16)) & 0xFFFFu), scale);
                    }
                    uint2 o;
                    o.x = (q[0] & 0xFFu) | ((q[1] & 0xFFu) << 8) | ((q[2] & 0xFFu) << 16) | (q[3] << 24);
                    o.y = (q[4] & 0xFFu) | ((q[5] & 0xFFu) << 8) | ((q[6] & 0xFFu) << 16) | (q[7] << 24);
                    enc_st8(rec + 512u * j + 8u * lane, o);
                }
                out_len = kBlockElems;
            } else if (mx == 0.0f) {
                // every element is +-0 (or a NaN, which quantises to 0 as well): all deltas are 0, so the record is
                // eight runs of 255 zeros and one of 8 (cache_engine.cpp:208-239) -- no need to encode anything
                if (lane < 16u)
                    reinterpret_cast<uint16_t*>(rec)[lane] = lane < 8u ? 0xFF00u : (lane == 8u ? 0x0800u : 0u);
                out_len = 18u;
            } else {
                uint8_t* wl = reinterpret_cast<uint8_t*>(lds) + wave * kEncWaveBytes;
                uint32_t nruns = finite ? encode_rle_fast<MODE>(raw, scale, rcp, wl, lane) : kEncFail;
                wave_lds_fence();
                if (nruns == kEncFail) {
                    if (finite) {                                        // long stretches: the fast path's SPLIT form, from the source again
                        uint4 again[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j) again[j] = enc_ld16(src + 2ull * (512u * j + 8u * lane));
                        nruns = encode_rle_fast<MODE, true>(again, scale, rcp, wl, lane);
                    } else {
                        nruns = encode_rle_general<MODE>(src, scale, wl, lane);
                    }
                    wave_lds_fence();
                }
                uint8_t* pairbuf = wl + kEncPairOff;
                // zero the tail of the last 16-byte chunk so the stored slot is deterministic
                {
                    const uint32_t idx = nruns + lane;
                    if (lane < 8u && idx < ((nruns + 7u) & ~7u)) reinterpret_cast<uint16_t*>(pairbuf)[idx] = 0;
                }
                wave_lds_fence();
                out_len = 2u * nruns;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const uint32_t b = 1024u * j + 16u * lane;
                    if (b < out_len)
                        enc_st16(rec + b, *reinterpret_cast<const uint4*>(pairbuf + b));
                }
                wave_lds_fence();
            }
        }
        if (lane == 0u) {
            if (entries) {
                gstore<uint32_t>(&entries[page].rec_bytes, out_len);
                if (SCHEME != kMxFp4) gstore<float>(&entries[page].scale, scale);       // (MXFP4: the word holds the code delta)
                if (SCHEME == kInt8DeltaRle && a.len_samples && (page & 1023u) == 0u) gstore<uint32_t>(&a.len_samples[(page >> 10) & 15u], out_len);
                if (scale_tab) {
                    const uint32_t j = static_cast<uint32_t>(page % region_pages) & 15u;
                    gstore<float>(&scale_tab[page - j + attend_tile_slot(j)], scale);
                    if (scale_run) gstore<float>(scale_tab + scale_run_index(page, scale_run), scale);
                }
            } else {
                a.rec_bytes[page] = out_len;
                if (a.scales) a.scales[page] = scale;
            }
        }
    }
}

// ===================================================================
// prefetch lookup  (prefetch_core.v:150-241 ; speckv_allocator.cpp:105-113)
// ===================================================================
// 32 lanes per request: lane c -> kind = c>>4, position cur_pos + (c&15) + 1.
// Pages of a position = pages covering its [head 0 .. head H-1] row in the
// shim layout; a lane emits only pages its predecessor lane did not cover.
struct Cand { uint32_t lo, hi; };   // half-open range of NEW pages of this lane (before residency filter)

__device__ __forceinline__ Cand candidate(const Layout& lay, uint32_t req, uint32_t layer,
                                          uint32_t pos, uint32_t depth, uint32_t c)
{
    Cand r{0u, 0u};
    const uint32_t kind = c >> 4, i = (c & 15u) + 1u;
    const uint64_t p = static_cast<uint64_t>(pos) + i;
    if (i > depth || p >= lay.num_tokens) return r;
    const uint64_t entry = static_cast<uint64_t>(lay.head_dim) * lay.bytes_per_element;
    const uint64_t row = entry * lay.num_heads;
    if (row == 0) return r;
    // vllm_speckv_backend.py:95-100 with head = 0
    const uint64_t off = ((((static_cast<uint64_t>(req) * lay.num_layers + layer) * 2 + kind)
                           * lay.num_tokens + p) * lay.num_heads) * entry;
    uint64_t pg0 = off / kPageSize;
    const uint64_t pg1 = (off + row - 1) / kPageSize;
    if (i > 1) {                       // predecessor position p-1 covered up to:
        const uint64_t prev_pg1 = (off - 1) / kPageSize;   // (off - row + row - 1)
        if (pg0 <= prev_pg1) pg0 = prev_pg1 + 1;
    }
    uint64_t hi = pg1 + 1;
    if (hi > lay.alloc_pages) hi = lay.alloc_pages;
    if (pg0 >= hi) return r;
    r.lo = static_cast<uint32_t>(pg0);
    r.hi = static_cast<uint32_t>(hi);
    return r;
}

template <bool WRITE>
__global__ __launch_bounds__(256) void k_prefetch_lookup(Layout lay, uint32_t n,
        const uint32_t* __restrict__ req, const uint32_t* __restrict__ layer,
        const uint32_t* __restrict__ pos, const uint32_t* __restrict__ depth,
        const uint32_t* __restrict__ flags, uint32_t* __restrict__ wave_tot,
        const uint32_t* __restrict__ wave_base, uint32_t* __restrict__ out, uint32_t cap)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t gw = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;   // global wave = 2 requests
    const uint32_t r = 2u * gw + (lane >> 5);
    uint32_t cnt = 0;
    Cand cd{0u, 0u};
    const bool live_wave = 2u * gw < n;
    if (r < n) {
        uint32_t dk = depth[r];
        if (dk > 16u) dk = 16u;
        cd = candidate(lay, req[r], layer[r], pos[r], dk, lane & 31u);
        for (uint32_t pg = cd.lo; pg < cd.hi; ++pg)
            cnt += (flags && (flags[pg] & 3u)) ? 0u : 1u;
    }
    const uint32_t incl = wave_incl_add(cnt);
    if (!WRITE) {
        if (lane == 63u && live_wave) wave_tot[gw] = incl;
    } else {
        uint32_t w = (live_wave ? wave_base[gw] : 0u) + incl - cnt;
        for (uint32_t pg = cd.lo; pg < cd.hi; ++pg)
            if (!(flags && (flags[pg] & 3u))) {
                if (w < cap) out[w] = pg;
                ++w;
            }
    }
}

// single-workgroup exclusive scan of the per-wave totals (n_w is small:
// requests/2); total is clamped to cap.
__global__ __launch_bounds__(1024) void k_scan_totals(const uint32_t* __restrict__ tot,
        uint32_t* __restrict__ base, uint32_t n_w, uint32_t* __restrict__ count, uint32_t cap)
{
    __shared__ uint32_t wsum[16];
    __shared__ uint32_t running;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) running = 0;
    __syncthreads();
    for (uint32_t i0 = 0; i0 < n_w; i0 += 1024u) {
        const uint32_t i = i0 + threadIdx.x;
        const uint32_t v = (i < n_w) ? tot[i] : 0u;
        const uint32_t incl = wave_incl_add(v);
        if (lane == 63u) wsum[wave] = incl;
        __syncthreads();
        uint32_t wbase = 0;
        for (uint32_t w = 0; w < wave; ++w) wbase += wsum[w];
        const uint32_t run = running;
        if (i < n_w) base[i] = run + wbase + incl - v;
        __syncthreads();
        if (threadIdx.x == 1023u) running = run + wbase + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) *count = running < cap ? running : cap;
}

// ===================================================================
// device-side prefetch flush  (prefetch_core.v:150-241: the whole loop without the host)
// ===================================================================
// Candidate words live at a fixed stride: request r, lane c (kind, look-ahead step), word t -> index
// (r*32 + c)*W + t, kNoSlot when unused.  Request order = index order, so "first occurrence" of a page is the
// smallest index naming it: every valid candidate does atomicMax(stamp[page], key(index)) with
// key = epoch<<24 | (0xFFFFFF - index); the candidate whose key survives is the one kept.  Stamps of older
// flushes carry an older epoch and lose against any key of this one (the host clears them when the 8-bit epoch wraps).
__device__ __forceinline__ uint32_t flush_key(uint32_t epoch, uint32_t index) { return (epoch << 24) | (0xFFFFFFu - index); }

__device__ __forceinline__ void flush_candidates_of(const FlushArgs& a, uint32_t gt)      // gt: one thread per (request, lane)
{
    const uint32_t r = gt >> 5, c = gt & 31u;
    if (r >= a.n) return;
    const uint32_t row = a.row[r];
    Cand cd{0u, 0u};
    DevAlloc t{};
    if (row != kNoSlot) {
        t = a.tab[row];
        if (t.entries) {
            uint32_t dk = a.depth[r];
            if (dk > 16u) dk = 16u;
            cd = candidate(t.layout, a.req[r], a.layer[r], a.pos[r], dk, c);
        }
    }
    const uint32_t base = gt * a.W;
    for (uint32_t w = 0; w < a.W; ++w) {
        uint32_t pg = kNoSlot;
        if (cd.lo + w < cd.hi && !(t.d_flags[cd.lo + w] & 3u)) {
            pg = cd.lo + w;
            atomicMax(&t.stamp[pg], flush_key(a.epoch, base + w));
        }
        a.cand[base + w] = pg;
    }
}
__global__ __launch_bounds__(256) void k_flush_candidates(FlushArgs a) { flush_candidates_of(a, blockIdx.x * blockDim.x + threadIdx.x); }

// candidate word i names page `pg` of allocation row `row` and is the first occurrence of that page in the flush
__device__ __forceinline__ bool flush_keeps(const FlushArgs& a, uint32_t i, uint32_t total, uint32_t& pg, uint32_t& row)
{
    pg = kNoSlot; row = kNoSlot;
    if (i >= total) return false;
    pg = a.cand[i];
    if (pg == kNoSlot) return false;
    row = a.row[i / (32u * a.W)];
    return a.tab[row].stamp[pg] == flush_key(a.epoch, i);
}
// Entry `rank` of the flush lands in ring slot base + rank.  Everything about it that is pointer chasing --
// its record descriptor, the slot's previous owner and that owner's residency words, the new owner, the
// page's slot words on both sides -- is done here, one THREAD per page, so that the fetch launch is the plain
// list form (descriptor + destination per block) at the bulk kernel's occupancy.  Done by the fetch kernel
// itself, one WAVE per page with a chain of ~8 dependent loads each, the 122 880-page flush of a 256-sequence
// decode step spent 210 us in the fetch; the chain now runs 64 pages per wave.
// (The words are final before the data has landed: the host waits for the flight's `done` event before it
// trusts a page whose slot lies in the flight's run, Engine::wait_landed.)
__device__ __forceinline__ void flush_place(const FlushArgs& a, const FlushResult& res, uint32_t rank, uint32_t row, uint32_t pg)
{
    const uint32_t slot = res.base + rank;
    const DevAlloc t = a.tab[row];
    a.final_entry[rank] = t.entries[pg];
    a.final_dst[rank] = reinterpret_cast<uint64_t>(a.ring_base + static_cast<uint64_t>(slot) * kPageSize);
    const uint64_t prev = a.ring_owner[slot];
    const uint64_t me = (static_cast<uint64_t>(row) << 32) | pg;
    if (prev != kNoOwner && prev != me) {
        const DevAlloc tp = a.tab[prev >> 32];
        const uint32_t pp = static_cast<uint32_t>(prev);
        // the row may have been recycled for a smaller allocation since the slot was filled
        if (tp.entries && pp < tp.layout.alloc_pages && tp.d_slot[pp] == slot)    // still pointing here: the page leaves L2
            atomicAnd(&tp.d_flags[pp], ~2u);
    }
    a.ring_owner[slot] = me;
    t.d_slot[pg] = slot;
    if (a.final_host) a.final_host[rank] = &t.h_slot[pg];     // stored by the fetch launch (CodecArgs::host_words)
    else t.h_slot[pg] = res.seq + rank;                       // the page's only host-visible word (Engine::l2_live)
    atomicOr(&t.d_flags[pg], 2u);
}
// the ring run of a flush that keeps `total` candidates: [base, base+m), a run never wraps (as Engine::take_l2_run on the
// host: same rule, so host and device agree on the hand).  *a.hand is the ring's sequence number: slot = seq % n_l2; the
// slots a run skips at the end of a lap count.
__device__ __forceinline__ FlushResult flush_take(const FlushArgs& a, uint32_t total)
{
    const uint32_t m = total < a.max_take ? total : a.max_take;
    const RingRun run = ring_take(*a.hand, m, a.n_l2);
    if (m) *a.hand = run.next;
    const FlushResult r{m, run.slot, total, run.seq};
    *a.result_dev = r;
    *a.result_host = r;
    return r;
}

// keep[i] = candidate i is the first occurrence of its page; WRITE = false: totals per workgroup (256 candidates: the
// single-workgroup scan of k_flush_assign is 4x shorter than over per-wave totals), true: ordered scatter (rank = the
// workgroup's base from k_flush_assign + the kept candidates before this one in it)
template <bool WRITE>
__global__ __launch_bounds__(256) void k_flush_mark(FlushArgs a, const FlushResult* __restrict__ res)
{
    __shared__ uint32_t wcount[4];
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t total = a.n * 32u * a.W;
    uint32_t pg, row;
    const bool keep = flush_keeps(a, i, total, pg, row);
    const unsigned long long mask = __ballot(keep);
    if (lane == 0u) wcount[wave] = static_cast<uint32_t>(__popcll(mask));
    __syncthreads();
    if (!WRITE) {
        if (threadIdx.x == 0u) a.wave_tot[blockIdx.x] = wcount[0] + wcount[1] + wcount[2] + wcount[3];
    } else if (keep) {
        uint32_t before = 0;
        for (uint32_t w = 0; w < wave; ++w) before += wcount[w];
        const uint32_t n_b = (total + 255u) >> 8;
        const uint32_t rank = a.wave_tot[n_b + blockIdx.x] + before + static_cast<uint32_t>(__popcll(mask & ((1ull << lane) - 1ull)));
        if (rank < res->m) flush_place(a, *res, rank, row, pg);
    }
}

// one workgroup: exclusive scan of the wave totals, then the ring run [base, base+m) (a run never wraps, as
// Engine::take_l2_run on the host: same rule, so host and device agree on the hand)
__global__ __launch_bounds__(1024) void k_flush_assign(FlushArgs a)
{
    __shared__ uint32_t wsum[16];
    __shared__ uint32_t running;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t n_w = (a.n * 32u * a.W + 255u) >> 8;       // totals per workgroup of k_flush_mark
    if (threadIdx.x == 0) running = 0;
    __syncthreads();
    for (uint32_t i0 = 0; i0 < n_w; i0 += 1024u) {
        const uint32_t i = i0 + threadIdx.x;
        const uint32_t v = (i < n_w) ? a.wave_tot[i] : 0u;
        const uint32_t incl = wave_incl_add(v);
        if (lane == 63u) wsum[wave] = incl;
        __syncthreads();
        uint32_t wbase = 0;
        for (uint32_t w = 0; w < wave; ++w) wbase += wsum[w];
        const uint32_t run = running;
        if (i < n_w) a.wave_tot[n_w + i] = run + wbase + incl - v;
        __syncthreads();
        if (threadIdx.x == 1023u) running = run + wbase + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) (void)flush_take(a, running);
}

// The whole pipeline in ONE workgroup for small flushes (the reference's own call pattern: speckv_prefetch per request and
// layer, flushed every num_layers requests -- 32 requests x 32 lanes x 2 words for an 8B-shaped model):
// four launches and their three hand-overs become phases between barriers; the kept flags stay in a register (one bit per
// pass of 1024 words), the counts per (pass, wave) in LDS.  Same arithmetic, same order of entries.
constexpr uint32_t kFlushSmallPasses = 16, kFlushSmallWords = kFlushSmallPasses * 1024u;
// measured on the MI355X (time until the pages have landed, W = 2): 4 .. 64 requests 38-42 us against 42-43 us for the four
// launches, 80 requests 54 against 45 (one workgroup then chases the pointers of ~500 pages alone); the host's submit time is
// 11-14 us against 19-23.  SPECKV_FLUSH_SMALL_WORDS moves the limit, SPECKV_FLUSH_NO_SMALL=1 removes the path.
// (Letting this kernel pull the request columns from the host's pinned slot itself, instead of the upload launch in front of
// it, saved the host another 2.5 us per flush and cost 4-5 us until landed, same box: 41.5-47 against 37.2-41.9.  Not kept.)
constexpr uint64_t kFlushSmallDefault = 4096;
__global__ __launch_bounds__(1024) void k_flush_small(FlushArgs a)
{
    __shared__ uint32_t wcnt[kFlushSmallPasses * 16u];      // kept candidates of (pass, wave) -> kept candidates before it
    __shared__ uint32_t s_part[4];
    __shared__ FlushResult s_res;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t total = a.n * 32u * a.W, passes = (total + 1023u) >> 10;
    for (uint32_t gt = tid; gt < a.n * 32u; gt += 1024u) flush_candidates_of(a, gt);
    __threadfence();                                        // the stamps' atomicMax and the candidate words are out
    __syncthreads();
    uint32_t keepbits = 0;
    for (uint32_t j = 0; j < passes; ++j) {
        uint32_t pg, row;
        const bool keep = flush_keeps(a, j * 1024u + tid, total, pg, row);
        const unsigned long long mask = __ballot(keep);
        if (lane == 0u) wcnt[j * 16u + wave] = static_cast<uint32_t>(__popcll(mask));
        keepbits |= (keep ? 1u : 0u) << j;
    }
    for (uint32_t e = passes * 16u + tid; e < kFlushSmallPasses * 16u; e += 1024u) wcnt[e] = 0u;
    __syncthreads();
    {   // exclusive scan of the 256 counts (entry order = candidate order) by the first four waves
        uint32_t v = 0, incl = 0;
        if (tid < 256u) { v = wcnt[tid]; incl = wave_incl_add(v); if (lane == 63u) s_part[wave] = incl; }
        __syncthreads();
        if (tid < 256u) {
            uint32_t before = 0;
            for (uint32_t w = 0; w < wave; ++w) before += s_part[w];
            wcnt[tid] = before + incl - v;
        }
        if (tid == 0u) s_res = flush_take(a, s_part[0] + s_part[1] + s_part[2] + s_part[3]);
        __syncthreads();
    }
    const FlushResult res = s_res;
    for (uint32_t j = 0; j < passes; ++j) {
        const bool keep = (keepbits >> j) & 1u;
        const unsigned long long mask = __ballot(keep);
        if (keep) {
            const uint32_t i = j * 1024u + tid;
            const uint32_t rank = wcnt[j * 16u + wave] + static_cast<uint32_t>(__popcll(mask & ((1ull << lane) - 1ull)));
            if (rank < res.m) flush_place(a, res, rank, a.row[i / (32u * a.W)], a.cand[i]);
        }
    }
}

// ===================================================================
// verify  (speculative_prefetcher.cpp:84-96): hit[r] = actual[r] in predicted[r][0..k)
// ===================================================================
// One request per lane; the 64-bit __ballot of the per-lane result is the
// wave's verify mask (its popcount feeds the hit counter).  A wave owns 64
// consecutive bytes of hit[] and a workgroup 256, so no two workgroups ever
// write into the same 128-byte line: the earlier layout (16 lanes per request,
// 4 result bytes per wave) let eight workgroups on eight XCDs share one line and
// showed rare wrong bytes on MI355X (tests/test_gpu_engine.py::test_verify_batch_kernel).
__global__ __launch_bounds__(256) void k_verify(uint32_t n, uint32_t k,
        const int32_t* __restrict__ actual, const int32_t* __restrict__ predicted,
        uint8_t* __restrict__ hit, uint32_t* __restrict__ hit_count)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    bool h = false;
    if (r < n) {
        const int32_t a = actual[r];
        const int32_t* p = predicted + static_cast<uint64_t>(r) * k;
        for (uint32_t j = 0; j < k; ++j) h = h || (p[j] == a);
        hit[r] = h ? 1 : 0;
    }
    const unsigned long long mask = __ballot(h);              // 64-bit verify mask of this wave
    if (lane == 0u && mask) atomicAdd(hit_count, static_cast<uint32_t>(__popcll(mask)));
}

__global__ void k_apply_updates(const DevAlloc* __restrict__ tab, const MirrorUpdate* __restrict__ up, uint32_t n)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const MirrorUpdate u = up[i];
    const DevAlloc t = tab[u.alloc_idx];
    if (!t.entries) return;
    if (u.and_mask != 0xFFFFFFFFu) atomicAnd(&t.d_flags[u.page], u.and_mask);
    if (u.or_mask) atomicOr(&t.d_flags[u.page], u.or_mask);
    if (u.slot != kKeepSlot) t.d_slot[u.page] = u.slot;
}

__global__ void k_init_entries(PageEntry* e, uint64_t n, uint64_t base, uint64_t stride, uint64_t rec0)
{
    const uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (stride == kPlanarMx4) { e[i].pool_addr = base + mx4_nib_off(rec0 + i); e[i].rec_bytes = 0; e[i].scale = __uint_as_float(mx4_code_delta(rec0 + i)); }
    else { e[i].pool_addr = base + i * stride; e[i].rec_bytes = 0; e[i].scale = 1.0f; }
}

// ===================================================================
// fused dequant-matvec (BASELINE config 5): q.K^T from FP8 records on the matrix cores
// ===================================================================
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ long pack64(uint32_t lo, uint32_t hi)
{
    return static_cast<long>(static_cast<uint64_t>(lo) | (static_cast<uint64_t>(hi) << 32));
}

// per query row: scale = max|q|/448 (1 if zero), e4m3 bytes of clamp(q/scale); rows >= g are zero
__global__ __launch_bounds__(64) void k_quantize_q_e4m3(const uint16_t* __restrict__ q16, uint32_t g,
                                                        uint32_t d, uint8_t* __restrict__ q8,
                                                        float* __restrict__ qs)
{
    // blockIdx.x runs over (layer, head, row): q16 is [layers][heads][g][d], q8 [layers][heads][16][d]
    const uint32_t lane = threadIdx.x, h = blockIdx.x / 16u, m = blockIdx.x % 16u;
    uint8_t* out = q8 + (static_cast<uint64_t>(h) * 16u + m) * d;
    if (m >= g) {
        for (uint32_t i = lane; i < d; i += 64u) out[i] = 0;
        if (lane == 0) qs[h * 16u + m] = 1.0f;
        return;
    }
    const uint16_t* row = q16 + (static_cast<uint64_t>(h) * g + m) * d;
    float mx = 0.0f;
    for (uint32_t i = lane; i < d; i += 64u) { const float a = fabsf(half_bits_to_float(row[i])); mx = (a > mx) ? a : mx; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const float t = __shfl_xor(mx, o); mx = (t > mx) ? t : mx; }
    const float sc = (mx > 0.0f) ? (mx / 448.0f) : 1.0f;
    for (uint32_t i = lane; i < d; i += 64u) {
        const float v = fminf(fmaxf(half_bits_to_float(row[i]) / sc, -448.0f), 448.0f);
        out[i] = static_cast<uint8_t>(__builtin_amdgcn_cvt_pk_fp8_f32(v, 0.0f, 0, false) & 0xFF);
    }
    if (lane == 0) qs[h * 16u + m] = sc;
}

// One wave = 8 pages = 16 positions, all 8 kv heads, K tile staged in LDS.
//   * fetch: the 16 KiB tile goes pool -> LDS with 16 global_load_lds_dwordx4 (1 KiB
//     each, no VGPRs); instruction i brings the row block of position i (8 heads x
//     128 B).  Whole 128-byte lines per instruction, pages read exactly once.
//   * LDS image: row block i sits at i*1024; its 16-byte chunks are XOR-swizzled with
//     i ON THE SOURCE SIDE (lane l fetches chunk l^i), because the MFMA reader walks
//     16 row blocks at the same in-row offset (1 KiB stride = one bank otherwise).
//   * MFMA 16x16x32 fp8: lane (c = l%16, kb = l/16) feeds 8 consecutive d of query
//     row c (A) / position c (B); the d axis is permuted so lane kb owns
//     d in [32kb, 32kb+32) -> two ds_read_b64 per 16-byte chunk.
__global__ __launch_bounds__(128) void k_qk_scores_fp8(const PageEntry* __restrict__ entries,
        uint64_t first_page, uint64_t layer_page_stride, uint32_t n_pages, uint32_t heads, uint32_t g,
        const uint8_t* __restrict__ q8, const float* __restrict__ qs, float* __restrict__ out)
{
    __shared__ __attribute__((aligned(1024))) uint8_t tiles[2][16384];
    // blockIdx.y = layer (several layers of one sequence in one launch)
    first_page += blockIdx.y * layer_page_stride;
    q8 += static_cast<uint64_t>(blockIdx.y) * heads * 16u * 128u;
    qs += static_cast<uint64_t>(blockIdx.y) * heads * 16u;
    out += static_cast<uint64_t>(blockIdx.y) * heads * g * 2u * n_pages;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t page0 = (blockIdx.x * 2u + wave) * 8u;            // wave-uniform
    if (page0 >= n_pages) return;
    uint8_t* tile = tiles[wave];
    const uint32_t n_pos = 2u * n_pages;
    // ---- fetch: 8 pages x 2 positions, descriptors through the scalar cache
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const uint32_t pg = page0 + j;
        PageEntry e{0, 0, 0.0f};
        if (pg < n_pages) e = entries[first_page + pg];
        const bool ok = pg < n_pages && e.rec_bytes >= kBlockElems;       // wave-uniform
#pragma unroll
        for (int sl = 0; sl < 2; ++sl) {
            const int i = 2 * j + sl;
            if (ok) {
                const uint8_t* src = reinterpret_cast<const uint8_t*>(e.pool_addr) + sl * 1024 + ((lane ^ i) * 16u);
                __builtin_amdgcn_global_load_lds(
                    (const __attribute__((address_space(1))) void*)(src),
                    (__attribute__((address_space(3))) void*)(tile + i * 1024), 16, 0, 0);
            } else {
                *reinterpret_cast<uint4*>(tile + i * 1024 + lane * 16u) = make_uint4(0u, 0u, 0u, 0u);
            }
        }
    }
    const uint32_t c = lane & 15u, kb = lane >> 4;
    const uint32_t pgc = page0 + (c >> 1);
    const bool live = pgc < n_pages;
    float ks = 0.0f;
    if (live) {
        const PageEntry ec = entries[first_page + pgc];
        ks = ec.rec_bytes >= kBlockElems ? ec.scale : 0.0f;
    }
    // query operands and row scales of all 8 heads: requested while the tile is in flight
    uint4 a0[8], a1[8];
    f32x4 qsc[8];
#pragma unroll
    for (int h = 0; h < 8; ++h) {
        const uint8_t* qrow = q8 + (static_cast<uint64_t>(h) * 16u + c) * 128u + kb * 32u;
        a0[h] = *reinterpret_cast<const uint4*>(qrow);
        a1[h] = *reinterpret_cast<const uint4*>(qrow + 16);
        qsc[h] = *reinterpret_cast<const f32x4*>(qs + h * 16u + 4u * kb);
    }
    const uint32_t t = page0 * 2u + c;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                    // DMA landed, operands loaded
    wave_lds_fence();
#pragma unroll
    for (int h = 0; h < 8; ++h) {
        // B: chunks h*8 + kb*2 (+1) of row block c, at their swizzled place
        const uint32_t q0 = (static_cast<uint32_t>(h) * 8u + kb * 2u) ^ c, q1 = (static_cast<uint32_t>(h) * 8u + kb * 2u + 1u) ^ c;
        const uint2 b00 = *reinterpret_cast<const uint2*>(tile + c * 1024u + q0 * 16u);
        const uint2 b01 = *reinterpret_cast<const uint2*>(tile + c * 1024u + q0 * 16u + 8u);
        const uint2 b10 = *reinterpret_cast<const uint2*>(tile + c * 1024u + q1 * 16u);
        const uint2 b11 = *reinterpret_cast<const uint2*>(tile + c * 1024u + q1 * 16u + 8u);
        f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
        acc = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(pack64(a0[h].x, a0[h].y), pack64(b00.x, b00.y), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(pack64(a0[h].z, a0[h].w), pack64(b01.x, b01.y), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(pack64(a1[h].x, a1[h].y), pack64(b10.x, b10.y), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(pack64(a1[h].z, a1[h].w), pack64(b11.x, b11.y), acc, 0, 0, 0);
        // accumulator: lane holds rows m = 4*kb + i (i = 0..3) of column c
        if (live) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const uint32_t m = 4u * kb + i;
                if (m < g) out[(static_cast<uint64_t>(h) * g + m) * n_pos + t] = acc[i] * ks * qsc[h][i];
            }
        }
    }
}

// ===================================================================
// token predictor  (src/prefetcher/lstm_predictor.cpp:40-188)
// ===================================================================
// The reference's "LSTM" is degenerate: gates fixed at 0.5, recurrent weights unused,
// candidate g = sum_j 0.1*embedding[token][j] (lstm_predictor.cpp:117-146).  These
// kernels compute exactly that maths for a batch of 16-token histories, then the
// 128 x vocab output mat-vec, softmax and top-k.  fp tolerance vs the oracle: the
// device tanhf/expf and the reduction order differ from glibc's (tests state 1e-4).
constexpr uint32_t kPredHist = 16, kPredEmb = 64, kPredHidden = 128;
// All-lanes reductions over the wave for the top-k rounds, written for latency (a round is a chain of six exchanges): the
// four steps inside a row of 16 lanes are DPP moves (quad_perm xor 1, xor 2, row_half_mirror, row_mirror: a few clocks each);
// rows 16 apart and the two halves of the wave meet through gfx950's v_permlane16_swap / v_permlane32_swap -- with both operands
// the same register they return the two rows (halves) side by side in every lane, still in the vector ALU.  (ds_swizzle and
// ds_bpermute, two trips through the LDS crossbar per reduction, were most of a one-request prediction: 15.7 us with them.)
template <int CTRL> __device__ __forceinline__ uint32_t tk_dpp(uint32_t v) { return static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), CTRL, 0xf, 0xf, false)); }
template <int STEP> __device__ __forceinline__ uint32_t tk_exchange(uint32_t v)
{
    static_assert(STEP < 4, "rows and halves: tk_rows / tk_halves");
    if constexpr (STEP == 0) return tk_dpp<0xB1>(v);                 // quad_perm [1,0,3,2]
    else if constexpr (STEP == 1) return tk_dpp<0x4E>(v);            // quad_perm [2,3,0,1]
    else if constexpr (STEP == 2) return tk_dpp<0x141>(v);           // row_half_mirror: the other quad of each 8
    else return tk_dpp<0x140>(v);                                    // row_mirror: the other 8 of each 16
}
struct TkPair { uint32_t a, b; };                                    // a lane's own value and its partner's (in no particular order)
__device__ __forceinline__ TkPair tk_rows(uint32_t v) { const auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false); return TkPair{r[0], r[1]}; }      // lane ^ 16
__device__ __forceinline__ TkPair tk_halves(uint32_t v) { const auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false); return TkPair{r[0], r[1]}; }    // lane ^ 32
__device__ __forceinline__ uint64_t wave_max_u64(uint64_t k)
{
#define SPECKV_TK_STEP(S) { const uint64_t other = (static_cast<uint64_t>(tk_exchange<S>(static_cast<uint32_t>(k >> 32))) << 32) | tk_exchange<S>(static_cast<uint32_t>(k)); k = other > k ? other : k; }
    SPECKV_TK_STEP(0) SPECKV_TK_STEP(1) SPECKV_TK_STEP(2) SPECKV_TK_STEP(3)
#undef SPECKV_TK_STEP
    {
        const TkPair hi = tk_rows(static_cast<uint32_t>(k >> 32)), lo = tk_rows(static_cast<uint32_t>(k));
        const uint64_t x = (static_cast<uint64_t>(hi.a) << 32) | lo.a, y = (static_cast<uint64_t>(hi.b) << 32) | lo.b;
        k = x > y ? x : y;
    }
    {
        const TkPair hi = tk_halves(static_cast<uint32_t>(k >> 32)), lo = tk_halves(static_cast<uint32_t>(k));
        const uint64_t x = (static_cast<uint64_t>(hi.a) << 32) | lo.a, y = (static_cast<uint64_t>(hi.b) << 32) | lo.b;
        k = x > y ? x : y;
    }
    return k;
}
__device__ __forceinline__ float wave_max_f32(float v)
{
    v = fmaxf(v, __uint_as_float(tk_exchange<0>(__float_as_uint(v))));
    v = fmaxf(v, __uint_as_float(tk_exchange<1>(__float_as_uint(v))));
    v = fmaxf(v, __uint_as_float(tk_exchange<2>(__float_as_uint(v))));
    v = fmaxf(v, __uint_as_float(tk_exchange<3>(__float_as_uint(v))));
    const TkPair r = tk_rows(__float_as_uint(v));
    v = fmaxf(__uint_as_float(r.a), __uint_as_float(r.b));
    const TkPair h = tk_halves(__float_as_uint(v));
    return fmaxf(__uint_as_float(h.a), __uint_as_float(h.b));
}
__device__ __forceinline__ float wave_sum_f32(float v)
{
    v += __uint_as_float(tk_exchange<0>(__float_as_uint(v)));
    v += __uint_as_float(tk_exchange<1>(__float_as_uint(v)));
    v += __uint_as_float(tk_exchange<2>(__float_as_uint(v)));
    v += __uint_as_float(tk_exchange<3>(__float_as_uint(v)));
    const TkPair r = tk_rows(__float_as_uint(v));
    v = __uint_as_float(r.a) + __uint_as_float(r.b);
    const TkPair h = tk_halves(__float_as_uint(v));
    return __uint_as_float(h.a) + __uint_as_float(h.b);
}
// tanh(x) = 1 - 2 / (exp(2x) + 1) on the hardware exponential and reciprocal: absolute error ~1e-7, i.e. 1e-5 relative at the
// |x| ~ 0.01 the reference's cell states have (tests: confidences within 5e-4 of the oracle).  libm's tanhf is ~100 instructions,
// and the recurrence is a chain of 16 x layers x 2 of them.
// exp(x) for the softmax terms (x <= 0): the hardware's exp2 on x log2(e), relative error ~1e-6 at |x| ~ 20 (libm's expf is ~20
// instructions and every logit of every request takes one; tests state 5e-4 on the confidences against the oracle).
__device__ __forceinline__ float pred_fast_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.4426950408889634f); }
__device__ __forceinline__ float pred_fast_tanh(float x)
{
    x = fminf(fmaxf(x, -15.0f), 15.0f);
    const float e = __builtin_amdgcn_exp2f(x * 2.8853900817779268f);         // exp(2x)
    return 1.0f - 2.0f * __builtin_amdgcn_rcpf(e + 1.0f);
}

// one wave per request: lane 0 walks the history; every lane stores 2 of the 128 hidden values
__global__ __launch_bounds__(64) void k_lstm_hidden(const int32_t* __restrict__ hist, uint32_t n,
        const float* __restrict__ emb, uint32_t vocab, uint32_t layers, float* __restrict__ hid)
{
    const uint32_t r = blockIdx.x, lane = threadIdx.x;
    if (r >= n) return;
    // candidate g_t = sum_j 0.1*embedding[token_t][j]: lane j holds entry j of every token's row
    // (16 independent loads in flight), one wave reduction per token
    float g[kPredHist];
#pragma unroll
    for (uint32_t t = 0; t < kPredHist; ++t) {
        const uint32_t tok = static_cast<uint32_t>(hist[r * kPredHist + t]);
        g[t] = (tok < vocab) ? emb[static_cast<uint64_t>(tok) * kPredEmb + lane] * 0.1f : 0.0f;
    }
    float tg[kPredHist];
#pragma unroll
    for (uint32_t t = 0; t < kPredHist; ++t) tg[t] = 0.5f * pred_fast_tanh(wave_sum_f32(g[t]));      // (independent of the chain)
    float h = 0.0f, c = 0.0f;
#pragma unroll
    for (uint32_t t = 0; t < kPredHist; ++t)
        for (uint32_t l = 0; l < layers; ++l) {
            c = 0.5f * c + tg[t];
            h = 0.5f * pred_fast_tanh(c);
        }
    h = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(h)));
    hid[static_cast<uint64_t>(r) * kPredHidden + lane] = h;
    hid[static_cast<uint64_t>(r) * kPredHidden + 64u + lane] = h;
}

// A REAL LSTM cell (the reference's is degenerate, above; SURVEY 8f N1: "semantics must be defined by us"): the standard
// cell with PyTorch's nn.LSTM conventions -- per layer  gates = W_ih x + W_hh h + b  (4 x 128 rows, order i, f, g, o),
// c = sigmoid(f) c + sigmoid(i) tanh(g),  h = sigmoid(o) tanh(c),  h_0 = c_0 = 0, layer l > 0 fed with layer l-1's h of the
// same time step; 16-token history, embedding width 64, hidden width 128.  Output: the top layer's last h.
//   A prediction is a chain of 16 x layers dependent steps, so the kernel is written for the length of a step, layer by layer:
//   * one workgroup = one request, 512 threads (256 requests = one workgroup per CU);
//   * the layer's input projections W_ih x_t + b of ALL 16 steps have no dependency: computed first, into LDS;
//   * a thread keeps, in 128 registers for the 16 recurrent steps, the weights of EIGHT gate rows over an eighth of the
//     columns (lstm_arranged_index; the host arranged them so that the 512 threads read coalesced).  A step is 64
//     packed fused multiply-adds per thread (v_pk_fma_f32 over two neighbouring columns) against its 16 values of h -- four
//     16-byte LDS reads -- then a reduction over the eight threads that share the rows (7 exchanges: DPP inside a quad,
//     ds_swizzle across, after which thread tid owns gate row tid);
//   * the kernel numbers gate rows 4 * unit + gate, so the four gates of a hidden unit end in the four lanes of a quad: each
//     lane applies its gate's non-linearity (tanh as 2 sigmoid(2x) - 1: one code path), the quad exchanges the four results
//     with DPP, and all four lanes carry c (in a register) and h; lane 0 writes h -- to the layer's output sequence, which
//     is also where the next step reads it, so a step has ONE barrier and no buffer is ever rewritten while it is read.
//   The forms before this one: a thread owning ONE whole gate row read all of h, 64 16-byte LDS reads per step and wave; the
//   LDS returns 128 bytes per clock however many lanes ask for the same word, so a step was 8 waves x 64 reads x 8 clocks =
//   1.7 us of LDS time against 0.4 us of arithmetic (0.083 ms per prediction; with separate multiply and add,
//   -ffp-contract=off as the reference's cell needs, 0.110-0.117 ms).  Sliced rows with the non-linearities on 256 threads
//   between two barriers (libm tanhf, IEEE division): 0.070 ms.  First version (weights streamed from L2 in every step,
//   8 requests per workgroup): 0.7-0.8 ms per prediction of 256 requests.
struct LstmWeights { const float* w_ih_t[4]; const float* w_hh_t[4]; const float* bias[4]; uint32_t layers; };   // bias = b_ih + b_hh
constexpr uint32_t kLstmPitch = kPredHidden + 4u * (kPredHidden / 16u);      // a 16-column slice starts 20 floats after the one before: the eight slices a wave reads fall in different banks
__device__ __forceinline__ constexpr uint32_t lstm_pad(uint32_t j) { return j + 4u * (j >> 4); }
__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false)); }
__device__ __forceinline__ float lane_xor7(float v) { return dpp_mov<0x141>(v); }    // row_half_mirror: lane 7 - s of each 8 (a DPP move; lane ^ 4 would be a ds_swizzle)
__device__ __forceinline__ float lane_xor2(float v) { return dpp_mov<0x4E>(v); }     // quad_perm [2,3,0,1]
__device__ __forceinline__ float lane_xor1(float v) { return dpp_mov<0xB1>(v); }     // quad_perm [1,0,3,2]
// v[i] of slice-thread s holds a partial sum of gate row 8 * group + (i ^ s): after three exchanges with the threads s ^ 7,
// s ^ 2, s ^ 1 the return value is the whole sum of row 8 * group + s, i.e. of row threadIdx.x.  (First exchange: thread s keeps
// the rows (i ^ s), i < 4; its partner 7 - s = s ^ 7 holds its share of row i ^ s in v[(i ^ s) ^ (s ^ 7)] = v[7 - i].  All three are
// DPP moves: the step of the recurrence has no trip through the LDS crossbar left.)
__device__ __forceinline__ float lstm_reduce8(float (&v)[8])
{
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] += lane_xor7(v[7 - i]);
#pragma unroll
    for (int i = 0; i < 2; ++i) v[i] += lane_xor2(v[i + 2]);
    return v[0] + lane_xor1(v[1]);
}
// sum over this thread's CS columns of  w[i][c] * x[c]  for its eight rows i; x: the thread's slice of the input vector
template <uint32_t CS>
__device__ __forceinline__ float lstm_slice_dot(const float (&w)[8u * CS], const float* x)
{
    f32x2 xv[CS / 2u];
#pragma unroll
    for (uint32_t k = 0; k < CS / 4u; ++k) {
        const float4 a = *reinterpret_cast<const float4*>(x + 4u * k);
        xv[2u * k] = f32x2{a.x, a.y}; xv[2u * k + 1u] = f32x2{a.z, a.w};
    }
    float red[8];
#pragma unroll
    for (uint32_t i = 0; i < 8u; ++i) {
        f32x2 a = {0.0f, 0.0f};                                         // (even columns, odd columns)
#pragma unroll
        for (uint32_t k = 0; k < CS / 2u; ++k) a = pk_fma(f32x2{w[i * CS + 2u * k], w[i * CS + 2u * k + 1u]}, xv[k], a);
        red[i] = a.x + a.y;
    }
    return lstm_reduce8(red);
}
template <uint32_t CS>
__device__ __forceinline__ void lstm_project(const float* __restrict__ wsrc, float b, const float (&seq)[kPredHist][kLstmPitch],
                                             float (&xp)[kPredHist][4 * kPredHidden])
{
    const uint32_t tid = threadIdx.x, s = tid & 7u;
    float w[8u * CS];
#pragma unroll
    for (uint32_t q = 0; q < 8u * CS; ++q) w[q] = wsrc[q * 512u + tid];
#pragma unroll 2
    for (uint32_t t = 0; t < kPredHist; ++t) xp[t][tid] = lstm_slice_dot<CS>(w, &seq[t][lstm_pad(CS * s)]) + b;
}
__device__ __forceinline__ float sigmoid_rcp(float x) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -1.4426950408889634f)); }     // the hardware's exp2 and reciprocal (1 ulp each; two of them on every step of the chain)
__global__ __launch_bounds__(512) void k_lstm_cell(const int32_t* __restrict__ hist, uint32_t n, const float* __restrict__ emb, uint32_t vocab,
                                                  LstmWeights w, float* __restrict__ hid)
{
    __shared__ __attribute__((aligned(16))) float seq[kPredHist][kLstmPitch];        // the layer's input sequence, then its own output (columns at lstm_pad)
    __shared__ float xp[kPredHist][4 * kPredHidden];                                // W_ih x_t + b of the current layer; [.][tid] is written and read by thread tid only
    const uint32_t tid = threadIdx.x, req = blockIdx.x, s = tid & 7u;
    const uint32_t unit = tid >> 2, gate = tid & 3u;                      // the gate row this thread owns after a reduction
    for (uint32_t i = tid; i < kPredHist * kPredEmb; i += 512u) {
        const uint32_t t = i / kPredEmb, j = i % kPredEmb;
        const uint32_t tok = static_cast<uint32_t>(hist[req * kPredHist + t]);
        seq[t][lstm_pad(j)] = tok < vocab ? emb[static_cast<uint64_t>(tok) * kPredEmb + j] : 0.0f;
    }
    __syncthreads();
    float h = 0.0f;
    for (uint32_t l = 0; l < w.layers; ++l) {
        const float b = w.bias[l][gate * kPredHidden + unit];
        if (l == 0) lstm_project<kPredEmb / 8u>(w.w_ih_t[l], b, seq, xp);
        else        lstm_project<kPredHidden / 8u>(w.w_ih_t[l], b, seq, xp);
        float wh[kPredHidden];                                            // eight rows x sixteen columns of W_hh
        {
            const float* whp = w.w_hh_t[l] + tid;
#pragma unroll
            for (uint32_t q = 0; q < kPredHidden; ++q) wh[q] = whp[q * 512u];
        }
        float c = 0.0f;
        __syncthreads();                                                  // everybody is done with seq as this layer's input
#pragma unroll 1
        for (uint32_t t = 0; t < kPredHist; ++t) {
            float g = xp[t][tid];
            if (t) g += lstm_slice_dot<kPredHidden / 8u>(wh, &seq[t - 1u][lstm_pad(16u * s)]);      // h_{-1} = 0
            const bool is_g = gate == 2u;
            const float sg = sigmoid_rcp(is_g ? g + g : g);
            const float act = is_g ? sg + sg - 1.0f : sg;                 // tanh(x) = 2 sigmoid(2x) - 1
            const float ai = dpp_mov<0x00>(act), af = dpp_mov<0x55>(act), ag = dpp_mov<0xAA>(act), ao = dpp_mov<0xFF>(act);   // quad_perm [k,k,k,k]
            c = af * c + ai * ag;
            const float sc = sigmoid_rcp(c + c);
            h = ao * (sc + sc - 1.0f);
            if (gate == 0u) seq[t][lstm_pad(unit)] = h;
            __syncthreads();
        }
    }
    if (gate == 0u) hid[static_cast<uint64_t>(req) * kPredHidden + unit] = h;
}

// logits[b][i] = sum_j hid[b][j] * wout[i][j] (+ bias[i]) on the fp32 matrix cores: a wave owns 32 output rows (16 KiB of
// weights, read once and kept in 64 registers) and walks the requests in tiles of 32 with v_mfma_f32_32x32x2_f32 -- the hidden
// vectors are the A operand (M = request), the weights the B operand (N = output row), so that an accumulator register holds
// 32 consecutive logits of one request per half-wave and every store instruction writes two whole 128-byte lines.
// The vector-ALU form this replaces (one quarter-row per lane, multiply and add per weight and request) needed ~80 VALU
// instructions per request and wave, 4 cycles each on a 16-lane SIMD: 0.115 ms for 256 requests against ~0.014 ms of matrix
// time (the instruction runs at 64 cycles back to back also on one accumulator: profiles/tools/probe/mfma_f32_rate.hip, 143-156
// TFLOP/s).  This kernel: 0.030 ms, of which 0.004 the stores and ~0.005 the weights' first read (one request: 0.0066 ms).
// The order of the 128 additions of one logit: k = 8j + 4*(lane/32) + e for j = 0..15, e = 0..3, the lower half-wave's k first
// inside each instruction (fused, unlike the oracle's mul + add: covered by the confidence tolerance of the parity tests).
typedef float f32x16 __attribute__((ext_vector_type(16)));
// tiles of 32 output rows, rounded up to the four waves of a workgroup of k_lstm_logits (every wave loads its tile unconditionally)
__host__ __device__ constexpr uint32_t logits_tiles_padded(uint32_t vocab) { return ((vocab + 31u) / 32u + 3u) & ~3u; }
// The output layer's weights in the order k_lstm_logits reads them: per 32 rows, float4 [j][lane] = row (lane % 32),
// columns 8j + 4 (lane / 32) .. + 3 -- a wave's load instruction is then one contiguous KiB (row-major, its 64 lanes touched
// 64 different lines 16 bytes at a time).  Once per predictor_load.
__global__ __launch_bounds__(256) void k_arrange_wout(const float* __restrict__ src, float4* __restrict__ dst, uint32_t vocab)
{
    const uint32_t lane = threadIdx.x & 63u, tile = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint32_t row = tile * 32u + (lane & 31u), kh = lane >> 5;
    if (tile >= logits_tiles_padded(vocab)) return;         // (tiles past the vocabulary, up to a whole workgroup of k_lstm_logits: zeros)
#pragma unroll
    for (uint32_t j = 0; j < 16u; ++j) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (row < vocab) v = *reinterpret_cast<const float4*>(src + static_cast<uint64_t>(row) * kPredHidden + 8u * j + 4u * kh);
        dst[(static_cast<uint64_t>(tile) * 16u + j) * 64u + lane] = v;
    }
}
constexpr uint32_t kLogitsTile = 32;        // requests per matrix tile
constexpr uint32_t kLogitsChunk = 128;      // requests per workgroup column (blockIdx.y): 2 waves per SIMD at 256 requests x 32 000 rows
constexpr uint32_t kLogitsPitch = kPredHidden + 4u;     // floats; 16 lanes x 16 B of one ds_read_b128 fall in 64 different banks
__global__ __launch_bounds__(256) void k_lstm_logits(const float* __restrict__ hid, uint32_t n,
        const float* __restrict__ wout, const float* __restrict__ out_bias, uint32_t vocab, float* __restrict__ logits)
{
    static_assert(kPredHidden == 128u, "16 float4 per lane and operand");
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t c = lane & 31u, kh = lane >> 5;
    const uint32_t row = (blockIdx.x * 4u + wave) * 32u + c;
    const bool live = row < vocab;
    // the weights arrive arranged (k_arrange_wout): the wave's 32 rows are 16 KiB in a row, [j][lane] float4, rows past the vocabulary zero
    float4 wq[16];
    const float4* wt = reinterpret_cast<const float4*>(wout) + static_cast<uint64_t>(blockIdx.x * 4u + wave) * (16u * 64u) + lane;
#pragma unroll
    for (int j = 0; j < 16; ++j) wq[j] = wt[j * 64];
    const float bias = (out_bias && live) ? out_bias[row] : 0.0f;
    const uint32_t b_begin = blockIdx.y * kLogitsChunk, b_end = min(n, b_begin + kLogitsChunk);
    // A tile of hidden vectors (32 requests, 16 KiB) goes through LDS, shared by the four waves; two buffers, so one barrier
    // per tile: a buffer is rewritten two tiles later, behind the barrier of the tile in between.
    __shared__ __attribute__((aligned(16))) float hs[2][kLogitsTile][kLogitsPitch];
    static_assert(kLogitsTile * kPredHidden / 4u == 4u * 256u, "four float4 per thread and tile");
    float4 nx[4];
    auto fetch = [&](uint32_t b0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint32_t e = threadIdx.x + 256u * static_cast<uint32_t>(q);
            const uint32_t r = e / (kPredHidden / 4u), c4 = e % (kPredHidden / 4u);
            nx[q] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (b0 + r < b_end) nx[q] = *reinterpret_cast<const float4*>(hid + static_cast<uint64_t>(b0 + r) * kPredHidden + 4u * c4);
        }
    };
    fetch(b_begin);
    uint32_t buf = 0;
    for (uint32_t b0 = b_begin; b0 < b_end; b0 += kLogitsTile, buf ^= 1u) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint32_t e = threadIdx.x + 256u * static_cast<uint32_t>(q);
            *reinterpret_cast<float4*>(&hs[buf][e / (kPredHidden / 4u)][4u * (e % (kPredHidden / 4u))]) = nx[q];
        }
        __syncthreads();
        if (b0 + kLogitsTile < b_end) fetch(b0 + kLogitsTile);
        f32x16 acc;
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[v] = 0.0f;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const float4 h4 = *reinterpret_cast<const float4*>(&hs[buf][c][8u * j + 4u * kh]);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(h4.x, wq[j].x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(h4.y, wq[j].y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(h4.z, wq[j].z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(h4.w, wq[j].w, acc, 0, 0, 0);
        }
        // acc[v]: request b0 + 8*(v/4) + 4*(lane/32) + v%4, output row `row`
        if (live) {
            float* o = logits + static_cast<uint64_t>(b0 + 4u * kh) * vocab + row;
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const uint32_t m = 8u * (v >> 2) + (v & 3);
                if (b0 + 4u * kh + m < b_end) o[static_cast<uint64_t>(m) * vocab] = out_bias ? acc[v] + bias : acc[v];
            }
        }
    }
}

// softmax + top-k of one request per workgroup (k <= 8).  Ties: lower token id first.
constexpr uint32_t kSmThreads = 1024;
__global__ __launch_bounds__(1024) void k_softmax_topk(const float* __restrict__ logits, uint32_t vocab,
        uint32_t k, int32_t* __restrict__ out_tok, float* __restrict__ out_conf)
{
    __shared__ float red[kSmThreads];
    __shared__ uint32_t redi[kSmThreads];
    const uint32_t b = blockIdx.x, tid = threadIdx.x;
    const float* l = logits + static_cast<uint64_t>(b) * vocab;
    float val[8];
    uint32_t idx[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { val[i] = -INFINITY; idx[i] = 0xFFFFFFFFu; }
    float mx = -INFINITY;
    for (uint32_t i = tid; i < vocab; i += kSmThreads) {
        const float v = l[i];
        mx = fmaxf(mx, v);
        // sorted insertion (descending value, ascending index)
        if (v > val[7] || (v == val[7] && i < idx[7])) {
            val[7] = v; idx[7] = i;
#pragma unroll
            for (int j = 7; j > 0; --j) {
                const bool sw = val[j] > val[j - 1] || (val[j] == val[j - 1] && idx[j] < idx[j - 1]);
                if (sw) { const float tv = val[j]; val[j] = val[j - 1]; val[j - 1] = tv;
                          const uint32_t ti = idx[j]; idx[j] = idx[j - 1]; idx[j - 1] = ti; }
            }
        }
    }
    red[tid] = mx; __syncthreads();
    for (uint32_t s = kSmThreads / 2; s > 0; s >>= 1) { if (tid < s) red[tid] = fmaxf(red[tid], red[tid + s]); __syncthreads(); }
    mx = red[0]; __syncthreads();
    float sum = 0.0f;
    for (uint32_t i = tid; i < vocab; i += kSmThreads) sum += expf(l[i] - mx);
    red[tid] = sum; __syncthreads();
    for (uint32_t s = kSmThreads / 2; s > 0; s >>= 1) { if (tid < s) red[tid] += red[tid + s]; __syncthreads(); }
    sum = red[0]; __syncthreads();
    uint32_t head = 0;
    for (uint32_t r = 0; r < k; ++r) {
        float cv = -INFINITY; uint32_t ci = 0xFFFFFFFFu;
#pragma unroll
        for (int j = 0; j < 8; ++j) if (static_cast<uint32_t>(j) == head) { cv = val[j]; ci = idx[j]; }
        red[tid] = cv; redi[tid] = ci; __syncthreads();
        for (uint32_t s = kSmThreads / 2; s > 0; s >>= 1) {
            if (tid < s) {
                const float ov = red[tid + s]; const uint32_t oi = redi[tid + s];
                if (ov > red[tid] || (ov == red[tid] && oi < redi[tid])) { red[tid] = ov; redi[tid] = oi; }
            }
            __syncthreads();
        }
        const float bv = red[0]; const uint32_t bi = redi[0];
        __syncthreads();
        if (ci == bi && ci != 0xFFFFFFFFu) ++head;                 // the owner of the winner advances
        if (tid == 0) {
            out_tok[b * k + r] = static_cast<int32_t>(bi);
            out_conf[b * k + r] = expf(bv - mx) / sum;
        }
    }
}

// The same for vocabularies of up to 262 144 tokens (kTkMaxParts parts), written for latency (one request per workgroup is a
// chain of dependent steps: the kernel above took 48-57 us per launch whatever the batch, a 1024-thread workgroup with 32
// logits per thread in registers 21-26 us).  A request is cut into parts of 4096 logits, one workgroup of 256 threads each, 16
// logits per thread:
//   * a wave finds ITS maximum, exp-sum (relative to its own maximum) and top k with shuffles only -- a candidate is one
//     64-bit key, the logit's bits made order-preserving above ~token id, so "value descending, token id ascending" (the
//     order of the sorted insertion above) is an unsigned maximum and a round is six exchange steps;
//   * one barrier, then wave 0 merges the four waves (maxima, rescaled sums, 4 k keys) and writes the part's result;
//   * a second kernel, one wave per request, merges the parts -- a lane holds one part's maximum, sum and k keys (already in
//     order: a round offers the lane's best key not yet taken) -- and writes tokens and confidences exp(logit - max) / sum.  (One kernel whose last-arriving workgroup merges was tried: 10 us for one request, but the
//     agent-scope release/acquire it needs writes back and invalidates the XCD's L2 once per workgroup -- 48 us for 256
//     requests against 26 us before.)
// A logit that is -inf or NaN is never chosen (as above: "v > best" is false for it); a rank without a candidate reports
// token -1 and confidence 0.
constexpr uint32_t kTkMaxParts = 64, kTkThreads = 256, kTkPer = 16, kTkSpan = kTkThreads * kTkPer;
static_assert(kTkSpan == kPredictTopkSpan && kTkMaxParts == kPredictTopkMaxParts, "predict_ws_bytes");
// workspace of one request: parts x (max, sum) | parts x 8 keys
__device__ __forceinline__ uint32_t tk_ws_stride(uint32_t parts) { return parts * kPredictWsPerPart; }
__device__ __forceinline__ uint64_t tk_key(float v, uint32_t i)
{
    if (!(v > -INFINITY)) return 0;
    uint32_t bits = __float_as_uint(v);
    bits ^= (bits >> 31) ? 0xFFFFFFFFu : 0x80000000u;
    return (static_cast<uint64_t>(bits) << 32) | (0xFFFFFFFFu - i);
}
__device__ __forceinline__ float tk_value(uint64_t key)
{
    uint32_t bits = static_cast<uint32_t>(key >> 32);
    bits ^= (bits >> 31) ? 0x80000000u : 0xFFFFFFFFu;
    return __uint_as_float(bits);
}
// merge of up to 64 (max, sum) pairs and 64 keys held one per lane; k rounds; lane 0 hands every round's winner to `put`
template <typename Put>
__device__ __forceinline__ void tk_merge(float m, float s, uint64_t key, uint32_t k, float& m_all, float& s_all, Put put)
{
    m_all = wave_max_f32(m);
    s_all = wave_sum_f32(m > -INFINITY ? s * pred_fast_exp(m - m_all) : 0.0f);
    for (uint32_t r = 0; r < k; ++r) {
        const uint64_t w = wave_max_u64(key);
        if (w == key) key = 0;                                          // keys are distinct (token ids are): one owner
        put(r, w);
    }
}
__global__ __launch_bounds__(256) void k_softmax_topk_small(const float* __restrict__ logits, uint32_t vocab,
        uint32_t k, uint8_t* __restrict__ ws)
{
    __shared__ float wm[4], wsum[4];
    __shared__ uint64_t wkey[4][8];
    const uint32_t b = blockIdx.x, part = blockIdx.y, tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6;      // (requests on x: no 65 535 limit)
    const float* l = logits + static_cast<uint64_t>(b) * vocab;
    const uint32_t base = part * kTkSpan + tid;
    float v[kTkPer];
#pragma unroll
    for (uint32_t j = 0; j < kTkPer; ++j) {
        const uint32_t i = base + j * kTkThreads;
        v[j] = i < vocab ? l[i] : -INFINITY;
    }
    float m = v[0];
#pragma unroll
    for (uint32_t j = 1; j < kTkPer; ++j) m = fmaxf(m, v[j]);
    m = wave_max_f32(m);
    float sum = 0.0f;
#pragma unroll
    for (uint32_t j = 0; j < kTkPer; ++j)
        if (base + j * kTkThreads < vocab && m > -INFINITY) sum += pred_fast_exp(v[j] - m);
    sum = wave_sum_f32(sum);
    for (uint32_t r = 0; r < k; ++r) {
        float bv = -INFINITY; int bj = -1;
#pragma unroll
        for (int j = 0; j < static_cast<int>(kTkPer); ++j)                // ascending token id: ">" keeps the lowest id among equals
            if (v[j] > bv) { bv = v[j]; bj = j; }
        const uint64_t key = bj >= 0 ? tk_key(bv, base + static_cast<uint32_t>(bj) * kTkThreads) : 0;
        const uint64_t w = wave_max_u64(key);
        if (w != 0 && w == key) {                                       // the owner retires the winner
#pragma unroll
            for (int j = 0; j < static_cast<int>(kTkPer); ++j) if (j == bj) v[j] = -INFINITY;
        }
        if (lane == 0u) wkey[wv][r] = w;
    }
    if (lane == 0u) { wm[wv] = m; wsum[wv] = sum; }
    __syncthreads();
    if (wv != 0u) return;
    const uint32_t parts = gridDim.y;
    uint8_t* mine = ws + static_cast<uint64_t>(b) * tk_ws_stride(parts);
    float* part_ms = reinterpret_cast<float*>(mine);                     // [part] (max, sum)
    uint64_t* part_key = reinterpret_cast<uint64_t*>(mine + parts * 8u);             // [part][8]
    float pm, ps;
    tk_merge(lane < 4u ? wm[lane] : -INFINITY, lane < 4u ? wsum[lane] : 0.0f, lane < 4u * k ? wkey[lane / k][lane % k] : 0, k, pm, ps,
             [&](uint32_t r, uint64_t w) { if (lane == 0u) part_key[part * 8u + r] = w; });
    if (lane == 0u) { part_ms[2u * part] = pm; part_ms[2u * part + 1u] = ps; }
}
__global__ __launch_bounds__(256) void k_softmax_topk_merge(const uint8_t* __restrict__ ws, uint32_t n, uint32_t k, uint32_t parts,
        int32_t* __restrict__ out_tok, float* __restrict__ out_conf)
{
    const uint32_t b = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63u;
    if (b >= n) return;
    const uint8_t* mine = ws + static_cast<uint64_t>(b) * tk_ws_stride(parts);
    const float* part_ms = reinterpret_cast<const float*>(mine);
    const uint64_t* part_key = reinterpret_cast<const uint64_t*>(mine + parts * 8u);
    const bool have = lane < parts;                                      // lane = part
    const float qm = have ? part_ms[2u * lane] : -INFINITY, qs = have ? part_ms[2u * lane + 1u] : 0.0f;
    uint64_t key[8];
#pragma unroll
    for (uint32_t r = 0; r < 8u; ++r) key[r] = (have && r < k) ? part_key[lane * 8u + r] : 0;       // descending: key[0] is the part's best not yet taken
    const float mx = wave_max_f32(qm);
    const float total = wave_sum_f32(qm > -INFINITY ? qs * pred_fast_exp(qm - mx) : 0.0f);
    float* conf = out_conf + static_cast<uint64_t>(b) * k;
    int32_t* tok = out_tok + static_cast<uint64_t>(b) * k;
    for (uint32_t r = 0; r < k; ++r) {
        const uint64_t w = wave_max_u64(key[0]);
        if (w != 0 && w == key[0]) {                                     // keys are distinct (token ids are): one owner, whose next key moves up
#pragma unroll
            for (uint32_t j = 0; j < 7u; ++j) key[j] = key[j + 1u];
            key[7] = 0;
        }
        if (lane == 0u) {
            tok[r] = w ? static_cast<int32_t>(0xFFFFFFFFu - static_cast<uint32_t>(w)) : -1;
            conf[r] = w ? pred_fast_exp(tk_value(w) - mx) / total : 0.0f;
        }
    }
}

// ---- a handful of requests (n <= kPredictSmallN): written for the length of the chain, not for throughput ----------------
// The batch path above is four launches (hidden state | logits on the matrix cores, 32 requests per tile | top-k of 4096-logit
// parts | merge) and writes n x vocab logits to memory in between: for ONE request that is four launch gaps around 16 MB of
// weights (23.7 us per prediction back to back, README's "< 10 us" claim of the reference's FPGA).  Here two launches:
//   k_predict_small:  a wave takes 32 output rows (its 16 KiB of arranged weights: the same k_arrange_wout layout), forms
//     their logits for every request with plain fused multiply-adds against the hidden vector in LDS -- which the workgroup
//     computes itself for the reference's degenerate cell (one wave per request: the loop of k_lstm_hidden) or reads from
//     k_lstm_cell's output for the real one -- and reduces them at once: maximum, exp-sum, top k of its 32 rows by wave
//     exchanges, then the four waves' results to one (max, sum, k keys) of the workgroup's 128 rows.  Logits never leave
//     the CU.
//   k_predict_small_merge:  one workgroup per request merges the workgroups' results (a lane per part, 64 parts per wave,
//     as k_softmax_topk_merge) and writes tokens and confidences.  (Merged by the workgroup that finishes last instead -- one
//     launch, write-through stores, arrival counter: 17.1 us per prediction against 15.7 with the second launch; the lone
//     workgroup's chain of counter, agent-scope loads and exchanges is longer than a launch gap.
//     profiles/experiments/r04_predict_small_last_arriver_merge.patch)
// Same arithmetic as the batch path up to the order of the dot product's additions (tests state 1e-4 on confidences, as for
// the batch path against the oracle); vocabularies up to kPredictSmallMaxParts x 128 rows, larger ones take the batch path.
__global__ __launch_bounds__(256) void k_predict_small(const int32_t* __restrict__ hist, uint32_t n, const float* __restrict__ emb,
        const float* __restrict__ hid_in, uint32_t layers, const float* __restrict__ wout, const float* __restrict__ out_bias,
        uint32_t vocab, uint32_t k, uint8_t* __restrict__ ws)
{
    __shared__ __attribute__((aligned(16))) float hs[kPredictSmallN][kPredHidden];
    __shared__ float wm[kPredictSmallN][4], wsum[kPredictSmallN][4];
    __shared__ uint64_t wkey[kPredictSmallN][4][8];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t c = lane & 31u, kh = lane >> 5;
    const uint32_t tile = blockIdx.x * 4u + wave;
    const uint32_t row = tile * 32u + c;
    const bool live = row < vocab && kh == 0u;                          // the two halves of the wave end with the same logit: one counts
    // the wave's weights first (they are the long pole: 16 KiB per wave from memory), then the hidden vectors under their flight
    float4 wq[16];
    const float4* wt = reinterpret_cast<const float4*>(wout) + static_cast<uint64_t>(tile) * (16u * 64u) + lane;
#pragma unroll
    for (int j = 0; j < 16; ++j) wq[j] = wt[j * 64];
    const float bias = (out_bias && row < vocab) ? out_bias[row] : 0.0f;
    if (hid_in) {                                                       // the real cell's output (k_lstm_cell)
        for (uint32_t e = threadIdx.x; e < n * kPredHidden; e += 256u) hs[e / kPredHidden][e % kPredHidden] = hid_in[e];
    } else if (wave < n) {                                              // the reference's degenerate cell: k_lstm_hidden's loop, request = wave
        float g[kPredHist];
#pragma unroll
        for (uint32_t t = 0; t < kPredHist; ++t) {
            const uint32_t tok = static_cast<uint32_t>(hist[wave * kPredHist + t]);
            g[t] = (tok < vocab) ? emb[static_cast<uint64_t>(tok) * kPredEmb + lane] * 0.1f : 0.0f;
        }
#pragma unroll
        for (uint32_t t = 0; t < kPredHist; ++t) g[t] = wave_sum_f32(g[t]);
        // (the recurrence: pred_fast_tanh, as k_lstm_hidden)
        float tg[kPredHist];
#pragma unroll
        for (uint32_t t = 0; t < kPredHist; ++t) tg[t] = 0.5f * pred_fast_tanh(g[t]);      // (independent of the chain)
        float h = 0.0f, cc = 0.0f;
#pragma unroll
        for (uint32_t t = 0; t < kPredHist; ++t)
            for (uint32_t l = 0; l < layers; ++l) {
                cc = 0.5f * cc + tg[t];
                h = 0.5f * pred_fast_tanh(cc);
            }
        h = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(h)));
        hs[wave][lane] = h;
        hs[wave][64u + lane] = h;
    }
    __syncthreads();
    for (uint32_t b = 0; b < n; ++b) {                                  // (n <= 4: the weights stay in registers)
        float acc = 0.0f;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const float4 h4 = *reinterpret_cast<const float4*>(&hs[b][8u * j + 4u * kh]);
            acc = __builtin_fmaf(h4.x, wq[j].x, acc);
            acc = __builtin_fmaf(h4.y, wq[j].y, acc);
            acc = __builtin_fmaf(h4.z, wq[j].z, acc);
            acc = __builtin_fmaf(h4.w, wq[j].w, acc);
        }
        { const TkPair h2 = tk_halves(__float_as_uint(acc)); acc = __uint_as_float(h2.a) + __uint_as_float(h2.b); }   // the other half of the columns
        const float v = live ? (out_bias ? acc + bias : acc) : -INFINITY;
        const float m = wave_max_f32(v);
        const float sum = wave_sum_f32((live && m > -INFINITY) ? pred_fast_exp(v - m) : 0.0f);
        uint64_t key = live ? tk_key(v, row) : 0;
        for (uint32_t r = 0; r < k; ++r) {
            const uint64_t w = wave_max_u64(key);
            if (w == key) key = 0;                                      // one row per lane: the owner retires it
            if (lane == 0u) wkey[b][wave][r] = w;
        }
        if (lane == 0u) { wm[b][wave] = m; wsum[b][wave] = sum; }
    }
    __syncthreads();
    if (wave >= n) return;                                              // wave b merges request b's four results
    const uint32_t b = wave, parts = gridDim.x, part = blockIdx.x;
    uint8_t* mine = ws + static_cast<uint64_t>(b) * tk_ws_stride(parts);
    float* part_ms = reinterpret_cast<float*>(mine);                     // [part] (max, sum)
    uint64_t* part_key = reinterpret_cast<uint64_t*>(mine + parts * 8u);             // [part][8]
    float pm, ps;
    tk_merge(lane < 4u ? wm[b][lane] : -INFINITY, lane < 4u ? wsum[b][lane] : 0.0f, lane < 4u * k ? wkey[b][lane / k][lane % k] : 0, k, pm, ps,
             [&](uint32_t r, uint64_t w) { if (lane == 0u) part_key[part * 8u + r] = w; });
    if (lane == 0u) { part_ms[2u * part] = pm; part_ms[2u * part + 1u] = ps; }
}
__global__ __launch_bounds__(256) void k_predict_small_merge(const uint8_t* __restrict__ ws, uint32_t k, uint32_t parts,
        int32_t* __restrict__ out_tok, float* __restrict__ out_conf)
{
    __shared__ float wm[4], wsum[4];
    __shared__ uint64_t wkey[4][8];
    const uint32_t b = blockIdx.x, lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint8_t* mine = ws + static_cast<uint64_t>(b) * tk_ws_stride(parts);
    const float* part_ms = reinterpret_cast<const float*>(mine);
    const uint64_t* part_key = reinterpret_cast<const uint64_t*>(mine + parts * 8u);
    const uint32_t part = wave * 64u + lane;                             // lane = part (up to 256 of them)
    const bool have = part < parts;
    const float qm = have ? part_ms[2u * part] : -INFINITY, qs = have ? part_ms[2u * part + 1u] : 0.0f;
    uint64_t key[8];
#pragma unroll
    for (uint32_t r = 0; r < 8u; ++r) key[r] = (have && r < k) ? part_key[part * 8u + r] : 0;       // descending: key[0] is the part's best not yet taken
    const float mx = wave_max_f32(qm);
    const float total = wave_sum_f32(qm > -INFINITY ? qs * pred_fast_exp(qm - mx) : 0.0f);
    for (uint32_t r = 0; r < k; ++r) {
        const uint64_t w = wave_max_u64(key[0]);
        if (w != 0 && w == key[0]) {                                     // keys are distinct (token ids are): one owner, whose next key moves up
#pragma unroll
            for (uint32_t j = 0; j < 7u; ++j) key[j] = key[j + 1u];
            key[7] = 0;
        }
        if (lane == 0u) wkey[wave][r] = w;
    }
    if (lane == 0u) { wm[wave] = mx; wsum[wave] = total; }
    __syncthreads();
    if (wave != 0u) return;
    float M, S;
    float* conf = out_conf + static_cast<uint64_t>(b) * k;
    int32_t* tok = out_tok + static_cast<uint64_t>(b) * k;
    tk_merge(lane < 4u ? wm[lane] : -INFINITY, lane < 4u ? wsum[lane] : 0.0f, lane < 4u * k ? wkey[lane / k][lane % k] : 0, k, M, S,
             [&](uint32_t r, uint64_t w) {
                 if (lane == 0u) {
                     tok[r] = w ? static_cast<int32_t>(0xFFFFFFFFu - static_cast<uint32_t>(w)) : -1;
                     conf[r] = w ? pred_fast_exp(tk_value(w) - M) / S : 0.0f;
                 }
             });
}

// Records move to new places (compaction into packed extents and back): one wave per page copies the record's bytes, rounded up
// to 16 (k_compress zero-pads a record's last 16-byte piece), then re-points the page's table entry.
__global__ __launch_bounds__(256) void k_repack(PageEntry* __restrict__ entries, const uint64_t* __restrict__ new_addr, uint64_t n)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t p = (static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x) >> 6;
    if (p >= n) return;
    const PageEntry e = entries[p];
    const uint8_t* src = reinterpret_cast<const uint8_t*>(e.pool_addr);
    uint8_t* dst = reinterpret_cast<uint8_t*>(new_addr[p]);
    const uint32_t bytes = (e.rec_bytes + 15u) & ~15u;
    for (uint32_t b = 16u * lane; b < bytes; b += 1024u) st16(dst + b, ld16(src + b));
    if (lane == 0u) entries[p].pool_addr = new_addr[p];
}

__global__ void k_retarget_entries(PageEntry* e, uint64_t n, uint64_t base, uint64_t stride, uint64_t rec0)
{
    const uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (stride == kPlanarMx4) { e[i].pool_addr = base + mx4_nib_off(rec0 + i); e[i].scale = __uint_as_float(mx4_code_delta(rec0 + i)); }
    else e[i].pool_addr = base + i * stride;
}

int g_cus = 0;
int num_cus()
{
    if (g_cus == 0) {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) == hipSuccess &&
            hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0)
            g_cus = cus;
        else
            g_cus = 256;
    }
    return g_cus;
}
// Launch shape: one block per wave up to per_cu workgroups per CU (short-lived waves even out the tail); beyond that
// every wave takes the same number of consecutive blocks (*per_wave) so that no round runs half empty.
// (A capped grid-stride loop: 655 360 blocks over 65 536 workgroups = 2.5 rounds, last one half empty: 0.73 of HBM
// peak; balanced 0.77-0.80.  Measured with profiles/tools/footprint.py, profiles/r02_footprint.md.)
uint32_t codec_grid(uint64_t n, uint64_t* per_wave)
{
    const int per_cu = tuning().wgs_per_cu > 0 ? tuning().wgs_per_cu : 256;
    const uint64_t want = (n + kWaves - 1) / kWaves;
    const uint64_t cap = static_cast<uint64_t>(num_cus()) * per_cu * (4 / kWaves);
    *per_wave = 1;
    if (want <= cap) return static_cast<uint32_t>(want ? want : 1);
    const uint64_t rounds = (want + cap - 1) / cap;
    *per_wave = rounds;
    return static_cast<uint32_t>((want + rounds - 1) / rounds);
}
// Multi-round launches: a wave's blocks are one grid apart (round r covers blocks [r*G, (r+1)*G) like a launch of its
// own).  Measured against `per_wave` consecutive blocks per wave on the same device (SPECKV_ROUNDS=consecutive):
// 0.775 / 0.772 / 0.770 of HBM peak vs 0.753 / 0.69 / 0.657 at 4 / 5 / 20 GiB of pool + destination.
bool round_strided()
{
    return tuning().rounds_consecutive == 0;
}

template <int SCHEME, int MODE>
hipError_t launch_dec2(const CodecArgs& a_in, hipStream_t s)
{
    CodecArgs a = a_in;
    const uint32_t grid = codec_grid(a.n, &a.per_wave);
    a.wave_step = (round_strided() && a.per_wave > 1) ? static_cast<uint64_t>(grid) * kWaves : 0;
    const int ext = a.host_words ? 3 : (a.alloc_list || a.ring_owner) ? 2 : a.stripe_n ? 1 : 0;
    if (ext == 2 && a.stripe_n) return hipErrorInvalidValue;
    if (a.ring_owner && a.alloc_list) return hipErrorInvalidValue;      // the ring form reads one allocation's row (load_desc_ring)
    if (a.done_flag && (ext != 2 || a.per_wave > 1 || a.n_dev || !a.done_count)) return hipErrorInvalidValue;
    if (ext == 3 && (a.out_f32 || a.alloc_list || a.ring_owner || a.stripe_n || !a.seq0_dev || !a.data_list)) return hipErrorInvalidValue;
#define SPECKV_LAUNCH_DEC(F32, EXT) \
    hipLaunchKernelGGL((k_fetch_decompress<SCHEME, MODE, F32, EXT>), dim3(grid), dim3(kThreads), 0, s, a)
    if (SCHEME == kInt8DeltaRle && a.structured_hint && ext <= 1) {      // data known to compress: the FLAT instantiation (plain and copy-engine-staged forms)
        if (a.out_f32 && ext == 0) hipLaunchKernelGGL((k_fetch_decompress_flat_f32<MODE>), dim3(grid), dim3(kThreads), 0, s, a);
        else if (a.out_f32)        hipLaunchKernelGGL((k_fetch_decompress_flat_f32_staged<MODE>), dim3(grid), dim3(kThreads), 0, s, a);
        else if (ext == 0)         hipLaunchKernelGGL((k_fetch_decompress_flat<MODE>), dim3(grid), dim3(kThreads), 0, s, a);
        else                       hipLaunchKernelGGL((k_fetch_decompress_flat_staged<MODE>), dim3(grid), dim3(kThreads), 0, s, a);
        return hipGetLastError();
    }
    if (a.out_f32) { if (ext == 2) SPECKV_LAUNCH_DEC(true, 2); else if (ext == 1) SPECKV_LAUNCH_DEC(true, 1); else SPECKV_LAUNCH_DEC(true, 0); }
    else           { if (ext == 3) SPECKV_LAUNCH_DEC(false, 3); else if (ext == 2) SPECKV_LAUNCH_DEC(false, 2); else if (ext == 1) SPECKV_LAUNCH_DEC(false, 1); else SPECKV_LAUNCH_DEC(false, 0); }
#undef SPECKV_LAUNCH_DEC
    return hipGetLastError();
}
template <int SCHEME>
hipError_t launch_dec1(const CodecArgs& a, hipStream_t s)
{
    return a.quant_mode == kIntent ? launch_dec2<SCHEME, kIntent>(a, s) : launch_dec2<SCHEME, kRefExact>(a, s);
}
template <int SCHEME>
hipError_t launch_enc1(const CodecArgs& a_in, hipStream_t s)
{
    CodecArgs a = a_in;
    const uint32_t grid = codec_grid(a.n, &a.per_wave);
    a.wave_step = (round_strided() && a.per_wave > 1) ? static_cast<uint64_t>(grid) * kWaves : 0;
    if (a.quant_mode == kIntent) hipLaunchKernelGGL((k_compress<SCHEME, kIntent>), dim3(grid), dim3(kThreads), 0, s, a);
    else                         hipLaunchKernelGGL((k_compress<SCHEME, kRefExact>), dim3(grid), dim3(kThreads), 0, s, a);
    return hipGetLastError();
}

} // namespace

hipError_t launch_decompress(const CodecArgs& a, hipStream_t s)
{
    if (a.n == 0) return hipSuccess;
    switch (a.scheme) {
    case kFp16: return launch_dec1<kFp16>(a, s);
    case kInt8: return launch_dec1<kInt8>(a, s);
    case kInt8DeltaRle: return launch_dec1<kInt8DeltaRle>(a, s);
    case kInt4G32: return launch_dec2<kInt4G32, kRefExact>(a, s);      // the quantiser mode does not apply
    case kFp8E4m3: return launch_dec2<kFp8E4m3, kRefExact>(a, s);
    case kMxFp4: return launch_dec2<kMxFp4, kRefExact>(a, s);
    default: return hipErrorInvalidValue;
    }
}

hipError_t launch_compress(const CodecArgs& a, hipStream_t s)
{
    if (a.n == 0) return hipSuccess;
    switch (a.scheme) {
    case kFp16: return launch_enc1<kFp16>(a, s);
    case kInt8: return launch_enc1<kInt8>(a, s);
    case kInt8DeltaRle: return launch_enc1<kInt8DeltaRle>(a, s);
    case kInt4G32: {
        CodecArgs b = a;
        const uint32_t grid = codec_grid(b.n, &b.per_wave);
        b.wave_step = (round_strided() && b.per_wave > 1) ? static_cast<uint64_t>(grid) * kWaves : 0;
        hipLaunchKernelGGL((k_compress<kInt4G32, kRefExact>), dim3(grid), dim3(kThreads), 0, s, b);
        return hipGetLastError();
    }
    case kFp8E4m3: {
        CodecArgs b = a;
        const uint32_t grid = codec_grid(b.n, &b.per_wave);
        b.wave_step = (round_strided() && b.per_wave > 1) ? static_cast<uint64_t>(grid) * kWaves : 0;
        hipLaunchKernelGGL((k_compress<kFp8E4m3, kRefExact>), dim3(grid), dim3(kThreads), 0, s, b);
        return hipGetLastError();
    }
    case kMxFp4: {
        CodecArgs b = a;
        const uint32_t grid = codec_grid(b.n, &b.per_wave);
        b.wave_step = (round_strided() && b.per_wave > 1) ? static_cast<uint64_t>(grid) * kWaves : 0;
        hipLaunchKernelGGL((k_compress<kMxFp4, kRefExact>), dim3(grid), dim3(kThreads), 0, s, b);
        return hipGetLastError();
    }
    default: return hipErrorInvalidValue;
    }
}

hipError_t launch_prefetch_lookup(const Layout& lay, uint32_t n, const uint32_t* d_req,
                                  const uint32_t* d_layer, const uint32_t* d_pos,
                                  const uint32_t* d_k, const uint32_t* d_flags, uint32_t* d_out,
                                  uint32_t cap, uint32_t* d_count, uint32_t* d_scratch,
                                  hipStream_t s)
{
    if (n == 0) return hipMemsetAsync(d_count, 0, sizeof(uint32_t), s);
    const uint32_t n_w = (n + 1u) / 2u;                 // 2 requests per wave
    uint32_t* tot = d_scratch;
    uint32_t* base = d_scratch + n_w;
    const uint32_t grid = (n_w * 64u + 255u) / 256u;
    hipLaunchKernelGGL((k_prefetch_lookup<false>), dim3(grid), dim3(256), 0, s, lay, n, d_req, d_layer,
                       d_pos, d_k, d_flags, tot, static_cast<const uint32_t*>(nullptr),
                       static_cast<uint32_t*>(nullptr), cap);
    hipLaunchKernelGGL(k_scan_totals, dim3(1), dim3(1024), 0, s, tot, base, n_w, d_count, cap);
    hipLaunchKernelGGL((k_prefetch_lookup<true>), dim3(grid), dim3(256), 0, s, lay, n, d_req, d_layer,
                       d_pos, d_k, d_flags, static_cast<uint32_t*>(nullptr), base, d_out, cap);
    return hipGetLastError();
}

hipError_t launch_flush_pipeline(const FlushArgs& a, hipStream_t s)
{
    if (a.n == 0 || a.W == 0) return hipErrorInvalidValue;
    const uint64_t total = static_cast<uint64_t>(a.n) * 32u * a.W;
    if (total >= (1ull << 24)) return hipErrorInvalidValue;        // the dedupe key carries 24 index bits
    // (test switches, tuning.hpp: the four-launch pipeline for every size / another limit for the one-workgroup form)
    const bool no_small = tuning().flush_no_small != 0;
    const uint64_t small_words = tuning().flush_small_words > 0 ? std::min<uint64_t>(static_cast<uint64_t>(tuning().flush_small_words), kFlushSmallWords) : kFlushSmallDefault;
    if (total <= small_words && !no_small) {
        hipLaunchKernelGGL(k_flush_small, dim3(1), dim3(1024), 0, s, a);
        return hipGetLastError();
    }
    const uint32_t g_lane = static_cast<uint32_t>((static_cast<uint64_t>(a.n) * 32u + 255u) / 256u);
    const uint32_t g_word = static_cast<uint32_t>((total + 255u) / 256u);
    hipLaunchKernelGGL(k_flush_candidates, dim3(g_lane), dim3(256), 0, s, a);
    hipLaunchKernelGGL((k_flush_mark<false>), dim3(g_word), dim3(256), 0, s, a, static_cast<const FlushResult*>(nullptr));
    hipLaunchKernelGGL(k_flush_assign, dim3(1), dim3(1024), 0, s, a);
    hipLaunchKernelGGL((k_flush_mark<true>), dim3(g_word), dim3(256), 0, s, a, static_cast<const FlushResult*>(a.result_dev));
    return hipGetLastError();
}

hipError_t launch_verify(uint32_t n, uint32_t k, const int32_t* d_actual,
                         const int32_t* d_predicted, uint8_t* d_hit, uint32_t* d_hit_count,
                         hipStream_t s)
{
    hipError_t e = hipMemsetAsync(d_hit_count, 0, sizeof(uint32_t), s);
    if (e != hipSuccess || n == 0) return e;
    if (k == 0 || k > 64u) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_verify, dim3((n + 255u) / 256u), dim3(256), 0, s, n, k, d_actual, d_predicted, d_hit,
                       d_hit_count);
    return hipGetLastError();
}

hipError_t launch_quantize_q_e4m3(const void* d_q_f16, uint32_t heads, uint32_t g, uint32_t d,
                                  uint8_t* d_q8, float* d_qs, hipStream_t s)
{
    if (heads == 0 || g == 0 || g > 16u || d != 128u) return hipErrorInvalidValue;
    // `heads` may be layers*heads: rows are independent
    hipLaunchKernelGGL(k_quantize_q_e4m3, dim3(heads * 16u), dim3(64), 0, s,
                       static_cast<const uint16_t*>(d_q_f16), g, d, d_q8, d_qs);
    return hipGetLastError();
}

hipError_t launch_qk_scores_fp8(const PageEntry* d_entries, uint64_t first_page, uint64_t layer_page_stride,
                                uint32_t n_layers, uint32_t n_pages, uint32_t heads, uint32_t g,
                                const uint8_t* d_q8, const float* d_qs, float* d_out, hipStream_t s)
{
    if (n_pages == 0 || n_layers == 0) return hipSuccess;
    const uint32_t waves = (n_pages + 7u) / 8u;
    hipLaunchKernelGGL(k_qk_scores_fp8, dim3((waves + 1u) / 2u, n_layers), dim3(128), 0, s, d_entries, first_page,
                       layer_page_stride, n_pages, heads, g, d_q8, d_qs, d_out);
    return hipGetLastError();
}

hipError_t launch_repack(PageEntry* d_entries, const uint64_t* d_new_addr, uint64_t n, hipStream_t s)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_repack, dim3(static_cast<uint32_t>((n + 3u) / 4u)), dim3(256), 0, s, d_entries, d_new_addr, n);
    return hipGetLastError();
}

hipError_t launch_retarget_entries(PageEntry* d_entries, uint64_t n, uint64_t base, uint64_t stride, hipStream_t s, uint64_t rec0)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_retarget_entries, dim3(static_cast<uint32_t>((n + 255u) / 256u)), dim3(256), 0, s,
                       d_entries, n, base, stride, rec0);
    return hipGetLastError();
}

size_t arranged_wout_bytes(uint32_t vocab) { return static_cast<size_t>(logits_tiles_padded(vocab)) * 32u * kPredHidden * sizeof(float); }
hipError_t launch_arrange_wout(const float* d_src, float* d_dst, uint32_t vocab, hipStream_t s)
{
    const uint32_t tiles = logits_tiles_padded(vocab);
    hipLaunchKernelGGL(k_arrange_wout, dim3(tiles / 4u), dim3(256), 0, s, d_src, reinterpret_cast<float4*>(d_dst), vocab);
    return hipGetLastError();
}

hipError_t launch_predict(uint32_t n, const int32_t* d_hist, const float* d_emb, const float* d_wout, uint32_t vocab,
                          uint32_t layers, uint32_t k, float* d_hid, float* d_logits, void* d_ws, int32_t* d_tok, float* d_conf,
                          hipStream_t s, const LstmParams* lstm)
{
    if (n == 0) return hipSuccess;
    if (k == 0 || k > 8u || vocab < k) return hipErrorInvalidValue;
    const uint32_t small_parts = logits_tiles_padded(vocab) / 4u;       // workgroups of 128 rows
    if (n <= kPredictSmallN && small_parts <= kPredictSmallMaxParts && !tuning().predict_batch_path) {
        // a handful of requests: two launches (three with the real cell), no logits in memory (k_predict_small)
        const bool real = lstm && lstm->layers;
        if (real) {
            if (lstm->layers > 4u) return hipErrorInvalidValue;
            LstmWeights w{};
            w.layers = lstm->layers;
            for (uint32_t l = 0; l < lstm->layers; ++l) { w.w_ih_t[l] = lstm->w_ih_t[l]; w.w_hh_t[l] = lstm->w_hh_t[l]; w.bias[l] = lstm->bias[l]; }
            hipLaunchKernelGGL(k_lstm_cell, dim3(n), dim3(512), 0, s, d_hist, n, d_emb, vocab, w, d_hid);
        }
        hipLaunchKernelGGL(k_predict_small, dim3(small_parts), dim3(256), 0, s, d_hist, n, d_emb, real ? d_hid : nullptr, layers, d_wout,
                           lstm ? lstm->out_bias : nullptr, vocab, k, static_cast<uint8_t*>(d_ws));
        hipLaunchKernelGGL(k_predict_small_merge, dim3(n), dim3(256), 0, s, static_cast<const uint8_t*>(d_ws), k, small_parts, d_tok, d_conf);
        return hipGetLastError();
    }
    if (lstm && lstm->layers) {                       // the real cell (speckv_ext_predictor_load_lstm)
        if (lstm->layers > 4u) return hipErrorInvalidValue;
        LstmWeights w{};
        w.layers = lstm->layers;
        for (uint32_t l = 0; l < lstm->layers; ++l) { w.w_ih_t[l] = lstm->w_ih_t[l]; w.w_hh_t[l] = lstm->w_hh_t[l]; w.bias[l] = lstm->bias[l]; }
        hipLaunchKernelGGL(k_lstm_cell, dim3(n), dim3(512), 0, s, d_hist, n, d_emb, vocab, w, d_hid);
    } else {
        hipLaunchKernelGGL(k_lstm_hidden, dim3(n), dim3(64), 0, s, d_hist, n, d_emb, vocab, layers, d_hid);
    }
    const uint32_t waves = (vocab + 31u) / 32u;              // 32 output rows per wave (k_lstm_logits)
    hipLaunchKernelGGL(k_lstm_logits, dim3((waves + 3u) / 4u, (n + kLogitsChunk - 1u) / kLogitsChunk), dim3(256), 0, s, d_hid, n, d_wout, lstm ? lstm->out_bias : nullptr, vocab, d_logits);
    const uint32_t parts = predict_topk_parts(vocab);
    if (parts) {
        hipLaunchKernelGGL(k_softmax_topk_small, dim3(n, parts), dim3(kTkThreads), 0, s, d_logits, vocab, k, static_cast<uint8_t*>(d_ws));
        hipLaunchKernelGGL(k_softmax_topk_merge, dim3((n + 3u) / 4u), dim3(256), 0, s, static_cast<const uint8_t*>(d_ws), n, k, parts, d_tok, d_conf);
    }
    else                           hipLaunchKernelGGL(k_softmax_topk, dim3(n), dim3(1024), 0, s, d_logits, vocab, k, d_tok, d_conf);
    return hipGetLastError();
}

hipError_t launch_apply_updates(const DevAlloc* d_tab, const MirrorUpdate* d_updates, uint32_t n, hipStream_t s)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_apply_updates, dim3((n + 255u) / 256u), dim3(256), 0, s, d_tab, d_updates, n);
    return hipGetLastError();
}

// 16-byte-per-lane copy (pinned host memory -> device): the request columns of a flush.  As a kernel on the flush's own
// stream it needs no copy engine and no cross-engine dependency in front of the first flush kernel.
__global__ __launch_bounds__(256) void k_copy16(const uint4* __restrict__ src, uint4* __restrict__ dst, uint32_t n16)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n16) dst[i] = src[i];
}
hipError_t launch_copy16(const void* src, void* dst, size_t bytes, hipStream_t s)
{
    const uint32_t n16 = static_cast<uint32_t>((bytes + 15u) / 16u);
    if (n16 == 0) return hipSuccess;
    hipLaunchKernelGGL(k_copy16, dim3((n16 + 255u) / 256u), dim3(256), 0, s, static_cast<const uint4*>(src), static_cast<uint4*>(dst), n16);
    return hipGetLastError();
}

hipError_t launch_init_entries(PageEntry* d_entries, uint64_t n, uint64_t base, uint64_t stride,
                               hipStream_t s, uint64_t rec0)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_init_entries, dim3(static_cast<uint32_t>((n + 255u) / 256u)), dim3(256), 0, s,
                       d_entries, n, base, stride, rec0);
    return hipGetLastError();
}

} // namespace speckv
