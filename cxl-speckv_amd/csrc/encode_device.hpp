// cxl-speckv_amd/csrc/encode_device.hpp -- the instruction-level helpers of the RLE encoder's fast path (device code only):
// packed-fp32 quantisation of 8 elements, SDWA byte arithmetic, run-start predicates kept in SGPR pairs, the EXEC-predicated
// pair scatter.  Used by k_compress (kernels.hip) and by the tile passes of the whole-tensor codec (tensor_codec.hip).
#pragma once
#include "codec_device.hpp"

namespace speckv {
namespace {

typedef __attribute__((address_space(3))) uint8_t lds_u8;
typedef float f32x2 __attribute__((ext_vector_type(2)));

// Wave LDS layout of the encoders (bytes): [0,16) lead (the count byte "before the first pair" lands here), [16, 16+4096) pair buffer.
constexpr uint32_t kEncPairOff = 16, kEncWaveBytes = 4128;
constexpr uint32_t kEncFail = 0xFFFFFFFFu;

__device__ __forceinline__ uint32_t lds_addr_of(const void* p)
{
    return static_cast<uint32_t>(reinterpret_cast<uintptr_t>((lds_u8*)p));
}

// The encoder is VALU-bound (rocprofv3, 131072 N(0,1) blocks: VALU issue 93 % busy at 923 instructions per block-wave,
// profiles/r02_compress_pmc.json), so this path is written for instruction count:
//   * quantisation in packed fp32 (v_pk_mul_f32 / v_pk_fma_f32 / v_pk_add_f32: two elements per instruction); the byte
//     is never masked out -- every consumer selects byte 0 (SDWA);
//   * delta = (q.b0 - prev.b0) as one SDWA subtract that writes a zero-padded byte;
//   * the eight "starts a run" predicates of a lane stay in SGPR pairs (v_cmp results); a carry chain
//     (v_addc: mask = 2*mask + predicate) turns them into an 8-bit mask -> run count by v_bcnt, last start by v_ffbl;
//   * the scatter is predicated by EXEC (scalar) instead of selecting a dummy address per element, the run index
//     advances by v_addc of the same predicate, count = k - (last start relative to the lane) with k an inline constant.
// 8 quantised elements of one lane (low byte of each q[k] is the int8; upper bits are the rest of the int32)
template <int MODE>
__device__ __forceinline__ void quantize8(const uint4 raw, float scale, float rcp, uint32_t (&q)[8])
{
    const uint32_t w[4] = {raw.x, raw.y, raw.z, raw.w};
    const f32x2 ss = {scale, scale}, rr = {rcp, rcp};
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        f32x2 x;
        x.x = half_bits_to_float(w[t] & 0xFFFFu);
        x.y = half_bits_to_float(w[t] >> 16);
        // x / scale through the block's reciprocal (div_by_scale: exact for every operand the codec sees)
        const f32x2 q0 = x * rr;
        const f32x2 e = __builtin_elementwise_fma(-q0, ss, x);
        f32x2 y = __builtin_elementwise_fma(e, rr, q0);
        if (MODE == kRefExact) { const f32x2 k127 = {127.0f, 127.0f}; y = y * k127; }      // cache_engine.cpp:190-191
        f32x2 h;
        h.x = __builtin_copysignf(0.5f, y.x);
        h.y = __builtin_copysignf(0.5f, y.y);
        const f32x2 r = y + h;                              // round half away from zero = truncate(y + copysign(0.5, y))
        int i0 = static_cast<int>(r.x), i1 = static_cast<int>(r.y);
        if (MODE != kRefExact) { i0 = min(max(i0, -127), 127); i1 = min(max(i1, -127), 127); }
        q[2 * t] = static_cast<uint32_t>(i0);
        q[2 * t + 1] = static_cast<uint32_t>(i1);
    }
}
// (a.b0 - b.b0) & 0xFF in one instruction
__device__ __forceinline__ uint32_t sub_bytes(uint32_t a, uint32_t b)
{
    uint32_t r;
    asm("v_sub_u32_sdwa %0, %1, %2 dst_sel:BYTE_0 dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:BYTE_0" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// v = 2*v + predicate / v += predicate, the predicate being a v_cmp result (lane mask in an SGPR pair)
__device__ __forceinline__ void shift_in(uint32_t& v, unsigned long long pred)
{
    asm("v_addc_co_u32 %0, vcc, %0, %0, %1" : "+v"(v) : "s"(pred) : "vcc");
}
__device__ __forceinline__ void add_pred(uint32_t& v, unsigned long long pred)
{
    asm("v_addc_co_u32 %0, vcc, 0, %0, %1" : "+v"(v) : "s"(pred) : "vcc");
}
// the two byte stores of a run start, executed by the lanes of `pred` only (EXEC is narrowed and restored here: no
// branch, no dummy address)
__device__ __forceinline__ void store_pair_if(unsigned long long pred, uint32_t addr, uint32_t count_prev, uint32_t value)
{
    unsigned long long saved;
    asm volatile("s_and_saveexec_b64 %0, %1\n\tds_write_b8 %2, %3\n\tds_write_b8 %2, %4 offset:1\n\ts_mov_b64 exec, %0"
                 : "=&s"(saved) : "s"(pred), "v"(addr), "v"(count_prev), "v"(value) : "memory");
}
__device__ __forceinline__ uint32_t lshl1_add(uint32_t a, uint32_t b)      // 2*a + b
{
    uint32_t r;
    asm("v_lshl_add_u32 %0, %1, 1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// max|x| of a block and "every element is finite" from the fp16 bit patterns: the largest |bits| of the 32 elements
// a lane holds (integer order = magnitude order for finite values), two elements per instruction.
typedef uint16_t u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t absmax_bits(const uint4 (&raw)[4])
{
    u16x2 m = {0, 0};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const uint32_t w[4] = {raw[j].x, raw[j].y, raw[j].z, raw[j].w};
#pragma unroll
        for (int t = 0; t < 4; ++t)
            m = __builtin_elementwise_max(m, __builtin_bit_cast(u16x2, w[t] & 0x7FFF7FFFu));
    }
    const uint32_t lanemax = m.x > m.y ? m.x : m.y;
    return lane63(wave_incl_max(lanemax));
}

} // namespace
} // namespace speckv
