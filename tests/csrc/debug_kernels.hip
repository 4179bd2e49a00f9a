// tests/csrc/debug_kernels.hip -- TEST-ONLY self-check kernels (built into tests/_build/libspeckv_debug.so, never
// into libcxlspeckv.so): exhaustive exactness check of the codec's reciprocal-based divide and cheap rounding, and a
// hardware self-test of the wave primitives, run against the very same device functions the product kernels use
// (cxl-speckv_amd/csrc/codec_device.hpp).
#include "../../cxl-speckv_amd/csrc/codec_device.hpp"

namespace speckv {
namespace {

// exhaustive check of div_by_scale: thread = one divisor (fp16 magnitude bits mbits in
// [1, 0x7BFF] divided by `den`), loop over all 65536 fp16 dividends; counts mismatches
// against the IEEE divide, and (second counter) mismatches of the REF_EXACT byte
// (roundf(x/s*127) & 0xFF) which is what the codec finally stores.
__global__ void k_debug_divcheck(float den, unsigned long long* counters)
{
    const uint32_t mbits = blockIdx.x * blockDim.x + threadIdx.x + 1u;
    if (mbits > 0x7BFFu) return;
    const float m = half_bits_to_float(mbits);
    // den > 0: s = m/den and |x| <= m (INT8 family: den 127, FP8: den 448).
    // den == 0: INT4 family: s = m is itself an fp16 value (the stored group scale) and |x| <= 7.5 m.
    const float s = den > 0.0f ? m / den : m;
    const float lim = den > 0.0f ? m : 7.5f * m;
    const float mul = den == 127.0f ? 127.0f : 1.0f;
    const float r = 1.0f / s;
    unsigned long long bad = 0, badq = 0, badr = 0;
    {                                                            // the encoders' short cuts, bit for bit against the IEEE divide
        unsigned long long badf = 0;
        if (den == 0.0f) {
            badf += (__float_as_uint(rcp_of_f16_value(m)) != __float_as_uint(1.0f / m)) ? 1ull : 0ull;
            badf += (__float_as_uint(div7_of_f16_value(m)) != __float_as_uint(m / 7.0f)) ? 1ull : 0ull;
        } else {
            const float fast = den == 127.0f ? div127_of_f16_value(m) : div448_of_f16_value(m);
            badf += (__float_as_uint(fast) != __float_as_uint(s)) ? 1ull : 0ull;
            badf += (__float_as_uint(rcp_of_scale(s)) != __float_as_uint(r)) ? 1ull : 0ull;
        }
        if (badf) atomicAdd(&counters[3], badf);
    }
    for (uint32_t xb = 0; xb < 65536u; ++xb) {
        if ((xb & 0x7C00u) == 0x7C00u) continue;                 // inf / nan dividends take the slow path
        const float x = half_bits_to_float(xb);
        if (fabsf(x) > lim) continue;                            // the block / group maximum bounds every |x|
        const float a = x / s, b = div_by_scale(x, s, r);
        bad += (__float_as_uint(a) != __float_as_uint(__builtin_copysignf(b, x))) ? 1ull : 0ull;
        badq += (static_cast<int>(roundf(a * mul)) != static_cast<int>(roundf(b * mul))) ? 1ull : 0ull;
        // candidate cheap rounding: truncate(y + copysign(0.5, y)) against roundf(y), for both products the codec rounds
        const float y1 = b * mul, y2 = b;
        badr += (static_cast<int>(roundf(y1)) != static_cast<int>(y1 + __builtin_copysignf(0.5f, y1))) ? 1ull : 0ull;
        badr += (static_cast<int>(roundf(y2)) != static_cast<int>(y2 + __builtin_copysignf(0.5f, y2))) ? 1ull : 0ull;
    }
    if (bad) atomicAdd(&counters[0], bad);
    if (badq) atomicAdd(&counters[1], badq);
    if (badr) atomicAdd(&counters[2], badr);
}

// div_f32_by_scale (codec_device.hpp; the tensor codec's fp32 sources) against the IEEE divide over EVERY fp32 bit pattern of x
// with |x| <= 127 s (what a tensor whose abs-max gave the scale s = mx / 127 can hold): counters[0] quotients that differ
// in their bits where |x / s| >= 2^-40, [1] stored bytes that differ in REF_EXACT (round(x / s * 127) & 0xFF), [2] in the
// saturating mode (clamp(round(x / s), -127, 127)), [3] elements checked.  One launch per divisor: 2^32 threads.
__global__ void k_debug_divcheck_f32(float s, unsigned long long* counters)
{
    unsigned long long bad = 0, bad_exact = 0, bad_sat = 0, n = 0;
    for (uint32_t sign = 0; sign < 2u; ++sign) {                        // (a grid holds fewer than 2^32 threads: both signs per thread)
    const uint32_t xb = (blockIdx.x * blockDim.x + threadIdx.x) | (sign << 31);
    const float x = __uint_as_float(xb);
    if ((xb & 0x7F800000u) != 0x7F800000u && fabsf(x) <= 127.0f * s * 1.0000002f) {
        const float r = 1.0f / s;
        const float a = x / s, b = div_f32_by_scale(x, s, r);
        n += 1;
        if (fabsf(a) >= 0x1p-40f) bad += (__float_as_uint(a) != __float_as_uint(b)) ? 1ull : 0ull;
        bad_exact += ((static_cast<uint32_t>(static_cast<int>(roundf(a * 127.0f))) & 0xFFu) != (static_cast<uint32_t>(round_to_int_f32(b * 127.0f)) & 0xFFu)) ? 1ull : 0ull;
        const int sa = static_cast<int>(fminf(fmaxf(roundf(a), -127.0f), 127.0f)), sb = min(max(round_to_int_f32(b), -127), 127);
        bad_sat += (sa != sb) ? 1ull : 0ull;
    }
    }
    // one atomic per wave
    for (int off = 32; off; off >>= 1) { bad += __shfl_xor(bad, off); bad_exact += __shfl_xor(bad_exact, off); bad_sat += __shfl_xor(bad_sat, off); n += __shfl_xor(n, off); }
    if ((threadIdx.x & 63u) == 0u) {
        if (bad) atomicAdd(&counters[0], bad);
        if (bad_exact) atomicAdd(&counters[1], bad_exact);
        if (bad_sat) atomicAdd(&counters[2], bad_sat);
        if (n) atomicAdd(&counters[3], n);
    }
}

// self-test of the wave primitives (tests/test_gpu_codec.py::test_wave_primitives)
__global__ void k_debug_dpp(const uint32_t* in, uint32_t* out)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t v = in[threadIdx.x];
    out[0 * 64 + lane] = wave_shr1(v, 0xABCDu);
    out[1 * 64 + lane] = wave_incl_add(v);
    out[2 * 64 + lane] = wave_incl_max(v);
    out[3 * 64 + lane] = lane63(v);
    // the dependent pattern used by the encoder: produce, shift, consume
    const uint32_t w = (v * 2654435761u) >> 24;
    const uint32_t prev = wave_shr1(w, 7u);
    out[4 * 64 + lane] = (w - prev) & 0xFFu;
}


} // namespace
} // namespace speckv

extern "C" {

int speckv_debug_divcheck(float den, unsigned long long* d_counters, void* stream)
{
    hipLaunchKernelGGL(speckv::k_debug_divcheck, dim3((0x7BFFu + 255u) / 256u), dim3(256), 0, static_cast<hipStream_t>(stream), den, d_counters);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

int speckv_debug_divcheck_f32(float s, unsigned long long* d_counters, void* stream)
{
    hipLaunchKernelGGL(speckv::k_debug_divcheck_f32, dim3(1u << 23), dim3(256), 0, static_cast<hipStream_t>(stream), s, d_counters);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

int speckv_debug_wave_primitives(const uint32_t* d_in, uint32_t* d_out, void* stream)
{
    hipLaunchKernelGGL(speckv::k_debug_dpp, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream), d_in, d_out);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

}
