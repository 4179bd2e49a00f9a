// cxl-speckv_amd/csrc/ring_rule.hpp -- the L2 prefetch ring's arithmetic, shared by the host (Engine::take_l2_run,
// Engine::l2_live), the device (k_flush_assign) and the CPU property tests (tests/csrc/host_rules_test.cpp): the two
// sides must agree exactly, because the host derives a page's residency from the sequence number a kernel stored for it.
//
// The ring has n slots.  A sequence number counts every slot the hand has passed -- slots skipped at the end of a lap
// included -- so slot = seq % n.  A run of m slots never wraps: if it does not fit before the end of the lap, the rest of
// the lap is skipped.  (Reference: the FIFO prefetch buffer of cxl_memory_manager.cpp:196-221 / prefetch_core.v:210-216,
// which moves nothing; here slots hold decompressed pages.)
#pragma once
#include <cstdint>
#ifndef __host__
#define __host__
#define __device__
#endif

namespace speckv {

struct RingRun { uint32_t seq; uint32_t slot; uint32_t next; };      // first sequence number, its slot, the hand afterwards

// take m <= n slots at hand position `seq`
__host__ __device__ inline RingRun ring_take(uint32_t seq, uint32_t m, uint32_t n)
{
    uint32_t slot = seq % n;
    if (slot + m > n) { seq += n - slot; slot = 0; }
    return RingRun{seq, slot, seq + m};
}

// A page stored under sequence number q is intact until the hand has passed q + n: with the hand at `seq`,
// live <=> 0 < seq - q <= n  (32-bit differences: the engine renumbers long before a wrap could alias).
__host__ __device__ inline bool ring_live(uint32_t seq, uint32_t q, uint32_t n)
{
    return static_cast<uint32_t>(seq - q) - 1u < n;
}

// n_tiles divided evenly over at most `want` splits: {tiles per split, number of splits}, every split non-empty
struct EvenSplit { uint32_t tiles_per_split; uint32_t n_splits; };
__host__ __device__ inline EvenSplit even_split(uint32_t n_tiles, uint32_t want)
{
    if (n_tiles == 0) return EvenSplit{want ? want : 1u, 0u};
    if (want == 0) want = 1;
    if (want > n_tiles) want = n_tiles;
    const uint32_t tps = (n_tiles + want - 1u) / want;
    return EvenSplit{tps, (n_tiles + tps - 1u) / tps};
}

} // namespace speckv
