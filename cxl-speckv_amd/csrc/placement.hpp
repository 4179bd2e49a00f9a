// cxl-speckv_amd/csrc/placement.hpp -- where the records of an allocation go.
// One rule, used by Engine::alloc, the copy-engine fetch (engine_io.cpp) and exported as speckv_ext_placement so the
// multi-rank tests can cross-check it without a GPU: page p of an allocation striped over D pool GPUs is record
// p / D of the run on pool p % D.  A logical page range [first, first+n) therefore maps, on every pool, to ONE
// contiguous run of records -- which is what lets the copy engine move it with one hipMemcpyPeerAsync per pool.
#pragma once
#include <cstdint>

namespace speckv {

struct Placement { uint32_t pool; uint64_t record; };

inline Placement place_page(uint64_t page, uint32_t n_pool)
{
    const uint32_t d = n_pool ? n_pool : 1u;
    return Placement{static_cast<uint32_t>(page % d), page / d};
}
// records pool k holds of an n_pages allocation: pages k, k+D, k+2D, ...
inline uint64_t shard_pages(uint64_t n_pages, uint32_t n_pool, uint32_t k)
{
    const uint64_t d = n_pool ? n_pool : 1u;
    return k >= d ? 0 : (n_pages + d - 1 - k) / d;
}
// the records of pool k inside the logical page range [first, first + n): [rec_begin, rec_begin + count)
inline void shard_range(uint64_t first, uint64_t n, uint32_t n_pool, uint32_t k, uint64_t* rec_begin, uint64_t* count)
{
    const uint64_t d = n_pool ? n_pool : 1u;
    // first page >= `first` with page % d == k
    const uint64_t p0 = first + ((k + d - first % d) % d);
    if (n == 0 || p0 >= first + n) { *rec_begin = p0 / d; *count = 0; return; }
    *rec_begin = p0 / d;
    *count = (first + n - 1 - p0) / d + 1;
}

} // namespace speckv
