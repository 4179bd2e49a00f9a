// oracle/coh_harness.cpp -- C wrappers over the reference's CoherenceManager, compiled TOGETHER with
// /root/reference/src/cxl_memory/coherence_manager.cpp (where it lies) into oracle/_ref/libspeckv_ref_coh.so.
// TEST INFRASTRUCTURE ONLY (see oracle/Makefile).
//
// The reference's C API (coherence_c_api.cpp) cannot be built: it constructs cxlspeckv::SpeckvDriver from the
// class declared in host/include/speckv_driver.h, which has no definition anywhere in the reference.  The
// manager itself only tests its driver pointer for null and never dereferences it, so the harness hands it a
// non-null pointer to a byte buffer with a no-op deleter; no reference code is replaced or stubbed.
#include "cxl_memory/coherence_manager.h"
#include <cstdint>
#include <cstring>
#include <memory>
#include <vector>

using cxlspeckv::CoherenceManager;
namespace cxlspeckv { class SpeckvDriver; }

static std::shared_ptr<cxlspeckv::SpeckvDriver> opaque_driver()
{
    alignas(64) static unsigned char token[64];
    return std::shared_ptr<cxlspeckv::SpeckvDriver>(reinterpret_cast<cxlspeckv::SpeckvDriver*>(token),
                                                    [](cxlspeckv::SpeckvDriver*) {});
}

extern "C" {
void* refcoh_new(size_t line, int with_driver)
{
    return new CoherenceManager(with_driver ? opaque_driver() : std::shared_ptr<cxlspeckv::SpeckvDriver>(), line);
}
void refcoh_delete(void* h) { delete static_cast<CoherenceManager*>(h); }
int refcoh_request_read(void* h, uint64_t a, void* out, size_t n) { return static_cast<CoherenceManager*>(h)->request_read(a, out, n); }
int refcoh_request_write(void* h, uint64_t a, const void* d, size_t n) { return static_cast<CoherenceManager*>(h)->request_write(a, d, n); }
int refcoh_invalidate(void* h, uint64_t a) { return static_cast<CoherenceManager*>(h)->invalidate(a); }
int refcoh_writeback(void* h, uint64_t a, const void* d, size_t n) { return static_cast<CoherenceManager*>(h)->writeback(a, d, n); }
int refcoh_flush_all(void* h) { return static_cast<CoherenceManager*>(h)->flush_all(); }
int refcoh_get_state(void* h, uint64_t a) { return static_cast<int>(static_cast<CoherenceManager*>(h)->get_state(a)); }
int refcoh_get_tier(void* h, uint64_t a) { return static_cast<int>(static_cast<CoherenceManager*>(h)->get_tier(a)); }
int refcoh_promote_to_l1(void* h, uint64_t a) { return static_cast<CoherenceManager*>(h)->promote_to_l1(a); }
int refcoh_demote_to_l3(void* h, uint64_t a) { return static_cast<CoherenceManager*>(h)->demote_to_l3(a); }
void refcoh_update_tier(void* h, uint64_t a, int tier) { static_cast<CoherenceManager*>(h)->update_tier(a, static_cast<CoherenceManager::MemoryTier>(tier)); }
int refcoh_batch_invalidate(void* h, const uint64_t* a, size_t n)
{
    return static_cast<CoherenceManager*>(h)->batch_invalidate(std::vector<uint64_t>(a, a + n));
}
void refcoh_get_statistics(void* h, uint64_t* out7)
{
    auto s = static_cast<CoherenceManager*>(h)->get_statistics();
    out7[0] = s.total_reads; out7[1] = s.total_writes; out7[2] = s.coherence_ops; out7[3] = s.invalidations_sent;
    out7[4] = s.writebacks_performed; out7[5] = s.directory_hits; out7[6] = s.directory_misses;
}
void refcoh_reset_statistics(void* h) { static_cast<CoherenceManager*>(h)->reset_statistics(); }
}
