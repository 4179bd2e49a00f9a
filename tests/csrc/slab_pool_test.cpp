// tests/csrc/slab_pool_test.cpp -- TEST-ONLY C wrapper around SlabPool with host-heap slabs
// (compiled with -DSPECKV_SLAB_HOST_BACKING into tests/_build/libslabpool_test.so).
#include "../../cxl-speckv_amd/csrc/slab_pool.hpp"

extern "C" {
void* slabtest_new(size_t slab_bytes, size_t cap_bytes) { return new speckv::SlabPool(0, slab_bytes, cap_bytes); }
void slabtest_delete(void* p) { delete static_cast<speckv::SlabPool*>(p); }
void* slabtest_alloc(void* p, size_t bytes) { return static_cast<speckv::SlabPool*>(p)->alloc(bytes); }
void* slabtest_alloc_up_to(void* p, size_t want, size_t granule, size_t* got) { return static_cast<speckv::SlabPool*>(p)->alloc_up_to(want, granule, got); }
void slabtest_free(void* p, void* addr, size_t bytes) { static_cast<speckv::SlabPool*>(p)->free(addr, bytes); }
size_t slabtest_used(void* p) { return static_cast<speckv::SlabPool*>(p)->used_bytes(); }
size_t slabtest_reserved(void* p) { return static_cast<speckv::SlabPool*>(p)->reserved_bytes(); }
size_t slabtest_free_runs(void* p) { return static_cast<speckv::SlabPool*>(p)->n_free_runs(); }
}
