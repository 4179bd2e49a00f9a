// cxl-speckv_amd/csrc/slab_pool.hpp -- HIP slab allocator for the KV pool.
//
// Replaces the bump allocators of the reference's CXLMemoryManager
// (src/cxl_memory/cxl_memory_manager.cpp:28-80: next_physical_addr_l{1,2,3}_ += bytes,
// never reused) and the FPGA-side "HBM page index" space of the host allocator
// (host/src/speckv_allocator.cpp:25).  One SlabPool per pool GPU: HBM is taken
// from the HIP runtime in large slabs (default 1 GiB, sized for 288 GB parts so
// a 70B-shaped sequence is two or three slabs) and handed out as contiguous
// page runs; freed runs coalesce and are reused first-fit.
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>
#include <map>
#include <vector>

namespace speckv {

class SlabPool {
public:
    SlabPool(int device, size_t slab_bytes, size_t capacity_bytes)
        : device_(device), slab_bytes_(slab_bytes), capacity_(capacity_bytes) {}
    ~SlabPool() { release(); }
    SlabPool(const SlabPool&) = delete;
    SlabPool& operator=(const SlabPool&) = delete;

    // Contiguous run of `bytes` (rounded up to whole 128-byte cache lines, starting on one).  nullptr = out of
    // memory.  free() takes any line-granular sub-range of what alloc returned.
    void* alloc(size_t bytes);
    // Fragmented fallback: the largest free run that is a multiple of `granule`
    // and at most `want` bytes (grows by one slab when nothing is free).
    // *got receives its size; nullptr = out of memory.
    void* alloc_up_to(size_t want, size_t granule, size_t* got);
    void free(void* p, size_t bytes);
    void release();                      // hipFree every slab

    int device() const { return device_; }
    size_t reserved_bytes() const { return reserved_; }
    size_t used_bytes() const { return used_; }
    size_t n_slabs() const { return slabs_.size(); }
    size_t n_free_runs() const { return free_.size(); }

private:
    struct Slab { uint8_t* base; size_t bytes; };
    int device_;
    size_t slab_bytes_;
    size_t capacity_;
    size_t reserved_ = 0, used_ = 0;
    std::vector<Slab> slabs_;
    std::map<uintptr_t, size_t> free_;   // address -> length, coalesced

    bool grow(size_t min_bytes);
    void insert_free(uintptr_t addr, size_t len);
    bool same_slab(uintptr_t a, uintptr_t b) const;
};

} // namespace speckv
