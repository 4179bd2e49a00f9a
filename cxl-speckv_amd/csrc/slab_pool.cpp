// cxl-speckv_amd/csrc/slab_pool.cpp -- see slab_pool.hpp
#include "slab_pool.hpp"

#include <cstdlib>

namespace speckv {

namespace {
#ifdef SPECKV_SLAB_HOST_BACKING
// test build only (tests/csrc/slab_pool_test.cpp, never libcxlspeckv.so): slabs come from the host heap so the
// allocator's bookkeeping can be property-tested where there is no GPU
hipError_t slab_malloc(void** p, size_t n) { *p = aligned_alloc(4096, (n + 4095) / 4096 * 4096); return *p ? hipSuccess : hipErrorOutOfMemory; }
void slab_free(void* p) { ::free(p); }
void use_device(int) {}
int current_device() { return 0; }
#else
hipError_t slab_malloc(void** p, size_t n) { return hipMalloc(p, n); }
void slab_free(void* p) { (void)hipFree(p); }
void use_device(int d) { (void)hipSetDevice(d); }
int current_device() { int d = 0; (void)hipGetDevice(&d); return d; }
#endif
// every unit the engine frees on its own is a multiple of one 128-byte cache line -- record strides 4096, 2048, 1152, packed
// records (128-byte aligned) and the 17 408-byte tiles of MXFP4 runs (kernels.hpp: 16 records = 136 lines) -- so with this
// granule a run of k units occupies exactly k units of bytes and any sub-run can go back to the pool (Engine::migrate) without
// the freed range reaching into live neighbours; every run starts on a line.
constexpr size_t kGranule = 128;
inline size_t round_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
}

bool SlabPool::same_slab(uintptr_t a, uintptr_t b) const
{
    for (const auto& s : slabs_) {
        const uintptr_t lo = reinterpret_cast<uintptr_t>(s.base), hi = lo + s.bytes;
        if (a >= lo && a < hi) return b >= lo && b <= hi;
    }
    return false;
}

void SlabPool::insert_free(uintptr_t addr, size_t len)
{
    // coalesce with neighbours, but never across two hipMalloc'ed slabs
    auto next = free_.lower_bound(addr);
    if (next != free_.begin()) {
        auto prev = std::prev(next);
        if (prev->first + prev->second == addr && same_slab(prev->first, addr)) {
            addr = prev->first;
            len += prev->second;
            free_.erase(prev);
        }
    }
    if (next != free_.end() && addr + len == next->first && same_slab(addr, next->first)) {
        len += next->second;
        free_.erase(next);
    }
    free_[addr] = len;
}

bool SlabPool::grow(size_t min_bytes)
{
    size_t want = min_bytes > slab_bytes_ ? round_up(min_bytes, kGranule) : slab_bytes_;
    if (capacity_ && reserved_ + want > capacity_) {
        if (reserved_ + min_bytes > capacity_) return false;
        want = round_up(min_bytes, kGranule);
    }
    const int prev = current_device();
    use_device(device_);
    void* p = nullptr;
    hipError_t e = slab_malloc(&p, want);
    if (e != hipSuccess && want > min_bytes) {       // fall back to an exact-size slab
        (void)hipGetLastError();
        want = round_up(min_bytes, kGranule);
        e = slab_malloc(&p, want);
    }
    use_device(prev);
    if (e != hipSuccess || !p) { (void)hipGetLastError(); return false; }
    slabs_.push_back({static_cast<uint8_t*>(p), want});
    reserved_ += want;
    insert_free(reinterpret_cast<uintptr_t>(p), want);
    return true;
}

void* SlabPool::alloc(size_t bytes)
{
    if (bytes == 0) return nullptr;
    bytes = round_up(bytes, kGranule);
    const size_t align = kGranule;
    for (int attempt = 0; attempt < 2; ++attempt) {
        for (auto it = free_.begin(); it != free_.end(); ++it) {
            const uintptr_t addr = round_up(it->first, align);
            const size_t head = addr - it->first;
            if (it->second >= head + bytes) {
                const uintptr_t start = it->first;
                const size_t rest = it->second - head - bytes;
                free_.erase(it);
                if (head) free_[start] = head;
                if (rest) free_[addr + bytes] = rest;
                used_ += bytes;
                return reinterpret_cast<void*>(addr);
            }
        }
        if (!grow(bytes)) return nullptr;
    }
    return nullptr;
}

void* SlabPool::alloc_up_to(size_t want, size_t granule, size_t* got)
{
    *got = 0;
    want = granule ? want / granule * granule : 0;      // whole granules only
    if (want == 0) return nullptr;
    for (int attempt = 0; attempt < 2; ++attempt) {
        auto best = free_.end();
        size_t best_len = 0;
        const size_t align = kGranule;
        for (auto it = free_.begin(); it != free_.end(); ++it) {
            const size_t head = round_up(it->first, align) - it->first;
            if (it->second <= head) continue;
            size_t usable = (it->second - head) / granule * granule;
            if (usable > want) usable = want;
            if (usable > best_len) { best_len = usable; best = it; }
        }
        if (best != free_.end() && best_len >= granule) {
            const uintptr_t start = best->first, addr = round_up(start, align);
            const size_t head = addr - start, rest = best->second - head - best_len;
            free_.erase(best);
            if (head) free_[start] = head;
            if (rest) free_[addr + best_len] = rest;
            used_ += best_len;
            *got = best_len;
            return reinterpret_cast<void*>(addr);
        }
        if (!grow(want < slab_bytes_ ? want : slab_bytes_)) return nullptr;
    }
    return nullptr;
}

void SlabPool::free(void* p, size_t bytes)
{
    if (!p || bytes == 0) return;
    bytes = round_up(bytes, kGranule);
    used_ = used_ >= bytes ? used_ - bytes : 0;
    insert_free(reinterpret_cast<uintptr_t>(p), bytes);
}

void SlabPool::release()
{
    if (slabs_.empty()) return;
    const int prev = current_device();
    use_device(device_);
    for (auto& s : slabs_) slab_free(s.base);
    use_device(prev);
    slabs_.clear();
    free_.clear();
    reserved_ = used_ = 0;
}

} // namespace speckv
