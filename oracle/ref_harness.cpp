// oracle/ref_harness.cpp -- C-linkage wrappers around the REFERENCE's own
// classes, compiled together with the reference translation units where they
// lie under /root/reference (see oracle/Makefile; nothing is copied).
//
// TEST INFRASTRUCTURE ONLY.  The resulting oracle/_ref/libspeckv_ref.so is
//   * the checker that pins oracle/speckv_oracle.c (tests/test_oracle_vs_ref.py),
//   * the generator of tests/golden/* (tests/golden/generate_golden.py),
//   * optionally bench.py's cpu_baseline leg (kind "reference").
// It is never loaded by the product library.
//
// Reference interfaces wrapped (paths relative to /root/reference):
//   src/fpga_engine/cache_engine.h:22-84      FPGACacheEngine
//   src/cxl_memory/cxl_memory_manager.h:40-90 CXLMemoryManager
//   src/prefetcher/speculative_prefetcher.h:32-78 SpeculativePrefetcher
//   src/integration/memory_allocator.h:19-52  CXLMemoryAllocator
// The reference's own C ABI (host/include/speckv.h) is exported by the same
// .so because host/src/*.cpp are linked in.
#include "fpga_engine/cache_engine.h"
#include "cxl_memory/cxl_memory_manager.h"
#include "prefetcher/speculative_prefetcher.h"
#include "integration/memory_allocator.h"

#include <cstdint>
#include <cstring>
#include <vector>

using namespace cxlspeckv;

extern "C" {

// ---------------------------------------------------------------- codec
void* ref_engine_new() { return new FPGACacheEngine(1, 800.0, 512, 16); }
void  ref_engine_delete(void* e) { delete static_cast<FPGACacheEngine*>(e); }

size_t ref_engine_compress(void* e, const float* x, size_t n, uint32_t layer,
                           float* scale, uint8_t* rle, size_t cap,
                           size_t* original_size)
{
    std::vector<float> v(x, x + n);
    auto c = static_cast<FPGACacheEngine*>(e)->compress(v, n, 1, layer);
    *scale = c.scale_factor;
    if (original_size) *original_size = c.original_size;
    size_t len = c.rle_data.size();
    if (len <= cap && len) std::memcpy(rle, c.rle_data.data(), len);
    return len;   // == compressed_size
}

size_t ref_engine_decompress(void* e, const uint8_t* rle, size_t len, float scale,
                             float* y, size_t cap)
{
    FPGACacheEngine::CompressedData c;
    c.scale_factor = scale;
    c.rle_data.assign(reinterpret_cast<const int8_t*>(rle),
                      reinterpret_cast<const int8_t*>(rle) + len);
    c.original_size = 0;
    c.compressed_size = len;
    auto out = static_cast<FPGACacheEngine*>(e)->decompress(c, 0, 0);
    size_t n = out.size() < cap ? out.size() : cap;
    if (n) std::memcpy(y, out.data(), n * sizeof(float));
    return out.size();
}

// Batched forms for timing the reference on many equally sized blocks
// (fp32 in, as the reference's API takes it).  Returns total compressed bytes.
size_t ref_engine_compress_blocks(void* e, const float* x, size_t n_blocks, size_t n,
                                  float* scales, uint8_t* recs, size_t stride,
                                  uint32_t* lens)
{
    auto eng = static_cast<FPGACacheEngine*>(e);
    size_t total = 0;
    std::vector<float> v(n);
    for (size_t b = 0; b < n_blocks; ++b) {
        std::memcpy(v.data(), x + b * n, n * sizeof(float));
        auto c = eng->compress(v, n, 1, 0);
        scales[b] = c.scale_factor;
        lens[b] = static_cast<uint32_t>(c.rle_data.size());
        if (c.rle_data.size() <= stride)
            std::memcpy(recs + b * stride, c.rle_data.data(), c.rle_data.size());
        total += c.rle_data.size();
    }
    return total;
}

size_t ref_engine_decompress_blocks(void* e, const uint8_t* recs, size_t stride,
                                    const uint32_t* lens, const float* scales,
                                    size_t n_blocks, float* y, size_t n)
{
    auto eng = static_cast<FPGACacheEngine*>(e);
    size_t total = 0;
    FPGACacheEngine::CompressedData c;
    for (size_t b = 0; b < n_blocks; ++b) {
        const int8_t* p = reinterpret_cast<const int8_t*>(recs + b * stride);
        c.scale_factor = scales[b];
        c.rle_data.assign(p, p + lens[b]);
        c.compressed_size = lens[b];
        c.original_size = n * 4;
        auto out = eng->decompress(c, 0, 0);
        size_t m = out.size() < n ? out.size() : n;
        if (y && m) std::memcpy(y + b * n, out.data(), m * sizeof(float));
        total += out.size();
    }
    return total;
}

double   ref_engine_ratio(void* e, uint32_t layer) { return static_cast<FPGACacheEngine*>(e)->get_compression_ratio(layer); }
double   ref_engine_throughput(void* e) { return static_cast<FPGACacheEngine*>(e)->get_statistics().throughput_gbps; }
uint64_t ref_engine_translate(void* e, uint64_t va) { return static_cast<FPGACacheEngine*>(e)->translate_address(va); }

// ------------------------------------------------------- memory manager
void* ref_mm_new(size_t l1, size_t l2, size_t l3) { return new CXLMemoryManager(l1, l2, l3); }
void  ref_mm_delete(void* m) { delete static_cast<CXLMemoryManager*>(m); }
#define MM(m) static_cast<CXLMemoryManager*>(m)
uint64_t ref_mm_allocate(void* m, size_t bytes, uint32_t layer, int tier) { return MM(m)->allocate(bytes, layer, static_cast<MemoryTier>(tier)); }
void     ref_mm_deallocate(void* m, uint64_t va) { MM(m)->deallocate(va); }
uint64_t ref_mm_translate(void* m, uint64_t va) { return MM(m)->translate_virtual_to_physical(va); }
int      ref_mm_is_in_cache(void* m, uint64_t va, int tier) { return MM(m)->is_in_cache(va, static_cast<MemoryTier>(tier)); }
int      ref_mm_promote_to_l1(void* m, uint64_t va) { return MM(m)->promote_to_l1(va); }
int      ref_mm_demote_to_l3(void* m, uint64_t va) { return MM(m)->demote_to_l3(va); }
void     ref_mm_invalidate_page(void* m, uint64_t va) { MM(m)->invalidate_page(va); }
void     ref_mm_mark_modified(void* m, uint64_t va) { MM(m)->mark_modified(va); }
int      ref_mm_get_page_state(void* m, uint64_t va) { return static_cast<int>(MM(m)->get_page_state(va)); }
void     ref_mm_update_access_tracking(void* m, uint64_t va) { MM(m)->update_access_tracking(va); }
int      ref_mm_is_hot_page(void* m, uint64_t va) { return MM(m)->is_hot_page(va); }
void     ref_mm_get_statistics(void* m, uint64_t* u7, double* d2)
{
    auto s = MM(m)->get_statistics();
    u7[0] = s.l1_hits; u7[1] = s.l1_misses; u7[2] = s.l2_hits; u7[3] = s.l2_misses;
    u7[4] = s.l3_accesses; u7[5] = s.migrations_l1_to_l3; u7[6] = s.migrations_l3_to_l1;
    d2[0] = s.l1_hit_rate; d2[1] = s.l2_hit_rate;
}

// ----------------------------------------------------------- prefetcher
void* ref_pf_new(void* mm, size_t depth, size_t hist) { return new SpeculativePrefetcher(MM(mm), depth, hist); }
void  ref_pf_delete(void* p) { delete static_cast<SpeculativePrefetcher*>(p); }
#define PF(p) static_cast<SpeculativePrefetcher*>(p)
size_t ref_pf_prefetch(void* p, const uint32_t* hist, size_t n_hist, uint32_t layer,
                       size_t depth, uint64_t* out_va, uint32_t* out_tok,
                       float* out_conf, size_t cap)
{
    std::vector<uint32_t> h(hist, hist + n_hist);
    auto reqs = PF(p)->prefetch(h, layer, depth);
    for (size_t i = 0; i < reqs.size() && i < cap; ++i) {
        out_va[i] = reqs[i].virtual_addr;
        if (out_tok) out_tok[i] = reqs[i].predicted_token_id;
        if (out_conf) out_conf[i] = reqs[i].confidence;
    }
    return reqs.size();
}
void   ref_pf_update_accuracy(void* p, uint32_t req, int ok) { PF(p)->update_prediction_accuracy(req, ok != 0); }
size_t ref_pf_adaptive_depth(void* p) { return PF(p)->get_adaptive_depth(); }
size_t ref_pf_handle_misprediction(void* p, uint32_t actual, const uint32_t* pred, size_t n)
{
    std::vector<uint32_t> v(pred, pred + n);
    PF(p)->handle_misprediction(actual, v);
    return PF(p)->get_statistics().mispredictions;
}

// ------------------------------------------- integration allocator policy
void* ref_ca_new(size_t l1, size_t l2, size_t l3)
{
    auto a = new CXLMemoryAllocator();
    if (!a->initialize(l1, l2, l3)) { delete a; return nullptr; }
    return a;
}
void  ref_ca_delete(void* a) { delete static_cast<CXLMemoryAllocator*>(a); }
#define CA(a) static_cast<CXLMemoryAllocator*>(a)
uint64_t ref_ca_malloc(void* a, size_t bytes, uint32_t layer) { return reinterpret_cast<uint64_t>(CA(a)->cxl_malloc(bytes, layer)); }
void     ref_ca_free(void* a, uint64_t p) { CA(a)->cxl_free(reinterpret_cast<void*>(p)); }
uint64_t ref_ca_access(void* a, uint64_t h, size_t off, size_t sz) { return reinterpret_cast<uint64_t>(CA(a)->cxl_access(reinterpret_cast<void*>(h), off, sz)); }
void     ref_ca_stats(void* a, uint64_t* u4)
{
    auto s = CA(a)->get_statistics();
    u4[0] = s.total_allocations; u4[1] = s.total_deallocations;
    u4[2] = s.current_allocated_bytes; u4[3] = s.peak_allocated_bytes;
}

} // extern "C"
