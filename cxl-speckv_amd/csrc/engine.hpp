// cxl-speckv_amd/csrc/engine.hpp -- the MI355X KV engine behind the C ABI.
//
// One Engine replaces, in one process and without a kernel module:
//   SpeckvAllocator  host/src/speckv_allocator.cpp   handle -> pages, residency flags, sync fetch
//   SpeckvDriver     host/src/speckv_driver.cpp      DMA batch / prefetch / poll_complete / params
//   kernel module    driver/speckv_kernel_module.c   descriptor ring + completion counter
//   CXLMemoryManager src/cxl_memory/cxl_memory_manager.cpp  L1/L2/L3 tiers, LRU, hot pages
//   CXLMemoryAllocator::cxl_access policy            src/integration/memory_allocator.cpp:105-143
//   SpeculativePrefetcher depth adaptation           src/prefetcher/speculative_prefetcher.cpp:84-137
//
// HBM layout (compute GPU = device the engine was opened on):
//   pool   (L3): SlabPool per pool GPU; every page of an allocation owns one
//                fixed-size record slot (4096 B for FP16 / INT8_DELTA_RLE worst
//                case, 2048 B for INT8); slots of one allocation are one
//                contiguous run per pool GPU, pages striped page % n_pool.
//   table      : per allocation, PageEntry[n_pages] (16 B: record address,
//                record bytes, scale) + uint32 residency mirror, both in HBM.
//   cache (L1+L2): one arena of 4 KiB slots on the compute GPU; slots
//                [0, n_l2) are the prefetch ring (FIFO), [n_l2, n_l2+n_l1)
//                the LRU-managed resident set.
// All tier bookkeeping is host-side (as in the reference); kernels see the
// page table and the residency mirror only.
#pragma once
#include "../../include/speckv_ext.h"
#include "kernels.hpp"
#include "slab_pool.hpp"

#include <deque>
#include <memory>
#include <string>
#include <unordered_map>
#include <vector>

namespace speckv {

// speculative_prefetcher.cpp:98-120
class AdaptiveDepth {
public:
    explicit AdaptiveDepth(uint32_t d = 4) : depth_(d) {}
    void set(uint32_t d) { depth_ = d; }
    uint32_t depth() const { return depth_; }
    void update(bool was_correct);
private:
    uint32_t depth_;
    std::deque<uint8_t> hist_;   // window of 100 outcomes
};

struct Allocation {
    uint64_t handle = 0;
    size_t size_bytes = 0;
    uint64_t n_pages = 0;
    int scheme = 0;
    uint32_t rec_stride = kPageSize;
    std::vector<uint32_t> flags;          // bit0 L1, bit1 L2, bit2 compressed (KvPageHandle::flags)
    std::vector<uint32_t> slot;           // cache slot when flags&3
    std::vector<uint32_t> access_count;   // MemoryPage::access_count
    std::vector<uint32_t> stamp;          // dedupe epoch for prefetch flushes
    // HIP mode
    struct Extent { int pool; void* base; size_t bytes; uint64_t n_pages; };
    std::vector<Extent> extents;
    std::vector<int> pool_of_residue;     // pool index serving pages with page % D == k (at allocation)
    std::vector<uint8_t> page_pool;       // pool index holding each page's record now
    PageEntry* d_entries = nullptr;
    uint32_t* d_flags = nullptr;
    // set while every record lies in ONE run of one local pool (record p at linear_base + p*rec_stride)
    // and, for the fixed-size formats, never-written records are zero bytes; cleared by a migration
    uint8_t* linear_base = nullptr;
    // FP8 allocations with a known layout: block scales in the fused attention's tile order (attend.hip), n_pages floats
    float* d_scale_tab = nullptr;
    uint32_t region_pages = 0;
    bool has_layout = false;
    Layout layout{};
};

class Engine {
public:
    // status: SPECKV_OK or the code speckv_init must return
    static std::unique_ptr<Engine> open(const char* dev_path, int* status);
    ~Engine();

    bool null_device() const { return null_; }

    int alloc(size_t bytes, const speckv_alloc_hint_t* hint, uint64_t* out);
    int free(uint64_t handle);
    int access(uint64_t handle, uint64_t off, size_t len, void** out);
    int prefetch(uint32_t req, uint16_t layer, uint32_t pos, uint32_t k,
                 const int32_t* tokens, uint32_t hist);
    int set_prefetch_depth(uint32_t k);
    int set_scheme(int scheme);
    int set_quant_mode(int mode);

    int translate(uint64_t handle, uint64_t off, speckv_ext_page_info_t* out);
    int fetch_desc(uint64_t handle, uint64_t off, speckv_dma_desc_t* out);
    int set_layout(uint64_t handle, uint32_t T, uint32_t L, uint32_t H, uint32_t D, uint32_t bpe);
    int write(uint64_t handle, uint64_t off, const void* src, size_t len, bool on_device);
    int read(uint64_t handle, uint64_t off, void* dst, size_t len, bool on_device);
    int fetch_range(uint64_t handle, uint64_t first, uint64_t n, void* d_dst, bool f32, hipStream_t s);
    int fetch_list(uint64_t handle, const uint32_t* d_pages, uint32_t n, void* d_dst, bool f32, hipStream_t s);
    int access_batch(uint64_t handle, const uint64_t* offs, uint32_t n, void** out);
    int prefetch_batch(uint32_t n, const uint32_t* req, const uint16_t* layer,
                       const uint32_t* pos, const uint32_t* k);
    int prefetch_flush(uint32_t* n_issued);
    int prefetch_lookup(uint64_t handle, uint32_t n, const uint32_t* d_req, const uint32_t* d_layer,
                        const uint32_t* d_pos, const uint32_t* d_k, uint32_t* d_out, uint32_t cap,
                        uint32_t* d_count, hipStream_t s);
    int verify(uint32_t req, int32_t actual, const int32_t* pred, uint32_t n,
               uint32_t* was_hit, uint32_t* new_depth);
    int qk_scores_fp8(uint64_t handle, uint32_t layer, uint32_t n_layers, const void* d_q_f16, uint32_t g,
                      uint32_t pos_begin, uint32_t pos_end, float* d_out, hipStream_t s);
    int attend_fp8(uint64_t handle, uint32_t layer, uint32_t n_layers, const void* d_q_f16, uint32_t g,
                   uint32_t pos_begin, uint32_t pos_end, float sm_scale, float* d_out, float* d_lse, hipStream_t s);
    int attend_fp8_batch(uint32_t n_seq, const uint64_t* handles, uint32_t layer, const void* d_q_f16, uint32_t g,
                         const uint32_t* pos_end, float sm_scale, float* d_out, float* d_lse, hipStream_t s);
    int attend_batch(int scheme, uint32_t n_seq, const uint64_t* handles, uint32_t layer, const void* d_q_f16, uint32_t g,
                     const uint32_t* pos_end, float sm_scale, float* d_out, float* d_lse, hipStream_t s);
    int attend_int4(uint64_t handle, uint32_t layer, uint32_t n_layers, const void* d_q_f16, uint32_t g,
                    uint32_t pos_begin, uint32_t pos_end, float sm_scale, float* d_out, float* d_lse, hipStream_t s);
    int migrate(uint64_t handle, uint64_t first_page, uint64_t n_pages, uint32_t target_pool);
    int predictor_load(const float* emb, const float* wout, uint32_t vocab, bool on_device);
    int predict_batch(uint32_t n, const int32_t* d_hist, uint32_t k, int32_t* d_tok, float* d_conf, hipStream_t s);
    int poll_complete(uint32_t* done);
    int sync();
    int promote_to_l1(uint64_t handle, uint64_t off);
    int demote_to_l3(uint64_t handle, uint64_t off);
    int stats(speckv_ext_stats_t* out);
    uint32_t prefetch_depth() const { return adapt_.depth(); }
    int compute_device() const { return device_; }

private:
    Engine() = default;

    bool null_ = false;
    int device_ = 0;
    hipStream_t stream_ = nullptr;         // fetch / codec side stream
    hipStream_t copy_stream_ = nullptr;    // peer copies (pool <-> pool migration)
    std::vector<std::unique_ptr<SlabPool>> pools_;

    std::unordered_map<uint64_t, std::unique_ptr<Allocation>> allocs_;
    uint64_t next_handle_ = 1;             // speckv_allocator.hpp:57
    uint64_t layout_handle_ = 0;

    int scheme_ = SPECKV_COMP_FP16;
    int quant_mode_ = SPECKV_QUANT_REF_EXACT;
    AdaptiveDepth adapt_{4};

    // cache arena
    struct Owner { Allocation* a; uint32_t page; };
    uint8_t* cache_base_ = nullptr;
    uint32_t n_l2_ = 0, n_l1_ = 0;
    uint64_t l2_hand_ = 0;
    std::vector<Owner> owner_;
    std::vector<uint32_t> lru_prev_, lru_next_, l1_free_;
    uint32_t lru_head_ = UINT32_MAX, lru_tail_ = UINT32_MAX;   // head = least recent

    // prefetch queue
    struct Req { uint32_t req, layer, pos, k; };
    std::vector<Req> queue_;
    uint32_t flush_threshold_ = 0;
    uint32_t epoch_ = 0;

    // residency-mirror maintenance
    std::unordered_map<Allocation*, std::vector<uint32_t>> pending_clear_;

    // completion accounting (speckv_kernel_module.c:194-215)
    struct Batch { hipEvent_t ev; uint32_t n; };
    std::deque<Batch> inflight_;
    std::vector<hipEvent_t> event_pool_;
    uint64_t completed_unpolled_ = 0;

    // device scratch
    struct Scratch { void* p = nullptr; size_t cap = 0; };
    Scratch s_pages_, s_dst_, s_req_, s_out_, s_tmp_, s_stage_;
    uint32_t* d_count_ = nullptr;

    // token predictor (lstm_predictor.cpp): weights in HBM, last history / prediction per request
    float* d_emb_ = nullptr;
    float* d_wout_ = nullptr;
    uint32_t vocab_ = 0;
    Scratch s_hid_, s_logits_, s_hist_, s_pred_;
    Scratch s_attn_, s_attn_seq_;
    // pinned staging for the batch descriptors: 4 slots in rotation, each guarded by an event (no stream sync per call)
    struct PinnedRing { void* base = nullptr; size_t slot_bytes = 0; hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr}; int next = 0; } seq_ring_;
    uint8_t* d_zero_page_ = nullptr;     // stands in for never-written pages in the fused attention
    std::unordered_map<uint32_t, std::vector<int32_t>> hist_;
    std::unordered_map<uint32_t, std::vector<int32_t>> pred_;
    std::vector<uint32_t> hist_dirty_;

    speckv_ext_stats_t st_{};

    Allocation* find(uint64_t h);
    int init_hip(int device);
    void* scratch(Scratch& s, size_t bytes);
    void release_allocation(Allocation* a);
    // tiers
    uint8_t* slot_ptr(uint32_t slot) const { return cache_base_ + static_cast<size_t>(slot) * kPageSize; }
    void drop_slot(uint32_t slot);                       // forget owner (page becomes non-resident)
    uint32_t take_l2_run(uint32_t n);                    // contiguous run of n ring slots
    uint32_t take_l1_slot();
    void lru_unlink(uint32_t slot);
    void lru_push_mru(uint32_t slot);
    void move_to_l1(Allocation* a, uint32_t page);
    void flush_mirror();
    // data movement
    int fetch_into_slots(Allocation* a, const std::vector<uint32_t>& pages,
                         const std::vector<uint32_t>& slots, bool wait);
    void reap(bool wait_all);
    int run_predictor_for_dirty();
    hipEvent_t get_event();
};

} // namespace speckv
