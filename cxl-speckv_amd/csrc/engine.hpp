// cxl-speckv_amd/csrc/engine.hpp -- the MI355X KV engine behind the C ABI.
//
// One Engine replaces, in one process and without a kernel module:
//   SpeckvAllocator  host/src/speckv_allocator.cpp   handle -> pages, residency flags, sync fetch
//   SpeckvDriver     host/src/speckv_driver.cpp      DMA batch / prefetch / poll_complete / params
//   kernel module    driver/speckv_kernel_module.c   descriptor ring + completion counter
//   CXLMemoryManager src/cxl_memory/cxl_memory_manager.cpp  L1/L2/L3 tiers, LRU, hot pages
//   CXLMemoryAllocator::cxl_access policy            src/integration/memory_allocator.cpp:105-143
//   SpeculativePrefetcher depth adaptation           src/prefetcher/speculative_prefetcher.cpp:84-137
//   prefetch_core lookup/issue loop                  hardware/rtl/prefetch_core.v:150-241 (device-side flush)
//   DMA engine, one descriptor per page              hardware/rtl/dma_engine.v:150-217 (copy-engine fetch of runs)
//
// HBM layout (compute GPU = device the engine was opened on):
//   pool   (L3): SlabPool per pool GPU; every page of an allocation owns one fixed-size record slot (4096 B for
//                FP16 / INT8_DELTA_RLE worst case, 2048 B for INT8 / FP8, 1152 B for INT4); pages striped
//                page % n_pool, so a logical page range is ONE contiguous record run on every pool GPU.
//   table      : per allocation PageEntry[n_pages] (16 B: record address, record bytes, scale), the residency
//                mirror (flags, slot) and the dedupe stamps, all in HBM; one DevAlloc row per allocation in the
//                device allocation table so kernels can work across allocations.
//   cache (L1+L2): one arena of 4 KiB slots on the compute GPU; slots [0, n_l2) are the prefetch ring (FIFO,
//                runs never wrap), [n_l2, n_l2+n_l1) the LRU-managed resident set.
// Who owns what: the L2 ring is managed ON THE DEVICE (owner table, eviction, slot / flag updates are done by the
// fetch kernel itself; the prefetch flush assigns its ring run on the device); the host keeps an exact mirror of
// the ring hand (same deterministic rule) and reads residency from a pinned, device-written copy of flags / slots.
// L1 (LRU) stays host-managed, as the reference's tier manager.
#pragma once
#include "../../include/speckv_ext.h"
#include "kernels.hpp"
#include "ring_rule.hpp"
#include "slab_pool.hpp"

#include <condition_variable>
#include <deque>
#include <memory>
#include <mutex>
#include <string>
#include <array>
#include <map>
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

constexpr uint32_t kLenSamples = 16;       // record-length samples per allocation (Allocation::len_samples)
struct Allocation {
    uint64_t handle = 0;
    size_t size_bytes = 0;
    uint64_t n_pages = 0;
    int scheme = 0;
    uint32_t rec_stride = kPageSize;
    uint32_t row = kNoSlot;               // row in the device allocation table
    // residency as the host sees it.  flags: bit0 L1, bit2 compressed (host-managed; bit1 L2 is stored on the fake device
    // only -- on a HIP device it is derived, Engine::res_flags).  slot (HIP device only): pinned host memory that the
    // kernels also write -- the ring sequence number of a page fetched into L2, or the L1 slot of a promoted page.
    uint32_t* flags = nullptr;
    uint32_t* slot = nullptr;
    std::vector<uint32_t> host_flags;     // backing store of flags
    void* pinned = nullptr;               // backing store of slot (+ kLenSamples words behind it)
    // INT8_DELTA_RLE: record lengths of every 1024th page, stored by the compress kernel itself (CodecArgs::len_samples; host-visible
    // memory, no copy back): looks_structured() = "the mean sample is under 512 B" picks the flat-run decoder for reads of an
    // allocation that was never sealed.  0 = no sample yet.
    uint32_t* len_samples = nullptr;
    std::vector<uint32_t> access_count;   // MemoryPage::access_count
    std::vector<uint32_t> stamp;          // dedupe epoch of access_batch
    uint32_t l1_pages = 0;
    // HIP mode
    struct Extent { int pool; void* base; size_t bytes; uint64_t n_pages; };
    std::vector<Extent> extents;
    // tile-planar MXFP4 runs (kernels.hpp) are returned to the pool tile by tile: tile address -> record slots that hold no page
    // of this allocation (the unused tail of a run's last tile, records that migrated away); a tile whose 16 slots are all
    // vacated goes back (Engine::migrate).  Only tiles with at least one vacated slot are listed.
    std::unordered_map<uint64_t, uint16_t> mx4_vacated;
    std::vector<int> pool_of_residue;     // pool index serving pages with page % D == k (at allocation)
    std::vector<uint8_t> page_pool;       // pool index holding each page's record now
    bool regular = false;                 // placement still is "page p = record p/D of the run on pool p%D"
    // compacted (speckv_ext_compact): the records lie back to back (128-byte aligned) in ONE packed extent per pool, in page
    // order; packed_off128[p] = offset of page p's record inside its pool's extent in units of 128 B, packed_bytes[k] = size of
    // pool residue k's extent (extents[k]); packed_regular: the pages still go to pool p % D (the copy engine's condition)
    bool packed = false, packed_regular = false;
    std::vector<uint32_t> packed_off128;
    std::vector<uint64_t> packed_bytes;
    PageEntry* d_entries = nullptr;
    uint32_t* d_flags = nullptr;
    uint32_t* d_slot = nullptr;
    uint32_t* d_stamp = nullptr;
    // set while every record lies in ONE run of one local pool (record p at linear_base + p*rec_stride)
    // and, for the fixed-size formats, never-written records are zero bytes; cleared by a migration
    uint8_t* linear_base = nullptr;
    // set while the placement is regular over 2..8 pools (and, for the fixed-size formats, never-written records are zero
    // bytes): device array of the 8 run bases, record of page p = d_stripe[p % stripe_n] + (p / stripe_n) * rec_stride --
    // the fused attention then computes addresses instead of chasing the page table; cleared by a migration
    uint64_t* d_stripe = nullptr;
    uint32_t stripe_n = 0;
    // FP8 allocations with a known layout: block scales in the fused attention's tile order (attend.hip), n_pages floats
    float* d_scale_tab = nullptr;
    float* d_scale_tab_base = nullptr;    // what was allocated: a striped FP8 allocation keeps its scales in run order in front of the table
    uint32_t scale_run = 0;               // ... CodecArgs::scale_run (D | cap << 4), 0 = no run-order part
    uint32_t region_pages = 0;
    bool has_layout = false;
    bool layout_inferred = false;
    Layout layout{};
    uint32_t entry_bytes_seen = 0;        // length_bytes of the first speckv_access (= head_dim * bytes_per_element in the shim)
    // caller streams that have been handed asynchronous work on this allocation (speckv_free waits for those only)
    std::vector<hipStream_t> user_streams;
};

class Engine {
public:
    // status: SPECKV_OK or the code speckv_init must return
    static std::unique_ptr<Engine> open(const char* dev_path, int* status);
    ~Engine();

    bool null_device() const { return null_; }
    // The C ABI serialises callers with one mutex (speckv_c_api.cpp:10).  Every entry hands its lock to the engine,
    // which releases it only while it waits for the GPU (state is re-validated afterwards).
    void enter(std::unique_lock<std::mutex>* lk) { lk_ = lk; }
    // speckv_finalize: returns (with `lk` held) once no thread is parked inside the engine with the lock released
    // (wait_event); the caller keeps new entries out meanwhile
    void wait_idle(std::unique_lock<std::mutex>& lk) { idle_cv_.wait(lk, [this] { return waiting_ == 0; }); }

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
    int bind_request(uint32_t req, uint64_t handle, uint32_t local_req);
    int write(uint64_t handle, uint64_t off, const void* src, size_t len, bool on_device);
    int write_strided(uint64_t handle, uint64_t first, uint64_t step, uint64_t n, const void* d_src, hipStream_t s);
    int write_async(uint64_t handle, uint64_t off, const void* d_src, size_t len, hipStream_t s);
    hipError_t upload_pinned(void* dst, const void* staged, size_t bytes, hipStream_t s);
    int attend_batch_plan(uint32_t n_seq, const uint64_t* handles, const uint32_t* pos_end, uint32_t max_pos_end, void* d_plan,
                          size_t plan_bytes, hipStream_t s);
    // the position a caller still keeps outside the pool (speckv_ext_attend_planned_tail): fp16 rows [n_tail][layers][heads][128]
    struct TailArgs { uint32_t n_tail; const uint32_t* d_tail_rows; const int32_t* d_tail_idx; const void* d_k_tail; const void* d_v_tail; uint64_t stride_elems; };
    int attend_planned(int scheme, const void* d_plan, uint32_t n_seq, uint32_t layer, const void* d_q_f16, uint32_t g,
                       uint32_t max_pos_end, float sm_scale, float* d_out, float* d_lse, hipStream_t s, const TailArgs* tail = nullptr, uint32_t n_layers = 1);
    int attend_planned_layers(int scheme, const void* d_plan, uint32_t n_seq, uint32_t layer_begin, uint32_t n_layers, const void* d_q_f16, uint32_t g,
                              uint32_t max_pos_end, float sm_scale, float* d_out, float* d_lse, hipStream_t s, const TailArgs* tail = nullptr);
    int attend_fold_tail(uint32_t n_rows, const uint32_t* d_rows, uint32_t heads, uint32_t g, const void* d_q_f16, const void* d_k_tail,
                         const void* d_v_tail, uint64_t tail_stride_elems, float sm_scale, float* d_out, float* d_lse, hipStream_t s);
    int write_strided_batch(const uint64_t* handles, const uint64_t* firsts, const void* const* d_srcs, uint32_t n_alloc, uint64_t step,
                            uint64_t n_each, hipStream_t s);
    int write_runs(uint64_t handle, const uint64_t* firsts, const void* const* d_srcs, uint32_t n_runs, uint64_t n_each, hipStream_t s);
    int read(uint64_t handle, uint64_t off, void* dst, size_t len, bool on_device);
    int fetch_range(uint64_t handle, uint64_t first, uint64_t n, void* d_dst, bool f32, hipStream_t s, int engine_choice);
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
    int attend_batch(int scheme, uint32_t n_seq, const uint64_t* handles, uint32_t layer, const void* d_q_f16, uint32_t g,
                     const uint32_t* pos_end, float sm_scale, float* d_out, float* d_lse, hipStream_t s);
    int attend_int4(uint64_t handle, uint32_t layer, uint32_t n_layers, const void* d_q_f16, uint32_t g,
                    uint32_t pos_begin, uint32_t pos_end, float sm_scale, float* d_out, float* d_lse, hipStream_t s);
    int attend_mx4(uint64_t handle, uint32_t layer, uint32_t n_layers, const void* d_q_f16, uint32_t g,
                   uint32_t pos_begin, uint32_t pos_end, float sm_scale, float* d_out, float* d_lse, hipStream_t s);
    int migrate(uint64_t handle, uint64_t first_page, uint64_t n_pages, uint32_t target_pool);
    int compact(uint64_t handle, uint64_t* bytes_before, uint64_t* bytes_after);
    int predictor_load(const float* emb, const float* wout, uint32_t vocab, bool on_device);
    int predictor_load_lstm(const float* emb, uint32_t vocab, uint32_t n_layers, const float* const* w_ih, const float* const* w_hh,
                            const float* const* b_ih, const float* const* b_hh, const float* wout, const float* out_bias, bool on_device);
    int predict_batch(uint32_t n, const int32_t* d_hist, uint32_t k, int32_t* d_tok, float* d_conf, hipStream_t s);
    int poll_complete(uint32_t* done);
    int sync();
    int promote_to_l1(uint64_t handle, uint64_t off);
    int demote_to_l3(uint64_t handle, uint64_t off);
    int stats(speckv_ext_stats_t* out);
    uint32_t prefetch_depth() const { return adapt_.depth(); }
    int compute_device() const { return device_; }
    // compute units of THIS engine's device (launch geometry of the attention kernels; ADVICE r4: not a process-wide guess)
    uint32_t cus()
    {
        if (!n_cus_) {
            int v = 0;
            if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, device_) != hipSuccess || v <= 0) { (void)hipGetLastError(); v = 256; }
            n_cus_ = static_cast<uint32_t>(v);
        }
        return n_cus_;
    }

private:
    Engine() = default;

    bool null_ = false;
    int device_ = 0;
    uint32_t n_cus_ = 0;
    bool dbg_unordered_scratch_ = false, access_spin_ok_ = true;      // SPECKV_DEBUG_UNORDERED_SCRATCH / SPECKV_ACCESS_NO_SPIN, read at open
    std::vector<int> default_layout_;                                 // SPECKV_LAYOUT, read at open
    std::unique_lock<std::mutex>* lk_ = nullptr;
    uint32_t waiting_ = 0;                 // threads inside wait_event with the ABI lock released
    std::condition_variable idle_cv_;
    hipStream_t stream_ = nullptr;         // fetch / codec side stream
    hipStream_t copy_stream_ = nullptr;    // peer copies (pool <-> pool migration)
    int flush_words_mode_ = 0;             // who stores a flush's host-visible words: 0 by size, 1 scatter kernel, 2 fetch launch
    std::vector<int> pool_devs_;           // pool GPUs in SPECKV_POOL_DEVICES order (also kept on the fake device)
    std::vector<std::unique_ptr<SlabPool>> pools_;

    std::unordered_map<uint64_t, std::unique_ptr<Allocation>> allocs_;
    uint64_t next_handle_ = 1;             // speckv_allocator.hpp:57
    uint64_t layout_handle_ = 0;

    int scheme_ = SPECKV_COMP_FP16;
    int quant_mode_ = SPECKV_QUANT_REF_EXACT;
    AdaptiveDepth adapt_{4};

    // device allocation table
    DevAlloc* d_tab_ = nullptr;
    DevAlloc* h_tab_ = nullptr;            // pinned mirror: the source of every (asynchronous) row update
    uint32_t tab_cap_ = 0;
    std::vector<uint32_t> free_rows_;
    std::vector<Allocation*> row_owner_;
    int publish_row(Allocation* a);        // (re)write the allocation's table row on the engine stream

    // cache arena
    struct Owner { Allocation* a; uint32_t page; };
    uint8_t* cache_base_ = nullptr;
    uint32_t n_l2_ = 0, n_l1_ = 0;
    // Ring sequence number: every slot the L2 ring's hand has passed (slots skipped at the end of a lap included), so
    // slot = seq % n_l2_.  The device keeps the same number (d_hand_) and stores, as a fetched page's ONLY host-visible
    // word, the sequence number of its slot (Allocation::slot); the host derives residency from it (l2_live).
    uint32_t ring_seq_ = 0;
    uint32_t ring_seq_limit_ = 0xC0000000u;   // renumber_ring_if_due: keeps sequence numbers clear of the 32-bit wrap
    uint64_t* d_owner_ = nullptr;          // [n_l2] (row << 32 | page), kNoOwner when free
    uint32_t* d_hand_ = nullptr;
    // completion word of small synchronous fetches (CodecArgs::done_flag): pinned host word, device counter, last token handed out
    uint32_t* h_done_ = nullptr;
    uint32_t* h_done_dev_ = nullptr;       // the same word as the device addresses it
    uint32_t* d_done_count_ = nullptr;
    uint32_t done_token_ = 0;
    std::vector<Owner> l1_owner_;          // [n_l1], slot - n_l2
    std::vector<uint32_t> lru_prev_, lru_next_, l1_free_;
    uint32_t lru_head_ = UINT32_MAX, lru_tail_ = UINT32_MAX;   // head = least recent

    // request -> allocation binding (speckv_ext_bind_request); unbound requests index into the last laid-out
    // allocation, as in the reference's single-allocation shim (vllm_speckv_backend.py:95-100)
    struct Binding { uint64_t handle; uint32_t local_req; };
    std::unordered_map<uint32_t, Binding> bindings_;

    // prefetch queue: requests are resolved when they arrive (columns ready for upload); the ones that arrive
    // before any geometry is known wait in q_unresolved_
    struct Req { uint32_t req, layer, pos, k; };
    struct Resolved { uint32_t req = kNoSlot, row = kNoSlot, local = 0, n_layers = 0, W = 0; int scheme = -1; uint64_t gen = 0;
                      bool ok = false, no_geometry = false; };
    Resolved last_res_;
    uint64_t res_gen_ = 1;
    std::vector<uint32_t> q_req_, q_layer_, q_pos_, q_k_, q_row_;
    std::vector<Req> q_unresolved_;
    int q_scheme_ = 0;
    uint32_t q_W_ = 0;
    uint64_t q_dropped_ = 0;
    bool in_flush_ = false;
    bool resolve(uint32_t req);
    void enqueue(uint32_t req, uint32_t layer, uint32_t pos, uint32_t k);
    uint32_t flush_threshold_ = 0;
    uint32_t access_epoch_ = 0;
    uint32_t flush_epoch_ = 0;             // 1..255 (dedupe stamps)
    uint32_t max_layer_seen_ = 0;
    bool warned_no_layout_ = false;

    // host-originated residency changes waiting to reach the device mirrors (pinned ring read by k_apply_updates)
    MirrorUpdate* upd_ring_ = nullptr;
    uint32_t upd_cap_ = 0, upd_head_ = 0, upd_pending_ = 0;      // [head - pending, head) not yet launched
    hipEvent_t upd_event_ = nullptr;       // last launch that read the ring

    // flushes whose slot assignment the host has not absorbed yet
    struct Flight {
        hipEvent_t assigned = nullptr;     // after the assign kernel: the result words are final
        hipEvent_t done = nullptr;         // after the fetch kernel
        FlushResult* result = nullptr;     // pinned
        bool absorbed = false;
        uint32_t base = 0, m = 0;
    };
    std::deque<Flight> flights_;
    FlushResult* res_ring_ = nullptr;      // pinned, kResSlots entries
    FlushResult* d_res_ring_ = nullptr;    // device twin
    uint32_t res_next_ = 0;
    void* req_stage_ = nullptr;            // pinned staging for request uploads (4 slots in rotation)
    size_t req_stage_bytes_ = 0;
    hipEvent_t req_stage_ev_[4] = {nullptr, nullptr, nullptr, nullptr};
    int req_stage_next_ = 0;

    // Asynchronous writes run on the CALLER's stream (write_strided / _batch / write_async); everything the engine reads
    // on its own stream afterwards (synchronous misses, speckv_ext_read, the prefetch flush, L1 promotion) must be
    // ordered behind them.  One event per caller stream, re-recorded by each write; `dirty` = the engine stream has not
    // waited for the latest record yet (order_after_writes, called in front of every engine-stream read of pool records).
    struct WriterEv { hipStream_t s; hipEvent_t ev; bool dirty; };
    std::vector<WriterEv> write_evs_;
    int note_async_write(hipStream_t s);
    int note_async_write_or_wait(hipStream_t s);
    int order_after_writes();

    // completion accounting (speckv_kernel_module.c:194-215): launches on the engine stream whose size is known when
    // they are submitted; a flush's pages are counted when its flight completes (harvest_flights)
    struct Batch { hipEvent_t ev; uint32_t n; };
    std::deque<Batch> inflight_;
    std::vector<hipEvent_t> event_pool_;
    uint64_t completed_unpolled_ = 0;

    // allocations freed while a stream may still read them
    struct Zombie { std::unique_ptr<Allocation> a; hipEvent_t engine_ev; };
    std::vector<Zombie> zombies_;

    // device scratch
    struct Scratch { void* p = nullptr; size_t cap = 0; bool in_graph = false; hipStream_t last = nullptr; };   // last: the stream of the latest user
    Scratch s_pages_, s_req_, s_tmp_, s_stage_, s_flush_;
    std::vector<void*> retired_;           // scratch buffers a captured graph may still reference
    uint32_t* d_count_ = nullptr;

    // copy-engine fetch: per-pool side streams, double-buffered staging on the compute GPU
    struct PeerLane { hipStream_t s = nullptr; hipEvent_t copied[2] = {nullptr, nullptr}; };
    std::vector<PeerLane> lanes_;
    uint8_t* stage_[2] = {nullptr, nullptr};
    size_t stage_bytes_ = 0;               // per buffer
    hipEvent_t stage_free_[2] = {nullptr, nullptr};

    // token predictor (lstm_predictor.cpp): weights in HBM, last history / prediction per request
    float* d_emb_ = nullptr;
    float* d_wout_ = nullptr;
    uint32_t vocab_ = 0;
    LstmParams lstm_{};                    // device copies of a real cell's weights (layers == 0: the reference's degenerate cell)
    std::vector<float*> lstm_bufs_;
    Scratch s_hid_, s_logits_, s_predict_ws_, s_hist_, s_pred_;
    Scratch s_attn_, s_attn_seq_;
    // pinned staging for the batch descriptors: slots in rotation, each guarded by an event (no stream sync per call).
    // The batch attention kernels read their slot IN PLACE, so its guard is passed only when the launches of the call that
    // last used it have finished: the ring is sized for the calls a caller runs ahead of the GPU -- one decode step of an
    // 80-layer model and some (with 4 slots the host, holding the ABI lock, waited for the kernels of the call four
    // layers back at every layer).  A 97th call in flight waits, under the lock, for the oldest.
    static constexpr int kSeqRingSlots = 96;
    template <int N> struct PinnedRingT { void* base = nullptr; size_t slot_bytes = 0; hipEvent_t ev[N] = {}; int next = 0; };
    PinnedRingT<kSeqRingSlots> seq_ring_;
    PinnedRingT<4> grp_ring_;
    struct PlanInfo { uint32_t n_seq; int scheme; uint32_t n_layers, max_pos_end; bool striped, table; uint32_t mx4_stripe_n_max; bool any_empty; bool ordered;
                      uint32_t max_splits; bool rows_first; uint32_t rule_tps, rule_splits; };      // (the launch geometry the FIRST plan of this shape in this buffer chose: attend_batch_plan)
    std::unordered_map<const void*, PlanInfo> plans_;      // device plan buffer -> what attend_batch_plan last wrote there
    // (buffer, members | format, bound | rule's piece length, rule's pieces) -> {room for pieces, rows-first grid}: what the FIRST plan of that shape in that
    // buffer chose -- kept per shape, so that a buffer that alternates between shapes keeps every shape's captured launches valid
    std::map<std::array<uint64_t, 4>, std::pair<uint32_t, bool>> plan_rooms_;
    CompressGroup* d_groups_ = nullptr;    // device twin of grp_ring_ (4 slots): descriptors of a grouped compress launch
    uint8_t* d_zero_page_ = nullptr;     // stands in for never-written pages in the fused attention
    std::unordered_map<uint32_t, std::vector<int32_t>> hist_;
    std::unordered_map<uint32_t, std::vector<int32_t>> pred_;
    std::vector<uint32_t> hist_dirty_;
    // The predictions of a flush run on a stream of their own and come back through pinned memory: nobody needs the tokens
    // before the next speckv_ext_verify (the pages a flush fetches are addressed by position, lstm_predictor's tokens only
    // feed the hit statistics -- speculative_prefetcher.cpp:84-96), so the flush does not wait for them.
    hipStream_t pred_stream_ = nullptr;
    hipEvent_t pred_ev_ = nullptr;
    int32_t* h_pred_io_ = nullptr;         // pinned: n x 16 history tokens in, n x k predicted tokens out
    size_t h_pred_cap_ = 0;                // bytes
    struct PendingPred { std::vector<uint32_t> reqs; uint32_t k = 0; uint64_t gen = 0; bool active = false; } pending_pred_;
    int harvest_predictions();             // waits for the prediction in flight (if any) and files its tokens under pred_

    speckv_ext_stats_t st_{};

    Allocation* find(uint64_t h);
    Allocation* default_target();
    int init_hip(int device);
    void* scratch(Scratch& s, size_t bytes, hipStream_t user = nullptr);
    void release_allocation(Allocation* a);
    void note_use(Allocation* a, hipStream_t s);
    bool quiet(const Zombie& z);
    void drain_zombies(bool wait);
    // waits that give up the ABI lock (lk_) while the GPU works
    int wait_event(hipEvent_t ev);
    int wait_stream();                                    // everything queued on stream_ so far
    // tiers
    uint8_t* slot_ptr(uint32_t slot) const { return cache_base_ + static_cast<size_t>(slot) * kPageSize; }
    void drop_page(Allocation* a, uint32_t page);         // page leaves the cache (any tier)
    uint32_t take_l2_run(uint32_t n);                     // host mirror of the ring rule: sequence number of the run's first slot
    // a page fetched into the ring with sequence number q is intact until the hand has passed q + n_l2_
    bool l2_live(uint32_t q) const { return q != kNoSlot && ring_live(ring_seq_, q, n_l2_); }
    // residency as the API reports it: bit0 L1 (host-managed), bit1 L2 (derived, see above), bit2 compressed
    uint32_t res_flags(const Allocation* a, uint64_t p) const
    {
        const uint32_t f = a->flags[p];
        if (null_) return f;
        return (f & ~2u) | ((!(f & 1u) && l2_live(a->slot[p])) ? 2u : 0u);
    }
    // cache slot of a resident page (L1 slots are numbered after the ring)
    uint32_t res_slot(const Allocation* a, uint64_t p) const { return (a->flags[p] & 1u) ? a->slot[p] : a->slot[p] % n_l2_; }
    int renumber_ring_if_due();
    uint32_t take_l1_slot();
    void lru_unlink(uint32_t slot);
    void lru_push_mru(uint32_t slot);
    int move_to_l1(Allocation* a, uint32_t page);
    void queue_update(Allocation* a, uint32_t page, uint32_t and_mask, uint32_t or_mask, uint32_t slot);
    int flush_mirror();
    int settle();                                         // absorb every flush assignment (waits for the assign kernels)
    int quiesce();                                        // no fetch kernel running or queued
    int prepare_ring_op();
    uint32_t ring_busy_ = 0;                              // synchronous ring fetches in their (unlocked) wait
    void absorb(Flight& f);
    void harvest_flights();                               // non-blocking: absorb assigned flights, retire and count finished ones
    int wait_landed(const Allocation* a, uint64_t p0, uint64_t p1);   // pages of an in-flight flush have arrived
    // data movement
    int fetch_into_ring(Allocation* a, const std::vector<uint32_t>& pages, uint32_t* base_out);
    int fetch_into_slot(Allocation* a, uint32_t page, uint32_t slot);
    int fetch_range_copy_engine(Allocation* a, uint64_t first, uint64_t n, void* d_dst, bool f32, hipStream_t st);
    bool infer_layout(Allocation* a);
    int migrate_mx4(Allocation* a, uint64_t first, uint64_t n, uint32_t target_pool);      // tile-planar records (engine_relocate.cpp)
    int unpack(Allocation* a);                            // a compacted allocation back into fixed slots (before any write / migration)
    int settle_for_relocation(Allocation*& a, uint64_t handle);
    int write_groups(const uint64_t* handles, const uint64_t* firsts, const void* const* d_srcs, uint32_t n_groups, uint64_t step,
                     uint64_t n_each, hipStream_t s, bool same_allocation);
    int flush_group(int scheme, const uint32_t* const cols[5], uint32_t n, uint32_t W, uint32_t* n_issued);
    void reap(bool wait_all);
    int run_predictor_for_dirty();
    hipEvent_t get_event();
    void put_event(hipEvent_t e) { if (e) event_pool_.push_back(e); }
};

} // namespace speckv
