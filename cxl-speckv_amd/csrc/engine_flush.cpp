// cxl-speckv_amd/csrc/engine_flush.cpp -- speculative look-ahead: request queue, device-side flush, token predictor, verification (Engine members)
#include "engine_internal.hpp"

namespace speckv {

// --------------------------------------------------------------- prefetch
// The allocation an unbound request id addresses: the one that last received a layout; failing that the newest
// allocation that has one, else the newest allocation at all (the reference shim keeps a single live allocation and
// sends no geometry, vllm_speckv_backend.py:26-43).
Allocation* Engine::default_target()
{
    if (layout_handle_)
        if (Allocation* a = find(layout_handle_)) return a;
    Allocation* with_layout = nullptr;
    Allocation* any = nullptr;
    for (auto& kv : allocs_) {
        Allocation* a = kv.second.get();
        if (!a->n_pages) continue;
        if (!any || a->handle > any->handle) any = a;
        if (a->has_layout && (!with_layout || a->handle > with_layout->handle)) with_layout = a;
    }
    return with_layout ? with_layout : any;
}

int Engine::bind_request(uint32_t req, uint64_t handle, uint32_t local_req)
{
    if (handle == 0) { bindings_.erase(req); ++res_gen_; return SPECKV_OK; }
    if (!find(handle)) return SPECKV_ERR_GENERAL;
    bindings_[req] = Binding{handle, local_req};
    ++res_gen_;
    return SPECKV_OK;
}

// Resolve a request id to (table row, request index inside the allocation, limits).  Decode loops send the requests
// of one sequence back to back, so the last resolution is cached (res_gen_ changes whenever a binding, a layout or
// the set of allocations does).
bool Engine::resolve(uint32_t req)
{
    last_res_ = Resolved{};
    last_res_.req = req;
    last_res_.gen = res_gen_;
    Allocation* a = nullptr;
    uint32_t lr = req;
    if (!bindings_.empty()) {
        auto b = bindings_.find(req);
        if (b != bindings_.end()) { a = find(b->second.handle); lr = b->second.local_req; }
    }
    if (!a) a = default_target();
    if (!a || a->n_pages == 0 || a->row == kNoSlot) return false;
    if (!a->has_layout) { last_res_.no_geometry = bindings_.empty(); return false; }
    const Layout& L = a->layout;
    const uint64_t per_req = 2ull * L.num_tokens * L.num_layers * L.num_heads * L.head_dim * L.bytes_per_element;
    const uint64_t n_req = per_req ? (a->size_bytes + per_req - 1) / per_req : 0;
    if (lr >= n_req) return false;
    last_res_.row = a->row;
    last_res_.local = lr;
    last_res_.n_layers = L.num_layers;
    last_res_.scheme = a->scheme;
    const uint64_t row_bytes = static_cast<uint64_t>(L.num_heads) * L.head_dim * L.bytes_per_element;
    last_res_.W = static_cast<uint32_t>(row_bytes / kPageSize + 2);
    last_res_.ok = true;
    return true;
}

// One request joins the queue (already resolved: the flush only uploads and launches).
void Engine::enqueue(uint32_t req, uint32_t layer, uint32_t pos, uint32_t k)
{
    if (last_res_.req != req || last_res_.gen != res_gen_) (void)resolve(req);
    if (!last_res_.ok) {
        if (last_res_.no_geometry) q_unresolved_.push_back({req, layer, pos, k});   // geometry may still be learnt before the flush
        else ++q_dropped_;
        return;
    }
    if (layer >= last_res_.n_layers) { ++q_dropped_; return; }
    if (k > 16u) {                       // the candidate kernel walks at most 16 look-ahead positions per request
        static bool warned = false;
        if (!warned) { warned = true; SPECKV_ERR("speckv_prefetch: look-ahead depth %u clamped to 16 (reported once)", k); }
        k = 16u;
    }
    if (!q_req_.empty() && q_scheme_ != last_res_.scheme) {
        (void)prefetch_flush(nullptr);
        (void)resolve(req);
        // (a flush in progress on another thread makes this one a no-op: a request of another format cannot join its queue)
        if (!last_res_.ok || (!q_req_.empty() && q_scheme_ != last_res_.scheme)) { ++q_dropped_; return; }
    }
    q_scheme_ = last_res_.scheme;
    q_W_ = std::max(q_W_, last_res_.W);
    q_req_.push_back(last_res_.local);
    q_layer_.push_back(layer);
    q_pos_.push_back(pos);
    q_k_.push_back(k);
    q_row_.push_back(last_res_.row);
}

int Engine::prefetch(uint32_t req, uint16_t layer, uint32_t pos, uint32_t k,
                     const int32_t* tokens, uint32_t hist)
{
    if (null_) return SPECKV_OK;                            // submit_prefetch result ignored, speckv_allocator.cpp:89
    // the history feeds the token predictor (it never influences the addressing,
    // speculative_prefetcher.cpp:48): last 16 tokens, zero-padded at the front (lstm_predictor.cpp:44-51)
    if (d_emb_ && tokens && hist) {
        std::vector<int32_t> h(16, 0);
        const uint32_t take = hist < 16 ? hist : 16;
        for (uint32_t i = 0; i < take; ++i) h[16 - take + i] = tokens[hist - take + i];
        auto it = hist_.find(req);
        if (it == hist_.end() || it->second != h) { hist_[req] = h; hist_dirty_.push_back(req); }
    }
    // Without a known geometry (a caller that speaks only the reference's 8 functions) the layer count is learnt
    // from the calls themselves: the shim walks layers 0..L-1 per token (vllm_speckv_backend.py:116-118), so the
    // step is complete when the layer index falls back; flush then, not after a fixed count.
    if (!q_unresolved_.empty() && layer <= q_unresolved_.back().layer) (void)prefetch_flush(nullptr);
    max_layer_seen_ = std::max<uint32_t>(max_layer_seen_, layer);
    enqueue(req, layer, pos, k ? k : adapt_.depth());
    uint32_t thr = flush_threshold_;
    if (thr == 0) thr = last_res_.ok ? last_res_.n_layers : 4096u;
    if (q_req_.size() + q_unresolved_.size() >= thr) (void)prefetch_flush(nullptr);   // driver result ignored, as in the reference
    return SPECKV_OK;
}

int Engine::prefetch_batch(uint32_t n, const uint32_t* req, const uint16_t* layer,
                           const uint32_t* pos, const uint32_t* k)
{
    if (null_) return SPECKV_OK;
    const size_t want = q_req_.size() + n;
    q_req_.reserve(want); q_layer_.reserve(want); q_pos_.reserve(want); q_k_.reserve(want); q_row_.reserve(want);
    const uint32_t dflt = adapt_.depth();
    for (uint32_t i = 0; i < n; ++i) {
        max_layer_seen_ = std::max<uint32_t>(max_layer_seen_, layer[i]);
        enqueue(req[i], layer[i], pos[i], (k && k[i]) ? k[i] : dflt);
    }
    return SPECKV_OK;
}

// Geometry for callers that never sent one (the reference's allocate() sends none; its hardware derives addresses
// itself, prefetch_core.v:92-98).  Prefetch is only a cache fill, so an assumed geometry can cost bandwidth but never
// correctness: entry size from the first speckv_access (length_bytes = head_dim * bytes_per_element in the shim,
// vllm_speckv_backend.py:57-64; 256 if none was seen), kv heads from SPECKV_KV_HEADS (8), layers from the calls.
bool Engine::infer_layout(Allocation* a)
{
    if (!a || a->n_pages == 0) return false;
    const uint64_t entry = a->entry_bytes_seen ? a->entry_bytes_seen : 256u;
    const uint64_t H = std::max<uint64_t>(1, env_mb("SPECKV_KV_HEADS", 8));
    const uint64_t L = static_cast<uint64_t>(max_layer_seen_) + 1;
    const uint64_t denom = 2 * L * H * entry;
    if (a->size_bytes == 0 || a->size_bytes % denom) return false;
    const uint64_t T = a->size_bytes / denom;
    const uint32_t bpe = (entry % 2 == 0) ? 2u : 1u;
    std::vector<Req> keep;
    keep.swap(q_unresolved_);                           // set_layout flushes the queue: not while we are re-resolving it
    const int rc = set_layout(a->handle, static_cast<uint32_t>(T), static_cast<uint32_t>(L), static_cast<uint32_t>(H),
                              static_cast<uint32_t>(entry / bpe), bpe);
    keep.swap(q_unresolved_);
    if (rc != SPECKV_OK) return false;
    a->layout_inferred = true;
    SPECKV_ERR("speckv_prefetch: no geometry was given for handle %llu (speckv_ext_set_layout / SPECKV_LAYOUT); assuming "
               "tokens=%llu layers=%llu kv_heads=%llu entry=%llu B from the calls seen so far",
               static_cast<unsigned long long>(a->handle), static_cast<unsigned long long>(T),
               static_cast<unsigned long long>(L), static_cast<unsigned long long>(H), static_cast<unsigned long long>(entry));
    return true;
}

int Engine::prefetch_flush(uint32_t* n_issued)
{
    if (n_issued) *n_issued = 0;
    if (null_) return SPECKV_OK;
    if (in_flush_) return SPECKV_OK;
    if (q_req_.empty() && q_unresolved_.empty() && q_dropped_ == 0) return SPECKV_OK;
    DeviceScope device_scope(device_);
    RC_TRY(renumber_ring_if_due());
    in_flush_ = true;
    const bool timing = g_verbose;                     // (SPECKV_LOG=1)
    const auto t_a = std::chrono::steady_clock::now();
    if (!q_unresolved_.empty()) {          // requests that arrived before any geometry was known
        Allocation* dflt = default_target();
        if (dflt && !dflt->has_layout) (void)infer_layout(dflt);
        std::vector<Req> again;
        again.swap(q_unresolved_);
        for (const Req& r : again) {
            if (last_res_.req != r.req || last_res_.gen != res_gen_) (void)resolve(r.req);
            if (last_res_.ok) enqueue(r.req, r.layer, r.pos, r.k); else ++q_dropped_;
        }
        q_dropped_ += q_unresolved_.size();
        q_unresolved_.clear();
    }
    if (q_dropped_) {
        st_.prefetch_dropped += q_dropped_;
        if (!warned_no_layout_) {
            warned_no_layout_ = true;
            SPECKV_ERR("speckv_prefetch: %llu request(s) could not be addressed (no geometry for the allocation, unknown request "
                       "binding, or layer / request index out of range) and were dropped; see speckv_ext_set_layout, "
                       "speckv_ext_bind_request, SPECKV_LAYOUT (reported once; counted in speckv_ext_stats.prefetch_dropped)",
                       static_cast<unsigned long long>(q_dropped_));
        }
        q_dropped_ = 0;
    }
    int rc = SPECKV_OK;
    uint32_t issued_total = 0;
    // The queue moves into locals first: flush_group may let go of the ABI lock while it waits for the GPU, and a thread
    // that calls speckv_prefetch meanwhile appends to the live (now empty) queue -- its requests wait for the next flush
    // (in_flush_ makes a nested flush a no-op) instead of reallocating the columns under this one or being cleared by it.
    std::vector<uint32_t> c_req, c_layer, c_pos, c_k, c_row;
    c_req.swap(q_req_); c_layer.swap(q_layer_); c_pos.swap(q_pos_); c_k.swap(q_k_); c_row.swap(q_row_);
    const int scheme = q_scheme_;
    const uint32_t q_w = q_W_;
    q_W_ = 0;
    const size_t total = c_req.size();
    if (total) {
        // at most 2^24 candidate words per pipeline run (dedupe key)
        const uint32_t W = std::max<uint32_t>(q_w, 2u);
        const uint32_t max_n = std::max<uint32_t>(1u, ((1u << 24) - 1u) / (32u * W));
        for (size_t b = 0; b < total && rc == SPECKV_OK; b += max_n) {
            const uint32_t n = static_cast<uint32_t>(std::min<size_t>(max_n, total - b));
            const uint32_t* cols[5] = {c_req.data() + b, c_layer.data() + b, c_pos.data() + b, c_k.data() + b, c_row.data() + b};
            uint32_t m = 0;
            rc = flush_group(scheme, cols, n, W, n_issued ? &m : nullptr);
            issued_total += m;
        }
    }
    in_flush_ = false;
    if (q_req_.empty() && q_req_.capacity() < c_req.capacity()) {      // keep the columns' capacity for the next step
        c_req.clear(); c_layer.clear(); c_pos.clear(); c_k.clear(); c_row.clear();
        c_req.swap(q_req_); c_layer.swap(q_layer_); c_pos.swap(q_pos_); c_k.swap(q_k_); c_row.swap(q_row_);
    }
    if (n_issued) *n_issued = issued_total;
    if (timing) {
        const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_a).count();
        fprintf(stderr, "[speckv timing] flush submit: %zu requests, %.1f us (host time; the GPU pipeline runs asynchronously)\n", total, us);
    }
    if (rc == SPECKV_OK) rc = run_predictor_for_dirty();
    return rc;
}

// One run of the device-side flush pipeline for requests of allocations that share a compression scheme:
// upload the requests, candidates -> dedupe -> ring assignment -> compaction (kernels.hip), then ONE fetch launch
// that reads its block count and first slot from device memory.  Nothing comes back to the host but 16 bytes
// (FlushResult, written to pinned memory by the assign kernel), read when somebody needs them.
// The request columns of a flush go to the device through a copy KERNEL on the flush's stream: on an idle stream a copy
// engine's transfer is followed by a cross-engine dependency of about 12 us in front of the first flush kernel.  (For the
// descriptors of the batch attention, between back-to-back launches, the two measured the same: they stay with the engine.)
// `staged` is pinned (hipHostMalloc) and padded to a multiple of 16 bytes, as is `dst`.
hipError_t Engine::upload_pinned(void* dst, const void* staged, size_t bytes, hipStream_t s)
{
    void* staged_dev = nullptr;
    const hipError_t e = hipHostGetDevicePointer(&staged_dev, const_cast<void*>(staged), 0);
    if (e != hipSuccess) return e;
    return launch_copy16(staged_dev, dst, bytes, s);
}

int Engine::flush_group(int scheme, const uint32_t* const cols[5], uint32_t n, uint32_t W, uint32_t* n_issued)
{
    reap(false);                               // a decode loop calls nothing else that retires finished flights and their events
    if (flights_.size() >= kMaxFlights) { RC_TRY(settle()); if (flights_.size() >= kMaxFlights) { RC_TRY(wait_stream()); RC_TRY(settle()); } }
    RC_TRY(flush_mirror());
    RC_TRY(order_after_writes());              // records appended on caller streams are in place before they are fetched
    if (++flush_epoch_ > 255u) {               // 8-bit epoch in the dedupe stamps: start over with clean stamps
        flush_epoch_ = 1;
        for (auto& kv : allocs_)
            if (kv.second->d_stamp) HIP_TRY(hipMemsetAsync(kv.second->d_stamp, 0, kv.second->n_pages * sizeof(uint32_t), stream_));
    }
    const uint64_t words = static_cast<uint64_t>(n) * 32u * W;
    const uint32_t n_w = static_cast<uint32_t>((words + 63u) >> 6);
    const uint32_t max_take = static_cast<uint32_t>(std::min<uint64_t>(n_l2_ / 2, words));   // never let one flush wipe the whole ring
    // The pages' host-visible words (one PCIe transaction each) are stored by the scatter kernel, in front of the fetch,
    // or -- large flushes -- by the fetch launch itself, spread over it: 20 480 requests -> 122 880 pages 0.213 -> 0.198 ms
    // until landed; at 8 192 requests -> 19 095 pages the fetch is too short to hide them (0.102 -> 0.104).
    const bool words_by_fetch = flush_words_mode_ == 2 || (flush_words_mode_ == 0 && words > (1u << 19));
    const size_t bytes = (5ull * n + words + 2ull * n_w + 8 + 6ull * max_take + 4 + (words_by_fetch ? 2ull * max_take + 2 : 0)) * sizeof(uint32_t);     // + descriptors (16 B), destinations (8 B), word addresses (8 B)
    uint32_t* buf = static_cast<uint32_t*>(scratch(s_flush_, bytes));
    if (!buf) return SPECKV_ERR_NOMEM;
    // request upload through a pinned slot (4 in rotation, each guarded by an event): no stream sync
    const size_t up = 5ull * n * sizeof(uint32_t);
    if (req_stage_bytes_ < up) {
        if (req_stage_) { RC_TRY(wait_stream()); (void)hipHostFree(req_stage_); req_stage_ = nullptr; }
        req_stage_bytes_ = std::max<size_t>(up * 2, 1 << 20);
        HIP_TRY(hipHostMalloc(&req_stage_, req_stage_bytes_ * 4, hipHostMallocDefault));
        for (auto& ev : req_stage_ev_)
            if (!ev) HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    }
    const int slot = req_stage_next_;
    req_stage_next_ = (slot + 1) & 3;
    RC_TRY(wait_event(req_stage_ev_[slot]));
    void* staged = static_cast<uint8_t*>(req_stage_) + static_cast<size_t>(slot) * req_stage_bytes_;
    for (int c = 0; c < 5; ++c) memcpy(static_cast<uint32_t*>(staged) + static_cast<size_t>(c) * n, cols[c], n * sizeof(uint32_t));
    // The columns are pulled over by a copy KERNEL on the flush's stream (16 bytes per lane from the pinned slot): a copy
    // engine's upload cost 9 us plus a 12 us cross-engine dependency in front of the first flush kernel -- 8 192 requests
    // 0.102 -> 0.094 ms until landed, 20 480 requests unchanged (the copy-engine form is gone: DESIGN_HISTORY.md).
    HIP_TRY(upload_pinned(buf, staged, up, stream_));
    HIP_TRY(hipEventRecord(req_stage_ev_[slot], stream_));

    const uint32_t rs = res_next_++ % kResSlots;
    void* dp = nullptr;
    HIP_TRY(hipHostGetDevicePointer(&dp, res_ring_, 0));
    FlushArgs f{};
    f.tab = d_tab_;
    f.n = n;
    f.W = W;
    f.req = buf; f.layer = buf + n; f.pos = buf + 2ull * n; f.depth = buf + 3ull * n; f.row = buf + 4ull * n;
    f.epoch = flush_epoch_;
    f.cand = buf + 5ull * n;
    f.wave_tot = f.cand + words;
    f.final_entry = reinterpret_cast<PageEntry*>((reinterpret_cast<uintptr_t>(f.wave_tot + 2ull * n_w + 8) + 15u) & ~uintptr_t(15));
    f.final_dst = reinterpret_cast<uint64_t*>(f.final_entry + max_take);
    if (words_by_fetch) f.final_host = reinterpret_cast<uint32_t**>(f.final_dst + max_take);
    f.ring_owner = d_owner_;
    f.ring_base = cache_base_;
    f.max_take = max_take;
    f.n_l2 = n_l2_;
    f.hand = d_hand_;
    f.result_dev = d_res_ring_ + rs;
    f.result_host = static_cast<FlushResult*>(dp) + rs;
    res_ring_[rs] = FlushResult{0, 0, 0, 0};
    HIP_TRY(launch_flush_pipeline(f, stream_));
    Flight fl;
    fl.assigned = get_event();
    fl.done = get_event();
    fl.result = res_ring_ + rs;
    struct EventGuard {                        // the flight's events go back to the pool on every error path
        Engine* e; Flight* f; bool keep = false;
        ~EventGuard() { if (!keep) { e->put_event(f->assigned); e->put_event(f->done); } }
    } guard{this, &fl};
    if (!fl.assigned || !fl.done) return SPECKV_ERR_DRIVER;
    HIP_TRY(hipEventRecord(fl.assigned, stream_));

    // the fetch itself: plain list form (the scatter kernel left a record descriptor and a destination per block)
    CodecArgs c{};
    c.trusted = 1;
    c.entries = f.final_entry;
    c.data_list = f.final_dst;
    c.n = max_take;
    c.n_dev = &f.result_dev->m;
    if (words_by_fetch) { c.host_words = f.final_host; c.seq0_dev = &f.result_dev->seq; }
    c.scheme = scheme;
    c.quant_mode = quant_mode_;
    HIP_TRY(launch_decompress(c, stream_));
    HIP_TRY(hipEventRecord(fl.done, stream_));
    guard.keep = true;
    flights_.push_back(fl);
    if (n_issued) {                       // the caller wants the page count now: wait for the assign kernel (not the data)
        RC_TRY(settle());
        *n_issued = fl.result->m;
    }
    return SPECKV_OK;
}

// ----------------------------------------------------------------- predictor
int Engine::predictor_load(const float* emb, const float* wout, uint32_t vocab, bool on_device)
{
    if (null_) return no_data_path("speckv_ext_predictor_load");
    if (!emb || !wout || vocab < 8) return SPECKV_ERR_INVAL;
    DeviceScope device_scope(device_);
    HIP_TRY(hipDeviceSynchronize());
    if (d_emb_) { (void)hipFree(d_emb_); d_emb_ = nullptr; }
    if (d_wout_) { (void)hipFree(d_wout_); d_wout_ = nullptr; }
    const size_t eb = static_cast<size_t>(vocab) * 64 * sizeof(float), wb = static_cast<size_t>(vocab) * 128 * sizeof(float);
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&d_emb_), eb));
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&d_wout_), arranged_wout_bytes(vocab)));
    const hipMemcpyKind kind = on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
    if (on_device) HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(d_emb_, emb, eb, kind));
    {
        // the output layer is kept in the order its kernel reads it (k_arrange_wout); a caller's device copy is read in place
        float* staged = nullptr;
        if (!on_device) {
            HIP_TRY(hipMalloc(reinterpret_cast<void**>(&staged), wb));
            const hipError_t ce = hipMemcpy(staged, wout, wb, hipMemcpyHostToDevice);
            if (ce != hipSuccess) { (void)hipFree(staged); HIP_TRY(ce); }
        }
        hipError_t e = launch_arrange_wout(on_device ? wout : staged, d_wout_, vocab, stream_);
        if (e == hipSuccess) e = hipStreamSynchronize(stream_);
        if (staged) (void)hipFree(staged);
        HIP_TRY(e);
    }
    vocab_ = vocab;
    pending_pred_.active = false;                            // (the device was synchronised above: nothing is in flight)
    hist_.clear(); pred_.clear(); hist_dirty_.clear();
    for (float* p : lstm_bufs_) (void)hipFree(p);       // back to the reference's cell
    lstm_bufs_.clear();
    lstm_ = LstmParams{};
    return SPECKV_OK;
}

// A real LSTM cell for the predictor (the reference's cell ignores its weights).  PyTorch nn.LSTM layout.
int Engine::predictor_load_lstm(const float* emb, uint32_t vocab, uint32_t n_layers, const float* const* w_ih, const float* const* w_hh,
                                const float* const* b_ih, const float* const* b_hh, const float* wout, const float* out_bias, bool on_device)
{
    if (null_) return no_data_path("speckv_ext_predictor_load_lstm");
    if (!emb || !wout || vocab < 8 || n_layers == 0 || n_layers > 4 || !w_ih || !w_hh || !b_ih || !b_hh) return SPECKV_ERR_INVAL;
    for (uint32_t l = 0; l < n_layers; ++l)
        if (!w_ih[l] || !w_hh[l] || !b_ih[l] || !b_hh[l]) return SPECKV_ERR_INVAL;
    RC_TRY(predictor_load(emb, wout, vocab, on_device));       // embedding + output layer, and the old cell's buffers released
    DeviceScope device_scope(device_);
    const hipMemcpyKind kind = on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
    auto upload = [&](const float* src, size_t n, float** out) -> int {
        float* d = nullptr;
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&d), n * sizeof(float)));
        lstm_bufs_.push_back(d);
        HIP_TRY(hipMemcpy(d, src, n * sizeof(float), kind));
        *out = d;
        return SPECKV_OK;
    };
    LstmParams p{};
    // the cell kernel reads weights as [register][thread] (coalesced over its 512 threads, lstm_arranged_index): arranged
    // here, once, through the host (under 1 MB per layer)
    auto upload_transposed = [&](const float* src, size_t rows, size_t cols, float** out) -> int {
        std::vector<float> a(rows * cols), t(rows * cols);
        HIP_TRY(hipMemcpy(a.data(), src, a.size() * sizeof(float), on_device ? hipMemcpyDeviceToHost : hipMemcpyHostToHost));
        for (size_t r = 0; r < rows; ++r)
            for (size_t c = 0; c < cols; ++c)
                t[lstm_arranged_index(static_cast<uint32_t>(r), static_cast<uint32_t>(c), static_cast<uint32_t>(cols))] = a[r * cols + c];
        float* d = nullptr;
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&d), t.size() * sizeof(float)));
        lstm_bufs_.push_back(d);
        HIP_TRY(hipMemcpy(d, t.data(), t.size() * sizeof(float), hipMemcpyHostToDevice));
        *out = d;
        return SPECKV_OK;
    };
    for (uint32_t l = 0; l < n_layers; ++l) {
        const size_t in_dim = l == 0 ? 64 : 128;
        float *wi = nullptr, *wh = nullptr, *bi = nullptr;
        RC_TRY(upload_transposed(w_ih[l], 512, in_dim, &wi));
        RC_TRY(upload_transposed(w_hh[l], 512, 128, &wh));
        // bias = b_ih + b_hh, summed once on the host side of the copy (exact: one fp32 addition, as the cell would do)
        std::vector<float> a(512), b(512);
        HIP_TRY(hipMemcpy(a.data(), b_ih[l], 512 * sizeof(float), on_device ? hipMemcpyDeviceToHost : hipMemcpyHostToHost));
        HIP_TRY(hipMemcpy(b.data(), b_hh[l], 512 * sizeof(float), on_device ? hipMemcpyDeviceToHost : hipMemcpyHostToHost));
        for (int i = 0; i < 512; ++i) a[i] += b[i];
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&bi), 512 * sizeof(float)));
        lstm_bufs_.push_back(bi);
        HIP_TRY(hipMemcpy(bi, a.data(), 512 * sizeof(float), hipMemcpyHostToDevice));
        p.w_ih_t[l] = wi; p.w_hh_t[l] = wh; p.bias[l] = bi;
    }
    if (out_bias) { float* ob = nullptr; RC_TRY(upload(out_bias, vocab, &ob)); p.out_bias = ob; }
    p.layers = n_layers;
    lstm_ = p;
    return SPECKV_OK;
}

int Engine::predict_batch(uint32_t n, const int32_t* d_hist, uint32_t k, int32_t* d_tok, float* d_conf, hipStream_t s)
{
    if (null_) return no_data_path("speckv_ext_predict_batch");
    if (!d_emb_) return SPECKV_ERR_INVAL;
    if (n == 0) return SPECKV_OK;
    if (!d_hist || !d_tok || !d_conf || k == 0 || k > 8) return SPECKV_ERR_INVAL;
    DeviceScope device_scope(device_);
    float* hid = static_cast<float*>(scratch(s_hid_, static_cast<size_t>(n) * 128 * sizeof(float), s));
    float* logits = static_cast<float*>(scratch(s_logits_, static_cast<size_t>(n) * vocab_ * sizeof(float), s));
    void* ws = scratch(s_predict_ws_, predict_ws_bytes(n, vocab_), s);
    if (!hid || !logits || !ws) return SPECKV_ERR_NOMEM;
    hipStream_t st = s ? s : stream_;
    HIP_TRY(launch_predict(n, d_hist, d_emb_, d_wout_, vocab_, 2, k, hid, logits, ws, d_tok, d_conf, st, &lstm_));
    if (!s) HIP_TRY(hipStreamSynchronize(stream_));
    return SPECKV_OK;
}

int Engine::harvest_predictions()
{
    if (!pending_pred_.active) return SPECKV_OK;
    const uint64_t gen = pending_pred_.gen;
    RC_TRY(wait_event(pred_ev_));                            // may let go of the ABI lock: another thread may have harvested, or started the next one
    if (!pending_pred_.active || pending_pred_.gen != gen) return SPECKV_OK;
    const uint32_t n = static_cast<uint32_t>(pending_pred_.reqs.size()), k = pending_pred_.k;
    const int32_t* tok = h_pred_io_ + static_cast<size_t>(n) * 16;
    for (uint32_t i = 0; i < n; ++i) pred_[pending_pred_.reqs[i]].assign(tok + static_cast<size_t>(i) * k, tok + static_cast<size_t>(i + 1) * k);
    pending_pred_.active = false;
    return SPECKV_OK;
}

int Engine::run_predictor_for_dirty()
{
    if (!d_emb_ || hist_dirty_.empty()) { hist_dirty_.clear(); return SPECKV_OK; }
    RC_TRY(harvest_predictions());                           // the one before (long finished as a rule): its staging is reused
    if (hist_dirty_.empty()) return SPECKV_OK;               // (another thread's flush took them while we waited)
    std::sort(hist_dirty_.begin(), hist_dirty_.end());
    hist_dirty_.erase(std::unique(hist_dirty_.begin(), hist_dirty_.end()), hist_dirty_.end());
    const uint32_t n = static_cast<uint32_t>(hist_dirty_.size());
    uint32_t k = adapt_.depth();
    if (k > 8) k = 8;
    if (k == 0) k = 1;
    if (!pred_stream_) HIP_TRY(hipStreamCreateWithFlags(&pred_stream_, hipStreamNonBlocking));
    if (!pred_ev_) HIP_TRY(hipEventCreateWithFlags(&pred_ev_, hipEventDisableTiming));
    const size_t hist_words = static_cast<size_t>(n) * 16, io_bytes = (hist_words + static_cast<size_t>(n) * k) * sizeof(int32_t);
    if (io_bytes > h_pred_cap_) {
        if (h_pred_io_) { (void)hipHostFree(h_pred_io_); h_pred_io_ = nullptr; h_pred_cap_ = 0; }
        const size_t want = std::max<size_t>(io_bytes + (io_bytes >> 1), 1 << 16);
        HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&h_pred_io_), want, hipHostMallocDefault));
        h_pred_cap_ = want;
    }
    for (uint32_t i = 0; i < n; ++i) memcpy(h_pred_io_ + static_cast<size_t>(i) * 16, hist_[hist_dirty_[i]].data(), 16 * sizeof(int32_t));
    int32_t* d_h = static_cast<int32_t*>(scratch(s_hist_, hist_words * sizeof(int32_t), pred_stream_));
    uint8_t* d_p = static_cast<uint8_t*>(scratch(s_pred_, static_cast<size_t>(n) * k * (sizeof(int32_t) + sizeof(float)), pred_stream_));
    if (!d_h || !d_p) return SPECKV_ERR_NOMEM;
    int32_t* d_tok = reinterpret_cast<int32_t*>(d_p);
    float* d_conf = reinterpret_cast<float*>(d_p + static_cast<size_t>(n) * k * sizeof(int32_t));
    HIP_TRY(hipMemcpyAsync(d_h, h_pred_io_, hist_words * sizeof(int32_t), hipMemcpyHostToDevice, pred_stream_));
    int rc = predict_batch(n, d_h, k, d_tok, d_conf, pred_stream_);
    if (rc != SPECKV_OK) return rc;
    HIP_TRY(hipMemcpyAsync(h_pred_io_ + hist_words, d_tok, static_cast<size_t>(n) * k * sizeof(int32_t), hipMemcpyDeviceToHost, pred_stream_));
    HIP_TRY(hipEventRecord(pred_ev_, pred_stream_));
    pending_pred_.reqs.swap(hist_dirty_);
    pending_pred_.k = k;
    ++pending_pred_.gen;
    pending_pred_.active = true;
    hist_dirty_.clear();
    return SPECKV_OK;
}

int Engine::prefetch_lookup(uint64_t handle, uint32_t n, const uint32_t* d_req, const uint32_t* d_layer,
                            const uint32_t* d_pos, const uint32_t* d_k, uint32_t* d_out, uint32_t cap,
                            uint32_t* d_count, hipStream_t s)
{
    if (null_) return no_data_path("speckv_ext_prefetch_lookup");
    Allocation* a = find(handle);
    if (!a) return SPECKV_ERR_GENERAL;
    if (!a->has_layout) return SPECKV_ERR_INVAL;
    DeviceScope device_scope(device_);
    RC_TRY(quiesce());                     // the residency mirror the kernel filters with is final
    if ((a = find(handle)) == nullptr) return SPECKV_ERR_GENERAL;
    RC_TRY(flush_mirror());
    uint32_t* scr = static_cast<uint32_t*>(scratch(s_tmp_, (2ull * n + 4) * sizeof(uint32_t), s));
    if (!scr) return SPECKV_ERR_NOMEM;
    hipStream_t st = s ? s : stream_;
    if (s) RC_TRY(wait_stream());          // the mirror updates above ran on the engine stream
    HIP_TRY(launch_prefetch_lookup(a->layout, n, d_req, d_layer, d_pos, d_k, a->d_flags, d_out, cap, d_count, scr, st));
    note_use(a, s);
    if (!s) RC_TRY(wait_stream());
    return SPECKV_OK;
}

int Engine::verify(uint32_t req, int32_t actual, const int32_t* pred, uint32_t n,
                   uint32_t* was_hit, uint32_t* new_depth)
{
    // no list given: verify against the prediction the engine made from the request's last history
    std::vector<int32_t> own;
    if (!pred || n == 0) {
        RC_TRY(harvest_predictions());                       // the last flush's prediction may still be on its way
        auto it = pred_.find(req);
        if (it == pred_.end()) return SPECKV_ERR_INVAL;
        own = it->second;
        pred = own.data();
        n = static_cast<uint32_t>(own.size());
    }
    bool hit = false;                                        // speculative_prefetcher.cpp:84-96
    for (uint32_t i = 0; i < n; ++i) if (pred[i] == actual) { hit = true; break; }
    if (!hit) st_.mispredictions++; else st_.successful_prefetches++;
    adapt_.update(hit);
    if (was_hit) *was_hit = hit ? 1u : 0u;
    if (new_depth) *new_depth = adapt_.depth();
    return SPECKV_OK;
}

} // namespace speckv
