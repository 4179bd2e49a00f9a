// cxl-speckv_amd/csrc/engine.cpp -- see engine.hpp
#include "engine_internal.hpp"
#include "tuning.hpp"

namespace speckv {


// ------------------------------------------------------------------ depth
void AdaptiveDepth::update(bool ok)
{   // speculative_prefetcher.cpp:98-120
    hist_.push_back(ok ? 1 : 0);
    if (hist_.size() > 100) hist_.pop_front();
    if (hist_.size() >= 10) {
        double acc = 0.0;
        for (size_t i = hist_.size() - 10; i < hist_.size(); ++i) acc += hist_[i] ? 1.0 : 0.0;
        acc /= 10.0;
        if (acc > 0.95 && depth_ < 8) ++depth_;
        else if (acc < 0.85 && depth_ > 2) --depth_;
    }
}

// ------------------------------------------------------------------- open
std::unique_ptr<Engine> Engine::open(const char* dev_path, int* status)
{
    std::unique_ptr<Engine> e(new Engine());
    const std::string path = dev_path ? dev_path : "";
    e->pool_devs_ = parse_int_list(getenv("SPECKV_POOL_DEVICES"));
    if (path == "/dev/null") {            // the reference's fake device
        e->null_ = true;
        *status = SPECKV_OK;
        return e;
    }
    int device = -1;
    if (path.rfind("hip:", 0) == 0) device = atoi(path.c_str() + 4);
    else if (path.rfind("/dev/speckv", 0) == 0 && path.size() > 11) device = atoi(path.c_str() + 11);
    if (const char* env = getenv("SPECKV_DEVICE")) device = atoi(env);
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) {
        (void)hipGetLastError();
        SPECKV_ERR("speckv_init(\"%s\"): no usable HIP device; the engine has no CPU data path "
                   "(use \"/dev/null\" for page-table-only emulation)", path.c_str());
        *status = SPECKV_ERR_GENERAL;     // reference: open() failure -> exception -> -1
        return nullptr;
    }
    if (device < 0) { if (hipGetDevice(&device) != hipSuccess) device = 0; }
    if (device >= count) {
        SPECKV_ERR("speckv_init(\"%s\"): HIP device %d does not exist (%d visible)", path.c_str(), device, count);
        *status = SPECKV_ERR_GENERAL;
        return nullptr;
    }
    int rc = e->init_hip(device);
    if (rc != SPECKV_OK) { *status = SPECKV_ERR_GENERAL; return nullptr; }
    *status = SPECKV_OK;
    return e;
}

int Engine::init_hip(int device)
{
    device_ = device;
    HIP_TRY(hipSetDevice(device_));
    HIP_TRY(hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking));
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&d_count_), 64));

    // pool devices: default = the compute GPU itself; SPECKV_POOL_DEVICES="1,2,3"
    // places the pool in peer HBM reached over xGMI.
    if (pool_devs_.empty()) pool_devs_.push_back(device_);
    int count = 0;
    HIP_TRY(hipGetDeviceCount(&count));
    const size_t slab = env_mb("SPECKV_SLAB_MB", 1024) << 20;
    const size_t cap = env_mb("SPECKV_POOL_CAP_MB", 0) << 20;
    for (int d : pool_devs_) {
        if (d < 0 || d >= count) { SPECKV_ERR("pool device %d does not exist", d); return SPECKV_ERR_DRIVER; }
        if (d != device_) {
            int can = 0;
            HIP_TRY(hipDeviceCanAccessPeer(&can, device_, d));
            if (!can) { SPECKV_ERR("device %d cannot access peer %d over xGMI", device_, d); return SPECKV_ERR_DRIVER; }
            hipError_t pe = hipDeviceEnablePeerAccess(d, 0);
            if (pe != hipSuccess && pe != hipErrorPeerAccessAlreadyEnabled) {
                SPECKV_ERR("hipDeviceEnablePeerAccess(%d) failed: %s", d, hipGetErrorString(pe));
                return SPECKV_ERR_DRIVER;
            }
            (void)hipGetLastError();
        }
        pools_.emplace_back(new SlabPool(d, slab, cap));
    }
    if (pools_.size() > 255) { SPECKV_ERR("at most 255 pool devices"); return SPECKV_ERR_DRIVER; }

    // cache arena on the compute GPU (reference defaults 12 GB L1 / 3 GB L2,
    // cxl_memory_manager.h:42-44; ours are env-tunable and allocated up front)
    const size_t l2_mb = env_mb("SPECKV_L2_MB", 256), l1_mb = env_mb("SPECKV_L1_MB", 256);
    if (const char* e = getenv("SPECKV_RING_SEQ_LIMIT")) ring_seq_limit_ = static_cast<uint32_t>(strtoul(e, nullptr, 0));   // tests
    if (const char* e = getenv("SPECKV_FLUSH_HOST_WORDS")) flush_words_mode_ = e[0] == 's' ? 1 : 2;                          // scatter / fetch (tests, A/B)
    // the remaining switches of this file, read here once (nothing behind an entry point walks the environment)
    dbg_unordered_scratch_ = getenv("SPECKV_DEBUG_UNORDERED_SCRATCH") != nullptr;        // test hook: shows that the ordering test can fail
    access_spin_ok_ = getenv("SPECKV_ACCESS_NO_SPIN") == nullptr;
    if (const char* e = getenv("SPECKV_LAYOUT")) default_layout_ = parse_int_list(e);     // T,L,H,D,bpe for callers of the reference's 8 functions
    (void)tuning();                                                                       // (and the launch-form switches: tuning.hpp)
    n_l2_ = static_cast<uint32_t>((l2_mb << 20) / kPageSize);
    n_l1_ = static_cast<uint32_t>((l1_mb << 20) / kPageSize);
    if (n_l2_ < 64) n_l2_ = 64;
    if (n_l1_ < 16) n_l1_ = 16;
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&cache_base_), static_cast<size_t>(n_l2_ + n_l1_) * kPageSize));
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&d_owner_), static_cast<size_t>(n_l2_) * sizeof(uint64_t)));
    HIP_TRY(hipMemsetAsync(d_owner_, 0xFF, static_cast<size_t>(n_l2_) * sizeof(uint64_t), stream_));
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&d_hand_), 64));
    HIP_TRY(hipMemsetAsync(d_hand_, 0, 64, stream_));
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&d_done_count_), 64));
    HIP_TRY(hipMemsetAsync(d_done_count_, 0, 64, stream_));
    HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&h_done_), 64, hipHostMallocMapped | hipHostMallocPortable));
    *h_done_ = 0;
    {
        void* dp = nullptr;
        HIP_TRY(hipHostGetDevicePointer(&dp, h_done_, 0));
        h_done_dev_ = static_cast<uint32_t*>(dp);
    }
    l1_owner_.assign(n_l1_, Owner{nullptr, 0});
    lru_prev_.assign(n_l2_ + n_l1_, UINT32_MAX);
    lru_next_.assign(n_l2_ + n_l1_, UINT32_MAX);
    l1_free_.reserve(n_l1_);
    for (uint32_t i = 0; i < n_l1_; ++i) l1_free_.push_back(n_l2_ + n_l1_ - 1 - i);

    // device allocation table, flush result words, mirror-update ring
    tab_cap_ = 4096;
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&d_tab_), tab_cap_ * sizeof(DevAlloc)));
    HIP_TRY(hipMemsetAsync(d_tab_, 0, tab_cap_ * sizeof(DevAlloc), stream_));
    HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&h_tab_), tab_cap_ * sizeof(DevAlloc), hipHostMallocDefault));
    memset(h_tab_, 0, tab_cap_ * sizeof(DevAlloc));
    row_owner_.assign(tab_cap_, nullptr);
    for (uint32_t i = 0; i < tab_cap_; ++i) free_rows_.push_back(tab_cap_ - 1 - i);
    HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&res_ring_), kResSlots * sizeof(FlushResult), hipHostMallocMapped | hipHostMallocPortable));
    memset(res_ring_, 0, kResSlots * sizeof(FlushResult));
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&d_res_ring_), kResSlots * sizeof(FlushResult)));
    HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&upd_ring_), kUpdCap * sizeof(MirrorUpdate), hipHostMallocMapped | hipHostMallocPortable));
    upd_cap_ = kUpdCap;
    HIP_TRY(hipEventCreateWithFlags(&upd_event_, hipEventDisableTiming));
    HIP_TRY(hipStreamSynchronize(stream_));

    st_.cache_bytes_reserved = static_cast<uint64_t>(n_l2_ + n_l1_) * kPageSize;
    st_.n_pool_devices = static_cast<uint32_t>(pools_.size());
    if (const char* env = getenv("SPECKV_PREFETCH_BATCH")) flush_threshold_ = static_cast<uint32_t>(atoi(env));
    SPECKV_LOGV("opened HIP device %d: %zu pool device(s), L2 %u slots, L1 %u slots", device_, pools_.size(), n_l2_, n_l1_);
    return SPECKV_OK;
}

Engine::~Engine()
{
    if (null_) return;
    DeviceScope device_scope(device_);
    (void)hipDeviceSynchronize();
    for (auto& f : flights_) { if (f.assigned) (void)hipEventDestroy(f.assigned); if (f.done) (void)hipEventDestroy(f.done); }
    for (auto& b : inflight_) (void)hipEventDestroy(b.ev);
    for (auto& w : write_evs_) if (w.ev) (void)hipEventDestroy(w.ev);
    for (auto ev : event_pool_) (void)hipEventDestroy(ev);
    for (auto& z : zombies_) { release_allocation(z.a.get()); if (z.engine_ev) (void)hipEventDestroy(z.engine_ev); }
    zombies_.clear();
    for (auto& kv : allocs_) release_allocation(kv.second.get());
    allocs_.clear();
    if (d_emb_) (void)hipFree(d_emb_);
    if (d_wout_) (void)hipFree(d_wout_);
    for (float* p : lstm_bufs_) (void)hipFree(p);
    for (Scratch* s : {&s_pages_, &s_req_, &s_tmp_, &s_stage_, &s_flush_, &s_hid_, &s_logits_, &s_predict_ws_, &s_hist_, &s_pred_, &s_attn_, &s_attn_seq_})
        if (s->p) (void)hipFree(s->p);
    for (void* p : retired_) (void)hipFree(p);
    for (auto& l : lanes_) {
        if (l.s) (void)hipStreamDestroy(l.s);
        for (auto ev : l.copied) if (ev) (void)hipEventDestroy(ev);
    }
    for (int b = 0; b < 2; ++b) {
        if (stage_[b]) (void)hipFree(stage_[b]);
        if (stage_free_[b]) (void)hipEventDestroy(stage_free_[b]);
    }
    if (d_count_) (void)hipFree(d_count_);
    if (d_zero_page_) (void)hipFree(d_zero_page_);
    if (seq_ring_.base) (void)hipHostFree(seq_ring_.base);
    for (auto ev : seq_ring_.ev) if (ev) (void)hipEventDestroy(ev);
    if (grp_ring_.base) (void)hipHostFree(grp_ring_.base);
    for (auto ev : grp_ring_.ev) if (ev) (void)hipEventDestroy(ev);
    if (d_groups_) (void)hipFree(d_groups_);
    if (req_stage_) (void)hipHostFree(req_stage_);
    for (auto ev : req_stage_ev_) if (ev) (void)hipEventDestroy(ev);
    if (res_ring_) (void)hipHostFree(res_ring_);
    if (d_res_ring_) (void)hipFree(d_res_ring_);
    if (upd_ring_) (void)hipHostFree(upd_ring_);
    if (upd_event_) (void)hipEventDestroy(upd_event_);
    if (d_tab_) (void)hipFree(d_tab_);
    if (h_tab_) (void)hipHostFree(h_tab_);
    if (d_owner_) (void)hipFree(d_owner_);
    if (d_hand_) (void)hipFree(d_hand_);
    if (d_done_count_) (void)hipFree(d_done_count_);
    if (h_done_) (void)hipHostFree(h_done_);
    if (pred_stream_) (void)hipStreamDestroy(pred_stream_);
    if (pred_ev_) (void)hipEventDestroy(pred_ev_);
    if (h_pred_io_) (void)hipHostFree(h_pred_io_);
    if (cache_base_) (void)hipFree(cache_base_);
    pools_.clear();
    if (copy_stream_) (void)hipStreamDestroy(copy_stream_);
    if (stream_) (void)hipStreamDestroy(stream_);
}

Allocation* Engine::find(uint64_t h)
{
    auto it = allocs_.find(h);
    return it == allocs_.end() ? nullptr : it->second.get();
}

// Scratch buffers grow on demand.  A buffer that a stream capture has used is never freed (the captured graph
// keeps its address): growth then retires it instead, and growth DURING a capture is refused (nullptr) -- warm the
// call up once outside the capture, as with any graph-captured library call.
// A scratch buffer is ONE buffer: a call that uses it on another stream than the previous user's is ordered behind
// that user (an event recorded at the old stream's tail; nothing when callers stay on one stream).  Captures are left
// alone: a captured call is ordered by whatever launches its graph.
void* Engine::scratch(Scratch& s, size_t bytes, hipStream_t user)
{
    const bool capturing = is_capturing(user);
    hipStream_t now = user ? user : stream_;
    const bool unordered = dbg_unordered_scratch_;
    if (s.last && s.last != now && s.p && !capturing && !unordered && !is_capturing(s.last)) {
        if (hipEvent_t ev = get_event()) {
            if (hipEventRecord(ev, s.last) != hipSuccess || hipStreamWaitEvent(now, ev, 0) != hipSuccess) {
                (void)hipGetLastError();
                (void)hipDeviceSynchronize();
            }
            event_pool_.push_back(ev);
        } else {
            (void)hipDeviceSynchronize();
        }
    }
    if (!capturing) s.last = now;
    if (bytes <= s.cap) { if (capturing) s.in_graph = true; return s.p; }
    if (capturing) {
        SPECKV_ERR("a call inside a stream capture needs %zu bytes of scratch but %zu are reserved: run it once outside the capture first",
                   bytes, s.cap);
        return nullptr;
    }
    if (s.p) {
        if (s.in_graph) retired_.push_back(s.p);
        else { (void)hipDeviceSynchronize(); (void)hipFree(s.p); }
        s.p = nullptr; s.cap = 0; s.in_graph = false;
    }
    size_t want = std::max<size_t>(bytes, 1 << 16);
    want = (want + (want >> 1) + 4095) & ~size_t(4095);
    if (hipMalloc(&s.p, want) != hipSuccess) { (void)hipGetLastError(); s.p = nullptr; return nullptr; }
    s.cap = want;
    return s.p;
}

hipEvent_t Engine::get_event()
{
    if (!event_pool_.empty()) { hipEvent_t e = event_pool_.back(); event_pool_.pop_back(); return e; }
    hipEvent_t e = nullptr;
    if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    return e;
}

// The ABI lock is released while the host only waits for the GPU: other threads may enter the engine meanwhile,
// so callers re-validate whatever they looked up before the wait.
int Engine::wait_event(hipEvent_t ev)
{
    if (!ev) return SPECKV_OK;
    if (hipEventQuery(ev) == hipSuccess) return SPECKV_OK;
    (void)hipGetLastError();
    std::unique_lock<std::mutex>* mine = lk_;
    if (mine && mine->owns_lock()) { ++waiting_; mine->unlock(); } else mine = nullptr;
    const hipError_t e = hipEventSynchronize(ev);
    if (mine) {
        mine->lock(); lk_ = mine; (void)hipSetDevice(device_);
        if (--waiting_ == 0) idle_cv_.notify_all();      // speckv_finalize may be waiting for the engine to empty
    }
    if (e != hipSuccess) {
        SPECKV_ERR("hipEventSynchronize failed: %s", hipGetErrorString(e));
        (void)hipGetLastError();
        return SPECKV_ERR_DRIVER;
    }
    return SPECKV_OK;
}

int Engine::wait_stream()
{
    hipEvent_t ev = get_event();
    if (!ev) { HIP_TRY(hipStreamSynchronize(stream_)); return SPECKV_OK; }
    HIP_TRY(hipEventRecord(ev, stream_));
    const int rc = wait_event(ev);
    put_event(ev);
    return rc;
}

// --------------------------------------------------------- allocation table
int Engine::publish_row(Allocation* a)
{
    DevAlloc r{};
    r.entries = a->d_entries;
    r.d_flags = a->d_flags;
    r.d_slot = a->d_slot;
    r.stamp = a->d_stamp;
    void* dp = nullptr;
    HIP_TRY(hipHostGetDevicePointer(&dp, a->pinned, 0));
    r.h_slot = static_cast<uint32_t*>(dp);
    r.layout = a->has_layout ? a->layout : Layout{0, 0, 0, 0, 0, a->n_pages};
    // the copy reads its source when the stream gets to it: the source is the row's own slot in a pinned mirror
    h_tab_[a->row] = r;
    HIP_TRY(hipMemcpyAsync(d_tab_ + a->row, h_tab_ + a->row, sizeof(r), hipMemcpyHostToDevice, stream_));
    return SPECKV_OK;
}

// ------------------------------------------------------------ alloc/free
int Engine::alloc(size_t bytes, const speckv_alloc_hint_t* hint, uint64_t* out)
{
    // speckv_allocator.cpp:11-38 : the handle is consumed even for 0 bytes
    std::unique_ptr<Allocation> a(new Allocation());
    a->size_bytes = bytes;
    a->n_pages = (bytes + kPageSize - 1) / kPageSize;
    a->scheme = scheme_;
    a->rec_stride = stride_for(scheme_);
    if (null_) {
        a->host_flags.assign(a->n_pages, 0u);
        a->flags = a->host_flags.data();
    } else {
        a->access_count.assign(a->n_pages, 0u);
        if (a->n_pages) {
            DeviceScope device_scope(device_);
            drain_zombies(false);
            // placement: preferred_node picks one pool GPU (1-based; 0 = stripe over all)
            std::vector<int> use;
            if (hint && hint->preferred_node >= 1 && hint->preferred_node <= pools_.size())
                use.push_back(static_cast<int>(hint->preferred_node) - 1);
            else
                for (size_t i = 0; i < pools_.size(); ++i) use.push_back(static_cast<int>(i));
            const uint32_t D = static_cast<uint32_t>(use.size());
            bool ok = true;
            bool single_run = (D == 1);
            bool regular = true;
            std::vector<PageEntry> host;
            for (int attempt = 0; attempt < 2; ++attempt) {
                ok = true; regular = true; single_run = (D == 1); host.clear();
                for (uint32_t k = 0; k < D && ok; ++k) {
                    const uint64_t np = shard_pages(a->n_pages, D, k);    // pages with page % D == k
                    if (np == 0) { a->extents.push_back({use[k], nullptr, 0, 0}); continue; }
                    // (INT4_G32 runs, and FP8 runs of a striped allocation, carry 15 records of slack: the class forms of the fused attention over
                    //  a striped pool fetch whole 16-record tiles, and the tile that holds a class's last record may reach past the run's --
                    //  k_attend_int4_wg8<.., CLS>, k_attend_fp8_dma<2>)
                    const size_t need = run_bytes_for(a->scheme, a->rec_stride, np) + ((a->scheme == SPECKV_COMP_INT4_G32 || (a->scheme == SPECKV_COMP_FP8_E4M3 && D > 1)) ? 15u * a->rec_stride : 0u);
                    void* base = pools_[use[k]]->alloc(need);
                    if (base) {
                        a->extents.push_back({use[k], base, need, np});
                        if (!single_run) {
                            if (host.empty()) host.resize(a->n_pages);
                            for (uint64_t j = 0; j < np; ++j)
                                host[k + j * D] = entry_at(a->scheme, a->rec_stride, reinterpret_cast<uint64_t>(base), j);
                        }
                        continue;
                    }
                    // fragmented pool: place the pages of this device in several runs
                    single_run = false; regular = false;
                    if (host.empty()) host.resize(a->n_pages);
                    uint64_t placed = 0;
                    while (ok && placed < np) {
                        size_t got = 0;
                        void* part = pools_[use[k]]->alloc_up_to(run_bytes_for(a->scheme, a->rec_stride, np - placed), run_granule_for(a->scheme, a->rec_stride), &got);
                        if (!part) { ok = false; break; }
                        const uint64_t cnt = std::min<uint64_t>(run_records_for(a->scheme, a->rec_stride, got), np - placed);
                        a->extents.push_back({use[k], part, got, cnt});
                        for (uint64_t j = 0; j < cnt; ++j)
                            host[k + (placed + j) * D] = entry_at(a->scheme, a->rec_stride, reinterpret_cast<uint64_t>(part), j);
                        placed += cnt;
                    }
                }
                if (ok || attempt == 1 || zombies_.empty()) break;
                // out of pool memory with freed allocations still waiting for their streams: wait for them, retry
                for (auto& ex : a->extents) if (ex.base) pools_[ex.pool]->free(ex.base, ex.bytes);
                a->extents.clear();
                a->mx4_vacated.clear();
                drain_zombies(true);
            }
            if (ok && planar_mx4(a->scheme))                  // the unused tail of every run's last tile holds no page
                for (const auto& ex : a->extents)
                    if (ex.base && ex.n_pages % kMx4TileRecs)
                        a->mx4_vacated[reinterpret_cast<uint64_t>(ex.base) + ex.n_pages / kMx4TileRecs * kMx4TileBytes] =
                            static_cast<uint16_t>(0xFFFFu << (ex.n_pages % kMx4TileRecs));
            uint32_t* dev3 = nullptr;
            if (ok) ok = hipMalloc(reinterpret_cast<void**>(&a->d_entries), a->n_pages * sizeof(PageEntry)) == hipSuccess;
            if (ok) ok = hipMalloc(reinterpret_cast<void**>(&dev3), 3 * a->n_pages * sizeof(uint32_t)) == hipSuccess;
            if (ok) {
                a->d_flags = dev3; a->d_slot = dev3 + a->n_pages; a->d_stamp = dev3 + 2 * a->n_pages;
                ok = hipMemsetAsync(a->d_flags, 0, a->n_pages * sizeof(uint32_t), stream_) == hipSuccess &&
                     hipMemsetAsync(a->d_slot, 0xFF, a->n_pages * sizeof(uint32_t), stream_) == hipSuccess &&
                     hipMemsetAsync(a->d_stamp, 0, a->n_pages * sizeof(uint32_t), stream_) == hipSuccess;
            }
            if (ok) ok = hipHostMalloc(&a->pinned, (a->n_pages + kLenSamples) * sizeof(uint32_t), hipHostMallocMapped | hipHostMallocPortable) == hipSuccess;      // + the record-length samples (Allocation::len_samples)
            if (ok) {
                a->host_flags.assign(a->n_pages, 0u);
                a->flags = a->host_flags.data();
                a->slot = static_cast<uint32_t*>(a->pinned);
                memset(a->slot, 0xFF, a->n_pages * sizeof(uint32_t));
                a->len_samples = a->slot + a->n_pages;
                memset(a->len_samples, 0, kLenSamples * sizeof(uint32_t));
            }
            if (ok) {
                if (single_run)
                    ok = launch_init_entries(a->d_entries, a->n_pages, reinterpret_cast<uint64_t>(a->extents[0].base),
                                             planar_mx4(a->scheme) ? kPlanarMx4 : a->rec_stride, stream_) == hipSuccess;
                else
                    ok = hipMemcpy(a->d_entries, host.data(), host.size() * sizeof(PageEntry), hipMemcpyHostToDevice) == hipSuccess;
            }
            // FP8 / INT4 pools start as zero bytes (= records of zeros), which lets the fused attention address them
            // arithmetically without a validity test per page: one run (linear form) or the regular striping over up to 8
            // pools (striped form).  Each run is cleared on the GPU that holds it.
            const bool fixed_fmt = a->scheme == SPECKV_COMP_FP8_E4M3 || a->scheme == SPECKV_COMP_INT4_G32 || a->scheme == SPECKV_COMP_MXFP4;
            if (ok && fixed_fmt && regular && D <= 8 && a->n_pages < (1ull << 28)) {
                for (const auto& ex : a->extents) {
                    if (!ex.base || !ok) continue;
                    if (pools_[ex.pool]->device() == device_) {
                        ok = hipMemsetAsync(ex.base, 0, ex.bytes, stream_) == hipSuccess;
                    } else {
                        DeviceScope owner(pools_[ex.pool]->device());
                        ok = hipMemset(ex.base, 0, ex.bytes) == hipSuccess;
                    }
                }
                if (ok && single_run) a->linear_base = static_cast<uint8_t*>(a->extents[0].base);
                if (ok) {                          // run bases for the striped form (D = 1: the single run, so that a batch may mix both)
                    uint64_t bases[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                    for (uint32_t k = 0; k < D; ++k) bases[k] = reinterpret_cast<uint64_t>(a->extents[k].base);   // extents[k] = the run of residue k
                    ok = hipMalloc(reinterpret_cast<void**>(&a->d_stripe), sizeof(bases)) == hipSuccess &&
                         hipMemcpy(a->d_stripe, bases, sizeof(bases), hipMemcpyHostToDevice) == hipSuccess;
                    if (ok) a->stripe_n = D;
                }
            }
            a->regular = regular;
            a->pool_of_residue.assign(D, 0);
            for (uint32_t k = 0; k < D; ++k) a->pool_of_residue[k] = use[k];
            a->page_pool.resize(a->n_pages);
            for (uint64_t i = 0; i < a->n_pages; ++i) a->page_pool[i] = static_cast<uint8_t>(use[i % D]);
            if (ok) {
                if (free_rows_.empty()) { SPECKV_ERR("speckv_alloc: more than %u live allocations", tab_cap_); ok = false; }
                else { a->row = free_rows_.back(); free_rows_.pop_back(); row_owner_[a->row] = a.get(); }
            }
            if (ok) ok = publish_row(a.get()) == SPECKV_OK;
            if (ok) ok = hipStreamSynchronize(stream_) == hipSuccess;
            if (!ok) {
                (void)hipGetLastError();
                release_allocation(a.get());
                SPECKV_ERR("speckv_alloc(%zu bytes): out of pool memory", bytes);
                return SPECKV_ERR_NOMEM;
            }
        }
    }
    a->handle = next_handle_++;
    ++res_gen_;
    *out = a->handle;
    st_.total_allocations++;
    st_.current_allocated_bytes += bytes;
    st_.peak_allocated_bytes = std::max(st_.peak_allocated_bytes, st_.current_allocated_bytes);
    Allocation* raw = a.get();
    allocs_[a->handle] = std::move(a);
    // SPECKV_LAYOUT=T,L,H,D,bpe: geometry for callers that only speak the reference's 8 functions (its
    // allocate() sends none, vllm_speckv_backend.py:26-43); applied when the size matches
    {
        const std::vector<int>& g = default_layout_;
        if (g.size() == 5 && g[0] > 0 && g[1] > 0 && g[2] > 0 && g[3] > 0 && g[4] > 0 &&
            2ull * g[0] * g[1] * g[2] * g[3] * g[4] == bytes)
            (void)set_layout(raw->handle, g[0], g[1], g[2], g[3], g[4]);
    }
    return SPECKV_OK;
}

void Engine::release_allocation(Allocation* a)
{
    if (null_) return;
    if (a->l1_pages)
        for (uint32_t i = 0; i < n_l1_; ++i)
            if (l1_owner_[i].a == a) {
                const uint32_t s = n_l2_ + i;
                lru_unlink(s);
                l1_free_.push_back(s);
                l1_owner_[i] = Owner{nullptr, 0};
            }
    a->l1_pages = 0;
    if (a->row != kNoSlot) {              // ring owners may have named the row until now
        row_owner_[a->row] = nullptr;
        free_rows_.push_back(a->row);
        a->row = kNoSlot;
    }
    for (auto& ex : a->extents)
        if (ex.base) pools_[ex.pool]->free(ex.base, ex.bytes);
    a->extents.clear();
    if (a->d_entries) (void)hipFree(a->d_entries);
    if (a->d_flags) (void)hipFree(a->d_flags);            // flags, slots and stamps are one block
    if (a->d_scale_tab) { (void)hipFree(a->d_scale_tab_base); a->d_scale_tab = a->d_scale_tab_base = nullptr; a->scale_run = 0; }
    if (a->d_stripe) { (void)hipFree(a->d_stripe); a->d_stripe = nullptr; a->stripe_n = 0; }
    if (a->pinned) { (void)hipHostFree(a->pinned); a->pinned = nullptr; }
    a->d_entries = nullptr;
    a->d_flags = a->d_slot = a->d_stamp = nullptr;
    a->flags = a->slot = nullptr;
}

void Engine::note_use(Allocation* a, hipStream_t s)
{
    if (!s || s == stream_) return;
    if (std::find(a->user_streams.begin(), a->user_streams.end(), s) == a->user_streams.end()) a->user_streams.push_back(s);
}

bool Engine::quiet(const Zombie& z)
{
    if (z.engine_ev && hipEventQuery(z.engine_ev) != hipSuccess) { (void)hipGetLastError(); return false; }
    for (hipStream_t s : z.a->user_streams) {
        const hipError_t q = hipStreamQuery(s);
        if (q == hipErrorNotReady) { (void)hipGetLastError(); return false; }
        if (q != hipSuccess) (void)hipGetLastError();       // a stream the caller destroyed has nothing queued
    }
    return true;
}

void Engine::drain_zombies(bool wait)
{
    for (size_t i = 0; i < zombies_.size();) {
        Zombie& z = zombies_[i];
        if (wait && !quiet(z)) {
            if (z.engine_ev) (void)hipEventSynchronize(z.engine_ev);
            for (hipStream_t s : z.a->user_streams) if (hipStreamSynchronize(s) != hipSuccess) (void)hipGetLastError();
        }
        if (wait || quiet(z)) {
            release_allocation(z.a.get());
            put_event(z.engine_ev);
            zombies_[i] = std::move(zombies_.back());
            zombies_.pop_back();
        } else {
            ++i;
        }
    }
}

int Engine::free(uint64_t handle)
{   // speckv_allocator.cpp:40-52 : unknown handle is a silent no-op
    auto it = allocs_.find(handle);
    if (it == allocs_.end()) return SPECKV_OK;
    st_.total_deallocations++;
    st_.current_allocated_bytes -= it->second->size_bytes;
    if (layout_handle_ == handle) layout_handle_ = 0;
    for (auto b = bindings_.begin(); b != bindings_.end();)
        b = b->second.handle == handle ? bindings_.erase(b) : std::next(b);
    ++res_gen_;
    std::unique_ptr<Allocation> a = std::move(it->second);
    allocs_.erase(it);
    if (a->row != kNoSlot)                                  // requests already queued for it address nothing now
        for (auto& r : q_row_) if (r == a->row) { r = kNoSlot; st_.prefetch_dropped++; }
    if (null_ || a->n_pages == 0) return SPECKV_OK;
    DeviceScope device_scope(device_);
    // The table row is cleared in stream order (kernels already queued still see it); the memory itself goes back to
    // the pool once the engine stream has passed this point and every caller stream that was handed work on the
    // allocation has drained -- without stalling the device, and without blocking this call when they have not.
    h_tab_[a->row] = DevAlloc{};
    (void)hipMemcpyAsync(d_tab_ + a->row, h_tab_ + a->row, sizeof(DevAlloc), hipMemcpyHostToDevice, stream_);
    Zombie z{std::move(a), get_event()};
    if (z.engine_ev) (void)hipEventRecord(z.engine_ev, stream_);
    else (void)hipStreamSynchronize(stream_);
    // the table row is recycled only when the allocation is really released
    zombies_.push_back(std::move(z));
    drain_zombies(false);
    return SPECKV_OK;
}

// ------------------------------------------------------------------ tiers
void Engine::lru_unlink(uint32_t s)
{
    const uint32_t p = lru_prev_[s], n = lru_next_[s];
    if (p != UINT32_MAX) lru_next_[p] = n; else if (lru_head_ == s) lru_head_ = n;
    if (n != UINT32_MAX) lru_prev_[n] = p; else if (lru_tail_ == s) lru_tail_ = p;
    lru_prev_[s] = lru_next_[s] = UINT32_MAX;
}

void Engine::lru_push_mru(uint32_t s)
{
    lru_prev_[s] = lru_tail_;
    lru_next_[s] = UINT32_MAX;
    if (lru_tail_ != UINT32_MAX) lru_next_[lru_tail_] = s;
    lru_tail_ = s;
    if (lru_head_ == UINT32_MAX) lru_head_ = s;
}

void Engine::queue_update(Allocation* a, uint32_t page, uint32_t and_mask, uint32_t or_mask, uint32_t slot)
{
    if (upd_pending_ == upd_cap_) (void)flush_mirror();
    upd_ring_[upd_head_ % upd_cap_] = MirrorUpdate{a->row, page, and_mask, or_mask, slot, 0u};
    ++upd_head_;
    ++upd_pending_;
}

// Host-originated residency changes reach the device mirrors before any kernel that reads them.  The kernel reads
// the pinned ring in place; the ring is not written again before that launch has finished (callers of queue_update
// run after quiesce(), and a full ring waits here).
int Engine::flush_mirror()
{
    if (upd_pending_ == 0) return SPECKV_OK;
    void* dp = nullptr;
    HIP_TRY(hipHostGetDevicePointer(&dp, upd_ring_, 0));
    const MirrorUpdate* d_ring = static_cast<const MirrorUpdate*>(dp);
    uint32_t start = (upd_head_ - upd_pending_) % upd_cap_, left = upd_pending_;
    while (left) {
        const uint32_t seg = std::min(left, upd_cap_ - start);
        HIP_TRY(launch_apply_updates(d_tab_, d_ring + start, seg, stream_));
        start = (start + seg) % upd_cap_;
        left -= seg;
    }
    HIP_TRY(hipEventRecord(upd_event_, stream_));
    const bool was_full = upd_pending_ == upd_cap_;
    upd_pending_ = 0;
    if (was_full) RC_TRY(wait_event(upd_event_));
    return SPECKV_OK;
}

// A page leaves the cache (host decision: invalidation by a write, demotion, LRU eviction, span re-fetch).
// Only called with no fetch in flight (quiesce), so host and device mirrors cannot race on the page's words.
void Engine::drop_page(Allocation* a, uint32_t page)
{
    if (null_) { a->flags[page] &= ~3u; return; }
    const uint32_t f = res_flags(a, page);
    if (!(f & 3u)) return;
    const uint32_t s = a->slot[page];
    if ((f & 1u) && s >= n_l2_) {
        lru_unlink(s);
        l1_free_.push_back(s);
        l1_owner_[s - n_l2_] = Owner{nullptr, 0};
        if (a->l1_pages) a->l1_pages--;
    }
    a->flags[page] &= ~3u;
    a->slot[page] = kNoSlot;
    queue_update(a, page, ~3u, 0u, kKeepSlot);     // a ring slot keeps naming the page until it is reused: harmless
}

uint32_t Engine::take_l2_run(uint32_t n)
{
    // FIFO ring; a run never wraps so multi-page spans stay contiguous (k_flush_assign applies the same rule).
    // ring_seq_ counts every slot the hand has passed, skipped ones included: slot = sequence number % n_l2_.
    const RingRun r = ring_take(ring_seq_, n, n_l2_);
    ring_seq_ = r.next;
    return r.seq;
}

// Sequence numbers are 32 bits: long before they wrap, every live one is moved down by a multiple of the ring size
// (slots unchanged) and every dead one cleared.  O(all pages), once per ~3 * 10^9 fetched pages.
int Engine::renumber_ring_if_due()
{
    if (ring_seq_ < ring_seq_limit_ || n_l2_ == 0) return SPECKV_OK;
    RC_TRY(quiesce());
    RC_TRY(flush_mirror());
    RC_TRY(wait_stream());
    const uint32_t shift = (ring_seq_ - std::min(ring_seq_, n_l2_)) / n_l2_ * n_l2_;
    if (!shift) return SPECKV_OK;
    for (auto& kv : allocs_) {
        Allocation* a = kv.second.get();
        if (!a->slot) continue;
        for (uint64_t p = 0; p < a->n_pages; ++p) {
            if (a->flags[p] & 1u) continue;                                  // L1 slot index
            const uint32_t q = a->slot[p];
            a->slot[p] = l2_live(q) ? q - shift : kNoSlot;
        }
    }
    ring_seq_ -= shift;
    HIP_TRY(hipMemcpyAsync(d_hand_, &ring_seq_, sizeof(uint32_t), hipMemcpyHostToDevice, stream_));
    RC_TRY(wait_stream());
    return SPECKV_OK;
}

uint32_t Engine::take_l1_slot()
{
    if (!l1_free_.empty()) { uint32_t s = l1_free_.back(); l1_free_.pop_back(); return s; }
    // evict_l1_lru -> demote_to_l3 (cxl_memory_manager.cpp:285-293)
    const uint32_t victim = lru_head_;
    const Owner o = l1_owner_[victim - n_l2_];
    if (o.a) drop_page(o.a, o.page); else { lru_unlink(victim); l1_free_.push_back(victim); }
    st_.migrations_l1_to_l3++;
    const uint32_t s = l1_free_.back();
    l1_free_.pop_back();
    return s;
}

int Engine::move_to_l1(Allocation* a, uint32_t page)
{   // promote_to_l1 (cxl_memory_manager.cpp:130-163) for a page that sits in the L2 ring
    const uint32_t from = res_slot(a, page);
    const uint32_t to = take_l1_slot();
    HIP_TRY(hipMemcpyAsync(slot_ptr(to), slot_ptr(from), kPageSize, hipMemcpyDeviceToDevice, stream_));
    l1_owner_[to - n_l2_] = Owner{a, page};
    a->l1_pages++;
    a->slot[page] = to;
    a->flags[page] = (a->flags[page] & ~2u) | 1u;
    queue_update(a, page, ~2u, 1u, to);
    lru_push_mru(to);
    return SPECKV_OK;
}

void Engine::absorb(Flight& f)
{
    f.m = f.result->m;
    f.base = f.result->base;
    f.absorbed = true;
    if (f.m) ring_seq_ = f.result->seq + f.m;          // the device applied take_l2_run's rule
    st_.total_prefetches += f.m;
    st_.dma_submitted += f.m;
    st_.total_decompressions += f.m;
    if (f.assigned) { put_event(f.assigned); f.assigned = nullptr; }
}

// Non-blocking: flights whose assign kernel has finished are absorbed (in order), finished flights are retired and
// their pages counted as completed descriptors (speckv_kernel_module.c:194-215).
void Engine::harvest_flights()
{
    for (auto& f : flights_) {
        if (f.absorbed) continue;
        if (hipEventQuery(f.assigned) != hipSuccess) { (void)hipGetLastError(); break; }
        absorb(f);
    }
    while (!flights_.empty() && flights_.front().absorbed) {
        if (hipEventQuery(flights_.front().done) != hipSuccess) { (void)hipGetLastError(); break; }
        completed_unpolled_ += flights_.front().m;
        st_.dma_completed += flights_.front().m;
        put_event(flights_.front().done);
        flights_.pop_front();
    }
}

// Every flush's slot assignment is known to the host (waits for the small assign kernels only, not for the data).
int Engine::settle()
{
    for (size_t i = 0; i < flights_.size(); ++i)
        if (!flights_[i].absorbed) {
            const hipEvent_t ev = flights_[i].assigned;
            RC_TRY(wait_event(ev));                          // may release the ABI lock: look the flight up again
            for (auto& f : flights_)
                if (!f.absorbed && f.assigned == ev) absorb(f);
            i = static_cast<size_t>(-1);
        }
    harvest_flights();
    return SPECKV_OK;
}

// No fetch kernel is running or queued: the precondition of every host-side change of a page's residency words
// (the fetch kernels update those words themselves, for the pages they bring in and for the ones they evict).
int Engine::quiesce()
{
    RC_TRY(settle());
    while (!flights_.empty() || ring_busy_ > 0) {
        RC_TRY(wait_stream());
        RC_TRY(settle());
        if (ring_busy_ > 0 && flights_.empty()) break;       // another thread's synchronous fetch: its kernel has finished too
    }
    return SPECKV_OK;
}

// before a host-initiated ring operation: device mirrors current, ring hand current
int Engine::prepare_ring_op()
{
    for (int spin = 0; spin < 8; ++spin) {
        RC_TRY(flush_mirror());
        RC_TRY(settle());
        bool clean = upd_pending_ == 0;
        for (auto& f : flights_) clean = clean && f.absorbed;
        if (clean) break;
    }
    return SPECKV_OK;
}

// The ring slots of a flush that is still in flight are changing hands: the previous owner of each slot is being
// evicted and the new page is landing, both done by the fetch kernel.  A page of [p0, p1] whose slot lies in such a
// run is either arriving or leaving -- wait for that flush, then its residency words are final.
int Engine::wait_landed(const Allocation* a, uint64_t p0, uint64_t p1)
{
    for (int spin = 0; spin < 64 && !flights_.empty(); ++spin) {
        hipEvent_t need = nullptr;
        for (const Flight& f : flights_) {
            if (!f.absorbed || f.m == 0) continue;
            for (uint64_t p = p0; p <= p1 && !need; ++p)
                if ((res_flags(a, p) & 2u) && res_slot(a, p) >= f.base && res_slot(a, p) < f.base + f.m) need = f.done;
            if (need) break;
        }
        if (!need) break;
        RC_TRY(wait_event(need));
        RC_TRY(settle());
    }
    return SPECKV_OK;
}

void Engine::reap(bool wait_all)
{
    while (!inflight_.empty()) {
        Batch& b = inflight_.front();
        hipError_t q = wait_all ? hipEventSynchronize(b.ev) : hipEventQuery(b.ev);
        if (q == hipErrorNotReady) { (void)hipGetLastError(); break; }
        completed_unpolled_ += b.n;
        st_.dma_completed += b.n;
        event_pool_.push_back(b.ev);
        inflight_.pop_front();
    }
    harvest_flights();
}

// ---- ordering of the engine stream behind asynchronous writes on caller streams (see WriterEv) -----------------------
int Engine::note_async_write(hipStream_t s)
{
    if (!s || s == stream_ || is_capturing(s)) return SPECKV_OK;       // (a captured write is ordered by its graph's launch stream)
    for (auto& w : write_evs_)
        if (w.s == s) { HIP_TRY(hipEventRecord(w.ev, s)); w.dirty = true; return SPECKV_OK; }
    if (write_evs_.size() >= 64) {               // streams long gone: everything they were handed has to be over first
        RC_TRY(order_after_writes());
        for (auto& w : write_evs_) put_event(w.ev);
        write_evs_.clear();
    }
    hipEvent_t ev = get_event();
    if (!ev) { HIP_TRY(hipStreamSynchronize(s)); return SPECKV_OK; }
    HIP_TRY(hipEventRecord(ev, s));
    write_evs_.push_back({s, ev, true});
    return SPECKV_OK;
}

// After a write kernel has been queued on `s`: note the ordering, and if that cannot be done (no event, a failed record,
// the 64-stream overflow path failing) wait for the stream instead -- the engine stream is then trivially ordered behind
// the write.  Fails only if the wait itself fails.
int Engine::note_async_write_or_wait(hipStream_t s)
{
    if (note_async_write(s) == SPECKV_OK) return SPECKV_OK;
    (void)hipGetLastError();
    // (note_async_write itself returns OK for a capturing stream: a captured write is ordered by its graph's launch stream.
    // Getting here while capturing means the bookkeeping failed in a state where the stream cannot be waited for either:
    // the ordering is NOT established, and the caller is told so.)
    if (is_capturing(s)) {
        SPECKV_ERR("a pool write on a capturing stream could not be ordered in front of the engine's own stream");
        return SPECKV_ERR_DRIVER;
    }
    HIP_TRY(hipStreamSynchronize(s));
    return SPECKV_OK;
}

int Engine::order_after_writes()
{
    for (auto& w : write_evs_)
        if (w.dirty) { HIP_TRY(hipStreamWaitEvent(stream_, w.ev, 0)); w.dirty = false; }
    return SPECKV_OK;
}

// Synchronous fetch of `pages` (in this order) into a fresh run of ring slots; *base_out = first slot.
// The fetch kernel does the ring bookkeeping (eviction of the previous owners in HBM, the new owner, the page's slot word
// in HBM and its sequence number in the host-visible word).
int Engine::fetch_into_ring(Allocation* a, const std::vector<uint32_t>& pages, uint32_t* base_out)
{
    const uint32_t n = static_cast<uint32_t>(pages.size());
    if (n == 0) return SPECKV_OK;
    if (n > n_l2_) return SPECKV_ERR_NOMEM;
    const uint64_t handle = a->handle;
    RC_TRY(prepare_ring_op());
    RC_TRY(renumber_ring_if_due());
    RC_TRY(order_after_writes());
    if (find(handle) != a) return SPECKV_ERR_GENERAL;     // both may wait (and let go of the ABI lock): freed meanwhile
    bool run = true;
    for (uint32_t i = 1; i < n && run; ++i) run = pages[i] == pages[0] + i;
    CodecArgs c{};
    c.entries = a->d_entries;
    c.trusted = 1;                       // pool records only ever come from k_compress
    if (run) {
        c.first = pages[0];
    } else {
        uint32_t* d_pages = static_cast<uint32_t*>(scratch(s_pages_, n * sizeof(uint32_t)));
        if (!d_pages) return SPECKV_ERR_NOMEM;
        HIP_TRY(hipMemcpyAsync(d_pages, pages.data(), n * sizeof(uint32_t), hipMemcpyHostToDevice, stream_));
        c.page_list = d_pages;
    }
    const uint32_t seq = take_l2_run(n);
    const uint32_t base = seq % n_l2_;
    c.n = n;
    c.scheme = a->scheme;
    c.quant_mode = quant_mode_;
    c.tab = d_tab_;
    c.alloc_idx = a->row;
    c.ring_owner = d_owner_;
    c.ring_base = cache_base_;
    c.slot0 = base;
    c.seq0 = seq;
    c.hand_ptr = d_hand_;
    c.new_hand = ring_seq_;
    // A miss of a page or a few: the kernel's last wave stores a token to a pinned host word and the host spins on it -- the
    // runtime's own completion path costs 4 us more for a launch this short (DESIGN.md sect. 5).  A spin that runs out
    // (a preempted GPU, a debugger) falls back to it.
    const bool spin_ok = access_spin_ok_;
    const bool spin = spin_ok && n <= 8u && h_done_dev_ && d_done_count_;
    if (spin) {
        c.done_flag = h_done_dev_;
        c.done_count = d_done_count_;
        c.done_token = ++done_token_ ? done_token_ : ++done_token_;      // never 0: the word's initial value
    }
    HIP_TRY(launch_decompress(c, stream_));
    st_.dma_submitted += n;
    st_.total_decompressions += n;
    ++ring_busy_;
    int wrc = SPECKV_OK;
    bool seen = false;
    if (spin) {
        const auto t_end = std::chrono::steady_clock::now() + std::chrono::microseconds(200);
        for (uint32_t it = 0; !seen; ++it) {
            seen = __atomic_load_n(h_done_, __ATOMIC_ACQUIRE) == c.done_token;
            if (!seen && (it & 63u) == 63u && std::chrono::steady_clock::now() > t_end) break;
        }
    }
    if (!seen) wrc = wait_stream();      // sync_fetch_page: submit, then spin on completion
    --ring_busy_;
    RC_TRY(wrc);
    completed_unpolled_ += n;
    st_.dma_completed += n;
    if (find(handle) != a) return SPECKV_ERR_GENERAL;     // freed by another thread while we waited
    *base_out = base;
    return SPECKV_OK;
}

// Synchronous fetch of one page into a host-managed (L1) slot.
int Engine::fetch_into_slot(Allocation* a, uint32_t page, uint32_t slot)
{
    RC_TRY(flush_mirror());
    RC_TRY(order_after_writes());
    CodecArgs c{};
    c.entries = a->d_entries;
    c.trusted = 1;
    c.first = page;
    c.n = 1;
    c.data = slot_ptr(slot);
    c.data_stride = kPageSize;
    c.scheme = a->scheme;
    c.quant_mode = quant_mode_;
    HIP_TRY(launch_decompress(c, stream_));
    st_.dma_submitted += 1;
    st_.total_decompressions += 1;
    RC_TRY(wait_stream());
    completed_unpolled_ += 1;
    st_.dma_completed += 1;
    return SPECKV_OK;
}

// ----------------------------------------------------------------- access
int Engine::access(uint64_t handle, uint64_t off, size_t len, void** out)
{
    Allocation* a = find(handle);
    if (!a) return SPECKV_ERR_GENERAL;                       // speckv_allocator.cpp:56
    const uint64_t p0 = off / kPageSize, poff = off % kPageSize;
    if (p0 >= a->n_pages) return SPECKV_ERR_GENERAL;         // speckv_allocator.cpp:62
    if (null_) {
        // is_in_l1_or_l2 / sync_fetch_page (speckv_allocator.cpp:66-73,105-138):
        // the ioctl fails on the fake device and the page is marked L2 anyway
        if ((a->flags[p0] & 3u) == 0) a->flags[p0] |= 2u;
        *out = reinterpret_cast<void*>(0x4000000000ULL + (handle << 20) + (p0 << 12) + poff);
        return SPECKV_OK;
    }
    if (!a->entry_bytes_seen && len && len <= kPageSize) a->entry_bytes_seen = static_cast<uint32_t>(len);
    uint64_t p1 = len ? (off + len - 1) / kPageSize : p0;
    if (p1 >= a->n_pages) p1 = a->n_pages - 1;
    if (p1 - p0 + 1 > n_l2_) return SPECKV_ERR_NOMEM;
    DeviceScope device_scope(device_);
    RC_TRY(settle());
    RC_TRY(wait_landed(a, p0, p1));                          // before residency is read: slots of a flush in flight are in transition
    if ((a = find(handle)) == nullptr) return SPECKV_ERR_GENERAL;
    auto resident_run = [&] {                                // a multi-page span must come back contiguous
        for (uint64_t p = p0; p <= p1; ++p)
            if (!(res_flags(a, p) & 3u) || res_slot(a, p) != res_slot(a, p0) + (p - p0)) return false;
        return true;
    };
    bool contiguous = resident_run();
    if (!contiguous && (!flights_.empty() || ring_busy_ > 0)) {
        // a flush in flight may be bringing these very pages: let it land before deciding to fetch (and before the
        // span's stale copies are dropped below)
        RC_TRY(quiesce());
        if ((a = find(handle)) == nullptr) return SPECKV_ERR_GENERAL;
        contiguous = resident_run();
    }
    std::vector<uint32_t> miss;
    for (uint64_t p = p0; p <= p1; ++p) {
        a->access_count[p]++;                                // update_access_tracking, cxl_memory_manager.cpp:223-245
        const uint32_t f = res_flags(a, p);
        if (f & 1u) { st_.l1_hits++; if (contiguous) { lru_unlink(a->slot[p]); lru_push_mru(a->slot[p]); } }
        else if (f & 2u) st_.l2_hits++;
        else { st_.l3_accesses++; st_.l2_misses++; }
        if (!contiguous || !(f & 3u)) miss.push_back(static_cast<uint32_t>(p));
    }
    int rc = SPECKV_OK;
    if (!miss.empty()) {
        if (p1 > p0)                                         // refetch the whole span into one run
            for (uint32_t p : miss) drop_page(a, p);
        uint32_t base = 0;
        rc = fetch_into_ring(a, miss, &base);                // sync_fetch_page: submit + spin on completion
        if (rc != SPECKV_OK) return rc;
    } else if (p1 == p0 && (res_flags(a, p0) & 3u) == 2u && a->access_count[p0] > 10) {
        // L2 hit on a hot page -> promote (memory_allocator.cpp:127-134, is_hot_page: count > 10)
        RC_TRY(quiesce());
        if ((a = find(handle)) == nullptr) return SPECKV_ERR_GENERAL;
        if ((res_flags(a, p0) & 3u) == 2u) { RC_TRY(move_to_l1(a, static_cast<uint32_t>(p0))); RC_TRY(wait_stream()); }
    }
    if ((a = find(handle)) == nullptr) return SPECKV_ERR_GENERAL;
    if (!(res_flags(a, p0) & 3u)) {                          // evicted again by a concurrent caller (cache far too small)
        SPECKV_ERR("speckv_access: page %llu of handle %llu is not resident after its fetch (flags %#x slot %u, %zu missed, hand %u)",
                   static_cast<unsigned long long>(p0), static_cast<unsigned long long>(handle), a->flags[p0], a->slot[p0],
                   miss.size(), ring_seq_);
        return SPECKV_ERR_GENERAL;
    }
    *out = slot_ptr(res_slot(a, p0)) + poff;
    return SPECKV_OK;
}

int Engine::access_batch(uint64_t handle, const uint64_t* offs, uint32_t n, void** out)
{
    Allocation* a = find(handle);
    if (!a) return SPECKV_ERR_GENERAL;
    for (uint32_t i = 0; i < n; ++i)
        if (offs[i] / kPageSize >= a->n_pages) return SPECKV_ERR_GENERAL;
    if (null_) {
        for (uint32_t i = 0; i < n; ++i) {
            const uint64_t p = offs[i] / kPageSize;
            if ((a->flags[p] & 3u) == 0) a->flags[p] |= 2u;
            out[i] = reinterpret_cast<void*>(0x4000000000ULL + (handle << 20) + (p << 12) + offs[i] % kPageSize);
        }
        return SPECKV_OK;
    }
    DeviceScope device_scope(device_);
    // everything a flush is still bringing in has to land before the pointers are handed out
    RC_TRY(quiesce());
    if ((a = find(handle)) == nullptr) return SPECKV_ERR_GENERAL;
    ++access_epoch_;
    if (a->stamp.size() != a->n_pages) a->stamp.assign(a->n_pages, 0u);
    std::vector<uint32_t> miss;
    for (uint32_t i = 0; i < n; ++i) {
        const uint32_t p = static_cast<uint32_t>(offs[i] / kPageSize);
        a->access_count[p]++;
        const uint32_t f = res_flags(a, p);
        if (f & 1u) st_.l1_hits++; else if (f & 2u) st_.l2_hits++; else { st_.l3_accesses++; st_.l2_misses++; }
        if (!(f & 3u) && a->stamp[p] != access_epoch_) { a->stamp[p] = access_epoch_; miss.push_back(p); }
    }
    int rc = SPECKV_OK;
    if (miss.size() > n_l2_) rc = SPECKV_ERR_NOMEM;
    if (rc == SPECKV_OK && !miss.empty()) {
        uint32_t base = 0;
        rc = fetch_into_ring(a, miss, &base);
        if (rc == SPECKV_OK && (a = find(handle)) == nullptr) rc = SPECKV_ERR_GENERAL;
    }
    if (rc == SPECKV_OK)
        for (uint32_t i = 0; i < n; ++i) {
            const uint64_t p = offs[i] / kPageSize;
            out[i] = (res_flags(a, p) & 3u) ? slot_ptr(res_slot(a, p)) + offs[i] % kPageSize : nullptr;
            if (!out[i]) rc = SPECKV_ERR_GENERAL;   // evicted inside this very batch (cache smaller than batch)
        }
    return rc;
}

// ------------------------------------------------------------------ knobs
int Engine::set_prefetch_depth(uint32_t k)
{
    if (null_) return SPECKV_ERR_DRIVER;     // ioctl on the fake device fails (speckv_c_api.cpp:108-109)
    adapt_.set(k);                           // SpeculativePrefetcher::set_prefetch_depth, speculative_prefetcher.cpp:144-147
    return SPECKV_OK;
}
int Engine::set_scheme(int scheme)
{
    if (null_) return SPECKV_ERR_DRIVER;
    if (scheme < 0 || scheme > SPECKV_COMP_MXFP4) return SPECKV_ERR_INVAL;
    scheme_ = scheme;
    return SPECKV_OK;
}
int Engine::set_quant_mode(int mode)
{
    if (mode != SPECKV_QUANT_REF_EXACT && mode != SPECKV_QUANT_INTENT) return SPECKV_ERR_INVAL;
    quant_mode_ = mode;
    return SPECKV_OK;
}

int Engine::set_layout(uint64_t handle, uint32_t T, uint32_t L, uint32_t H, uint32_t D, uint32_t bpe)
{
    Allocation* a = find(handle);
    if (!a) return SPECKV_ERR_GENERAL;
    if (!T || !L || !H || !D || !bpe) return SPECKV_ERR_INVAL;
    if (!q_req_.empty()) {                                    // queued requests were resolved against the old geometry
        (void)prefetch_flush(nullptr);
        if ((a = find(handle)) == nullptr) return SPECKV_ERR_GENERAL;       // the flush may have let go of the ABI lock
    }
    ++res_gen_;
    a->layout = Layout{T, L, H, D, bpe, a->n_pages};
    a->has_layout = true;
    a->layout_inferred = false;
    layout_handle_ = handle;
    if (null_ || a->n_pages == 0) return SPECKV_OK;
    DeviceScope device_scope(device_);
    RC_TRY(publish_row(a));
    // fused-attention scale table (FP8 records, 2 positions per page, regions aligned to 16-page tiles)
    if (a->scheme == SPECKV_COMP_FP8_E4M3 && static_cast<uint64_t>(H) * D * bpe == 2048u && T % 32u == 0u) {
        if (!a->d_scale_tab) {
            // a regularly striped allocation keeps the scales a second time, in run order (entry p / D of run p % D, like the records),
            // in front of the table: the residue-class forms of the attention then find a class tile's 16 scales in one line
            const uint32_t Dn = a->regular ? a->stripe_n : 0u;
            const uint64_t cap = Dn >= 2u ? (a->n_pages + Dn - 1u) / Dn : 0u;
            a->scale_run = (Dn >= 2u && Dn <= 8u && cap < (1ull << 28)) ? (Dn | static_cast<uint32_t>(cap << 4)) : 0u;
            const size_t front = a->scale_run ? static_cast<size_t>(Dn) * cap : 0u;
            HIP_TRY(hipMalloc(reinterpret_cast<void**>(&a->d_scale_tab_base), (front + a->n_pages) * sizeof(float)));
            a->d_scale_tab = a->d_scale_tab_base + front;
        }
        a->region_pages = T / 2u;
        HIP_TRY(launch_build_scale_tab(a->d_entries, a->n_pages, a->region_pages, a->d_scale_tab, stream_, a->scale_run));
        HIP_TRY(hipStreamSynchronize(stream_));
    } else if (a->d_scale_tab) {
        HIP_TRY(hipDeviceSynchronize());
        (void)hipFree(a->d_scale_tab_base);
        a->d_scale_tab = a->d_scale_tab_base = nullptr;
        a->scale_run = 0;
        a->region_pages = 0;
    }
    return SPECKV_OK;
}

// ---------------------------------------------------------- introspection
int Engine::translate(uint64_t handle, uint64_t off, speckv_ext_page_info_t* o)
{
    Allocation* a = find(handle);
    if (!a) return SPECKV_ERR_GENERAL;
    const uint64_t p = off / kPageSize;
    if (p >= a->n_pages) return SPECKV_ERR_GENERAL;
    memset(o, 0, sizeof(*o));
    o->virt_page_id = (handle << 32) | (p << 12);                        // speckv_allocator.cpp:24
    o->phys_page_id = 0x4000000000ULL + (handle << 20) + (p << 12);      // speckv_allocator.cpp:25
    o->page_size = kPageSize;
    o->scheme = static_cast<uint32_t>(a->scheme);
    o->scale = 1.0f;
    if (null_) {
        o->flags = a->flags[p];
        // the pool GPU the page WOULD live on (placement rule only; nothing is stored on the fake device)
        o->pool_device = pool_devs_.empty() ? -1 : pool_devs_[place_page(p, static_cast<uint32_t>(pool_devs_.size())).pool];
        return SPECKV_OK;
    }
    DeviceScope device_scope(device_);
    RC_TRY(quiesce());                      // residency words are final (a flush in flight may be writing them)
    if ((a = find(handle)) == nullptr) return SPECKV_ERR_GENERAL;
    o->flags = res_flags(a, p);
    PageEntry e{};
    HIP_TRY(hipMemcpy(&e, a->d_entries + p, sizeof(e), hipMemcpyDeviceToHost));
    o->pool_device = pools_[a->page_pool[p]]->device();
    o->rec_bytes = e.rec_bytes;
    o->scale = planar_mx4(a->scheme) ? 1.0f : e.scale;
    o->pool_addr = e.pool_addr;
    o->aux_offset = planar_mx4(a->scheme) ? float_bits(e.scale) : 0u;      // tile-planar MXFP4: the record's 64 codes lie this far behind its nibbles
    o->cache_addr = (res_flags(a, p) & 3u) ? reinterpret_cast<uint64_t>(slot_ptr(res_slot(a, p))) : 0;
    o->access_count = a->access_count[p];
    return SPECKV_OK;
}

int Engine::fetch_desc(uint64_t handle, uint64_t off, speckv_dma_desc_t* o)
{   // speckv_allocator.cpp:115-127
    Allocation* a = find(handle);
    if (!a) return SPECKV_ERR_GENERAL;
    const uint64_t p = off / kPageSize;
    if (p >= a->n_pages) return SPECKV_ERR_GENERAL;
    const uint64_t virt = (handle << 32) | (p << 12);
    o->fpga_addr = 0x4000000000ULL + (handle << 20) + (p << 12);
    o->gpu_addr = 0x8000000000ULL + (virt & 0xFFFFFFFFFFFFULL);
    o->bytes = kPageSize;
    o->flags = 0;
    return SPECKV_OK;
}

int Engine::poll_complete(uint32_t* done)
{   // SPECKV_IOCTL_POLL_DONE: completions since the previous poll, then cleared
    if (null_) return SPECKV_ERR_DRIVER;
    DeviceScope device_scope(device_);
    reap(false);
    *done = static_cast<uint32_t>(std::min<uint64_t>(completed_unpolled_, UINT32_MAX));
    completed_unpolled_ = 0;
    return SPECKV_OK;
}

int Engine::sync()
{
    if (null_) return SPECKV_OK;
    DeviceScope device_scope(device_);
    int rc = prefetch_flush(nullptr);
    RC_TRY(wait_stream());
    RC_TRY(settle());
    reap(false);
    drain_zombies(false);
    return rc;
}

int Engine::promote_to_l1(uint64_t handle, uint64_t off)
{
    if (null_) return no_data_path("speckv_ext_promote_to_l1");
    Allocation* a = find(handle);
    if (!a) return SPECKV_ERR_GENERAL;
    const uint64_t p = off / kPageSize;
    if (p >= a->n_pages) return SPECKV_ERR_GENERAL;
    DeviceScope device_scope(device_);
    RC_TRY(quiesce());
    if ((a = find(handle)) == nullptr) return SPECKV_ERR_GENERAL;
    if (a->flags[p] & 1u) return SPECKV_ERR_GENERAL;          // already there -> false (cxl_memory_manager.cpp:134-136)
    if (res_flags(a, p) & 2u) {
        RC_TRY(move_to_l1(a, static_cast<uint32_t>(p)));
        return wait_stream();
    }
    const uint32_t s = take_l1_slot();
    const int rc = fetch_into_slot(a, static_cast<uint32_t>(p), s);
    if ((a = find(handle)) == nullptr) { l1_free_.push_back(s); return SPECKV_ERR_GENERAL; }
    if (rc != SPECKV_OK) { l1_free_.push_back(s); return rc; }
    a->slot[p] = s;
    a->flags[p] = (a->flags[p] & ~3u) | 1u;
    l1_owner_[s - n_l2_] = Owner{a, static_cast<uint32_t>(p)};
    a->l1_pages++;
    queue_update(a, static_cast<uint32_t>(p), ~3u, 1u, s);
    lru_push_mru(s);
    st_.migrations_l3_to_l1++;
    return SPECKV_OK;
}

int Engine::demote_to_l3(uint64_t handle, uint64_t off)
{
    if (null_) return no_data_path("speckv_ext_demote_to_l3");
    Allocation* a = find(handle);
    if (!a) return SPECKV_ERR_GENERAL;
    const uint64_t p = off / kPageSize;
    if (p >= a->n_pages) return SPECKV_ERR_GENERAL;
    DeviceScope device_scope(device_);
    RC_TRY(quiesce());
    if ((a = find(handle)) == nullptr) return SPECKV_ERR_GENERAL;
    if (!(res_flags(a, p) & 3u)) return SPECKV_ERR_GENERAL;   // already in the pool only
    if (a->flags[p] & 1u) st_.migrations_l1_to_l3++;
    drop_page(a, static_cast<uint32_t>(p));
    return SPECKV_OK;
}

int Engine::stats(speckv_ext_stats_t* out)
{
    st_.prefetch_depth = adapt_.depth();
    st_.compression_scheme = static_cast<uint32_t>(scheme_);
    st_.quant_mode = static_cast<uint32_t>(quant_mode_);
    st_.pool_bytes_reserved = 0;
    for (auto& p : pools_) st_.pool_bytes_reserved += p->reserved_bytes();
    if (!null_) {
        // compressed bytes = sum of record lengths currently stored
        DeviceScope device_scope(device_);
        RC_TRY(settle());                   // prefetch counters of the flushes submitted so far
        reap(false);
        uint64_t comp = 0, in_use = 0, written = 0, sealed = 0;
        std::vector<PageEntry> host;
        for (auto& kv : allocs_) {
            Allocation* a = kv.second.get();
            if (!a->n_pages) continue;
            for (const auto& ex : a->extents) in_use += ex.bytes;
            sealed += a->packed ? 1u : 0u;
            host.resize(a->n_pages);
            if (hipMemcpy(host.data(), a->d_entries, a->n_pages * sizeof(PageEntry), hipMemcpyDeviceToHost) == hipSuccess)
                for (auto& e : host) { comp += e.rec_bytes; written += e.rec_bytes ? 1u : 0u; }
        }
        st_.compressed_bytes = comp;
        st_.pool_bytes_in_use = in_use;
        st_.written_pages = written;
        st_.sealed_allocations = sealed;
    }
    *out = st_;
    return SPECKV_OK;
}

} // namespace speckv
