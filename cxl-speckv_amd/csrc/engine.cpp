// cxl-speckv_amd/csrc/engine.cpp -- see engine.hpp
#include "engine.hpp"
#include "placement.hpp"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace speckv {

namespace {

bool g_verbose = [] { const char* e = getenv("SPECKV_LOG"); return e && *e && *e != '0'; }();

#define SPECKV_ERR(...) do { fprintf(stderr, "[libcxlspeckv] " __VA_ARGS__); fputc('\n', stderr); } while (0)
#define SPECKV_LOGV(...) do { if (g_verbose) { fprintf(stderr, "[libcxlspeckv] " __VA_ARGS__); fputc('\n', stderr); } } while (0)

#define HIP_TRY(expr)                                                              \
    do {                                                                           \
        hipError_t _e = (expr);                                                    \
        if (_e != hipSuccess) {                                                    \
            SPECKV_ERR("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            (void)hipGetLastError();                                               \
            return SPECKV_ERR_DRIVER;                                              \
        }                                                                          \
    } while (0)
#define RC_TRY(expr) do { int _rc = (expr); if (_rc != SPECKV_OK) return _rc; } while (0)

constexpr uint32_t kResSlots = 64;          // flush result words in rotation
constexpr uint32_t kMaxFlights = 16;        // flushes in flight before the oldest is waited for
constexpr uint32_t kUpdCap = 1u << 16;      // mirror-update ring entries

size_t env_mb(const char* name, size_t def_mb)
{
    const char* e = getenv(name);
    if (!e || !*e) return def_mb;
    return static_cast<size_t>(strtoull(e, nullptr, 10));
}

// The C ABI may be called with any HIP device current (SURVEY 8b "Threading"): every entry that touches the
// GPU makes the engine's device current for its own duration and restores the caller's on every exit path.
struct DeviceScope {
    int prev = -1;
    bool switched = false;
    explicit DeviceScope(int device)
    {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != device) switched = hipSetDevice(device) == hipSuccess;
    }
    ~DeviceScope() { if (switched && prev >= 0) (void)hipSetDevice(prev); }
    DeviceScope(const DeviceScope&) = delete;
    DeviceScope& operator=(const DeviceScope&) = delete;
};

uint32_t stride_for(int scheme)
{
    switch (scheme) {
    case SPECKV_COMP_INT8: return 2048u;
    case SPECKV_COMP_FP8_E4M3: return 2048u;
    case SPECKV_COMP_INT4_G32: return kInt4RecBytes;      // 1152 B: the 4:1 format (3.56:1 with scales)
    default: return kPageSize;
    }
}

int no_data_path(const char* what)
{
    static bool warned = false;
    if (!warned) {
        SPECKV_ERR("%s: the \"/dev/null\" device has no data path (page-table emulation only); "
                   "open a HIP device to move or decode KV blocks", what);
        warned = true;
    }
    return SPECKV_ERR_DRIVER;
}

bool is_capturing(hipStream_t s)
{
    if (!s) return false;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &cs) != hipSuccess) { (void)hipGetLastError(); return false; }
    return cs != hipStreamCaptureStatusNone;
}

std::vector<int> parse_int_list(const char* env)
{
    std::vector<int> out;
    if (!env) return out;
    std::string s(env);
    size_t i = 0;
    while (i < s.size()) {
        size_t j = s.find(',', i);
        if (j == std::string::npos) j = s.size();
        if (j > i) out.push_back(atoi(s.substr(i, j - i).c_str()));
        i = j + 1;
    }
    return out;
}

} // namespace

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
    static const bool unordered = getenv("SPECKV_DEBUG_UNORDERED_SCRATCH") != nullptr;      // test hook: shows that the test can fail
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
                    const size_t need = np * a->rec_stride;
                    void* base = pools_[use[k]]->alloc(need);
                    if (base) {
                        a->extents.push_back({use[k], base, need, np});
                        if (!single_run) {
                            if (host.empty()) host.resize(a->n_pages);
                            for (uint64_t j = 0; j < np; ++j)
                                host[k + j * D] = PageEntry{reinterpret_cast<uint64_t>(base) + j * a->rec_stride, 0u, 1.0f};
                        }
                        continue;
                    }
                    // fragmented pool: place the pages of this device in several runs
                    single_run = false; regular = false;
                    if (host.empty()) host.resize(a->n_pages);
                    uint64_t placed = 0;
                    while (ok && placed < np) {
                        size_t got = 0;
                        void* part = pools_[use[k]]->alloc_up_to((np - placed) * a->rec_stride, a->rec_stride, &got);
                        if (!part) { ok = false; break; }
                        const uint64_t cnt = got / a->rec_stride;
                        a->extents.push_back({use[k], part, got, cnt});
                        for (uint64_t j = 0; j < cnt; ++j)
                            host[k + (placed + j) * D] = PageEntry{reinterpret_cast<uint64_t>(part) + j * a->rec_stride, 0u, 1.0f};
                        placed += cnt;
                    }
                }
                if (ok || attempt == 1 || zombies_.empty()) break;
                // out of pool memory with freed allocations still waiting for their streams: wait for them, retry
                for (auto& ex : a->extents) if (ex.base) pools_[ex.pool]->free(ex.base, ex.bytes);
                a->extents.clear();
                drain_zombies(true);
            }
            uint32_t* dev3 = nullptr;
            if (ok) ok = hipMalloc(reinterpret_cast<void**>(&a->d_entries), a->n_pages * sizeof(PageEntry)) == hipSuccess;
            if (ok) ok = hipMalloc(reinterpret_cast<void**>(&dev3), 3 * a->n_pages * sizeof(uint32_t)) == hipSuccess;
            if (ok) {
                a->d_flags = dev3; a->d_slot = dev3 + a->n_pages; a->d_stamp = dev3 + 2 * a->n_pages;
                ok = hipMemsetAsync(a->d_flags, 0, a->n_pages * sizeof(uint32_t), stream_) == hipSuccess &&
                     hipMemsetAsync(a->d_slot, 0xFF, a->n_pages * sizeof(uint32_t), stream_) == hipSuccess &&
                     hipMemsetAsync(a->d_stamp, 0, a->n_pages * sizeof(uint32_t), stream_) == hipSuccess;
            }
            if (ok) ok = hipHostMalloc(&a->pinned, a->n_pages * sizeof(uint32_t), hipHostMallocMapped | hipHostMallocPortable) == hipSuccess;
            if (ok) {
                a->host_flags.assign(a->n_pages, 0u);
                a->flags = a->host_flags.data();
                a->slot = static_cast<uint32_t*>(a->pinned);
                memset(a->slot, 0xFF, a->n_pages * sizeof(uint32_t));
            }
            if (ok) {
                if (single_run)
                    ok = launch_init_entries(a->d_entries, a->n_pages, reinterpret_cast<uint64_t>(a->extents[0].base),
                                             a->rec_stride, stream_) == hipSuccess;
                else
                    ok = hipMemcpy(a->d_entries, host.data(), host.size() * sizeof(PageEntry), hipMemcpyHostToDevice) == hipSuccess;
            }
            // FP8 / INT4 pools start as zero bytes (= records of zeros), which lets the fused attention address them
            // arithmetically without a validity test per page: one run (linear form) or the regular striping over up to 8
            // pools (striped form).  Each run is cleared on the GPU that holds it.
            const bool fixed_fmt = a->scheme == SPECKV_COMP_FP8_E4M3 || a->scheme == SPECKV_COMP_INT4_G32;
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
    if (const char* env = getenv("SPECKV_LAYOUT")) {
        const std::vector<int> g = parse_int_list(env);
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
    if (a->d_scale_tab) { (void)hipFree(a->d_scale_tab); a->d_scale_tab = nullptr; }
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
    if (is_capturing(s)) return SPECKV_OK;
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
    static const bool spin_ok = getenv("SPECKV_ACCESS_NO_SPIN") == nullptr;
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
    static const bool timing = getenv("SPECKV_TIMING") != nullptr;
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
    static const bool by_kernel = [] { const char* e = getenv("SPECKV_FLUSH_UPLOAD"); return !(e && e[0] == 'c'); }();
    if (!by_kernel) return hipMemcpyAsync(dst, staged, bytes, hipMemcpyHostToDevice, s);
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
    // 0.102 -> 0.094 ms until landed, 20 480 requests unchanged (SPECKV_FLUSH_UPLOAD=copy for the A/B).
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
    if (scheme < 0 || scheme > SPECKV_COMP_FP8_E4M3) return SPECKV_ERR_INVAL;
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
        if (!a->d_scale_tab) HIP_TRY(hipMalloc(reinterpret_cast<void**>(&a->d_scale_tab), a->n_pages * sizeof(float)));
        a->region_pages = T / 2u;
        HIP_TRY(launch_build_scale_tab(a->d_entries, a->n_pages, a->region_pages, a->d_scale_tab, stream_));
        HIP_TRY(hipStreamSynchronize(stream_));
    } else if (a->d_scale_tab) {
        HIP_TRY(hipDeviceSynchronize());
        (void)hipFree(a->d_scale_tab);
        a->d_scale_tab = nullptr;
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
    o->scale = e.scale;
    o->pool_addr = e.pool_addr;
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

// -------------------------------------------------------------- data path
int Engine::write(uint64_t handle, uint64_t off, const void* src, size_t len, bool on_device)
{
    if (null_) return no_data_path("speckv_ext_write");
    Allocation* a = find(handle);
    if (!a) return SPECKV_ERR_GENERAL;
    if (off % kPageSize || !src) return SPECKV_ERR_INVAL;
    if (off > a->size_bytes || len > a->size_bytes - off) return SPECKV_ERR_GENERAL;
    const bool to_end = (off + len == a->size_bytes);
    if (len % kPageSize && !to_end) return SPECKV_ERR_INVAL;
    if (len == 0) return SPECKV_OK;
    const uint64_t p0 = off / kPageSize;
    const uint64_t full = len / kPageSize, tail = len % kPageSize;
    DeviceScope device_scope(device_);
    if (a->packed) { RC_TRY(unpack(a)); if ((a = find(handle)) == nullptr) return SPECKV_ERR_GENERAL; }   // a sealed allocation goes back into slots first
    RC_TRY(quiesce());
    if ((a = find(handle)) == nullptr) return SPECKV_ERR_GENERAL;
    reap(false);
    // the source may have been produced on any stream of the caller: this call is
    // synchronous anyway, so order it after everything queued on the device
    if (on_device) HIP_TRY(hipDeviceSynchronize());
    CodecArgs c{};
    c.entries = a->d_entries;
    c.scale_tab = a->d_scale_tab;        // fused-attention scale table follows every write
    c.region_pages = a->region_pages;
    c.data_stride = kPageSize;
    c.scheme = a->scheme;
    c.quant_mode = quant_mode_;
    const uint8_t* s8 = static_cast<const uint8_t*>(src);
    if (on_device) {
        if (full) {
            c.first = p0; c.n = full; c.data = const_cast<uint8_t*>(s8);
            HIP_TRY(launch_compress(c, stream_));
        }
        if (tail) {
            uint8_t* st = static_cast<uint8_t*>(scratch(s_stage_, kPageSize));
            if (!st) return SPECKV_ERR_NOMEM;
            HIP_TRY(hipMemsetAsync(st, 0, kPageSize, stream_));
            HIP_TRY(hipMemcpyAsync(st, s8 + full * kPageSize, tail, hipMemcpyDeviceToDevice, stream_));
            c.first = p0 + full; c.n = 1; c.data = st;
            HIP_TRY(launch_compress(c, stream_));
        }
    } else {
        const uint64_t total = full + (tail ? 1 : 0);
        const uint64_t chunk_pages = std::min<uint64_t>(total, 16384);      // 64 MiB staging
        uint8_t* st = static_cast<uint8_t*>(scratch(s_stage_, chunk_pages * kPageSize));
        if (!st) return SPECKV_ERR_NOMEM;
        for (uint64_t done = 0; done < total; done += chunk_pages) {
            const uint64_t np = std::min(chunk_pages, total - done);
            const size_t bytes = static_cast<size_t>(std::min<uint64_t>(np * kPageSize, len - done * kPageSize));
            if (bytes < np * kPageSize) HIP_TRY(hipMemsetAsync(st + (np - 1) * kPageSize, 0, kPageSize, stream_));
            HIP_TRY(hipMemcpyAsync(st, s8 + done * kPageSize, bytes, hipMemcpyHostToDevice, stream_));
            c.first = p0 + done; c.n = np; c.data = st;
            HIP_TRY(launch_compress(c, stream_));
            HIP_TRY(hipStreamSynchronize(stream_));       // the staging buffer is shared: keep the ABI lock
        }
    }
    HIP_TRY(hipStreamSynchronize(stream_));
    const uint64_t np = full + (tail ? 1 : 0);
    for (uint64_t p = p0; p < p0 + np; ++p) {
        drop_page(a, static_cast<uint32_t>(p));             // a cached copy is stale now
        if (a->scheme != SPECKV_COMP_FP16) a->flags[p] |= 4u; else a->flags[p] &= ~4u;
    }
    st_.total_compressions += np;
    st_.original_bytes += np * kPageSize;
    return SPECKV_OK;
}

// Asynchronous page writes for a decode loop: n pages first, first+step, first+2*step, ... (the pages of one position
// pair in every (layer, kind) region of the shim layout are `num_tokens/2` pages apart) compressed from a contiguous
// device buffer on the caller's stream.  Pages that are cached right now would go stale: that case takes the
// synchronous path (a decode loop appends positions nobody has fetched yet).
int Engine::write_strided(uint64_t handle, uint64_t first, uint64_t step, uint64_t n, const void* d_src, hipStream_t s)
{
    if (null_) return no_data_path("speckv_ext_write_strided");
    Allocation* a = find(handle);
    if (!a) return SPECKV_ERR_GENERAL;
    if (!d_src || step == 0) return SPECKV_ERR_INVAL;
    if (n == 0) return SPECKV_OK;
    if (first >= a->n_pages || (n - 1) > (a->n_pages - 1 - first) / step) return SPECKV_ERR_GENERAL;
    // a last page that is only partly inside the allocation would need zero padding of the source: not here
    if (a->size_bytes % kPageSize && first + (n - 1) * step == a->n_pages - 1) return SPECKV_ERR_INVAL;
    DeviceScope device_scope(device_);
    if (a->packed) { RC_TRY(unpack(a)); if ((a = find(handle)) == nullptr) return SPECKV_ERR_GENERAL; }
    // NULL = the engine's stream: the source may have been produced on any stream of the caller, order after all of them
    if (!s) HIP_TRY(hipDeviceSynchronize());
    bool cached = false;
    for (uint64_t i = 0; i < n && !cached; ++i) cached = (res_flags(a, first + i * step) & 3u) != 0;
    if (cached || !flights_.empty() || ring_busy_ > 0) {
        RC_TRY(quiesce());
        if ((a = find(handle)) == nullptr) return SPECKV_ERR_GENERAL;
        for (uint64_t i = 0; i < n; ++i) drop_page(a, static_cast<uint32_t>(first + i * step));
        RC_TRY(flush_mirror());
        if (s) RC_TRY(wait_stream());
    }
    CodecArgs c{};
    c.entries = a->d_entries;
    c.scale_tab = a->d_scale_tab;
    c.region_pages = a->region_pages;
    c.data_stride = kPageSize;
    c.scheme = a->scheme;
    c.quant_mode = quant_mode_;
    c.first = first;
    c.page_step = step;
    c.n = n;
    c.data = static_cast<uint8_t*>(const_cast<void*>(d_src));
    hipStream_t st = s ? s : stream_;
    HIP_TRY(launch_compress(c, st));
    // The kernel is queued: from here on the host mirror follows it whatever else fails (an early return between
    // the launch and these lines left the device table and the host flags disagreeing).
    note_use(a, s);
    for (uint64_t i = 0; i < n; ++i) {
        uint32_t& f = a->flags[first + i * step];
        if (a->scheme != SPECKV_COMP_FP16) f |= 4u; else f &= ~4u;
    }
    st_.total_compressions += n;
    st_.original_bytes += n * kPageSize;
    RC_TRY(note_async_write_or_wait(s));
    if (!s) RC_TRY(wait_stream());
    return SPECKV_OK;
}

// speckv_ext_write_async: a contiguous page range from a device buffer, on the caller's stream, no device-wide wait.
int Engine::write_async(uint64_t handle, uint64_t off, const void* d_src, size_t len, hipStream_t s)
{
    if (null_) return no_data_path("speckv_ext_write_async");
    if (off % kPageSize || len % kPageSize) return SPECKV_ERR_INVAL;
    if (len == 0) return find(handle) ? SPECKV_OK : SPECKV_ERR_GENERAL;
    return write_strided(handle, off / kPageSize, 1, len / kPageSize, d_src, s);
}

// write_strided for a batch of allocations in one launch (the append of a decode step: SURVEY 8f row N2), and several
// page runs of ONE allocation in one launch (a prompt's K / V regions: speckv_ext_write_runs).  Host side as in
// write_strided per group (cached pages are invalidated first); the kernel takes one descriptor per group.
int Engine::write_groups(const uint64_t* handles, const uint64_t* firsts, const void* const* d_srcs, uint32_t n_groups,
                         uint64_t step, uint64_t n_each, hipStream_t s, bool same_allocation)
{
    if (!handles || !firsts || !d_srcs || step == 0 || !s) return SPECKV_ERR_INVAL;
    if (n_groups == 0 || n_each == 0) return SPECKV_OK;
    std::vector<Allocation*> as(n_groups);
    bool cached = false;
    for (uint32_t i = 0; i < n_groups; ++i) {
        Allocation* a = find(handles[same_allocation ? 0 : i]);
        if (!a) return SPECKV_ERR_GENERAL;
        if (!d_srcs[i]) return SPECKV_ERR_INVAL;
        if (a->scheme != find(handles[0])->scheme) return SPECKV_ERR_INVAL;
        if (firsts[i] >= a->n_pages || (n_each - 1) > (a->n_pages - 1 - firsts[i]) / step) return SPECKV_ERR_GENERAL;
        if (a->size_bytes % kPageSize && firsts[i] + (n_each - 1) * step == a->n_pages - 1) return SPECKV_ERR_INVAL;
        if (!same_allocation)
            for (uint32_t k = 0; k < i; ++k) if (as[k] == a) return SPECKV_ERR_INVAL;   // one descriptor per allocation
        as[i] = a;
        for (uint64_t j = 0; j < n_each && !cached; ++j) cached = (res_flags(a, firsts[i] + j * step) & 3u) != 0;
    }
    for (uint32_t i = 0; i < n_groups; ++i)
        if (as[i]->packed) {                                    // sealed allocations go back into slots first
            DeviceScope scope(device_);
            RC_TRY(unpack(as[i]));
            for (uint32_t j = 0; j < n_groups; ++j)
                if ((as[j] = find(handles[same_allocation ? 0 : j])) == nullptr) return SPECKV_ERR_GENERAL;
        }
    if (same_allocation && n_groups > 1) {                      // the runs of one allocation must not overlap (racing writers)
        std::vector<uint64_t> order(firsts, firsts + n_groups);
        std::sort(order.begin(), order.end());
        const uint64_t span = (n_each - 1) * step;
        for (uint32_t i = 1; i < n_groups; ++i)
            if (step == 1 ? order[i] <= order[i - 1] + span : order[i] == order[i - 1]) return SPECKV_ERR_INVAL;
        // (strided groups that start on different pages interleave without touching: page = first + j * step)
        if (step != 1)
            for (uint32_t i = 1; i < n_groups; ++i)
                if ((order[i] - order[0]) % step == 0 && order[i] - order[0] <= span) return SPECKV_ERR_INVAL;
    }
    DeviceScope device_scope(device_);
    if (cached || !flights_.empty() || ring_busy_ > 0) {
        RC_TRY(quiesce());
        for (uint32_t i = 0; i < n_groups; ++i) {
            if ((as[i] = find(handles[same_allocation ? 0 : i])) == nullptr) return SPECKV_ERR_GENERAL;
            for (uint64_t j = 0; j < n_each; ++j) drop_page(as[i], static_cast<uint32_t>(firsts[i] + j * step));
        }
        RC_TRY(flush_mirror());
        RC_TRY(wait_stream());
    }
    // descriptors: pinned slot -> device slot (4 of each in rotation, guarded by an event on the caller's stream)
    const size_t bytes = static_cast<size_t>(n_groups) * sizeof(CompressGroup);
    if (grp_ring_.slot_bytes < bytes) {
        HIP_TRY(hipDeviceSynchronize());
        if (grp_ring_.base) { (void)hipHostFree(grp_ring_.base); grp_ring_.base = nullptr; }
        if (d_groups_) { (void)hipFree(d_groups_); d_groups_ = nullptr; }
        grp_ring_.slot_bytes = std::max<size_t>(bytes * 2, 16384);
        HIP_TRY(hipHostMalloc(&grp_ring_.base, grp_ring_.slot_bytes * 4, hipHostMallocDefault));
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&d_groups_), grp_ring_.slot_bytes * 4));
        for (auto& ev : grp_ring_.ev)
            if (!ev) HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    }
    const int slot = grp_ring_.next;
    grp_ring_.next = (slot + 1) & 3;
    RC_TRY(wait_event(grp_ring_.ev[slot]));                   // may release the ABI lock
    for (uint32_t i = 0; i < n_groups; ++i)
        if ((as[i] = find(handles[same_allocation ? 0 : i])) == nullptr) return SPECKV_ERR_GENERAL;
    CompressGroup* staged = reinterpret_cast<CompressGroup*>(static_cast<uint8_t*>(grp_ring_.base) + static_cast<size_t>(slot) * grp_ring_.slot_bytes);
    CompressGroup* d_slot = reinterpret_cast<CompressGroup*>(reinterpret_cast<uint8_t*>(d_groups_) + static_cast<size_t>(slot) * grp_ring_.slot_bytes);
    for (uint32_t i = 0; i < n_groups; ++i) {
        const Allocation* a = as[i];
        staged[i] = CompressGroup{a->d_entries, a->d_scale_tab, a->region_pages, 0u, firsts[i],
                                  static_cast<const uint8_t*>(d_srcs[i])};
    }
    HIP_TRY(hipMemcpyAsync(d_slot, staged, bytes, hipMemcpyHostToDevice, s));
    CodecArgs c{};
    c.groups = d_slot;
    c.group_n = n_each;
    c.page_step = step;
    c.data_stride = kPageSize;
    c.scheme = as[0]->scheme;
    c.quant_mode = quant_mode_;
    c.n = static_cast<uint64_t>(n_groups) * n_each;
    HIP_TRY(launch_compress(c, s));
    for (uint32_t i = 0; i < n_groups; ++i) {           // the kernel is queued: host mirror first, then the orderings
        Allocation* a = as[i];
        note_use(a, s);
        for (uint64_t j = 0; j < n_each; ++j) {
            uint32_t& f = a->flags[firsts[i] + j * step];
            if (a->scheme != SPECKV_COMP_FP16) f |= 4u; else f &= ~4u;
        }
    }
    st_.total_compressions += c.n;
    st_.original_bytes += c.n * kPageSize;
    if (hipEventRecord(grp_ring_.ev[slot], s) != hipSuccess) {      // the staging slot must not be reused under the kernel
        (void)hipGetLastError();
        HIP_TRY(hipStreamSynchronize(s));
    }
    RC_TRY(note_async_write_or_wait(s));
    return SPECKV_OK;
}

int Engine::write_strided_batch(const uint64_t* handles, const uint64_t* firsts, const void* const* d_srcs, uint32_t n_alloc,
                                uint64_t step, uint64_t n_each, hipStream_t s)
{
    if (null_) return no_data_path("speckv_ext_write_strided_batch");
    return write_groups(handles, firsts, d_srcs, n_alloc, step, n_each, s, false);
}

int Engine::write_runs(uint64_t handle, const uint64_t* firsts, const void* const* d_srcs, uint32_t n_runs, uint64_t n_each, hipStream_t s)
{
    if (null_) return no_data_path("speckv_ext_write_runs");
    return write_groups(&handle, firsts, d_srcs, n_runs, 1, n_each, s, true);
}

int Engine::read(uint64_t handle, uint64_t off, void* dst, size_t len, bool on_device)
{
    if (null_) return no_data_path("speckv_ext_read");
    Allocation* a = find(handle);
    if (!a) return SPECKV_ERR_GENERAL;
    if (off % kPageSize || !dst) return SPECKV_ERR_INVAL;
    if (off > a->size_bytes || len > a->size_bytes - off) return SPECKV_ERR_GENERAL;
    if (len % kPageSize && off + len != a->size_bytes) return SPECKV_ERR_INVAL;
    if (len == 0) return SPECKV_OK;
    const uint64_t p0 = off / kPageSize, full = len / kPageSize, tail = len % kPageSize;
    DeviceScope device_scope(device_);
    if (on_device) HIP_TRY(hipDeviceSynchronize());    // dst may still be in use on a caller stream
    else RC_TRY(order_after_writes());                 // records being written asynchronously on a caller stream
    CodecArgs c{};
    c.entries = a->d_entries;
    c.trusted = 1;                       // pool records only ever come from k_compress
    c.data_stride = kPageSize;
    c.scheme = a->scheme;
    c.quant_mode = quant_mode_;
    uint8_t* d8 = static_cast<uint8_t*>(dst);
    if (on_device && !tail) {
        c.first = p0; c.n = full; c.data = d8;
        HIP_TRY(launch_decompress(c, stream_));
    } else {
        const uint64_t total = full + (tail ? 1 : 0);
        const uint64_t chunk_pages = std::min<uint64_t>(total, 16384);
        uint8_t* st = static_cast<uint8_t*>(scratch(s_stage_, chunk_pages * kPageSize));
        if (!st) return SPECKV_ERR_NOMEM;
        for (uint64_t done = 0; done < total; done += chunk_pages) {
            const uint64_t np = std::min(chunk_pages, total - done);
            const size_t bytes = static_cast<size_t>(std::min<uint64_t>(np * kPageSize, len - done * kPageSize));
            c.first = p0 + done; c.n = np; c.data = st;
            HIP_TRY(launch_decompress(c, stream_));
            HIP_TRY(hipMemcpyAsync(d8 + done * kPageSize, st, bytes,
                                   on_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, stream_));
            HIP_TRY(hipStreamSynchronize(stream_));
        }
    }
    HIP_TRY(hipStreamSynchronize(stream_));
    const uint64_t np = full + (tail ? 1 : 0);
    st_.total_decompressions += np;
    st_.dma_submitted += np; st_.dma_completed += np; completed_unpolled_ += np;
    return SPECKV_OK;
}

// Copy-engine fetch of a logical page range (the reference's DMA path: one descriptor per 4 KiB page through the
// DMA engine, speckv_allocator.cpp:115-138, dma_engine.v:150-217 -- here one hipMemcpyPeerAsync per POOL GPU and
// chunk, because striping makes the range one contiguous record run on every pool): the runs are copied over xGMI
// into local staging on per-peer side streams, then decompressed locally from there.  Two staging buffers in
// rotation: the copies of chunk c+1 overlap the decompression of chunk c.
int Engine::fetch_range_copy_engine(Allocation* a, uint64_t first, uint64_t n, void* d_dst, bool f32, hipStream_t st)
{
    const uint32_t D = static_cast<uint32_t>(a->pool_of_residue.size());
    const bool packed = a->packed && a->packed_regular;
    if (D == 0 || D > 8 || !(a->regular || packed)) return SPECKV_ERR_INVAL;
    const size_t stride = a->rec_stride;
    if (!stage_[0]) {
        stage_bytes_ = env_mb("SPECKV_STAGE_MB", 64) << 20;
        for (int b = 0; b < 2; ++b) {
            HIP_TRY(hipMalloc(reinterpret_cast<void**>(&stage_[b]), stage_bytes_));
            HIP_TRY(hipEventCreateWithFlags(&stage_free_[b], hipEventDisableTiming));
        }
    }
    if (lanes_.size() < pools_.size()) {
        const size_t old = lanes_.size();
        lanes_.resize(pools_.size());
        for (size_t i = old; i < lanes_.size(); ++i) {
            HIP_TRY(hipStreamCreateWithFlags(&lanes_[i].s, hipStreamNonBlocking));
            for (auto& ev : lanes_[i].copied) HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        }
    }
    // records per pool and chunk: the staging buffer is cut into D equal regions
    const uint64_t region = (stage_bytes_ / D) / stride * stride;
    const uint64_t recs_per_region = region / stride;
    if (recs_per_region == 0) return SPECKV_ERR_NOMEM;
    const uint64_t chunk_pages = recs_per_region * D;     // logical pages per chunk (each pool gets <= recs_per_region of them)
    // the source records must be in place: everything queued on the engine stream (writes are synchronous) and on
    // the caller's stream so far is ordered before the first copy
    hipEvent_t start = get_event();
    if (!start) return SPECKV_ERR_DRIVER;
    HIP_TRY(hipEventRecord(start, st));
    uint64_t done = 0;
    int chunk = 0;
    while (done < n) {
        const uint64_t f0 = first + done, nc = std::min(chunk_pages, n - done);
        const int b = chunk & 1;
        CodecArgs c{};
        c.entries = a->d_entries;
        c.trusted = 1;
        c.first = f0;
        c.n = nc;
        c.data = static_cast<uint8_t*>(d_dst) + done * (f32 ? 2ull * kPageSize : kPageSize);
        c.data_stride = f32 ? 2ull * kPageSize : kPageSize;
        c.scheme = a->scheme;
        c.quant_mode = quant_mode_;
        c.out_f32 = f32 ? 1 : 0;
        c.stripe_n = D;
        c.stripe_magic = (1ull << 35) / D + 1;
        for (uint32_t k = 0; k < D; ++k) {
            uint64_t rb = 0, cnt = 0;
            shard_range(f0, nc, D, k, &rb, &cnt);
            c.stripe_delta[k] = 0;
            if (cnt == 0) continue;
            const int pool = a->pool_of_residue[k];
            // the byte run of this pool's records of the chunk: fixed slots, or -- sealed allocation -- the packed records
            // themselves (record rb of residue k is page rb * D + k; the run ends where the next record of the pool starts)
            const uint8_t* src;
            size_t run_bytes;
            if (packed) {
                const uint64_t p_first = rb * D + k, p_next = (rb + cnt) * D + k;
                const uint64_t lo = static_cast<uint64_t>(a->packed_off128[p_first]) << 7;
                const uint64_t hi = p_next < a->n_pages ? static_cast<uint64_t>(a->packed_off128[p_next]) << 7 : a->packed_bytes[k];
                src = static_cast<const uint8_t*>(a->extents[k].base) + lo;
                run_bytes = static_cast<size_t>(hi - lo);
            } else {
                src = static_cast<const uint8_t*>(a->extents[k].base) + rb * stride;
                run_bytes = cnt * stride;
            }
            uint8_t* dstk = stage_[b] + k * region;
            c.stripe_delta[k] = static_cast<int64_t>(reinterpret_cast<intptr_t>(dstk) - reinterpret_cast<intptr_t>(src));
            if (run_bytes == 0) continue;                                   // (records of zero length: nothing to move)
            PeerLane& lane = lanes_[pool];
            if (chunk == 0) HIP_TRY(hipStreamWaitEvent(lane.s, start, 0));
            HIP_TRY(hipStreamWaitEvent(lane.s, stage_free_[b], 0));        // the decompression that last read this buffer
            HIP_TRY(hipMemcpyPeerAsync(dstk, device_, src, pools_[pool]->device(), run_bytes, lane.s));
            HIP_TRY(hipEventRecord(lane.copied[b], lane.s));
            HIP_TRY(hipStreamWaitEvent(st, lane.copied[b], 0));
            st_.copy_engine_bytes += run_bytes;
        }
        HIP_TRY(launch_decompress(c, st));
        HIP_TRY(hipEventRecord(stage_free_[b], st));
        done += nc;
        ++chunk;
    }
    put_event(start);
    st_.copy_engine_runs += static_cast<uint64_t>(chunk) * D;
    return SPECKV_OK;
}

int Engine::fetch_range(uint64_t handle, uint64_t first, uint64_t n, void* d_dst, bool f32, hipStream_t s, int engine_choice)
{
    if (null_) return no_data_path("speckv_ext_fetch_range");
    Allocation* a = find(handle);
    if (!a) return SPECKV_ERR_GENERAL;
    if (first > a->n_pages || n > a->n_pages - first) return SPECKV_ERR_GENERAL;
    if (!d_dst) return SPECKV_ERR_INVAL;
    if (engine_choice < 0 || engine_choice > 2) return SPECKV_ERR_INVAL;
    if (n == 0) return SPECKV_OK;
    DeviceScope device_scope(device_);
    if (!s) HIP_TRY(hipDeviceSynchronize());      // NULL = synchronous call on the engine's stream: d_dst may be in use on any stream
    hipStream_t st = s ? s : stream_;
    // which engine moves the records: the fused peer-load kernel (the wave loads the record over xGMI and
    // decompresses in registers) or the copy engines (SDMA runs into local staging, then a local decompress).
    // Per batch: long runs on remote pools go to the copy engines, short ones to the kernel; 1 / 2 force a choice
    // (SPECKV_REMOTE_ENGINE=kernel|copy overrides "auto").
    static const int env_choice = [] {
        const char* e = getenv("SPECKV_REMOTE_ENGINE");
        return !e ? 0 : !strcmp(e, "kernel") ? 1 : !strcmp(e, "copy") ? 2 : 0;
    }();
    int choice = engine_choice ? engine_choice : env_choice;
    const uint32_t D = static_cast<uint32_t>(a->pool_of_residue.size());
    const bool can_copy = (a->regular || (a->packed && a->packed_regular)) && D >= 1 && D <= 8 && a->n_pages < (1ull << 28) && !is_capturing(st);
    if (choice == 0) {
        bool remote = false;
        for (int p : a->pool_of_residue) remote = remote || pools_[p]->device() != device_;
        static const uint64_t min_run = env_mb("SPECKV_COPY_MIN_RUN_KB", 1024) << 10;
        choice = (remote && can_copy && (n / D) * a->rec_stride >= min_run) ? 2 : 1;
    }
    if (choice == 2 && !can_copy) {
        if (engine_choice == 2) return SPECKV_ERR_INVAL;     // asked for explicitly on a placement that has no runs
        choice = 1;
    }
    if (choice == 2) {
        RC_TRY(fetch_range_copy_engine(a, first, n, d_dst, f32, st));
    } else {
        CodecArgs c{};
        c.entries = a->d_entries;
        c.trusted = 1;                       // pool records only ever come from k_compress
        c.first = first;
        c.n = n;
        c.data = static_cast<uint8_t*>(d_dst);
        c.data_stride = f32 ? 2ull * kPageSize : kPageSize;
        c.scheme = a->scheme;
        c.quant_mode = quant_mode_;
        c.out_f32 = f32 ? 1 : 0;
        if (a->packed && a->n_pages) {                       // sealed: the packed size is known -- short records take the flat-run decoder
            uint64_t packed = 0;
            for (uint64_t b : a->packed_bytes) packed += b;
            c.structured_hint = packed / a->n_pages < 512u ? 1 : 0;
        }
        HIP_TRY(launch_decompress(c, st));
    }
    note_use(a, s);
    st_.dma_submitted += n;
    st_.total_decompressions += n;
    if (!s) {
        hipEvent_t ev = get_event();
        if (ev) { HIP_TRY(hipEventRecord(ev, stream_)); inflight_.push_back({ev, static_cast<uint32_t>(n)}); }
    } else {
        st_.dma_completed += n;            // completion belongs to the caller's stream
    }
    return SPECKV_OK;
}

int Engine::fetch_list(uint64_t handle, const uint32_t* d_pages, uint32_t n, void* d_dst, bool f32, hipStream_t s)
{
    if (null_) return no_data_path("speckv_ext_fetch_list");
    Allocation* a = find(handle);
    if (!a) return SPECKV_ERR_GENERAL;
    if (!d_dst || (!d_pages && n)) return SPECKV_ERR_INVAL;
    if (n == 0) return SPECKV_OK;
    DeviceScope device_scope(device_);
    if (!s) HIP_TRY(hipDeviceSynchronize());      // as in fetch_range
    CodecArgs c{};
    c.entries = a->d_entries;
    c.trusted = 1;                       // pool records only ever come from k_compress
    c.page_list = d_pages;
    c.n = n;
    c.data = static_cast<uint8_t*>(d_dst);
    c.data_stride = f32 ? 2ull * kPageSize : kPageSize;
    c.scheme = a->scheme;
    c.quant_mode = quant_mode_;
    c.out_f32 = f32 ? 1 : 0;
    hipStream_t st = s ? s : stream_;
    HIP_TRY(launch_decompress(c, st));
    note_use(a, s);
    st_.dma_submitted += n;
    st_.total_decompressions += n;
    if (!s) {
        hipEvent_t ev = get_event();
        if (ev) { HIP_TRY(hipEventRecord(ev, stream_)); inflight_.push_back({ev, n}); }
    } else {
        st_.dma_completed += n;
    }
    return SPECKV_OK;
}

int Engine::qk_scores_fp8(uint64_t handle, uint32_t layer, uint32_t n_layers, const void* d_q_f16, uint32_t g,
                          uint32_t pos_begin, uint32_t pos_end, float* d_out, hipStream_t s)
{
    if (null_) return no_data_path("speckv_ext_qk_scores_fp8");
    Allocation* a = find(handle);
    if (!a) return SPECKV_ERR_GENERAL;
    if (!a->has_layout || a->scheme != SPECKV_COMP_FP8_E4M3) return SPECKV_ERR_INVAL;
    const Layout& L = a->layout;
    // one K row (all heads of a position) must be 2048 B: two positions per page
    if (L.head_dim != 128 || L.bytes_per_element != 2 || L.num_heads * L.head_dim != 1024) return SPECKV_ERR_INVAL;
    if (n_layers == 0 || layer >= L.num_layers || n_layers > L.num_layers - layer || pos_begin % 2 || pos_begin > pos_end ||
        pos_end > L.num_tokens || pos_end % 2)
        return SPECKV_ERR_INVAL;
    if (g == 0 || g > 16 || !d_q_f16 || !d_out) return SPECKV_ERR_INVAL;
    const uint32_t n_pages = (pos_end - pos_begin) / 2;
    if (n_pages == 0) return SPECKV_OK;
    // shim layout [req 0][layer][kind 0 = K][pos][head]: page of (layer, pos)
    const uint64_t first_page = (static_cast<uint64_t>(layer) * 2 * L.num_tokens + pos_begin) / 2;
    const uint64_t layer_stride = static_cast<uint64_t>(L.num_tokens);      // pages per layer: K + V = 2*T/2
    if (first_page + (n_layers - 1) * layer_stride + n_pages > a->n_pages) return SPECKV_ERR_GENERAL;
    DeviceScope device_scope(device_);
    // NULL = the engine's stream and a synchronous call: the query may have been produced on any stream of the caller
    if (!s) HIP_TRY(hipDeviceSynchronize());
    hipStream_t st = s ? s : stream_;
    {   // linear form (records in one run, scale table, tile-aligned range inside the layer's region): direct loads
        const uint32_t n_tiles = (n_pages + 15u) / 16u;
        const bool fits = pos_begin % 32u == 0u && a->d_scale_tab && a->linear_base && !getenv("SPECKV_ATTEND_GENERAL") &&
                          static_cast<uint64_t>(pos_begin) + static_cast<uint64_t>(n_tiles) * 32u <= L.num_tokens;
        if (fits) {
            AttendArgs k{};
            k.k_first = first_page;
            k.layer_stride = layer_stride;
            k.n_pages = n_pages;
            k.heads = L.num_heads;
            k.g = g;
            k.tiles_per_split = 16;
            if (const char* env = getenv("SPECKV_QK_TILES_PER_WAVE")) k.tiles_per_split = std::max(1, atoi(env));
            k.lin_base = a->linear_base;
            k.scale_tab = a->d_scale_tab;
            k.q16 = static_cast<const uint16_t*>(d_q_f16);
            HIP_TRY(launch_qk_scores_fp8_linear(k, n_layers, d_out, st));
            note_use(a, s);
            if (!s) RC_TRY(wait_stream());
            return SPECKV_OK;
        }
    }
    const size_t rows = static_cast<size_t>(n_layers) * L.num_heads * 16;
    uint8_t* q8 = static_cast<uint8_t*>(scratch(s_req_, rows * 128 + rows * sizeof(float), s));
    if (!q8) return SPECKV_ERR_NOMEM;
    float* qs = reinterpret_cast<float*>(q8 + rows * 128);
    HIP_TRY(launch_quantize_q_e4m3(d_q_f16, n_layers * L.num_heads, g, L.head_dim, q8, qs, st));
    HIP_TRY(launch_qk_scores_fp8(a->d_entries, first_page, layer_stride, n_layers, n_pages, L.num_heads, g, q8, qs, d_out, st));
    note_use(a, s);
    if (!s) RC_TRY(wait_stream());
    return SPECKV_OK;
}

// Fused decode attention over the FP8 K and V regions of [layer, layer+n_layers) (attend.hip).
int Engine::attend_fp8(uint64_t handle, uint32_t layer, uint32_t n_layers, const void* d_q_f16, uint32_t g,
                       uint32_t pos_begin, uint32_t pos_end, float sm_scale, float* d_out, float* d_lse, hipStream_t s)
{
    if (null_) return no_data_path("speckv_ext_attend_fp8");
    Allocation* a = find(handle);
    if (!a) return SPECKV_ERR_GENERAL;
    if (!a->has_layout || a->scheme != SPECKV_COMP_FP8_E4M3) return SPECKV_ERR_INVAL;
    const Layout& L = a->layout;
    if (L.head_dim != 128 || L.bytes_per_element != 2 || L.num_heads * L.head_dim != 1024 || L.num_tokens % 2) return SPECKV_ERR_INVAL;
    if (n_layers == 0 || layer >= L.num_layers || n_layers > L.num_layers - layer || pos_begin % 2 || pos_begin > pos_end ||
        pos_end > L.num_tokens || pos_end % 2)
        return SPECKV_ERR_INVAL;
    if (g == 0 || g > 16 || !d_q_f16 || !d_out) return SPECKV_ERR_INVAL;
    DeviceScope device_scope(device_);
    // NULL = the engine's stream and a synchronous call: the query may have been produced on any stream of the caller
    if (!s) HIP_TRY(hipDeviceSynchronize());
    hipStream_t st = s ? s : stream_;
    const size_t out_elems = static_cast<size_t>(n_layers) * L.num_heads * g * 128;
    if (pos_end == pos_begin) {  // empty range: softmax over nothing -> zeros (and -inf lse is left to the caller)
        HIP_TRY(hipMemsetAsync(d_out, 0, out_elems * sizeof(float), st));
        if (!s) HIP_TRY(hipStreamSynchronize(stream_));
        return SPECKV_OK;
    }
    // The scale table is laid out in tiles of 32 positions from the start of a region: a range that starts inside a tile is
    // attended from the tile's start with its leading positions masked (AttendArgs::skip_pages), so every range of a layout
    // with a scale table takes the tile forms (linear / striped / table) -- the per-wave page-table kernel is left with the
    // layouts that have none (num_tokens not a multiple of 32).
    const bool has_tab = a->d_scale_tab != nullptr;
    const uint32_t begin_al = has_tab ? (pos_begin & ~31u) : pos_begin;
    const uint32_t skip_pages = (pos_begin - begin_al) / 2;
    const uint32_t n_pages = (pos_end - begin_al) / 2;
    pos_begin = begin_al;
    // shim layout [req 0][layer][kind][pos][head]: K pages of a layer, then its V pages
    const uint64_t k_first = (static_cast<uint64_t>(layer) * 2 * L.num_tokens + pos_begin) / 2;
    const uint64_t v_first = k_first + L.num_tokens / 2;
    const uint64_t layer_stride = static_cast<uint64_t>(L.num_tokens);
    if (v_first + (n_layers - 1) * layer_stride + n_pages > a->n_pages) return SPECKV_ERR_GENERAL;
    if (!d_zero_page_) {
        if (is_capturing(s)) return SPECKV_ERR_INVAL;        // first call must run outside a capture (see scratch())
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&d_zero_page_), kPageSize));
        HIP_TRY(hipMemset(d_zero_page_, 0, kPageSize));
    }
    // splits: ~20 waves per CU over the launch (the LDS-DMA kernel keeps 8 resident; measured at 70B-shaped, 80 layers:
    // 8 splits/row 0.72 of HBM peak at 32k and 0.63 at 8k, 16 splits 0.705 / 0.61, 4 splits 0.71 / 0.63, 2: 0.63 / 0.58)
    // A launch that already has 128+ workgroup columns (layers x head quads) is best left unsplit: each workgroup then
    // streams one long run, the rows are final (no partials, no merge launch) -- 80 layers: 1 split 0.73 / 0.70 / 0.64 of
    // HBM peak at 32k / 8k / 2k context against 0.71 / 0.62 / 0.48 with 8 splits.
    const uint32_t n_tiles = (n_pages + 15u) / 16u;
    const uint32_t rows = n_layers * L.num_heads;
    // linear form: records in one run, scale table present, tiles aligned with the table's (pos_begin a multiple of 32),
    // and the last (possibly ragged) 32-position tile must not read past the K / V region of its layer
    // (with the range aligned as above and num_tokens a multiple of 32 the tiles never leave the region)
    const bool fits = has_tab && static_cast<uint64_t>(pos_begin) + static_cast<uint64_t>(n_tiles) * 32u <= L.num_tokens;
    const char* general_env = getenv("SPECKV_ATTEND_GENERAL");
    const uint8_t* lin_base = (general_env || !fits) ? nullptr : a->linear_base;
    // regular striping over several pools: the same kernel with computed record addresses (no page-table chase)
    const bool striped = !lin_base && fits && a->stripe_n >= 2 && !general_env;
    // no regular placement (pages migrated one by one), or SPECKV_ATTEND_GENERAL set (measurements, tests): the fast kernel
    // with its record addresses from the page table, looked up one request ahead
    const bool table = fits && !lin_base && !striped;
    // (the page-table form has nothing to gain from whole rows: it hides its look-ups behind other waves and always
    // goes through the merge -- 80 layers x 8k: one split 0.13 of HBM peak, eight 0.18+)
    uint32_t want = (lin_base && rows / 4u >= 128u && n_tiles < 768u) ? 1u : (5120u + rows - 1u) / rows;     // (32k and beyond: 8 splits, below)
    // per-layer calls are latency-bound: short contexts want short splits (measured best: 2 tiles per split at 2k
    // context, 4 at 8k, 8 at 32k), long multi-layer launches are bounded by `want` above
    const uint32_t min_tiles = std::min(8u, std::max(2u, n_tiles / 64u));
    want = std::min(want, std::max(1u, n_tiles / min_tiles));
    if (const char* env = getenv("SPECKV_ATTEND_SPLITS")) want = static_cast<uint32_t>(atoi(env));
    const EvenSplit es = even_split(n_tiles, std::max(1u, std::min(want, 2048u)));
    const uint32_t n_splits = es.n_splits, tiles_per_split = es.tiles_per_split;
    const size_t q_bytes = static_cast<size_t>(rows) * 16 * 128, qs_bytes = static_cast<size_t>(rows) * 16 * sizeof(float);
    const size_t acc_bytes = static_cast<size_t>(rows) * n_splits * 16 * 128 * sizeof(float);
    const size_t ml_bytes = static_cast<size_t>(rows) * n_splits * 32 * sizeof(float);
    uint8_t* buf = static_cast<uint8_t*>(scratch(s_attn_, q_bytes + qs_bytes + acc_bytes + ml_bytes, s));
    if (!buf) return SPECKV_ERR_NOMEM;
    AttendArgs k{};
    k.entries = a->d_entries;
    k.k_first = k_first;
    k.v_first = v_first;
    k.layer_stride = layer_stride;
    k.n_pages = n_pages;
    k.skip_pages = skip_pages;
    k.heads = L.num_heads;
    k.g = g;
    k.n_splits = n_splits;
    k.tiles_per_split = tiles_per_split;
    k.q8 = buf;
    k.qs = reinterpret_cast<float*>(buf + q_bytes);
    k.scale_log2e = sm_scale * 1.4426950408889634f;
    k.zero_page = d_zero_page_;
    k.scale_tab = a->d_scale_tab;
    k.q16 = static_cast<const uint16_t*>(d_q_f16);
    k.lin_base = lin_base;
    if (striped) {
        k.stripe_bases = a->d_stripe;
        k.stripe_n = a->stripe_n;
        k.stripe_magic = static_cast<uint32_t>((1ull << 32) / a->stripe_n + 1u);
    }
    k.part_acc = reinterpret_cast<float*>(buf + q_bytes + qs_bytes);
    k.part_ml = reinterpret_cast<float*>(buf + q_bytes + qs_bytes + acc_bytes);
    if (table) { k.table_form = 1u; k.lin_base = nullptr; k.stripe_bases = nullptr; }
    if (!k.lin_base && !striped && !table)         // the linear / striped / table forms quantise the query in their own prologue
        HIP_TRY(launch_quantize_q_e4m3(d_q_f16, rows, g, L.head_dim, buf, reinterpret_cast<float*>(buf + q_bytes), st));
    if (n_splits == 1u) { k.direct_out = d_out; k.direct_lse = d_lse; }      // no merge launch (linear / striped form)
    HIP_TRY(launch_attend_fp8(k, n_layers, d_out, d_lse, st));
    note_use(a, s);
    if (!s) RC_TRY(wait_stream());
    return SPECKV_OK;
}

// One decode step of a batch: the fused attention of ONE layer for many sequences (allocations) in one launch
// (BASELINE configs[3] shape: 256 sequences).  Every allocation must qualify for the linear form.
// Not capturable into a HIP graph: the per-call descriptors travel through a pinned slot that later calls reuse, so a
// replay would read other calls' descriptors -- the call refuses to run on a capturing stream (the per-sequence
// entry points speckv_ext_attend_fp8 / _int4 are capturable).
// Split length of a batch launch (see the measurements quoted in attend_batch).  seqs[i].n_splits holds the tile count of
// sequence i (null: n_seq sequences of uniform_tiles each, the bound a plan is sized for).
// INT4 batch launches between half a machine and a whole one of workgroup columns: every long sequence in a long and a short
// piece, dispatched rows-first (ring_rule.hpp: int4_unequal_fraction / unequal_pieces).  The environment switches are for
// measurement runs.
using UnequalSplit = UnequalFraction;
static UnequalSplit int4_unequal_split(uint32_t n_seq, uint32_t hq, uint32_t tiles_max)
{
    if (getenv("SPECKV_ATTEND_TILES_PER_SPLIT") || getenv("SPECKV_ATTEND_WG_TARGET") || getenv("SPECKV_ATTEND_WHOLE_SEQUENCES")) return {false, 1.0};
    UnequalSplit u = int4_unequal_fraction(n_seq * hq, tiles_max);
    if (u.on) if (const char* env = getenv("SPECKV_ATTEND_UNEQUAL_A")) u.a = atof(env);
    return u;
}

static uint32_t batch_tiles_per_split(bool fp8, uint32_t n_seq, uint32_t heads, uint64_t total_tiles, const AttendSeq* seqs,
                                      uint32_t uniform_tiles)
{
    if (const char* env = getenv("SPECKV_ATTEND_TILES_PER_SPLIT")) return std::max(1, atoi(env));
    const char* target_env = getenv("SPECKV_ATTEND_WG_TARGET");                // (measurement runs: the plain workgroup target)
    const uint32_t hq = heads / 4u;
    if (fp8 && !target_env) {
        // FP8: the busiest-CU cost rule of ring_rule.hpp (48 sequences x 16k: 288 workgroups 0.50 of HBM peak, 192: 0.64,
        // 768: 0.71; 32 x 32k: 256 workgroups 0.79, 512: 0.76, 384: 0.63; 128 x 2k: unsplit 0.73, two splits 0.56)
        std::vector<uint32_t> tiles;
        if (seqs) { tiles.resize(n_seq); for (uint32_t i = 0; i < n_seq; ++i) tiles[i] = seqs[i].n_splits; }
        const char* mc = getenv("SPECKV_FP8_MERGE_COST");                     // (measurement runs)
        return fp8_batch_tiles_per_split(seqs ? tiles.data() : nullptr, n_seq, uniform_tiles, hq, 256u, mc ? static_cast<uint32_t>(atoi(mc)) : 8u);
    }
    uint64_t wg_target = fp8 ? 512u : 768u;
    if (target_env) wg_target = std::max(1, atoi(target_env));
    uint32_t tps = static_cast<uint32_t>(std::max<uint64_t>(8, (total_tiles * hq + wg_target - 1u) / wg_target));
    if (!fp8) tps = (static_cast<uint64_t>(n_seq) * hq >= 384u) ? 256u : std::min(tps, 256u);   // enough columns: whole sequences
    return tps;
}

// INT4 batches on the whole-record kernel (k_attend_int4_wg8<2>: workgroups = sequences x splits, one 16-wave workgroup per
// CU resident, its two halves merged in LDS): one round of resident workgroups when the batch is smaller than that, whole
// sequences otherwise (a whole sequence is final: no partials, no merge launch); never under 32 tiles a split.
static bool int4_batch_wg8() { static const bool on = !getenv("SPECKV_INT4_WG4"); return on; }
static uint32_t int4_wg8_batch_tps(uint32_t n_seq, uint32_t tiles_max)
{
    if (const char* env = getenv("SPECKV_ATTEND_TILES_PER_SPLIT")) return std::max(1, atoi(env));      // (measurement runs)
    const uint32_t resident = 256u;                                       // 16-wave workgroups (two halves each), one per CU
    const uint32_t splits = std::max(1u, resident / std::max(1u, n_seq));
    return std::max(32u, (tiles_max + splits - 1u) / splits);
}

int Engine::attend_batch(int scheme, uint32_t n_seq, const uint64_t* handles, uint32_t layer, const void* d_q_f16, uint32_t g,
                         const uint32_t* pos_end, float sm_scale, float* d_out, float* d_lse, hipStream_t s)
{
    const bool fp8 = scheme == SPECKV_COMP_FP8_E4M3;
    if (null_) return no_data_path("speckv_ext_attend_*_batch");
    if (n_seq == 0) return SPECKV_OK;
    if (is_capturing(s)) {
        SPECKV_ERR("speckv_ext_attend_*_batch cannot be captured into a HIP graph (its descriptors are staged per call); "
                   "capture the per-sequence speckv_ext_attend_fp8 / _int4 calls instead");
        return SPECKV_ERR_INVAL;
    }
    if (!handles || !pos_end || !d_q_f16 || !d_out || g == 0 || g > 16) return SPECKV_ERR_INVAL;
    std::vector<AttendSeq> seqs(n_seq);
    uint64_t total_tiles = 0;
    uint32_t heads = 0;
    bool any_striped = false;                 // then the whole launch takes the striped kernels (a single run is "striped over 1")
    bool any_table = false;                   // ... or, with a member that has no regular placement, the table forms
    std::vector<const PageEntry*> ents(n_seq);
    for (uint32_t i = 0; i < n_seq; ++i) {
        Allocation* a = find(handles[i]);
        if (!a) return SPECKV_ERR_GENERAL;
        if (!a->has_layout || a->scheme != scheme) return SPECKV_ERR_INVAL;
        const Layout& L = a->layout;
        if (L.head_dim != 128 || L.bytes_per_element != 2 || L.num_heads * L.head_dim != 1024 || L.num_tokens % 2) return SPECKV_ERR_INVAL;
        if (layer >= L.num_layers || pos_end[i] % 2 || pos_end[i] > L.num_tokens) return SPECKV_ERR_INVAL;
        const uint32_t n_pages = pos_end[i] / 2, n_tiles = (n_pages + 15u) / 16u;
        if ((fp8 && !a->d_scale_tab) || static_cast<uint64_t>(n_tiles) * 32u > L.num_tokens) {
            SPECKV_ERR("speckv_ext_attend_*_batch: sequence %u does not qualify for the tile-aligned forms (pos_end rounded up "
                       "to 32 inside the layer%s)", i, fp8 ? ", layout with num_tokens %% 32 == 0" : "");
            return SPECKV_ERR_INVAL;
        }
        heads = L.num_heads;
        note_use(a, s);
        any_table = any_table || !a->stripe_n;                 // no regular placement (migrated pages): the launch reads addresses from the page tables
        ents[i] = a->d_entries;
        any_striped = any_striped || !a->linear_base;
        seqs[i].stripe_bases = a->d_stripe;
        seqs[i].stripe_n = a->stripe_n;
        seqs[i].lin_base = a->linear_base;
        seqs[i].scale_tab = a->d_scale_tab;
        seqs[i].k_first = static_cast<uint64_t>(layer) * L.num_tokens;       // (layer*2*T)/2
        seqs[i].v_first = seqs[i].k_first + L.num_tokens / 2;
        seqs[i].n_pages = n_pages;
        seqs[i].layer_pages = L.num_tokens;                                   // K + V pages of one layer
        seqs[i].n_splits = n_tiles;                                           // tiles for now, splits below
        total_tiles += n_tiles;
    }
    DeviceScope device_scope(device_);
    // NULL = the engine's stream and a synchronous call: the query may have been produced on any stream of the caller
    if (!s) HIP_TRY(hipDeviceSynchronize());
    hipStream_t st = s ? s : stream_;
    const size_t out_elems = static_cast<size_t>(n_seq) * heads * g * 128;
    if (total_tiles == 0) {
        HIP_TRY(hipMemsetAsync(d_out, 0, out_elems * sizeof(float), st));
        if (!s) HIP_TRY(hipStreamSynchronize(stream_));
        return SPECKV_OK;
    }
    // one split length for the whole batch.  A batch brings its own parallelism: the fewer, longer splits the better, down
    // to about one round of resident workgroups (256 sequences x 8k context, one layer, FP8: 8 tiles per split 0.50 of
    // HBM peak, 32: 0.59, 64: 0.67, 128: 0.72, 256 = no split: 0.74; INT4: 64..128 best, 0.59; at 2k context both
    // formats want no split at all).  INT4 target: 768 workgroups, never under 8 tiles per split; FP8: the cost rule of
    // batch_tiles_per_split.
    // INT4 (arithmetic-bound kernel): splits longer than 256 tiles stop paying (256 sequences x 32k: 256 tiles per split
    // 0.67, 512: 0.65, 1024 = no split: 0.60), shorter sequences are best left whole (8k 0.63 against 0.59 in two
    // splits, 4k 0.60 / 0.52, 2k 0.58 / 0.43: single-split rows are final, no partials and no merge).
    if (any_table)                                           // (AttendSeq::lin_base carries the page table in table launches)
        for (uint32_t i = 0; i < n_seq; ++i) seqs[i].lin_base = reinterpret_cast<const uint8_t*>(ents[i]);
    uint32_t tiles_max = 0;
    for (uint32_t i = 0; i < n_seq; ++i) tiles_max = std::max(tiles_max, seqs[i].n_splits);
    const bool wg8 = !fp8 && !any_striped && !any_table && heads == 8u && int4_batch_wg8();
    const uint32_t tps = wg8 ? int4_wg8_batch_tps(n_seq, tiles_max) : batch_tiles_per_split(fp8, n_seq, heads, total_tiles, seqs.data(), 0);
    const UnequalSplit unequal = (fp8 || wg8) ? UnequalSplit{false, 1.0} : int4_unequal_split(n_seq, heads / 4u, tiles_max);
    uint32_t max_splits = 0;
    uint64_t parts = 0;
    for (uint32_t i = 0; i < n_seq; ++i) {
        const uint32_t n_tiles = seqs[i].n_splits;
        if ((n_tiles + tps - 1u) / tps > 2048u) return SPECKV_ERR_INVAL;
        // the sequence's tiles divided evenly over its splits (171 + 85 tiles instead of 128 + 128 cost 15 %)
        EvenSplit es = even_split(n_tiles, (n_tiles + tps - 1u) / tps);
        if (unequal.on) es = unequal_pieces(unequal, n_tiles);
        seqs[i].tiles_per_split = n_tiles ? es.tiles_per_split : tps;
        seqs[i].n_splits = es.n_splits;
        seqs[i].part_base = static_cast<uint32_t>(parts);
        parts += static_cast<uint64_t>(heads) * seqs[i].n_splits;
        max_splits = std::max(max_splits, seqs[i].n_splits);
    }
    const size_t acc_bytes = static_cast<size_t>(parts) * 16 * 128 * sizeof(float), ml_bytes = static_cast<size_t>(parts) * 32 * sizeof(float);
    uint8_t* buf = static_cast<uint8_t*>(scratch(s_attn_, acc_bytes + ml_bytes, s));
    AttendSeq* d_seqs = static_cast<AttendSeq*>(scratch(s_attn_seq_, seqs.size() * sizeof(AttendSeq), s));
    if (!buf || !d_seqs) return SPECKV_ERR_NOMEM;
    // descriptors go through a pinned slot so the call can return without waiting for the copy
    const size_t seq_bytes = seqs.size() * sizeof(AttendSeq);
    if (seq_ring_.slot_bytes < seq_bytes) {
        if (seq_ring_.base) { HIP_TRY(hipDeviceSynchronize()); (void)hipHostFree(seq_ring_.base); seq_ring_.base = nullptr; }
        seq_ring_.slot_bytes = std::max<size_t>(seq_bytes * 2, 16384);
        HIP_TRY(hipHostMalloc(&seq_ring_.base, seq_ring_.slot_bytes * 4, hipHostMallocDefault));
        for (auto& ev : seq_ring_.ev)
            if (!ev) HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    }
    const int slot = seq_ring_.next;
    seq_ring_.next = (slot + 1) & 3;
    HIP_TRY(hipEventSynchronize(seq_ring_.ev[slot]));          // the copy that last used this slot has finished
    void* staged = static_cast<uint8_t*>(seq_ring_.base) + static_cast<size_t>(slot) * seq_ring_.slot_bytes;
    memcpy(staged, seqs.data(), seq_bytes);
    HIP_TRY(hipMemcpyAsync(d_seqs, staged, seq_bytes, hipMemcpyHostToDevice, st));
    HIP_TRY(hipEventRecord(seq_ring_.ev[slot], st));
    // sequences without positions have no splits: their rows are written as zeros by the merge (L == 0)
    AttendArgs k{};
    k.heads = heads;
    k.g = g;
    k.n_splits = max_splits;
    k.tiles_per_split = tps;
    k.layer_stride = 0;
    k.q16 = static_cast<const uint16_t*>(d_q_f16);
    k.q8 = static_cast<const uint8_t*>(d_q_f16);          // the INT4 kernel reads the fp16 query through q8
    k.scale_log2e = sm_scale * 1.4426950408889634f;
    k.lin_base = (any_striped || any_table) ? nullptr : seqs[0].lin_base;           // (overridden per sequence)
    if (any_striped) k.stripe_bases = seqs[0].stripe_bases;          // marks a striped launch (each sequence brings its own table)
    if (any_table) {
        k.table_form = 1u; k.lin_base = nullptr; k.stripe_bases = nullptr;
        if (!d_zero_page_) {
            HIP_TRY(hipMalloc(reinterpret_cast<void**>(&d_zero_page_), kPageSize));
            HIP_TRY(hipMemset(d_zero_page_, 0, kPageSize));
        }
        k.zero_page = d_zero_page_;
    }
    k.seqs = d_seqs;
    k.part_acc = reinterpret_cast<float*>(buf);
    k.part_ml = reinterpret_cast<float*>(buf + acc_bytes);
    bool one_split_each = true;                           // then the attention kernel writes the final rows itself
    for (uint32_t i = 0; i < n_seq; ++i) one_split_each = one_split_each && seqs[i].n_splits == 1u;
    if (one_split_each) { k.direct_out = d_out; k.direct_lse = d_lse; }
    if (unequal.on) k.rows_first = 1u;
    if (wg8) k.wg8 = 1u;
    if (fp8) {
        HIP_TRY(launch_attend_fp8_batch(k, n_seq, d_out, d_lse, st));
    } else {
        HIP_TRY(launch_attend_int4(k, n_seq, st));        // grid y = sequences x head groups, as for layers
        if (!k.direct_out) HIP_TRY(launch_attend_combine(k, n_seq, d_out, d_lse, st));
    }
    if (!s) HIP_TRY(hipStreamSynchronize(stream_));
    return SPECKV_OK;
}

// ---- planned batches: descriptors resident on the device, the launches capturable ---------------------------------------
// A decode step under a HIP graph replays the same launches with new sequence lengths.  speckv_ext_attend_batch_plan
// (outside the graph, once per step) writes one descriptor per sequence -- valid for every layer -- into a device buffer
// of the caller; speckv_ext_attend_*_planned is kernel launches only: no handle look-ups, no staging, grid and scratch
// sized from max_pos_end alone, so a captured launch stays valid for as long as the lengths stay within that bound.
struct PlanGeometry { uint32_t tps, max_splits; uint64_t parts_bound; UnequalSplit unequal; };
static PlanGeometry plan_geometry(bool fp8, uint32_t n_seq, uint32_t heads, uint32_t max_pos_end)
{
    const uint32_t tiles_max = (max_pos_end / 2u + 15u) / 16u;
    PlanGeometry g{};
    if (!fp8 && heads == 8u && int4_batch_wg8()) {            // the whole-record kernel's geometry (a striped / table launch runs it on the 4-head kernels)
        g.unequal = UnequalSplit{false, 1.0};
        g.tps = int4_wg8_batch_tps(n_seq, tiles_max);
        g.max_splits = std::max(1u, (tiles_max + g.tps - 1u) / g.tps);
        g.parts_bound = static_cast<uint64_t>(n_seq) * heads * g.max_splits;
        return g;
    }
    g.unequal = fp8 ? UnequalSplit{false, 1.0} : int4_unequal_split(n_seq, heads / 4u, tiles_max);
    if (g.unequal.on) {                                       // (the rule depends on the plan's bound only: plan and launch agree)
        g.tps = tiles_max;
        g.max_splits = 2u;
        g.parts_bound = static_cast<uint64_t>(n_seq) * heads * 2u;
        return g;
    }
    g.tps = batch_tiles_per_split(fp8, n_seq, heads, static_cast<uint64_t>(tiles_max) * n_seq, nullptr, tiles_max);
    g.max_splits = std::max(1u, (tiles_max + g.tps - 1u) / g.tps);
    g.parts_bound = static_cast<uint64_t>(n_seq) * heads * g.max_splits;
    return g;
}

int Engine::attend_batch_plan(uint32_t n_seq, const uint64_t* handles, const uint32_t* pos_end, uint32_t max_pos_end,
                              void* d_plan, size_t plan_bytes, hipStream_t s)
{
    if (null_) return no_data_path("speckv_ext_attend_batch_plan");
    if (n_seq == 0) return SPECKV_OK;
    if (!handles || !pos_end || !d_plan || !s || max_pos_end % 2 || plan_bytes < n_seq * sizeof(AttendSeq)) return SPECKV_ERR_INVAL;
    if (is_capturing(s)) return SPECKV_ERR_INVAL;            // the plan is what changes between replays: it stays outside the graph
    std::vector<AttendSeq> seqs(n_seq);
    int scheme = -1;
    bool any_striped = false, any_table = false;
    std::vector<const PageEntry*> ents(n_seq);
    uint32_t heads = 0, min_layers = UINT32_MAX;
    for (uint32_t i = 0; i < n_seq; ++i) {
        Allocation* a = find(handles[i]);
        if (!a) return SPECKV_ERR_GENERAL;
        if (scheme < 0) scheme = a->scheme;
        if (!a->has_layout || a->scheme != scheme || (scheme != SPECKV_COMP_FP8_E4M3 && scheme != SPECKV_COMP_INT4_G32)) return SPECKV_ERR_INVAL;
        const Layout& L = a->layout;
        if (L.head_dim != 128 || L.bytes_per_element != 2 || L.num_heads * L.head_dim != 1024 || L.num_tokens % 2) return SPECKV_ERR_INVAL;
        if (pos_end[i] % 2 || pos_end[i] > L.num_tokens || pos_end[i] > max_pos_end) return SPECKV_ERR_INVAL;
        const uint32_t n_pages = pos_end[i] / 2, n_tiles = (n_pages + 15u) / 16u;
        const bool fp8 = scheme == SPECKV_COMP_FP8_E4M3;
        if ((fp8 && !a->d_scale_tab) || static_cast<uint64_t>(n_tiles) * 32u > L.num_tokens) return SPECKV_ERR_INVAL;
        min_layers = std::min(min_layers, L.num_layers);
        heads = L.num_heads;
        note_use(a, s);
        any_table = any_table || !a->stripe_n;
        ents[i] = a->d_entries;
        any_striped = any_striped || !a->linear_base;
        seqs[i].stripe_bases = a->d_stripe;
        seqs[i].stripe_n = a->stripe_n;
        seqs[i].lin_base = a->linear_base;
        seqs[i].scale_tab = a->d_scale_tab;
        seqs[i].k_first = 0;                                   // layer 0; the launch adds layer * layer_pages
        seqs[i].v_first = L.num_tokens / 2;
        seqs[i].layer_pages = L.num_tokens;
        seqs[i].n_pages = n_pages;
        seqs[i].n_splits = n_tiles;
    }
    const PlanGeometry g = plan_geometry(scheme == SPECKV_COMP_FP8_E4M3, n_seq, heads, max_pos_end);
    if (g.max_splits > 2048u) return SPECKV_ERR_INVAL;
    if (plans_.size() >= 64 && !plans_.count(d_plan)) plans_.clear();        // (buffers of long-gone steps)
    if (any_table)
        for (uint32_t i = 0; i < n_seq; ++i) seqs[i].lin_base = reinterpret_cast<const uint8_t*>(ents[i]);
    if (any_table && !d_zero_page_) {
        DeviceScope zero_scope(device_);
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&d_zero_page_), kPageSize));
        HIP_TRY(hipMemset(d_zero_page_, 0, kPageSize));
    }
    plans_[d_plan] = PlanInfo{n_seq, scheme, min_layers, max_pos_end, any_striped, any_table};
    uint64_t parts = 0;
    for (uint32_t i = 0; i < n_seq; ++i) {
        const uint32_t n_tiles = seqs[i].n_splits;
        const EvenSplit es = g.unequal.on ? unequal_pieces(g.unequal, n_tiles) : even_split(n_tiles, (n_tiles + g.tps - 1u) / g.tps);
        seqs[i].tiles_per_split = n_tiles ? es.tiles_per_split : g.tps;
        seqs[i].n_splits = es.n_splits;
        seqs[i].part_base = static_cast<uint32_t>(parts);
        parts += static_cast<uint64_t>(heads) * seqs[i].n_splits;
    }
    DeviceScope device_scope(device_);
    const size_t seq_bytes = seqs.size() * sizeof(AttendSeq);
    if (seq_ring_.slot_bytes < seq_bytes) {
        if (seq_ring_.base) { HIP_TRY(hipDeviceSynchronize()); (void)hipHostFree(seq_ring_.base); seq_ring_.base = nullptr; }
        seq_ring_.slot_bytes = std::max<size_t>(seq_bytes * 2, 16384);
        HIP_TRY(hipHostMalloc(&seq_ring_.base, seq_ring_.slot_bytes * 4, hipHostMallocDefault));
        for (auto& ev : seq_ring_.ev)
            if (!ev) HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    }
    const int slot = seq_ring_.next;
    seq_ring_.next = (slot + 1) & 3;
    HIP_TRY(hipEventSynchronize(seq_ring_.ev[slot]));
    void* staged = static_cast<uint8_t*>(seq_ring_.base) + static_cast<size_t>(slot) * seq_ring_.slot_bytes;
    memcpy(staged, seqs.data(), seq_bytes);
    HIP_TRY(hipMemcpyAsync(d_plan, staged, seq_bytes, hipMemcpyHostToDevice, s));
    HIP_TRY(hipEventRecord(seq_ring_.ev[slot], s));
    return SPECKV_OK;
}

int Engine::attend_planned(int scheme, const void* d_plan, uint32_t n_seq, uint32_t layer, const void* d_q_f16, uint32_t g,
                           uint32_t max_pos_end, float sm_scale, float* d_out, float* d_lse, hipStream_t s)
{
    const bool fp8 = scheme == SPECKV_COMP_FP8_E4M3;
    if (null_) return no_data_path("speckv_ext_attend_*_planned");
    if (n_seq == 0) return SPECKV_OK;
    if (!d_plan || !d_q_f16 || !d_out || !s || g == 0 || g > 16 || max_pos_end % 2 || max_pos_end == 0) return SPECKV_ERR_INVAL;
    const uint32_t heads = 8;                                  // the page-wise layout: 8 kv heads x 128
    const auto plan = plans_.find(d_plan);                     // what speckv_ext_attend_batch_plan last wrote there
    if (plan == plans_.end() || plan->second.n_seq != n_seq || plan->second.scheme != scheme || plan->second.max_pos_end != max_pos_end ||
        layer >= plan->second.n_layers) {
        SPECKV_ERR("speckv_ext_attend_*_planned: no plan of this shape at %p (n_seq, format and max_pos_end as planned, layer inside every layout)", d_plan);
        return SPECKV_ERR_INVAL;
    }
    const PlanGeometry pg = plan_geometry(fp8, n_seq, heads, max_pos_end);
    DeviceScope device_scope(device_);
    const size_t acc_bytes = static_cast<size_t>(pg.parts_bound) * 16 * 128 * sizeof(float), ml_bytes = static_cast<size_t>(pg.parts_bound) * 32 * sizeof(float);
    uint8_t* buf = static_cast<uint8_t*>(scratch(s_attn_, acc_bytes + ml_bytes, s));      // (growth during a capture is refused: warm up once)
    if (!buf) return is_capturing(s) ? SPECKV_ERR_INVAL : SPECKV_ERR_NOMEM;
    AttendArgs k{};
    k.heads = heads;
    k.g = g;
    k.n_splits = pg.max_splits;
    k.tiles_per_split = pg.tps;
    k.layer_stride = 0;
    k.q16 = static_cast<const uint16_t*>(d_q_f16);
    k.q8 = static_cast<const uint8_t*>(d_q_f16);
    k.scale_log2e = sm_scale * 1.4426950408889634f;
    if (plan->second.table) { k.table_form = 1u; k.zero_page = d_zero_page_; }           // a member without a regular placement: addresses from the page tables
    else if (plan->second.striped) k.stripe_bases = reinterpret_cast<const uint64_t*>(1);     // striped launch: every descriptor brings its table
    else { k.lin_base = reinterpret_cast<const uint8_t*>(1); if (!fp8 && int4_batch_wg8()) k.wg8 = 1u; }     // non-null: linear form (the real base comes from the descriptor)
    k.seqs = static_cast<const AttendSeq*>(d_plan);
    k.part_acc = reinterpret_cast<float*>(buf);
    k.part_ml = reinterpret_cast<float*>(buf + acc_bytes);
    k.batch_layer = layer;
    k.direct_out = d_out;                                      // sequences with a single split are written directly ...
    k.direct_lse = d_lse;
    // ... decided per sequence on the device, the merge skips those; a geometry of one split at most needs no merge at all
    k.direct_per_seq = pg.max_splits == 1u ? 2u : 1u;
    if (pg.unequal.on) k.rows_first = 1u;
    if (fp8) {
        HIP_TRY(launch_attend_fp8_batch(k, n_seq, d_out, d_lse, s));
    } else {
        HIP_TRY(launch_attend_int4(k, n_seq, s));
        if (k.direct_per_seq != 2u) HIP_TRY(launch_attend_combine(k, n_seq, d_out, d_lse, s));
    }
    return SPECKV_OK;
}

int Engine::attend_fold_tail(uint32_t n_rows, const uint32_t* d_rows, uint32_t heads, uint32_t g, const void* d_q_f16, const void* d_k_tail,
                             const void* d_v_tail, uint64_t tail_stride_elems, float sm_scale, float* d_out, float* d_lse, hipStream_t s)
{
    if (null_) return no_data_path("speckv_ext_attend_fold_tail");
    if (n_rows == 0) return SPECKV_OK;
    if (!d_q_f16 || !d_k_tail || !d_v_tail || !d_out || !d_lse || heads == 0 || g == 0 || g > 16 || tail_stride_elems % 2 ||
        tail_stride_elems < static_cast<uint64_t>(heads) * 128u)
        return SPECKV_ERR_INVAL;
    DeviceScope device_scope(device_);
    if (!s) HIP_TRY(hipDeviceSynchronize());                   // NULL: the engine's stream, synchronous (include/speckv_ext.h)
    HIP_TRY(launch_attend_fold_tail(n_rows, d_rows, heads, g, d_q_f16, d_k_tail, d_v_tail, tail_stride_elems, sm_scale, d_out, d_lse,
                                    s ? s : stream_));
    if (!s) HIP_TRY(hipStreamSynchronize(stream_));
    return SPECKV_OK;
}

// Launch geometry of the whole-record INT4 kernel (k_attend_int4_wg8; 512-thread workgroups, two resident per CU).
static int device_cus()
{
    static const int n_cus = [] {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) { (void)hipGetLastError(); v = 256; }
        return v;
    }();
    return n_cus;
}
// Stream form (many layers of one sequence): the launch's n_layers x n_tiles tiles, layer-major, in as many equal pieces as
// workgroups are resident at once -- one pipeline fill per workgroup, no partial last round, few partials per layer.  Worth
// it when a piece is long enough to amortise its fill (>= 16 tiles); *max_slots = most pieces any layer is cut into.
static bool int4_wg8_stream(uint32_t n_layers, uint32_t n_tiles, AttendArgs::Stream* out)
{
    const uint64_t total = static_cast<uint64_t>(n_layers) * n_tiles;
    uint64_t wgs = 2ull * static_cast<uint64_t>(device_cus());
    if (const char* env = getenv("SPECKV_INT4_STREAM_WGS")) wgs = std::max(1, atoi(env));
    if (n_layers < 2 || total < 16u * wgs || getenv("SPECKV_INT4_NO_STREAM")) return false;
    out->n_wgs = static_cast<uint32_t>(wgs);
    out->len = static_cast<uint32_t>(total / wgs);
    out->rem = static_cast<uint32_t>(total % wgs);
    out->max_slots = 1;
    for (uint32_t l = 0; l < n_layers; ++l) out->max_slots = std::max(out->max_slots, attend_stream_count(l, n_tiles, out->len, out->rem));
    return true;
}
// Fixed grid (per-layer calls, short launches): splits x layers workgroups, in whole rounds of the resident set when the
// launch is that long, else as many 8-tile pieces as there are.
static uint32_t int4_wg8_splits(uint32_t n_layers, uint32_t n_tiles)
{
    const uint64_t resident = static_cast<uint64_t>(device_cus());                 // (16-wave workgroups, two halves each: one per CU)
    const uint64_t total = static_cast<uint64_t>(n_layers) * n_tiles;
    uint64_t wgs = std::max<uint64_t>(1, total / 64u);
    if (wgs >= resident) wgs = (wgs + resident / 2u) / resident * resident;       // whole rounds
    else wgs = std::min<uint64_t>(resident, std::max<uint64_t>(wgs, total / 8u));
    const uint64_t per_layer = std::max<uint64_t>(1, (wgs + n_layers / 2u) / n_layers);
    return static_cast<uint32_t>(std::min<uint64_t>(per_layer, std::max<uint32_t>(1u, n_tiles / 4u)));
}

// Fused decode attention over INT4_G32 K and V records (attend_int4.hip): linear placement only.
int Engine::attend_int4(uint64_t handle, uint32_t layer, uint32_t n_layers, const void* d_q_f16, uint32_t g,
                        uint32_t pos_begin, uint32_t pos_end, float sm_scale, float* d_out, float* d_lse, hipStream_t s)
{
    if (null_) return no_data_path("speckv_ext_attend_int4");
    Allocation* a = find(handle);
    if (!a) return SPECKV_ERR_GENERAL;
    if (!a->has_layout || a->scheme != SPECKV_COMP_INT4_G32) return SPECKV_ERR_INVAL;
    const Layout& L = a->layout;
    if (L.head_dim != 128 || L.bytes_per_element != 2 || L.num_heads * L.head_dim != 1024 || L.num_tokens % 2) return SPECKV_ERR_INVAL;
    if (n_layers == 0 || layer >= L.num_layers || n_layers > L.num_layers - layer || pos_begin % 2 || pos_begin > pos_end ||
        pos_end > L.num_tokens || pos_end % 2)
        return SPECKV_ERR_INVAL;
    if (g == 0 || g > 16 || !d_q_f16 || !d_out) return SPECKV_ERR_INVAL;
    const uint32_t n_pages = (pos_end - pos_begin) / 2;
    DeviceScope device_scope(device_);
    // NULL = the engine's stream and a synchronous call: the query may have been produced on any stream of the caller
    if (!s) HIP_TRY(hipDeviceSynchronize());
    hipStream_t st = s ? s : stream_;
    const size_t out_elems = static_cast<size_t>(n_layers) * L.num_heads * g * 128;
    if (n_pages == 0) {
        HIP_TRY(hipMemsetAsync(d_out, 0, out_elems * sizeof(float), st));
        if (!s) HIP_TRY(hipStreamSynchronize(stream_));
        return SPECKV_OK;
    }
    const uint64_t k_first = (static_cast<uint64_t>(layer) * 2 * L.num_tokens + pos_begin) / 2;
    const uint64_t v_first = k_first + L.num_tokens / 2;
    const uint64_t layer_stride = static_cast<uint64_t>(L.num_tokens);
    if (v_first + (n_layers - 1) * layer_stride + n_pages > a->n_pages) return SPECKV_ERR_GENERAL;
    const uint32_t n_tiles = (n_pages + 15u) / 16u;
    // linear form: records in one local run and every 32-position tile inside the layer's K / V region; otherwise the
    // page-table form of the same kernel
    const bool fits = static_cast<uint64_t>(pos_begin) + static_cast<uint64_t>(n_tiles) * 32u <= L.num_tokens;
    const char* general_env = getenv("SPECKV_ATTEND_GENERAL");
    const bool linear = a->linear_base && fits && !general_env;
    const bool striped = !linear && a->stripe_n >= 2 && fits && !general_env;
    // everything else -- no regular placement, a last tile that would leave the region, SPECKV_ATTEND_GENERAL (measurements,
    // tests) -- takes the workgroup kernel with its record addresses from the page table: its look-ups are clamped to the
    // range, so a ragged last tile never reads a record it has no business with.  (The per-wave page-table kernel of rounds
    // 1-3, 0.37 of HBM peak, is gone.)
    const bool table = !linear && !striped;
    if (!linear && !d_zero_page_) {
        if (is_capturing(s)) return SPECKV_ERR_INVAL;
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&d_zero_page_), kPageSize));
        HIP_TRY(hipMemset(d_zero_page_, 0, kPageSize));
    }
    const uint32_t rows = n_layers * L.num_heads;
    uint32_t want = (5120u + rows - 1u) / rows;      // VALU-bound kernel: fewer, longer splits measured best
    const uint32_t min_tiles = std::min(8u, std::max(2u, n_tiles / 64u));      // per-layer calls: see attend_fp8
    want = std::min(want, std::max(1u, n_tiles / min_tiles));
    // Workgroups go to the 8 XCDs round-robin by linear id = split + n_splits * (layer, head quad): with a split count that
    // is a multiple of 8 the two workgroups that share a page's scale line (head quads 0 and 1) run on the same XCD, next
    // to each other (measured at 32k x 80 layers: 8 splits 0.598, 10 or 12 splits 0.56, 16 splits 0.595)
    if (want > 8u) want &= ~7u;
    // whole-record kernel (8 waves = 8 heads, one workgroup per CU): workgroups = splits x layers, in whole rounds of the CUs
    const bool wg8 = linear && L.num_heads == 8 && !getenv("SPECKV_INT4_WG4");
    if (wg8) want = int4_wg8_splits(n_layers, n_tiles);
    if (const char* env = getenv("SPECKV_ATTEND_SPLITS")) want = static_cast<uint32_t>(atoi(env));
    EvenSplit es = even_split(n_tiles, std::max(1u, std::min(want, 2048u)));
    if (!wg8 && es.n_splits > 8u && (es.n_splits & 7u) && !getenv("SPECKV_ATTEND_SPLITS"))      // the rounding can fall off a multiple of 8
        es = even_split(n_tiles, es.n_splits & ~7u);
    AttendArgs k{};
    const bool stream = wg8 && !getenv("SPECKV_ATTEND_SPLITS") && int4_wg8_stream(n_layers, n_tiles, &k.stream);
    const uint32_t n_splits = stream ? k.stream.max_slots : es.n_splits, tiles_per_split = es.tiles_per_split;      // (stream: slots per row)
    const size_t acc_bytes = static_cast<size_t>(rows) * n_splits * 16 * 128 * sizeof(float);
    const size_t ml_bytes = static_cast<size_t>(rows) * n_splits * 32 * sizeof(float);
    uint8_t* buf = static_cast<uint8_t*>(scratch(s_attn_, acc_bytes + ml_bytes, s));
    if (!buf) return SPECKV_ERR_NOMEM;
    k.entries = a->d_entries;
    k.k_first = k_first;
    k.v_first = v_first;
    k.layer_stride = layer_stride;
    k.n_pages = n_pages;
    k.heads = L.num_heads;
    k.g = g;
    k.n_splits = n_splits;
    k.tiles_per_split = tiles_per_split;
    k.q8 = static_cast<const uint8_t*>(d_q_f16);
    k.scale_log2e = sm_scale * 1.4426950408889634f;
    k.lin_base = linear ? a->linear_base : nullptr;
    if (table) k.table_form = 1u;
    if (wg8) k.wg8 = 1u;
    if (striped) {
        k.stripe_bases = a->d_stripe;
        k.stripe_n = a->stripe_n;
        k.stripe_magic = static_cast<uint32_t>((1ull << 32) / a->stripe_n + 1u);
    }
    k.zero_page = d_zero_page_;
    k.part_acc = reinterpret_cast<float*>(buf);
    k.part_ml = reinterpret_cast<float*>(buf + acc_bytes);
    if (n_splits == 1u && !stream) { k.direct_out = d_out; k.direct_lse = d_lse; }      // no merge launch
    HIP_TRY(launch_attend_int4(k, n_layers, st));
    if (!k.direct_out) HIP_TRY(launch_attend_combine(k, n_layers, d_out, d_lse, st));
    note_use(a, s);
    if (!s) RC_TRY(wait_stream());
    return SPECKV_OK;
}

// Migration of pool records between pool GPUs (the data-moving counterpart of the
// reference's tier flips, cxl_memory_manager.cpp:130-194, which move nothing):
// hipMemcpyPeerAsync on a dedicated copy stream, one copy per contiguous source run,
// then the device page table is re-pointed and the old slots return to their slab.
int Engine::migrate(uint64_t handle, uint64_t first, uint64_t n, uint32_t target_pool)
{
    if (null_) return no_data_path("speckv_ext_migrate");
    Allocation* a = find(handle);
    if (!a) return SPECKV_ERR_GENERAL;
    if (first > a->n_pages || n > a->n_pages - first) return SPECKV_ERR_GENERAL;
    if (target_pool >= pools_.size()) return SPECKV_ERR_INVAL;
    if (n == 0) return SPECKV_OK;
    DeviceScope device_scope(device_);
    if (a->packed) { RC_TRY(unpack(a)); if ((a = find(handle)) == nullptr) return SPECKV_ERR_GENERAL; }
    RC_TRY(quiesce());
    RC_TRY(wait_stream());
    if ((a = find(handle)) == nullptr) return SPECKV_ERR_GENERAL;
    // asynchronous entry points on caller streams (fetch_range / fetch_list / attend_*) may still be reading the
    // records that are about to move: wait for exactly those streams (the ABI lock stays held: the allocation's
    // placement must not change under another caller)
    for (hipStream_t us : a->user_streams)
        if (hipStreamSynchronize(us) != hipSuccess) (void)hipGetLastError();
    reap(false);
    if (!copy_stream_) HIP_TRY(hipStreamCreateWithFlags(&copy_stream_, hipStreamNonBlocking));
    const size_t stride = a->rec_stride;
    uint8_t* dst = static_cast<uint8_t*>(pools_[target_pool]->alloc(n * stride));
    if (!dst) return SPECKV_ERR_NOMEM;
    struct PoolGuard {                      // the new run goes back to the pool on every error path
        SlabPool* pool; void* p; size_t bytes; bool keep = false;
        ~PoolGuard() { if (!keep) pool->free(p, bytes); }
    } guard{pools_[target_pool].get(), dst, n * stride};
    std::vector<PageEntry> cur(n);
    HIP_TRY(hipMemcpy(cur.data(), a->d_entries + first, n * sizeof(PageEntry), hipMemcpyDeviceToHost));
    const int dst_dev = pools_[target_pool]->device();
    struct Run { uint64_t addr; size_t bytes; int pool; };
    std::vector<Run> old;
    for (uint64_t i = 0; i < n;) {
        uint64_t j = i + 1;
        while (j < n && cur[j].pool_addr == cur[j - 1].pool_addr + stride && a->page_pool[first + j] == a->page_pool[first + i]) ++j;
        const int src_pool = a->page_pool[first + i];
        const size_t bytes = (j - i) * stride;
        HIP_TRY(hipMemcpyPeerAsync(dst + i * stride, dst_dev, reinterpret_cast<const void*>(cur[i].pool_addr),
                                   pools_[src_pool]->device(), bytes, copy_stream_));
        old.push_back({cur[i].pool_addr, bytes, src_pool});
        i = j;
    }
    HIP_TRY(hipStreamSynchronize(copy_stream_));
    HIP_TRY(launch_retarget_entries(a->d_entries + first, n, reinterpret_cast<uint64_t>(dst), stride, stream_));
    HIP_TRY(hipStreamSynchronize(stream_));
    guard.keep = true;
    // bookkeeping: the old runs leave the allocation's extent list (split where needed).  Record strides are
    // multiples of the pool's 128-byte granule, so a sub-run is freed exactly (never reaching into live neighbours).
    for (const Run& r : old) {
        std::vector<Allocation::Extent> next;
        for (const auto& ex : a->extents) {
            const uint64_t lo = reinterpret_cast<uint64_t>(ex.base), hi = lo + ex.bytes;
            if (!ex.base || ex.pool != r.pool || r.addr >= hi || r.addr + r.bytes <= lo) { next.push_back(ex); continue; }
            if (r.addr > lo) next.push_back({ex.pool, ex.base, static_cast<size_t>(r.addr - lo), (r.addr - lo) / stride});
            if (r.addr + r.bytes < hi)
                next.push_back({ex.pool, reinterpret_cast<void*>(r.addr + r.bytes), static_cast<size_t>(hi - r.addr - r.bytes),
                                (hi - r.addr - r.bytes) / stride});
        }
        a->extents.swap(next);
        pools_[r.pool]->free(reinterpret_cast<void*>(r.addr), r.bytes);
    }
    a->extents.push_back({static_cast<int>(target_pool), dst, n * stride, n});
    for (uint64_t i = 0; i < n; ++i) a->page_pool[first + i] = static_cast<uint8_t>(target_pool);
    a->linear_base = nullptr;                     // records no longer lie in one run
    a->regular = false;                           // nor in the striping order the copy engine relies on
    a->stripe_n = 0;                              // (nor the fused attention's striped form; its table goes with the allocation)
    st_.pool_migrated_pages += n;
    if (first == 0 && n == a->n_pages) {
        // The WHOLE allocation moved (a hot sequence pulled onto one pool GPU, typically the compute GPU itself): its records
        // are one run again -- page p at dst + p * stride, never-written slots copied along as the zero bytes they were -- so
        // the placement is regular "over one pool" and every arithmetic-address path applies again: the linear form of the
        // fused attention, the copy engine, the batch descriptors.
        a->extents.clear();
        a->extents.push_back({static_cast<int>(target_pool), dst, n * stride, n});
        a->pool_of_residue.assign(1, static_cast<int>(target_pool));
        a->regular = true;
        const bool fixed_fmt = a->scheme == SPECKV_COMP_FP8_E4M3 || a->scheme == SPECKV_COMP_INT4_G32;
        if (fixed_fmt && a->d_stripe) {
            uint64_t bases[8] = {reinterpret_cast<uint64_t>(dst), 0, 0, 0, 0, 0, 0, 0};
            HIP_TRY(hipMemcpy(a->d_stripe, bases, sizeof(bases), hipMemcpyHostToDevice));
            a->linear_base = dst;
            a->stripe_n = 1;
        }
    }
    return SPECKV_OK;
}

// ---------------------------------------------------------------- compaction (packed INT8_DELTA_RLE records)
// The pool gives every page a worst-case 4 KiB slot, so on its own the reference's variable-length scheme buys no capacity
// (cache_engine.cpp:62-78 only COUNTS compressed_size).  speckv_ext_compact packs the records of an allocation back to back
// (128-byte aligned, page order, one extent per pool GPU) and hands the slot runs back to the slab pool: the allocation is
// "sealed".  Everything that reads goes through the page table and does not care; the copy-engine fetch then moves record
// bytes, not slots.  A write (or a migration) to a sealed allocation first unpacks it into slots again -- sealing is meant
// for sequences that are parked in the pool, not for ones a decode loop appends to.
int Engine::settle_for_relocation(Allocation*& a, uint64_t handle)
{
    RC_TRY(quiesce());
    RC_TRY(order_after_writes());
    RC_TRY(wait_stream());
    if ((a = find(handle)) == nullptr) return SPECKV_ERR_GENERAL;
    for (hipStream_t us : a->user_streams)            // asynchronous readers / writers on caller streams (ABI lock stays held)
        if (hipStreamSynchronize(us) != hipSuccess) (void)hipGetLastError();
    reap(false);
    return SPECKV_OK;
}

int Engine::compact(uint64_t handle, uint64_t* bytes_before, uint64_t* bytes_after)
{
    if (null_) return no_data_path("speckv_ext_compact");
    Allocation* a = find(handle);
    if (!a) return SPECKV_ERR_GENERAL;
    uint64_t before = 0;
    for (const auto& ex : a->extents) before += ex.bytes;
    if (bytes_before) *bytes_before = before;
    if (bytes_after) *bytes_after = before;
    if (a->scheme != SPECKV_COMP_INT8_DELTA_RLE || a->packed || a->n_pages == 0) return SPECKV_OK;   // fixed-size formats: slot == record
    DeviceScope device_scope(device_);
    RC_TRY(settle_for_relocation(a, handle));
    const uint32_t D = static_cast<uint32_t>(a->pool_of_residue.size());
    if (D == 0 || D > 255) return SPECKV_ERR_INVAL;
    std::vector<PageEntry> cur(a->n_pages);
    HIP_TRY(hipMemcpy(cur.data(), a->d_entries, a->n_pages * sizeof(PageEntry), hipMemcpyDeviceToHost));
    // packed offsets per pool (the pool a page lives on NOW: a migration may have moved it), page order, 128-byte aligned
    std::vector<uint64_t> total(pools_.size(), 0), new_addr(a->n_pages);
    std::vector<uint32_t> off128(a->n_pages);
    for (uint64_t p = 0; p < a->n_pages; ++p) {
        const uint32_t k = a->page_pool[p];
        off128[p] = static_cast<uint32_t>(total[k] >> 7);
        total[k] += (static_cast<uint64_t>(cur[p].rec_bytes) + 127u) & ~127ull;
        if ((total[k] >> 7) > 0xFFFFFFFFull) return SPECKV_ERR_NOMEM;
    }
    std::vector<Allocation::Extent> fresh;
    auto undo = [&] { for (auto& ex : fresh) if (ex.base) pools_[ex.pool]->free(ex.base, ex.bytes); };
    std::vector<uint8_t*> base(pools_.size(), nullptr);
    // extents in residue order first (fetch_range_copy_engine reads extents[k] as "the run of residue k"), then any other pool
    std::vector<int> order;
    for (uint32_t k = 0; k < D; ++k) order.push_back(a->pool_of_residue[k]);
    for (size_t k = 0; k < pools_.size(); ++k) if (std::find(order.begin(), order.end(), static_cast<int>(k)) == order.end()) order.push_back(static_cast<int>(k));
    bool regular_pools = true;
    for (uint32_t k = 0; k < D; ++k) for (uint32_t j = 0; j < k; ++j) regular_pools = regular_pools && a->pool_of_residue[k] != a->pool_of_residue[j];
    std::vector<uint64_t> pbytes;
    std::vector<bool> have(pools_.size(), false);
    for (int k : order) {
        // ONE extent per distinct pool.  A pool that stands for several residues (pool_of_residue may repeat one) keeps its
        // place in the list with an empty extent, so that extents[j] / packed_bytes[j] still belong to order[j]; the
        // residue-indexed reader (fetch_range_copy_engine) only runs when the residues' pools are distinct (packed_regular).
        const uint64_t need = have[k] ? 0 : total[k];
        have[k] = true;
        uint8_t* b = need ? static_cast<uint8_t*>(pools_[k]->alloc(need)) : nullptr;
        if (need && !b) { undo(); return SPECKV_ERR_NOMEM; }
        if (need) base[k] = b;
        fresh.push_back({k, b, static_cast<size_t>(need), 0});
        pbytes.push_back(need);
    }
    for (uint64_t p = 0; p < a->n_pages; ++p)
        new_addr[p] = reinterpret_cast<uint64_t>(base[a->page_pool[p]]) + (static_cast<uint64_t>(off128[p]) << 7);
    uint64_t* d_new = static_cast<uint64_t*>(scratch(s_pages_, a->n_pages * sizeof(uint64_t)));
    if (!d_new) { undo(); return SPECKV_ERR_NOMEM; }
    if (hipMemcpy(d_new, new_addr.data(), a->n_pages * sizeof(uint64_t), hipMemcpyHostToDevice) != hipSuccess ||
        launch_repack(a->d_entries, d_new, a->n_pages, stream_) != hipSuccess || hipStreamSynchronize(stream_) != hipSuccess) {
        (void)hipGetLastError();
        undo();
        return SPECKV_ERR_DRIVER;
    }
    for (auto& ex : a->extents) if (ex.base) pools_[ex.pool]->free(ex.base, ex.bytes);
    // the copy engine's condition: pages still striped page % D over D distinct pools (no migration since the allocation)
    bool striped = regular_pools;
    for (uint64_t p = 0; p < a->n_pages && striped; ++p) striped = a->page_pool[p] == static_cast<uint8_t>(a->pool_of_residue[p % D]);
    a->extents.swap(fresh);
    a->packed = true;
    a->packed_regular = striped;
    a->packed_off128.swap(off128);
    a->packed_bytes.swap(pbytes);
    a->regular = false;
    a->linear_base = nullptr;
    a->stripe_n = 0;
    uint64_t after = 0;
    for (const auto& ex : a->extents) after += ex.bytes;
    if (bytes_after) *bytes_after = after;
    st_.compactions++;
    return SPECKV_OK;
}

// A sealed allocation back into fixed slots (the placement of a fresh allocation: page p -> record p / D of the run on pool
// residue p % D when it was striped that way, else one run per pool in page order).
int Engine::unpack(Allocation* a)
{
    if (!a->packed) return SPECKV_OK;
    const uint64_t handle = a->handle;
    RC_TRY(settle_for_relocation(a, handle));
    if (!a->packed) return SPECKV_OK;                   // another thread got here first while we waited
    const uint32_t D = static_cast<uint32_t>(a->pool_of_residue.size());
    const size_t stride = a->rec_stride;
    std::vector<uint64_t> count(pools_.size(), 0), new_addr(a->n_pages);
    for (uint64_t p = 0; p < a->n_pages; ++p) count[a->page_pool[p]]++;
    std::vector<Allocation::Extent> fresh;
    auto undo = [&] { for (auto& ex : fresh) if (ex.base) pools_[ex.pool]->free(ex.base, ex.bytes); };
    std::vector<uint8_t*> base(pools_.size(), nullptr);
    std::vector<int> order;
    for (uint32_t k = 0; k < D; ++k) order.push_back(a->pool_of_residue[k]);
    for (size_t k = 0; k < pools_.size(); ++k) if (std::find(order.begin(), order.end(), static_cast<int>(k)) == order.end()) order.push_back(static_cast<int>(k));
    std::vector<bool> have(pools_.size(), false);
    for (int k : order) {
        const size_t need = have[k] ? 0 : count[k] * stride;            // one extent per distinct pool (see compact())
        const uint64_t recs = have[k] ? 0 : count[k];
        have[k] = true;
        uint8_t* b = need ? static_cast<uint8_t*>(pools_[k]->alloc(need)) : nullptr;
        if (need && !b) { undo(); SPECKV_ERR("a write to a compacted allocation needs %zu bytes of slots again: out of pool memory", need); return SPECKV_ERR_NOMEM; }
        if (need) base[k] = b;
        fresh.push_back({k, b, need, recs});
    }
    std::vector<uint64_t> next(pools_.size(), 0);
    for (uint64_t p = 0; p < a->n_pages; ++p) {
        const uint32_t k = a->page_pool[p];
        const uint64_t rec = a->packed_regular ? p / D : next[k]++;
        new_addr[p] = reinterpret_cast<uint64_t>(base[k]) + rec * stride;
    }
    uint64_t* d_new = static_cast<uint64_t*>(scratch(s_pages_, a->n_pages * sizeof(uint64_t)));
    if (!d_new) { undo(); return SPECKV_ERR_NOMEM; }
    if (hipMemcpy(d_new, new_addr.data(), a->n_pages * sizeof(uint64_t), hipMemcpyHostToDevice) != hipSuccess ||
        launch_repack(a->d_entries, d_new, a->n_pages, stream_) != hipSuccess || hipStreamSynchronize(stream_) != hipSuccess) {
        (void)hipGetLastError();
        undo();
        return SPECKV_ERR_DRIVER;
    }
    for (auto& ex : a->extents) if (ex.base) pools_[ex.pool]->free(ex.base, ex.bytes);
    a->extents.swap(fresh);
    a->regular = a->packed_regular;
    a->packed = a->packed_regular = false;
    a->packed_off128.clear(); a->packed_off128.shrink_to_fit();
    a->packed_bytes.clear();
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
