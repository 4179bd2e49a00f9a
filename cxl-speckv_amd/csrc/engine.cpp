// cxl-speckv_amd/csrc/engine.cpp -- see engine.hpp
#include "engine.hpp"

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
    std::vector<int> devs;
    if (const char* env = getenv("SPECKV_POOL_DEVICES")) {
        std::string s(env);
        size_t i = 0;
        while (i < s.size()) {
            size_t j = s.find(',', i);
            if (j == std::string::npos) j = s.size();
            if (j > i) devs.push_back(atoi(s.substr(i, j - i).c_str()));
            i = j + 1;
        }
    }
    if (devs.empty()) devs.push_back(device_);
    int count = 0;
    HIP_TRY(hipGetDeviceCount(&count));
    const size_t slab = env_mb("SPECKV_SLAB_MB", 1024) << 20;
    const size_t cap = env_mb("SPECKV_POOL_CAP_MB", 0) << 20;
    for (int d : devs) {
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

    // cache arena on the compute GPU (reference defaults 12 GB L1 / 3 GB L2,
    // cxl_memory_manager.h:42-44; ours are env-tunable and allocated up front)
    const size_t l2_mb = env_mb("SPECKV_L2_MB", 256), l1_mb = env_mb("SPECKV_L1_MB", 256);
    n_l2_ = static_cast<uint32_t>((l2_mb << 20) / kPageSize);
    n_l1_ = static_cast<uint32_t>((l1_mb << 20) / kPageSize);
    if (n_l2_ < 64) n_l2_ = 64;
    if (n_l1_ < 16) n_l1_ = 16;
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&cache_base_), static_cast<size_t>(n_l2_ + n_l1_) * kPageSize));
    owner_.assign(n_l2_ + n_l1_, Owner{nullptr, 0});
    lru_prev_.assign(n_l2_ + n_l1_, UINT32_MAX);
    lru_next_.assign(n_l2_ + n_l1_, UINT32_MAX);
    l1_free_.reserve(n_l1_);
    for (uint32_t i = 0; i < n_l1_; ++i) l1_free_.push_back(n_l2_ + n_l1_ - 1 - i);
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
    if (stream_) (void)hipStreamSynchronize(stream_);
    for (auto& b : inflight_) (void)hipEventDestroy(b.ev);
    for (auto ev : event_pool_) (void)hipEventDestroy(ev);
    for (auto& kv : allocs_) release_allocation(kv.second.get());
    allocs_.clear();
    if (d_emb_) (void)hipFree(d_emb_);
    if (d_wout_) (void)hipFree(d_wout_);
    for (Scratch* s : {&s_pages_, &s_dst_, &s_req_, &s_out_, &s_tmp_, &s_stage_, &s_hid_, &s_logits_, &s_hist_, &s_pred_, &s_attn_, &s_attn_seq_})
        if (s->p) (void)hipFree(s->p);
    if (d_count_) (void)hipFree(d_count_);
    if (d_zero_page_) (void)hipFree(d_zero_page_);
    if (seq_ring_.base) (void)hipHostFree(seq_ring_.base);
    for (auto ev : seq_ring_.ev) if (ev) (void)hipEventDestroy(ev);
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

void* Engine::scratch(Scratch& s, size_t bytes)
{
    if (bytes <= s.cap) return s.p;
    if (s.p) { (void)hipStreamSynchronize(stream_); (void)hipFree(s.p); s.p = nullptr; s.cap = 0; }
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

// ------------------------------------------------------------ alloc/free
int Engine::alloc(size_t bytes, const speckv_alloc_hint_t* hint, uint64_t* out)
{
    // speckv_allocator.cpp:11-38 : the handle is consumed even for 0 bytes
    std::unique_ptr<Allocation> a(new Allocation());
    a->size_bytes = bytes;
    a->n_pages = (bytes + kPageSize - 1) / kPageSize;
    a->scheme = scheme_;
    a->rec_stride = stride_for(scheme_);
    a->flags.assign(a->n_pages, 0u);
    if (!null_) {
        a->slot.assign(a->n_pages, 0u);
        a->access_count.assign(a->n_pages, 0u);
        if (a->n_pages) {
            DeviceScope device_scope(device_);
            // placement: preferred_node picks one pool GPU (1-based; 0 = stripe over all)
            std::vector<int> use;
            if (hint && hint->preferred_node >= 1 && hint->preferred_node <= pools_.size())
                use.push_back(static_cast<int>(hint->preferred_node) - 1);
            else
                for (size_t i = 0; i < pools_.size(); ++i) use.push_back(static_cast<int>(i));
            const uint64_t D = use.size();
            bool ok = true;
            bool single_run = (D == 1);
            std::vector<PageEntry> host;
            for (uint64_t k = 0; k < D && ok; ++k) {
                const uint64_t np = (a->n_pages + D - 1 - k) / D;     // pages with page % D == k
                if (np == 0) continue;
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
                single_run = false;
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
            if (ok) ok = hipMalloc(reinterpret_cast<void**>(&a->d_entries), a->n_pages * sizeof(PageEntry)) == hipSuccess;
            if (ok) ok = hipMalloc(reinterpret_cast<void**>(&a->d_flags), a->n_pages * sizeof(uint32_t)) == hipSuccess;
            if (ok) ok = hipMemsetAsync(a->d_flags, 0, a->n_pages * sizeof(uint32_t), stream_) == hipSuccess;
            if (ok) {
                if (single_run) {
                    ok = launch_init_entries(a->d_entries, a->n_pages, reinterpret_cast<uint64_t>(a->extents[0].base),
                                             a->rec_stride, stream_) == hipSuccess;
                    // FP8 / INT4 pools start as zero bytes (= records of zeros), which lets the fused attention address
                    // them arithmetically without a validity test per page
                    if (ok && (a->scheme == SPECKV_COMP_FP8_E4M3 || a->scheme == SPECKV_COMP_INT4_G32) &&
                        pools_[a->extents[0].pool]->device() == device_) {
                        ok = hipMemsetAsync(a->extents[0].base, 0, a->extents[0].bytes, stream_) == hipSuccess;
                        if (ok) a->linear_base = static_cast<uint8_t*>(a->extents[0].base);
                    }
                }
                else {
                    ok = hipMemcpy(a->d_entries, host.data(), host.size() * sizeof(PageEntry), hipMemcpyHostToDevice) == hipSuccess;
                }
            }
            a->pool_of_residue.assign(D, 0);
            for (uint64_t k = 0; k < D; ++k) a->pool_of_residue[k] = use[k];
            a->page_pool.resize(a->n_pages);
            for (uint64_t i = 0; i < a->n_pages; ++i) a->page_pool[i] = static_cast<uint8_t>(use[i % D]);
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
    *out = a->handle;
    st_.total_allocations++;
    st_.current_allocated_bytes += bytes;
    st_.peak_allocated_bytes = std::max(st_.peak_allocated_bytes, st_.current_allocated_bytes);
    allocs_[a->handle] = std::move(a);
    return SPECKV_OK;
}

void Engine::release_allocation(Allocation* a)
{
    if (null_) return;
    for (uint64_t p = 0; p < a->n_pages && !a->slot.empty(); ++p)
        if (a->flags[p] & 3u) {
            const uint32_t s = a->slot[p];
            if (s >= n_l2_) { lru_unlink(s); l1_free_.push_back(s); }
            owner_[s] = Owner{nullptr, 0};
        }
    pending_clear_.erase(a);
    for (auto& ex : a->extents)
        if (ex.base) pools_[ex.pool]->free(ex.base, ex.bytes);
    a->extents.clear();
    if (a->d_entries) (void)hipFree(a->d_entries);
    if (a->d_flags) (void)hipFree(a->d_flags);
    if (a->d_scale_tab) { (void)hipFree(a->d_scale_tab); a->d_scale_tab = nullptr; }
    a->d_entries = nullptr;
    a->d_flags = nullptr;
}

int Engine::free(uint64_t handle)
{   // speckv_allocator.cpp:40-52 : unknown handle is a silent no-op
    auto it = allocs_.find(handle);
    if (it == allocs_.end()) return SPECKV_OK;
    if (!null_) {
        DeviceScope device_scope(device_);
        reap(true);
        // asynchronous entry points (fetch_range / fetch_list / attend_* on a caller stream) may still be reading this
        // allocation's records: wait for the whole device, as hipFree would, before the pool memory is recycled
        (void)hipDeviceSynchronize();
        release_allocation(it->second.get());
    }
    st_.total_deallocations++;
    st_.current_allocated_bytes -= it->second->size_bytes;
    if (layout_handle_ == handle) layout_handle_ = 0;
    allocs_.erase(it);
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

void Engine::drop_slot(uint32_t s)
{
    Owner& o = owner_[s];
    if (!o.a) return;
    o.a->flags[o.page] &= ~3u;
    pending_clear_[o.a].push_back(o.page);
    o = Owner{nullptr, 0};
}

uint32_t Engine::take_l2_run(uint32_t n)
{
    // FIFO ring; a run never wraps so multi-page spans stay contiguous
    uint32_t start = static_cast<uint32_t>(l2_hand_ % n_l2_);
    if (start + n > n_l2_) start = 0;
    for (uint32_t i = 0; i < n; ++i) drop_slot(start + i);
    l2_hand_ = start + n;
    return start;
}

uint32_t Engine::take_l1_slot()
{
    if (!l1_free_.empty()) { uint32_t s = l1_free_.back(); l1_free_.pop_back(); return s; }
    // evict_l1_lru -> demote_to_l3 (cxl_memory_manager.cpp:285-293)
    const uint32_t victim = lru_head_;
    lru_unlink(victim);
    drop_slot(victim);
    st_.migrations_l1_to_l3++;
    return victim;
}

void Engine::move_to_l1(Allocation* a, uint32_t page)
{   // promote_to_l1 (cxl_memory_manager.cpp:130-163) for a page that sits in the L2 ring
    const uint32_t from = a->slot[page];
    const uint32_t to = take_l1_slot();
    (void)hipMemcpyAsync(slot_ptr(to), slot_ptr(from), kPageSize, hipMemcpyDeviceToDevice, stream_);
    owner_[from] = Owner{nullptr, 0};
    owner_[to] = Owner{a, page};
    a->slot[page] = to;
    a->flags[page] = (a->flags[page] & ~2u) | 1u;
    lru_push_mru(to);
}

void Engine::flush_mirror()
{
    for (auto& kv : pending_clear_) {
        auto& v = kv.second;
        if (v.empty()) continue;
        void* d = scratch(s_tmp_, v.size() * sizeof(uint32_t));
        if (!d) continue;
        (void)hipMemcpyAsync(d, v.data(), v.size() * sizeof(uint32_t), hipMemcpyHostToDevice, stream_);
        (void)launch_update_flags(kv.first->d_flags, static_cast<const uint32_t*>(d),
                                  static_cast<uint32_t>(v.size()), ~3u, 0u, stream_);
        (void)hipStreamSynchronize(stream_);     // scratch is reused by the next group
        v.clear();
    }
    pending_clear_.clear();
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
}

int Engine::fetch_into_slots(Allocation* a, const std::vector<uint32_t>& pages,
                             const std::vector<uint32_t>& slots, bool wait)
{
    const uint32_t n = static_cast<uint32_t>(pages.size());
    if (n == 0) return SPECKV_OK;
    flush_mirror();
    bool run = true;                     // consecutive pages into consecutive slots: no descriptor upload
    for (uint32_t i = 1; i < n && run; ++i) run = pages[i] == pages[0] + i && slots[i] == slots[0] + i;
    CodecArgs c{};
    c.entries = a->d_entries;
    c.trusted = 1;                       // pool records only ever come from k_compress
    if (run) {
        c.first = pages[0];
        c.data = slot_ptr(slots[0]);
        c.data_stride = kPageSize;
    } else {
        std::vector<uint64_t> dst(n);
        for (uint32_t i = 0; i < n; ++i) dst[i] = reinterpret_cast<uint64_t>(slot_ptr(slots[i]));
        uint32_t* d_pages = static_cast<uint32_t*>(scratch(s_pages_, n * sizeof(uint32_t)));
        uint64_t* d_dst = static_cast<uint64_t*>(scratch(s_dst_, n * sizeof(uint64_t)));
        if (!d_pages || !d_dst) return SPECKV_ERR_NOMEM;
        HIP_TRY(hipMemcpyAsync(d_pages, pages.data(), n * sizeof(uint32_t), hipMemcpyHostToDevice, stream_));
        HIP_TRY(hipMemcpyAsync(d_dst, dst.data(), n * sizeof(uint64_t), hipMemcpyHostToDevice, stream_));
        c.page_list = d_pages;
        c.data_list = d_dst;
    }
    c.n = n;
    c.flags = a->d_flags;
    c.set_flags = 2u;
    c.scheme = a->scheme;
    c.quant_mode = quant_mode_;
    HIP_TRY(launch_decompress(c, stream_));
    st_.dma_submitted += n;
    st_.total_decompressions += n;
    if (wait) {                          // sync_fetch_page: submit, then spin on completion
        HIP_TRY(hipStreamSynchronize(stream_));
        reap(true);
        completed_unpolled_ += n;
        st_.dma_completed += n;
        return SPECKV_OK;
    }
    hipEvent_t ev = get_event();
    if (ev) { HIP_TRY(hipEventRecord(ev, stream_)); inflight_.push_back({ev, n}); }
    else { HIP_TRY(hipStreamSynchronize(stream_)); completed_unpolled_ += n; st_.dma_completed += n; }
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
    uint64_t p1 = len ? (off + len - 1) / kPageSize : p0;
    if (p1 >= a->n_pages) p1 = a->n_pages - 1;
    if (p1 - p0 + 1 > n_l2_) return SPECKV_ERR_NOMEM;
    DeviceScope device_scope(device_);
    int rc = SPECKV_OK;
    // a multi-page span must come back contiguous
    bool contiguous = true;
    for (uint64_t p = p0; p <= p1; ++p)
        if (!(a->flags[p] & 3u) || a->slot[p] != a->slot[p0] + (p - p0)) { contiguous = false; break; }
    std::vector<uint32_t> miss;
    for (uint64_t p = p0; p <= p1; ++p) {
        a->access_count[p]++;                                // update_access_tracking, cxl_memory_manager.cpp:223-245
        const uint32_t f = a->flags[p];
        if (f & 1u) { st_.l1_hits++; if (contiguous) { lru_unlink(a->slot[p]); lru_push_mru(a->slot[p]); } }
        else if (f & 2u) st_.l2_hits++;
        else { st_.l3_accesses++; st_.l2_misses++; }
        if (!contiguous || !(f & 3u)) miss.push_back(static_cast<uint32_t>(p));
    }
    if (!miss.empty()) {
        if (p1 > p0) {                                       // refetch the whole span into one run
            for (uint32_t p : miss)
                if (a->flags[p] & 3u) {
                    const uint32_t s = a->slot[p];
                    if (s >= n_l2_) { lru_unlink(s); l1_free_.push_back(s); }
                    drop_slot(s);
                }
        }
        const uint32_t run = take_l2_run(static_cast<uint32_t>(miss.size()));
        std::vector<uint32_t> slots(miss.size());
        for (size_t i = 0; i < miss.size(); ++i) slots[i] = run + static_cast<uint32_t>(i);
        rc = fetch_into_slots(a, miss, slots, true);         // sync_fetch_page: submit + spin on completion
        if (rc == SPECKV_OK)
            for (size_t i = 0; i < miss.size(); ++i) {
                a->slot[miss[i]] = slots[i];
                a->flags[miss[i]] |= 2u;                     // speckv_allocator.cpp:135
                owner_[slots[i]] = Owner{a, miss[i]};
            }
    } else if (p1 == p0 && (a->flags[p0] & 3u) == 2u && a->access_count[p0] > 10) {
        // L2 hit on a hot page -> promote (memory_allocator.cpp:127-134, is_hot_page: count > 10)
        move_to_l1(a, static_cast<uint32_t>(p0));
    }
    if (rc == SPECKV_OK && !inflight_.empty()) {             // a prefetched page may still be landing
        HIP_TRY(hipStreamSynchronize(stream_));
        reap(true);
    }
    if (rc == SPECKV_OK) *out = slot_ptr(a->slot[p0]) + poff;
    return rc;
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
    ++epoch_;
    if (a->stamp.size() != a->n_pages) a->stamp.assign(a->n_pages, 0u);
    std::vector<uint32_t> miss;
    for (uint32_t i = 0; i < n; ++i) {
        const uint32_t p = static_cast<uint32_t>(offs[i] / kPageSize);
        a->access_count[p]++;
        const uint32_t f = a->flags[p];
        if (f & 1u) st_.l1_hits++; else if (f & 2u) st_.l2_hits++; else { st_.l3_accesses++; st_.l2_misses++; }
        if (!(f & 3u) && a->stamp[p] != epoch_) { a->stamp[p] = epoch_; miss.push_back(p); }
    }
    int rc = SPECKV_OK;
    if (miss.size() > n_l2_) rc = SPECKV_ERR_NOMEM;
    if (rc == SPECKV_OK && !miss.empty()) {
        const uint32_t run = take_l2_run(static_cast<uint32_t>(miss.size()));
        std::vector<uint32_t> slots(miss.size());
        for (size_t i = 0; i < miss.size(); ++i) slots[i] = run + static_cast<uint32_t>(i);
        rc = fetch_into_slots(a, miss, slots, true);
        if (rc == SPECKV_OK)
            for (size_t i = 0; i < miss.size(); ++i) {
                a->slot[miss[i]] = slots[i];
                a->flags[miss[i]] |= 2u;
                owner_[slots[i]] = Owner{a, miss[i]};
            }
    }
    if (rc == SPECKV_OK && !inflight_.empty()) { (void)hipStreamSynchronize(stream_); reap(true); }
    if (rc == SPECKV_OK)
        for (uint32_t i = 0; i < n; ++i) {
            const uint64_t p = offs[i] / kPageSize;
            out[i] = (a->flags[p] & 3u) ? slot_ptr(a->slot[p]) + offs[i] % kPageSize : nullptr;
            if (!out[i]) rc = SPECKV_ERR_GENERAL;   // evicted inside this very batch (cache smaller than batch)
        }
    return rc;
}

// --------------------------------------------------------------- prefetch
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
    queue_.push_back({req, layer, pos, k ? k : adapt_.depth()});
    uint32_t thr = flush_threshold_;
    if (thr == 0) {
        Allocation* a = layout_handle_ ? find(layout_handle_) : nullptr;
        thr = a && a->has_layout ? a->layout.num_layers : 32u;
    }
    if (queue_.size() >= thr) { uint32_t n = 0; (void)prefetch_flush(&n); }   // driver result ignored, as in the reference
    return SPECKV_OK;
}

int Engine::prefetch_batch(uint32_t n, const uint32_t* req, const uint16_t* layer,
                           const uint32_t* pos, const uint32_t* k)
{
    if (null_) return SPECKV_OK;
    queue_.reserve(queue_.size() + n);
    for (uint32_t i = 0; i < n; ++i) queue_.push_back({req[i], layer[i], pos[i], (k && k[i]) ? k[i] : adapt_.depth()});
    return SPECKV_OK;
}

int Engine::prefetch_flush(uint32_t* n_issued)
{
    if (n_issued) *n_issued = 0;
    if (null_) return SPECKV_OK;
    if (queue_.empty()) return SPECKV_OK;
    Allocation* a = layout_handle_ ? find(layout_handle_) : nullptr;
    if (!a || !a->has_layout || a->n_pages == 0) {
        // no geometry known: nothing can be addressed (the reference would have sent the
        // request to the FPGA, whose ATU maps (req,layer,pos) itself)
        queue_.clear();
        return SPECKV_OK;
    }
    DeviceScope device_scope(device_);
    static const bool timing = getenv("SPECKV_TIMING") != nullptr;
    auto tnow = [] { return std::chrono::steady_clock::now(); };
    auto t_a = tnow();
    const uint32_t n = static_cast<uint32_t>(queue_.size());
    std::vector<uint32_t> soa(4ull * n);
    for (uint32_t i = 0; i < n; ++i) {
        soa[i] = queue_[i].req; soa[n + i] = queue_[i].layer;
        soa[2ull * n + i] = queue_[i].pos; soa[3ull * n + i] = queue_[i].k;
    }
    queue_.clear();
    const uint64_t row = static_cast<uint64_t>(a->layout.num_heads) * a->layout.head_dim * a->layout.bytes_per_element;
    const uint32_t cap = static_cast<uint32_t>(std::min<uint64_t>(n * 32ull * (row / kPageSize + 2), 1ull << 30));
    uint32_t* d_req = static_cast<uint32_t*>(scratch(s_req_, soa.size() * sizeof(uint32_t)));
    uint32_t* d_out = static_cast<uint32_t*>(scratch(s_out_, (static_cast<size_t>(cap) + 2ull * n + 4) * sizeof(uint32_t)));
    if (!d_req || !d_out) return SPECKV_ERR_NOMEM;
    uint32_t* d_scr = d_out + cap;
    flush_mirror();
    HIP_TRY(hipMemcpyAsync(d_req, soa.data(), soa.size() * sizeof(uint32_t), hipMemcpyHostToDevice, stream_));
    HIP_TRY(launch_prefetch_lookup(a->layout, n, d_req, d_req + n, d_req + 2ull * n, d_req + 3ull * n,
                                   a->d_flags, d_out, cap, d_count_, d_scr, stream_));
    uint32_t count = 0;
    HIP_TRY(hipMemcpyAsync(&count, d_count_, sizeof(uint32_t), hipMemcpyDeviceToHost, stream_));
    HIP_TRY(hipStreamSynchronize(stream_));
    auto t_b = tnow();
    std::vector<uint32_t> cand(count);
    if (count) HIP_TRY(hipMemcpy(cand.data(), d_out, count * sizeof(uint32_t), hipMemcpyDeviceToHost));
    auto t_c = tnow();
    // host side: dedupe across requests, assign ring slots, submit one fetch batch
    ++epoch_;
    if (a->stamp.size() != a->n_pages) a->stamp.assign(a->n_pages, 0u);
    std::vector<uint32_t> pages;
    pages.reserve(count);
    for (uint32_t p : cand)
        if (p < a->n_pages && !(a->flags[p] & 3u) && a->stamp[p] != epoch_) { a->stamp[p] = epoch_; pages.push_back(p); }
    if (pages.size() > n_l2_ / 2) pages.resize(n_l2_ / 2);   // never let one flush wipe the whole ring
    auto t_d = tnow(), t_e = t_d, t_f = t_d;
    int rc = SPECKV_OK;
    if (!pages.empty()) {
        const uint32_t m = static_cast<uint32_t>(pages.size());
        const uint32_t run = take_l2_run(m);
        std::vector<uint32_t> slots(m);
        for (uint32_t i = 0; i < m; ++i) slots[i] = run + i;
        t_e = tnow();
        rc = fetch_into_slots(a, pages, slots, false);       // asynchronous: overlaps the caller's compute
        t_f = tnow();
        if (rc == SPECKV_OK) {
            for (uint32_t i = 0; i < m; ++i) {
                a->slot[pages[i]] = slots[i];
                a->flags[pages[i]] |= 2u;
                owner_[slots[i]] = Owner{a, pages[i]};
            }
            st_.total_prefetches += m;
            if (n_issued) *n_issued = m;
        }
    }
    if (timing) {
        auto us = [](auto x, auto y) { return std::chrono::duration<double, std::micro>(y - x).count(); };
        fprintf(stderr, "[speckv timing] flush n=%u: lookup+sync %.1f us, list d2h %.1f, dedupe %.1f, ring %.1f, submit %.1f, bookkeeping %.1f\n",
                n, us(t_a, t_b), us(t_b, t_c), us(t_c, t_d), us(t_d, t_e), us(t_e, t_f), us(t_f, tnow()));
    }
    if (rc == SPECKV_OK) rc = run_predictor_for_dirty();
    return rc;
}

// ----------------------------------------------------------------- predictor
int Engine::predictor_load(const float* emb, const float* wout, uint32_t vocab, bool on_device)
{
    if (null_) return no_data_path("speckv_ext_predictor_load");
    if (!emb || !wout || vocab < 8) return SPECKV_ERR_INVAL;
    DeviceScope device_scope(device_);
    HIP_TRY(hipStreamSynchronize(stream_));
    if (d_emb_) { (void)hipFree(d_emb_); d_emb_ = nullptr; }
    if (d_wout_) { (void)hipFree(d_wout_); d_wout_ = nullptr; }
    const size_t eb = static_cast<size_t>(vocab) * 64 * sizeof(float), wb = static_cast<size_t>(vocab) * 128 * sizeof(float);
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&d_emb_), eb));
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&d_wout_), wb));
    const hipMemcpyKind kind = on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
    if (on_device) HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(d_emb_, emb, eb, kind));
    HIP_TRY(hipMemcpy(d_wout_, wout, wb, kind));
    vocab_ = vocab;
    hist_.clear(); pred_.clear(); hist_dirty_.clear();
    return SPECKV_OK;
}

int Engine::predict_batch(uint32_t n, const int32_t* d_hist, uint32_t k, int32_t* d_tok, float* d_conf, hipStream_t s)
{
    if (null_) return no_data_path("speckv_ext_predict_batch");
    if (!d_emb_) return SPECKV_ERR_INVAL;
    if (n == 0) return SPECKV_OK;
    if (!d_hist || !d_tok || !d_conf || k == 0 || k > 8) return SPECKV_ERR_INVAL;
    DeviceScope device_scope(device_);
    float* hid = static_cast<float*>(scratch(s_hid_, static_cast<size_t>(n) * 128 * sizeof(float)));
    float* logits = static_cast<float*>(scratch(s_logits_, static_cast<size_t>(n) * vocab_ * sizeof(float)));
    if (!hid || !logits) return SPECKV_ERR_NOMEM;
    hipStream_t st = s ? s : stream_;
    HIP_TRY(launch_predict(n, d_hist, d_emb_, d_wout_, vocab_, 2, k, hid, logits, d_tok, d_conf, st));
    if (!s) HIP_TRY(hipStreamSynchronize(stream_));
    return SPECKV_OK;
}

int Engine::run_predictor_for_dirty()
{
    if (!d_emb_ || hist_dirty_.empty()) { hist_dirty_.clear(); return SPECKV_OK; }
    std::sort(hist_dirty_.begin(), hist_dirty_.end());
    hist_dirty_.erase(std::unique(hist_dirty_.begin(), hist_dirty_.end()), hist_dirty_.end());
    const uint32_t n = static_cast<uint32_t>(hist_dirty_.size());
    uint32_t k = adapt_.depth();
    if (k > 8) k = 8;
    if (k == 0) k = 1;
    std::vector<int32_t> h(static_cast<size_t>(n) * 16);
    for (uint32_t i = 0; i < n; ++i) memcpy(&h[i * 16], hist_[hist_dirty_[i]].data(), 16 * sizeof(int32_t));
    int32_t* d_h = static_cast<int32_t*>(scratch(s_hist_, h.size() * sizeof(int32_t)));
    uint8_t* d_p = static_cast<uint8_t*>(scratch(s_pred_, static_cast<size_t>(n) * k * (sizeof(int32_t) + sizeof(float))));
    if (!d_h || !d_p) return SPECKV_ERR_NOMEM;
    int32_t* d_tok = reinterpret_cast<int32_t*>(d_p);
    float* d_conf = reinterpret_cast<float*>(d_p + static_cast<size_t>(n) * k * sizeof(int32_t));
    HIP_TRY(hipMemcpyAsync(d_h, h.data(), h.size() * sizeof(int32_t), hipMemcpyHostToDevice, stream_));
    int rc = predict_batch(n, d_h, k, d_tok, d_conf, stream_);
    if (rc != SPECKV_OK) return rc;
    std::vector<int32_t> tok(static_cast<size_t>(n) * k);
    HIP_TRY(hipMemcpyAsync(tok.data(), d_tok, tok.size() * sizeof(int32_t), hipMemcpyDeviceToHost, stream_));
    HIP_TRY(hipStreamSynchronize(stream_));
    reap(true);
    for (uint32_t i = 0; i < n; ++i) pred_[hist_dirty_[i]].assign(tok.begin() + static_cast<size_t>(i) * k, tok.begin() + static_cast<size_t>(i + 1) * k);
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
    flush_mirror();
    uint32_t* scr = static_cast<uint32_t*>(scratch(s_tmp_, (2ull * n + 4) * sizeof(uint32_t)));
    if (!scr) return SPECKV_ERR_NOMEM;
    hipStream_t st = s ? s : stream_;
    HIP_TRY(launch_prefetch_lookup(a->layout, n, d_req, d_layer, d_pos, d_k, a->d_flags, d_out, cap, d_count, scr, st));
    if (!s) HIP_TRY(hipStreamSynchronize(stream_));
    return SPECKV_OK;
}

int Engine::verify(uint32_t req, int32_t actual, const int32_t* pred, uint32_t n,
                   uint32_t* was_hit, uint32_t* new_depth)
{
    // no list given: verify against the prediction the engine made from the request's last history
    std::vector<int32_t> own;
    if (!pred || n == 0) {
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
    a->layout = Layout{T, L, H, D, bpe, a->n_pages};
    a->has_layout = true;
    layout_handle_ = handle;
    // fused-attention scale table (FP8 records, 2 positions per page, regions aligned to 16-page tiles)
    if (!null_ && a->scheme == SPECKV_COMP_FP8_E4M3 && a->n_pages && static_cast<uint64_t>(H) * D * bpe == 2048u && T % 32u == 0u) {
        DeviceScope device_scope(device_);
        if (!a->d_scale_tab) HIP_TRY(hipMalloc(reinterpret_cast<void**>(&a->d_scale_tab), a->n_pages * sizeof(float)));
        a->region_pages = T / 2u;
        HIP_TRY(launch_build_scale_tab(a->d_entries, a->n_pages, a->region_pages, a->d_scale_tab, stream_));
        HIP_TRY(hipStreamSynchronize(stream_));
    } else if (a->d_scale_tab) {
        DeviceScope device_scope(device_);
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
    o->flags = a->flags[p];
    o->scheme = static_cast<uint32_t>(a->scheme);
    o->pool_device = -1;
    o->scale = 1.0f;
    if (!null_) {
        DeviceScope device_scope(device_);
        PageEntry e{};
        HIP_TRY(hipMemcpy(&e, a->d_entries + p, sizeof(e), hipMemcpyDeviceToHost));
        o->pool_device = pools_[a->page_pool[p]]->device();
        o->rec_bytes = e.rec_bytes;
        o->scale = e.scale;
        o->pool_addr = e.pool_addr;
        o->cache_addr = (a->flags[p] & 3u) ? reinterpret_cast<uint64_t>(slot_ptr(a->slot[p])) : 0;
        o->access_count = a->access_count[p];
    }
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
    reap(true);
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
            HIP_TRY(hipStreamSynchronize(stream_));
        }
    }
    HIP_TRY(hipStreamSynchronize(stream_));
    const uint64_t np = full + (tail ? 1 : 0);
    for (uint64_t p = p0; p < p0 + np; ++p) {
        if (a->flags[p] & 3u) {                             // a cached copy is stale now
            const uint32_t s = a->slot[p];
            if (s >= n_l2_) { lru_unlink(s); l1_free_.push_back(s); }
            drop_slot(s);
        }
        if (a->scheme != SPECKV_COMP_FP16) a->flags[p] |= 4u; else a->flags[p] &= ~4u;
    }
    st_.total_compressions += np;
    st_.original_bytes += np * kPageSize;
    return SPECKV_OK;
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

int Engine::fetch_range(uint64_t handle, uint64_t first, uint64_t n, void* d_dst, bool f32, hipStream_t s)
{
    if (null_) return no_data_path("speckv_ext_fetch_range");
    Allocation* a = find(handle);
    if (!a) return SPECKV_ERR_GENERAL;
    if (first > a->n_pages || n > a->n_pages - first) return SPECKV_ERR_GENERAL;
    if (!d_dst) return SPECKV_ERR_INVAL;
    if (n == 0) return SPECKV_OK;
    DeviceScope device_scope(device_);
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
    hipStream_t st = s ? s : stream_;
    HIP_TRY(launch_decompress(c, st));
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
            if (!s) HIP_TRY(hipStreamSynchronize(stream_));
            return SPECKV_OK;
        }
    }
    const size_t rows = static_cast<size_t>(n_layers) * L.num_heads * 16;
    uint8_t* q8 = static_cast<uint8_t*>(scratch(s_req_, rows * 128 + rows * sizeof(float)));
    if (!q8) return SPECKV_ERR_NOMEM;
    float* qs = reinterpret_cast<float*>(q8 + rows * 128);
    HIP_TRY(launch_quantize_q_e4m3(d_q_f16, n_layers * L.num_heads, g, L.head_dim, q8, qs, st));
    HIP_TRY(launch_qk_scores_fp8(a->d_entries, first_page, layer_stride, n_layers, n_pages, L.num_heads, g, q8, qs, d_out, st));
    if (!s) HIP_TRY(hipStreamSynchronize(stream_));
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
    const uint32_t n_pages = (pos_end - pos_begin) / 2;
    DeviceScope device_scope(device_);
    hipStream_t st = s ? s : stream_;
    const size_t out_elems = static_cast<size_t>(n_layers) * L.num_heads * g * 128;
    if (n_pages == 0) {          // empty range: softmax over nothing -> zeros (and -inf lse is left to the caller)
        HIP_TRY(hipMemsetAsync(d_out, 0, out_elems * sizeof(float), st));
        if (!s) HIP_TRY(hipStreamSynchronize(stream_));
        return SPECKV_OK;
    }
    // shim layout [req 0][layer][kind][pos][head]: K pages of a layer, then its V pages
    const uint64_t k_first = (static_cast<uint64_t>(layer) * 2 * L.num_tokens + pos_begin) / 2;
    const uint64_t v_first = k_first + L.num_tokens / 2;
    const uint64_t layer_stride = static_cast<uint64_t>(L.num_tokens);
    if (v_first + (n_layers - 1) * layer_stride + n_pages > a->n_pages) return SPECKV_ERR_GENERAL;
    if (!d_zero_page_) {
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&d_zero_page_), kPageSize));
        HIP_TRY(hipMemset(d_zero_page_, 0, kPageSize));
    }
    // splits: ~40 waves per CU over the launch (a few rounds of the 16 resident ones; measured best at
    // 70B@32k: 16 splits/row), at least 8 tiles (256 positions) per split so the partials stay small
    const uint32_t n_tiles = (n_pages + 15u) / 16u;
    const uint32_t rows = n_layers * L.num_heads;
    uint32_t want = (10240u + rows - 1u) / rows;
    // per-layer calls are latency-bound: short contexts want short splits (measured best: 2 tiles per split at 2k
    // context, 4 at 8k, 8 at 32k), long multi-layer launches are bounded by `want` above
    const uint32_t min_tiles = std::min(8u, std::max(2u, n_tiles / 64u));
    want = std::min(want, std::max(1u, n_tiles / min_tiles));
    if (const char* env = getenv("SPECKV_ATTEND_SPLITS")) want = static_cast<uint32_t>(atoi(env));
    uint32_t n_splits = std::max(1u, std::min(std::min(want, n_tiles), 2048u));
    const uint32_t tiles_per_split = (n_tiles + n_splits - 1u) / n_splits;
    n_splits = (n_tiles + tiles_per_split - 1u) / tiles_per_split;
    const size_t q_bytes = static_cast<size_t>(rows) * 16 * 128, qs_bytes = static_cast<size_t>(rows) * 16 * sizeof(float);
    const size_t acc_bytes = static_cast<size_t>(rows) * n_splits * 16 * 128 * sizeof(float);
    const size_t ml_bytes = static_cast<size_t>(rows) * n_splits * 32 * sizeof(float);
    uint8_t* buf = static_cast<uint8_t*>(scratch(s_attn_, q_bytes + qs_bytes + acc_bytes + ml_bytes));
    if (!buf) return SPECKV_ERR_NOMEM;
    AttendArgs k{};
    k.entries = a->d_entries;
    k.k_first = k_first;
    k.v_first = v_first;
    k.layer_stride = layer_stride;
    k.n_pages = n_pages;
    k.heads = L.num_heads;
    k.g = g;
    k.n_splits = n_splits;
    k.tiles_per_split = tiles_per_split;
    k.q8 = buf;
    k.qs = reinterpret_cast<float*>(buf + q_bytes);
    k.scale_log2e = sm_scale * 1.4426950408889634f;
    k.zero_page = d_zero_page_;
    // linear form: records in one run, scale table present, tiles aligned with the table's (pos_begin a multiple of 32),
    // and the last (possibly ragged) 32-position tile must not read past the K / V region of its layer
    const bool fits = pos_begin % 32u == 0u && a->d_scale_tab &&
                      static_cast<uint64_t>(pos_begin) + static_cast<uint64_t>(n_tiles) * 32u <= L.num_tokens;
    k.scale_tab = a->d_scale_tab;
    k.q16 = static_cast<const uint16_t*>(d_q_f16);
    k.lin_base = (getenv("SPECKV_ATTEND_GENERAL") || !fits) ? nullptr : a->linear_base;
    k.part_acc = reinterpret_cast<float*>(buf + q_bytes + qs_bytes);
    k.part_ml = reinterpret_cast<float*>(buf + q_bytes + qs_bytes + acc_bytes);
    if (!k.lin_base)                     // the linear form quantises the query in its own prologue
        HIP_TRY(launch_quantize_q_e4m3(d_q_f16, rows, g, L.head_dim, buf, reinterpret_cast<float*>(buf + q_bytes), st));
    HIP_TRY(launch_attend_fp8(k, n_layers, d_out, d_lse, st));
    if (!s) HIP_TRY(hipStreamSynchronize(stream_));
    return SPECKV_OK;
}

// One decode step of a batch: the fused FP8 attention of ONE layer for many sequences (allocations) in one launch
// (BASELINE configs[3] shape: 256 sequences).  Every allocation must qualify for the linear form.
int Engine::attend_fp8_batch(uint32_t n_seq, const uint64_t* handles, uint32_t layer, const void* d_q_f16, uint32_t g,
                             const uint32_t* pos_end, float sm_scale, float* d_out, float* d_lse, hipStream_t s)
{
    return attend_batch(SPECKV_COMP_FP8_E4M3, n_seq, handles, layer, d_q_f16, g, pos_end, sm_scale, d_out, d_lse, s);
}

int Engine::attend_batch(int scheme, uint32_t n_seq, const uint64_t* handles, uint32_t layer, const void* d_q_f16, uint32_t g,
                         const uint32_t* pos_end, float sm_scale, float* d_out, float* d_lse, hipStream_t s)
{
    const bool fp8 = scheme == SPECKV_COMP_FP8_E4M3;
    if (null_) return no_data_path("speckv_ext_attend_*_batch");
    if (n_seq == 0) return SPECKV_OK;
    if (!handles || !pos_end || !d_q_f16 || !d_out || g == 0 || g > 16) return SPECKV_ERR_INVAL;
    std::vector<AttendSeq> seqs(n_seq);
    uint64_t total_tiles = 0;
    uint32_t heads = 0;
    for (uint32_t i = 0; i < n_seq; ++i) {
        Allocation* a = find(handles[i]);
        if (!a) return SPECKV_ERR_GENERAL;
        if (!a->has_layout || a->scheme != scheme) return SPECKV_ERR_INVAL;
        const Layout& L = a->layout;
        if (L.head_dim != 128 || L.bytes_per_element != 2 || L.num_heads * L.head_dim != 1024 || L.num_tokens % 2) return SPECKV_ERR_INVAL;
        if (layer >= L.num_layers || pos_end[i] % 2 || pos_end[i] > L.num_tokens) return SPECKV_ERR_INVAL;
        const uint32_t n_pages = pos_end[i] / 2, n_tiles = (n_pages + 15u) / 16u;
        if (!a->linear_base || (fp8 && !a->d_scale_tab) || static_cast<uint64_t>(n_tiles) * 32u > L.num_tokens) {
            SPECKV_ERR("speckv_ext_attend_*_batch: sequence %u does not qualify for the linear form "
                       "(records in one local run, pos_end rounded up to 32 inside the layer%s)", i,
                       fp8 ? ", layout with num_tokens %% 32 == 0" : "");
            return SPECKV_ERR_INVAL;
        }
        heads = L.num_heads;
        seqs[i].lin_base = a->linear_base;
        seqs[i].scale_tab = a->d_scale_tab;
        seqs[i].k_first = static_cast<uint64_t>(layer) * L.num_tokens;       // (layer*2*T)/2
        seqs[i].v_first = seqs[i].k_first + L.num_tokens / 2;
        seqs[i].n_pages = n_pages;
        seqs[i].n_splits = n_tiles;                                           // tiles for now, splits below
        total_tiles += n_tiles;
    }
    DeviceScope device_scope(device_);
    hipStream_t st = s ? s : stream_;
    const size_t out_elems = static_cast<size_t>(n_seq) * heads * g * 128;
    if (total_tiles == 0) {
        HIP_TRY(hipMemsetAsync(d_out, 0, out_elems * sizeof(float), st));
        if (!s) HIP_TRY(hipStreamSynchronize(stream_));
        return SPECKV_OK;
    }
    // one split length for the whole batch: ~10k waves over the launch, at least 8 tiles per split
    uint32_t tps = static_cast<uint32_t>(std::max<uint64_t>(8, (total_tiles * heads + 10239u) / 10240u));
    if (const char* env = getenv("SPECKV_ATTEND_TILES_PER_SPLIT")) tps = std::max(1, atoi(env));
    uint32_t max_splits = 0;
    uint64_t parts = 0;
    for (uint32_t i = 0; i < n_seq; ++i) {
        const uint32_t n_tiles = seqs[i].n_splits;
        seqs[i].n_splits = (n_tiles + tps - 1u) / tps;
        if (seqs[i].n_splits > 2048u) return SPECKV_ERR_INVAL;
        seqs[i].part_base = static_cast<uint32_t>(parts);
        parts += static_cast<uint64_t>(heads) * seqs[i].n_splits;
        max_splits = std::max(max_splits, seqs[i].n_splits);
    }
    const size_t acc_bytes = static_cast<size_t>(parts) * 16 * 128 * sizeof(float), ml_bytes = static_cast<size_t>(parts) * 32 * sizeof(float);
    uint8_t* buf = static_cast<uint8_t*>(scratch(s_attn_, acc_bytes + ml_bytes));
    AttendSeq* d_seqs = static_cast<AttendSeq*>(scratch(s_attn_seq_, seqs.size() * sizeof(AttendSeq)));
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
    k.lin_base = seqs[0].lin_base;           // (overridden per sequence)
    k.seqs = d_seqs;
    k.part_acc = reinterpret_cast<float*>(buf);
    k.part_ml = reinterpret_cast<float*>(buf + acc_bytes);
    if (fp8) {
        HIP_TRY(launch_attend_fp8_batch(k, n_seq, d_out, d_lse, st));
    } else {
        HIP_TRY(launch_attend_int4(k, n_seq, st));        // grid y = sequences x head groups, as for layers
        HIP_TRY(launch_attend_combine(k, n_seq, d_out, d_lse, st));
    }
    if (!s) HIP_TRY(hipStreamSynchronize(stream_));
    return SPECKV_OK;
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
    const bool linear = a->linear_base && fits && !getenv("SPECKV_ATTEND_GENERAL");
    if (!linear && !d_zero_page_) {
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&d_zero_page_), kPageSize));
        HIP_TRY(hipMemset(d_zero_page_, 0, kPageSize));
    }
    const uint32_t rows = n_layers * L.num_heads;
    uint32_t want = (5120u + rows - 1u) / rows;      // VALU-bound kernel: fewer, longer splits measured best
    const uint32_t min_tiles = std::min(8u, std::max(2u, n_tiles / 64u));      // per-layer calls: see attend_fp8
    want = std::min(want, std::max(1u, n_tiles / min_tiles));
    if (const char* env = getenv("SPECKV_ATTEND_SPLITS")) want = static_cast<uint32_t>(atoi(env));
    uint32_t n_splits = std::max(1u, std::min(std::min(want, n_tiles), 2048u));
    const uint32_t tiles_per_split = (n_tiles + n_splits - 1u) / n_splits;
    n_splits = (n_tiles + tiles_per_split - 1u) / tiles_per_split;
    const size_t acc_bytes = static_cast<size_t>(rows) * n_splits * 16 * 128 * sizeof(float);
    const size_t ml_bytes = static_cast<size_t>(rows) * n_splits * 32 * sizeof(float);
    uint8_t* buf = static_cast<uint8_t*>(scratch(s_attn_, acc_bytes + ml_bytes));
    if (!buf) return SPECKV_ERR_NOMEM;
    AttendArgs k{};
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
    k.zero_page = d_zero_page_;
    k.part_acc = reinterpret_cast<float*>(buf);
    k.part_ml = reinterpret_cast<float*>(buf + acc_bytes);
    HIP_TRY(launch_attend_int4(k, n_layers, st));
    HIP_TRY(launch_attend_combine(k, n_layers, d_out, d_lse, st));
    if (!s) HIP_TRY(hipStreamSynchronize(stream_));
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
    HIP_TRY(hipStreamSynchronize(stream_));
    reap(true);
    if (!copy_stream_) HIP_TRY(hipStreamCreateWithFlags(&copy_stream_, hipStreamNonBlocking));
    const size_t stride = a->rec_stride;
    uint8_t* dst = static_cast<uint8_t*>(pools_[target_pool]->alloc(n * stride));
    if (!dst) return SPECKV_ERR_NOMEM;
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
    // bookkeeping: the old runs leave the allocation's extent list (split where needed)
    for (const Run& r : old) {
        std::vector<Allocation::Extent> next;
        for (const auto& ex : a->extents) {
            const uint64_t lo = reinterpret_cast<uint64_t>(ex.base), hi = lo + ex.bytes;
            if (ex.pool != r.pool || r.addr >= hi || r.addr + r.bytes <= lo) { next.push_back(ex); continue; }
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
    st_.pool_migrated_pages += n;
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
    uint32_t n = 0;
    int rc = prefetch_flush(&n);
    HIP_TRY(hipStreamSynchronize(stream_));
    reap(true);
    return rc;
}

int Engine::promote_to_l1(uint64_t handle, uint64_t off)
{
    if (null_) return no_data_path("speckv_ext_promote_to_l1");
    Allocation* a = find(handle);
    if (!a) return SPECKV_ERR_GENERAL;
    const uint64_t p = off / kPageSize;
    if (p >= a->n_pages) return SPECKV_ERR_GENERAL;
    if (a->flags[p] & 1u) return SPECKV_ERR_GENERAL;          // already there -> false (cxl_memory_manager.cpp:134-136)
    DeviceScope device_scope(device_);
    int rc = SPECKV_OK;
    if (a->flags[p] & 2u) {
        move_to_l1(a, static_cast<uint32_t>(p));
    } else {
        const uint32_t s = take_l1_slot();
        rc = fetch_into_slots(a, {static_cast<uint32_t>(p)}, {s}, true);
        if (rc == SPECKV_OK) {
            a->slot[p] = s; a->flags[p] |= 1u; owner_[s] = Owner{a, static_cast<uint32_t>(p)};
            lru_push_mru(s);
            st_.migrations_l3_to_l1++;
        } else {
            l1_free_.push_back(s);
        }
    }
    HIP_TRY(hipStreamSynchronize(stream_));
    return rc;
}

int Engine::demote_to_l3(uint64_t handle, uint64_t off)
{
    if (null_) return no_data_path("speckv_ext_demote_to_l3");
    Allocation* a = find(handle);
    if (!a) return SPECKV_ERR_GENERAL;
    const uint64_t p = off / kPageSize;
    if (p >= a->n_pages) return SPECKV_ERR_GENERAL;
    if (!(a->flags[p] & 3u)) return SPECKV_ERR_GENERAL;       // already in the pool only
    const uint32_t s = a->slot[p];
    if (a->flags[p] & 1u) { lru_unlink(s); l1_free_.push_back(s); st_.migrations_l1_to_l3++; }
    drop_slot(s);
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
        uint64_t comp = 0;
        std::vector<PageEntry> host;
        for (auto& kv : allocs_) {
            Allocation* a = kv.second.get();
            if (!a->n_pages) continue;
            host.resize(a->n_pages);
            if (hipMemcpy(host.data(), a->d_entries, a->n_pages * sizeof(PageEntry), hipMemcpyDeviceToHost) == hipSuccess)
                for (auto& e : host) comp += e.rec_bytes;
        }
        st_.compressed_bytes = comp;
    }
    *out = st_;
    return SPECKV_OK;
}

} // namespace speckv
