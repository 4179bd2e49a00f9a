// cxl-speckv_amd/csrc/engine_io.cpp -- the data path: writes (compress into the pool), reads, bulk fetch + decompress (Engine members)
#include "engine_internal.hpp"
#include "tuning.hpp"

namespace speckv {

// device-visible address of an allocation's record-length samples (null where there are none: other schemes, the fake device)
static uint32_t* len_samples_dev(const Allocation* a)
{
    if (!a->len_samples || a->scheme != SPECKV_COMP_INT8_DELTA_RLE) return nullptr;
    void* dp = nullptr;
    if (hipHostGetDevicePointer(&dp, a->len_samples, 0) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    return static_cast<uint32_t*>(dp);
}
// "this allocation holds data that compresses": the mean of the record lengths the compress kernel left behind (every 1024th
// page) is under 512 B.  Read without any ordering -- samples of writes still in flight may be missing -- because it only picks
// between two decoders that produce the same bytes.
static bool looks_structured(const Allocation* a)
{
    if (!a->len_samples || a->scheme != SPECKV_COMP_INT8_DELTA_RLE) return false;
    uint64_t sum = 0, cnt = 0;
    for (uint32_t i = 0; i < kLenSamples; ++i) {
        const uint32_t v = *const_cast<volatile uint32_t*>(a->len_samples + i);
        if (v) { sum += v; ++cnt; }
    }
    return cnt && sum / cnt < 512u;
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
    c.scale_run = a->scale_run;
    c.len_samples = len_samples_dev(a);
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
    c.scale_run = a->scale_run;
    c.len_samples = len_samples_dev(a);
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
        staged[i] = CompressGroup{a->d_entries, a->d_scale_tab, a->region_pages, a->scale_run, firsts[i],
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
    // data that compresses takes the flat-run decoder here too: a sealed allocation by its packed size, another by its length samples
    int hint = 0;
    if (a->scheme == SPECKV_COMP_INT8_DELTA_RLE && a->n_pages) {
        if (a->packed) { uint64_t pk = 0; for (uint64_t b : a->packed_bytes) pk += b; hint = pk / a->n_pages < 512u ? 1 : 0; }
        else hint = looks_structured(a) ? 1 : 0;
    }
    if (hint) st_.flat_decoder_fetches++;
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
    // records per pool and chunk: the staging buffer is cut into D equal regions.  Tile-planar MXFP4 runs move as whole tiles
    // (the staged copy keeps every record's place inside its tile, so "pool address + delta" still finds nibbles and codes):
    // a chunk's records may begin and end inside a tile -- one tile of the region is kept for that
    const bool planar = planar_mx4(a->scheme);
    const uint64_t region = planar ? (stage_bytes_ / D) / kMx4TileBytes * kMx4TileBytes : (stage_bytes_ / D) / stride * stride;
    const uint64_t recs_per_region = planar ? (region / kMx4TileBytes >= 2 ? (region / kMx4TileBytes - 1) * kMx4TileRecs : 0) : region / stride;
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
        c.structured_hint = hint;                            // (the flat-run decoder also reads staged records: k_fetch_decompress_flat_staged)
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
            } else if (planar) {
                const uint64_t t0 = rb / kMx4TileRecs, t1 = (rb + cnt + kMx4TileRecs - 1) / kMx4TileRecs;
                src = static_cast<const uint8_t*>(a->extents[k].base) + t0 * kMx4TileBytes;
                run_bytes = static_cast<size_t>(t1 - t0) * kMx4TileBytes;
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
    int choice = engine_choice ? engine_choice : tuning().remote_engine;
    const uint32_t D = static_cast<uint32_t>(a->pool_of_residue.size());
    const bool can_copy = (a->regular || (a->packed && a->packed_regular)) && D >= 1 && D <= 8 && a->n_pages < (1ull << 28) && !is_capturing(st);
    if (choice == 0) {
        bool remote = false;
        for (int p : a->pool_of_residue) remote = remote || pools_[p]->device() != device_;
        const uint64_t min_run = static_cast<uint64_t>(tuning().copy_min_run_kb > 0 ? tuning().copy_min_run_kb : 1024) << 10;
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
        } else {                                             // never sealed: the compress kernel's own length samples
            c.structured_hint = looks_structured(a) ? 1 : 0;
        }
        if (c.structured_hint) st_.flat_decoder_fetches++;
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

} // namespace speckv
