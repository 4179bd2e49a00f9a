// cxl-speckv_amd/csrc/engine_relocate.cpp -- records that move: migration between pool GPUs, compaction into packed extents and back (Engine members)
#include "engine_internal.hpp"

namespace speckv {

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
    if (planar_mx4(a->scheme)) return migrate_mx4(a, first, n, target_pool);
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
    // multiples of the pool's 64-byte granule, so a sub-run is freed exactly (never reaching into live neighbours).
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
        const bool fixed_fmt = a->scheme == SPECKV_COMP_FP8_E4M3 || a->scheme == SPECKV_COMP_INT4_G32 || a->scheme == SPECKV_COMP_MXFP4;
        if (fixed_fmt && a->d_stripe) {
            uint64_t bases[8] = {reinterpret_cast<uint64_t>(dst), 0, 0, 0, 0, 0, 0, 0};
            HIP_TRY(hipMemcpy(a->d_stripe, bases, sizeof(bases), hipMemcpyHostToDevice));
            a->linear_base = dst;
            a->stripe_n = 1;
        }
    }
    return SPECKV_OK;
}

// The same for tile-planar MXFP4 records (kernels.hpp: 16 records = 16 nibble rows + 16 code rows, 136 whole lines).  A source
// run = consecutive pages whose records are consecutive slots of one pool's run.  The destination is one new run, records
// dense in page order -- except that a long source run (>= 32 records) starts at its source's slot phase (up to 15 slots of
// padding), so that all its full tiles move as ONE hipMemcpyPeerAsync; everything else moves in pieces cut at the tile
// boundaries of either side (a piece = its nibble rows + its code rows: two copies).  Old tiles go back to their pool when
// their last record has left (Allocation::mx4_vacated).
int Engine::migrate_mx4(Allocation* a, uint64_t first, uint64_t n, uint32_t target_pool)
{
    std::vector<PageEntry> cur(n);
    HIP_TRY(hipMemcpy(cur.data(), a->d_entries + first, n * sizeof(PageEntry), hipMemcpyDeviceToHost));
    struct Src { uint64_t i0, cnt, tile; uint32_t j0; int pool; uint64_t d0; };      // records i0 .. i0+cnt of the range: slot j0 of the tile at `tile` on
    auto slot_of = [](const PageEntry& e) { return (kMx4CodePlane - float_bits(e.scale)) / 960u; };
    std::vector<Src> runs;
    for (uint64_t i = 0; i < n; ++i) {
        const uint32_t j = slot_of(cur[i]);
        const uint64_t tile = cur[i].pool_addr - 1024ull * j;
        const int pool = a->page_pool[first + i];
        if (!runs.empty()) {
            const Src& r = runs.back();
            const uint64_t next = r.j0 + r.cnt;
            if (r.pool == pool && tile == r.tile + next / kMx4TileRecs * kMx4TileBytes && j == next % kMx4TileRecs) { runs.back().cnt++; continue; }
        }
        runs.push_back({i, 1, tile, j, pool, 0});
    }
    uint64_t total = 0;
    for (Src& r : runs) {
        if (r.cnt >= 2 * kMx4TileRecs) total += (r.j0 + kMx4TileRecs - total % kMx4TileRecs) % kMx4TileRecs;      // the source's phase
        r.d0 = total;
        total += r.cnt;
    }
    const size_t dst_bytes = static_cast<size_t>(mx4_run_bytes(total));
    uint8_t* dst = static_cast<uint8_t*>(pools_[target_pool]->alloc(dst_bytes));
    if (!dst) return SPECKV_ERR_NOMEM;
    struct PoolGuard {
        SlabPool* pool; void* p; size_t bytes; bool keep = false;
        ~PoolGuard() { if (!keep) pool->free(p, bytes); }
    } guard{pools_[target_pool].get(), dst, dst_bytes};
    const int dst_dev = pools_[target_pool]->device();
    std::vector<uint16_t> used((total + kMx4TileRecs - 1) / kMx4TileRecs, 0);            // slots of the new run that hold a page
    for (const Src& r : runs) {
        const int src_dev = pools_[r.pool]->device();
        const uint8_t* sb = reinterpret_cast<const uint8_t*>(r.tile);
        uint64_t s = r.j0, t = r.d0, left = r.cnt;
        while (left) {
            if (s % kMx4TileRecs == 0 && t % kMx4TileRecs == 0 && left >= kMx4TileRecs) {
                const uint64_t m = left / kMx4TileRecs;
                HIP_TRY(hipMemcpyPeerAsync(dst + t / kMx4TileRecs * kMx4TileBytes, dst_dev, sb + s / kMx4TileRecs * kMx4TileBytes, src_dev,
                                           static_cast<size_t>(m) * kMx4TileBytes, copy_stream_));
                s += m * kMx4TileRecs; t += m * kMx4TileRecs; left -= m * kMx4TileRecs;
                continue;
            }
            const uint64_t len = std::min<uint64_t>(left, std::min<uint64_t>(kMx4TileRecs - s % kMx4TileRecs, kMx4TileRecs - t % kMx4TileRecs));
            HIP_TRY(hipMemcpyPeerAsync(dst + mx4_nib_off(t), dst_dev, sb + mx4_nib_off(s), src_dev, static_cast<size_t>(len) * 1024u, copy_stream_));
            HIP_TRY(hipMemcpyPeerAsync(dst + mx4_nib_off(t) + mx4_code_delta(t), dst_dev, sb + mx4_nib_off(s) + mx4_code_delta(s), src_dev,
                                       static_cast<size_t>(len) * 64u, copy_stream_));
            s += len; t += len; left -= len;
        }
        for (uint64_t c = 0; c < r.cnt; ++c) {
            used[(r.d0 + c) / kMx4TileRecs] |= static_cast<uint16_t>(1u << ((r.d0 + c) % kMx4TileRecs));
            cur[r.i0 + c] = PageEntry{reinterpret_cast<uint64_t>(dst) + mx4_nib_off(r.d0 + c), cur[r.i0 + c].rec_bytes, bits_as_float(mx4_code_delta(r.d0 + c))};
        }
    }
    HIP_TRY(hipStreamSynchronize(copy_stream_));
    HIP_TRY(hipMemcpy(a->d_entries + first, cur.data(), n * sizeof(PageEntry), hipMemcpyHostToDevice));
    guard.keep = true;
    // the slots the records left: a tile whose 16 slots are all vacated goes back to its pool, adjacent ones as one range
    std::vector<std::pair<uint64_t, int>> gone;
    for (const Src& r : runs)
        for (uint64_t c = 0; c < r.cnt; ++c) {
            const uint64_t tile = r.tile + (r.j0 + c) / kMx4TileRecs * kMx4TileBytes;
            uint16_t& m = a->mx4_vacated[tile];
            m |= static_cast<uint16_t>(1u << ((r.j0 + c) % kMx4TileRecs));
            if (m == 0xFFFFu) { gone.push_back({tile, r.pool}); a->mx4_vacated.erase(tile); }
        }
    std::sort(gone.begin(), gone.end());
    for (size_t g = 0; g < gone.size();) {
        size_t h = g + 1;
        while (h < gone.size() && gone[h].second == gone[g].second && gone[h].first == gone[h - 1].first + kMx4TileBytes) ++h;
        const uint64_t addr = gone[g].first, bytes = (h - g) * static_cast<uint64_t>(kMx4TileBytes);
        const int pool = gone[g].second;
        // (adjacent tiles may still belong to two extents of a fragmented allocation: every extent gives up its share)
        std::vector<Allocation::Extent> next;
        for (const auto& ex : a->extents) {
            const uint64_t lo = reinterpret_cast<uint64_t>(ex.base), hi = lo + ex.bytes;
            if (!ex.base || ex.pool != pool || addr >= hi || addr + bytes <= lo) { next.push_back(ex); continue; }
            const uint64_t f0 = std::max(lo, addr), f1 = std::min(hi, addr + bytes);
            if (f0 > lo) next.push_back({ex.pool, ex.base, static_cast<size_t>(f0 - lo), (f0 - lo) / kMx4TileBytes * kMx4TileRecs});
            if (f1 < hi) next.push_back({ex.pool, reinterpret_cast<void*>(f1), static_cast<size_t>(hi - f1), (hi - f1) / kMx4TileBytes * kMx4TileRecs});
            pools_[pool]->free(reinterpret_cast<void*>(f0), static_cast<size_t>(f1 - f0));
        }
        a->extents.swap(next);
        g = h;
    }
    a->extents.push_back({static_cast<int>(target_pool), dst, dst_bytes, total});
    for (size_t t = 0; t < used.size(); ++t)
        if (used[t] != 0xFFFFu) a->mx4_vacated[reinterpret_cast<uint64_t>(dst) + t * kMx4TileBytes] = static_cast<uint16_t>(~used[t]);
    for (uint64_t i = 0; i < n; ++i) a->page_pool[first + i] = static_cast<uint8_t>(target_pool);
    a->linear_base = nullptr;
    a->regular = false;
    a->stripe_n = 0;
    st_.pool_migrated_pages += n;
    if (first == 0 && n == a->n_pages && total == n) {
        // the WHOLE allocation moved and landed dense (page p = record p of the new run): regular "over one pool" again
        a->pool_of_residue.assign(1, static_cast<int>(target_pool));
        a->regular = true;
        if (a->d_stripe) {
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


} // namespace speckv
