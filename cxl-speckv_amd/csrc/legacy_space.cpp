// cxl-speckv_amd/csrc/legacy_space.cpp -- the reference's legacy 3-tier address space as pure host arithmetic
// behind speckv_ext_mm_* (SURVEY 8a rows A10-A14).
//
// What it reproduces, call for call (reference src/cxl_memory/cxl_memory_manager.cpp):
//   allocate/deallocate     :28-104   bump addresses: virt from 0x100000000, phys L1 0x8000000000,
//                                      L2 0x10000000000, L3 0x20000000000; L1 requests that do not fit fall to L3;
//                                      deallocate(base) forgets the FIRST page only (Appendix B item 10, replicated)
//   translate / is_in_cache :106-128
//   promote / demote / LRU  :130-194,285-322   (tier tags only -- the engine's slot cache moves the data)
//   states, access tracking, hot (> 10 touches), statistics :196-277
//   cxl_access policy       src/integration/memory_allocator.cpp:105-143
// Own structure: virtual addresses are handed out by one bump pointer, so the page table is a flat vector indexed by
// (va - base) / page_size; the tier lists of the reference are kept as multisets of the raw addresses they were
// given (their SIZE is what can_fit_in_tier reads), the LRU order as a linked list with an index.
// The HIP engine does not use this object: its own page table / slot cache implements the same policy on real
// memory (engine.cpp).  This is the logical-id side of "bit-exact indexing" for the legacy manager.
#pragma GCC visibility push(default)
#include "../../include/speckv_ext.h"
#pragma GCC visibility pop

#include <list>
#include <new>
#include <unordered_map>
#include <vector>

namespace {

constexpr uint64_t kVirtBase = 0x100000000ULL;
constexpr uint64_t kPhysBase[3] = {0x8000000000ULL, 0x10000000000ULL, 0x20000000000ULL};

struct TierList {                        // a std::vector<uint64_t> used only through push_back / remove-all / size()
    std::unordered_map<uint64_t, uint32_t> copies;
    uint64_t total = 0;
    void push(uint64_t va) { ++copies[va]; ++total; }
    void remove_all(uint64_t va)
    {
        auto it = copies.find(va);
        if (it == copies.end()) return;
        total -= it->second;
        copies.erase(it);
    }
};

struct Page {
    uint64_t phys = 0;
    uint32_t access_count = 0;
    uint8_t tier = 2, state = 0, present = 0, hot = 0;
};

struct Space {
    uint64_t page_size = 4096;
    uint64_t cap[3] = {0, 0, 0};
    uint64_t next_virt = kVirtBase;
    uint64_t next_phys[3] = {kPhysBase[0], kPhysBase[1], kPhysBase[2]};
    std::vector<Page> pages;
    TierList tier[3];
    std::list<uint64_t> lru;             // front = least recently used (raw addresses)
    std::unordered_map<uint64_t, std::list<uint64_t>::iterator> lru_at;
    speckv_ext_mm_stats_t st{};

    Page* page_of(uint64_t va)
    {
        if (va < kVirtBase) return nullptr;
        const uint64_t i = (va / page_size * page_size - kVirtBase) / page_size;
        if ((va / page_size * page_size - kVirtBase) % page_size) return nullptr;   // page size not dividing the base
        if (i >= pages.size() || !pages[i].present) return nullptr;
        return &pages[i];
    }
    bool fits(int t, uint64_t bytes) const { return tier[t].total * page_size + bytes <= cap[t]; }
    void lru_drop(uint64_t va)
    {
        auto it = lru_at.find(va);
        if (it == lru_at.end()) return;
        lru.erase(it->second);
        lru_at.erase(it);
    }
    void lru_touch(uint64_t va) { lru_drop(va); lru.push_back(va); lru_at[va] = std::prev(lru.end()); }

    bool demote(uint64_t va)
    {
        Page* p = page_of(va);
        if (!p || p->tier == 2) return false;
        const int old = p->tier;
        p->tier = 2;
        tier[old].remove_all(va);
        if (old == 0) { lru_drop(va); st.migrations_l1_to_l3++; }
        tier[2].push(va);
        return true;
    }
    bool promote(uint64_t va)
    {
        Page* p = page_of(va);
        if (!p || p->tier == 0) return false;
        if (!fits(0, page_size) && !lru.empty()) {       // evict_l1_lru
            const uint64_t victim = lru.front();
            lru_drop(victim);
            demote(victim);
        }
        const int old = p->tier;
        p->tier = 0;
        tier[old].remove_all(va);
        if (old == 2) st.migrations_l3_to_l1++;
        tier[0].push(va);
        lru_touch(va);
        return true;
    }
    void touch(uint64_t va)
    {
        Page* p = page_of(va);
        if (!p) return;
        p->access_count++;
        if (p->tier == 0) st.l1_hits++; else if (p->tier == 1) st.l2_hits++; else st.l3_accesses++;
        lru_touch(va);
    }
};

Space* S(speckv_ext_mm_t* m) { return reinterpret_cast<Space*>(m); }

} // namespace

extern "C" {

speckv_ext_mm_t* speckv_ext_mm_new(uint64_t l1_gb, uint64_t l2_gb, uint64_t l3_gb, uint64_t page_size)
{
    if (page_size == 0) return nullptr;
    Space* s = new (std::nothrow) Space();
    if (!s) return nullptr;
    s->page_size = page_size;
    s->cap[0] = l1_gb << 30; s->cap[1] = l2_gb << 30; s->cap[2] = l3_gb << 30;
    return reinterpret_cast<speckv_ext_mm_t*>(s);
}

void speckv_ext_mm_delete(speckv_ext_mm_t* m) { delete S(m); }

uint64_t speckv_ext_mm_allocate(speckv_ext_mm_t* m, uint64_t size_bytes, uint32_t layer_id, int preferred_tier)
{
    (void)layer_id;                                      // recorded by the reference, never read
    Space* s = S(m);
    if (!s || preferred_tier < 0 || preferred_tier > 2) return 0;
    try {
        const uint64_t n = (size_bytes + s->page_size - 1) / s->page_size, bytes = n * s->page_size;
        int t = preferred_tier;
        if (t == 0 && !s->fits(0, bytes)) t = 2;
        const uint64_t va = s->next_virt, pa = s->next_phys[t];
        s->next_phys[t] += bytes;
        s->tier[t].push(va);
        const uint64_t first = (va - kVirtBase) / s->page_size;
        s->pages.resize(first + n);
        for (uint64_t i = 0; i < n; ++i) {
            Page& p = s->pages[first + i];
            p.phys = pa + i * s->page_size;
            p.tier = static_cast<uint8_t>(t);
            p.state = 2;                                 // EXCLUSIVE
            p.present = 1;
        }
        s->next_virt += bytes;
        return va;
    } catch (...) {
        return 0;
    }
}

void speckv_ext_mm_deallocate(speckv_ext_mm_t* m, uint64_t va)
{
    Space* s = S(m);
    if (!s || va < kVirtBase || (va - kVirtBase) % s->page_size) return;   // exact page address only (map lookup by key)
    Page* p = s->page_of(va);
    if (!p) return;
    try {
        s->tier[p->tier].remove_all(va);
        if (p->tier == 0) s->lru_drop(va);
        p->present = 0;
    } catch (...) {}
}

uint64_t speckv_ext_mm_translate(speckv_ext_mm_t* m, uint64_t va)
{
    Space* s = S(m);
    Page* p = s ? s->page_of(va) : nullptr;
    return p ? p->phys + va % s->page_size : 0;
}

int speckv_ext_mm_is_in_cache(speckv_ext_mm_t* m, uint64_t va, int tier)
{
    Space* s = S(m);
    Page* p = s ? s->page_of(va) : nullptr;
    return p && p->tier == tier ? 1 : 0;
}

int speckv_ext_mm_promote_to_l1(speckv_ext_mm_t* m, uint64_t va)
{
    try { return S(m) && S(m)->promote(va) ? 1 : 0; } catch (...) { return 0; }
}

int speckv_ext_mm_demote_to_l3(speckv_ext_mm_t* m, uint64_t va)
{
    try { return S(m) && S(m)->demote(va) ? 1 : 0; } catch (...) { return 0; }
}

void speckv_ext_mm_invalidate_page(speckv_ext_mm_t* m, uint64_t va)
{
    Page* p = S(m) ? S(m)->page_of(va) : nullptr;
    if (p) p->state = 0;
}

void speckv_ext_mm_mark_modified(speckv_ext_mm_t* m, uint64_t va)
{
    Page* p = S(m) ? S(m)->page_of(va) : nullptr;
    if (p) p->state = 3;
}

int speckv_ext_mm_get_page_state(speckv_ext_mm_t* m, uint64_t va)
{
    Page* p = S(m) ? S(m)->page_of(va) : nullptr;
    return p ? p->state : 0;
}

void speckv_ext_mm_update_access_tracking(speckv_ext_mm_t* m, uint64_t va)
{
    try { if (S(m)) S(m)->touch(va); } catch (...) {}
}

int speckv_ext_mm_is_hot_page(speckv_ext_mm_t* m, uint64_t va)
{
    Page* p = S(m) ? S(m)->page_of(va) : nullptr;
    if (!p) return 0;
    p->hot = p->access_count > 10;
    return p->hot;
}

void speckv_ext_mm_get_statistics(speckv_ext_mm_t* m, speckv_ext_mm_stats_t* out)
{
    if (!S(m) || !out) return;
    *out = S(m)->st;
    const uint64_t t1 = out->l1_hits + out->l1_misses, t2 = out->l2_hits + out->l2_misses;
    out->l1_hit_rate = t1 ? static_cast<double>(out->l1_hits) / static_cast<double>(t1) : 0.0;
    out->l2_hit_rate = t2 ? static_cast<double>(out->l2_hits) / static_cast<double>(t2) : 0.0;
}

uint64_t speckv_ext_mm_cxl_access(speckv_ext_mm_t* m, uint64_t base_va, uint64_t offset)
{   // memory_allocator.cpp:105-143
    Space* s = S(m);
    if (!s) return 0;
    const uint64_t va = base_va + offset;
    try {
        s->touch(va);
        Page* p = s->page_of(va);
        if (p && p->tier == 0) return va;
        if (p && p->tier == 1) {
            if (speckv_ext_mm_is_hot_page(m, va)) s->promote(va);
            return va;
        }
        s->promote(va);
    } catch (...) {}
    return va;
}

} // extern "C"
