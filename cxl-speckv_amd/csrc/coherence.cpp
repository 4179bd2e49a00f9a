// cxl-speckv_amd/csrc/coherence.cpp -- coherence shadow directory behind the reference's
// coherence_manager_* C ABI (include/speckv_coherence.h; reference: coherence_c_api.cpp:33-209 over
// CoherenceManager, coherence_manager.cpp).  Host bookkeeping only -- the reference's device operations are
// stubs that always succeed (coherence_manager.cpp:398-434) and no call moves data.
#pragma GCC visibility push(default)
#include "../../include/speckv_coherence.h"
#pragma GCC visibility pop
#include <hip/hip_runtime.h>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

namespace {

enum State : uint8_t { kInvalid = 0, kShared = 1, kExclusive = 2, kModified = 3 };
enum Tier : uint8_t { kL1 = 0, kL2 = 1, kL3 = 2 };
enum Op { kRead, kWrite, kInvalidate, kWriteback };

struct Line {
    uint64_t addr;
    uint32_t access_count;
    uint8_t live, state, tier;
};

class Directory {
public:
    explicit Directory(size_t line_size) : mask_(~(static_cast<uint64_t>(line_size) - 1u)), slots_(1024) {}

    bool read(uint64_t addr)
    {   // coherence_manager.cpp:33-68
        std::lock_guard<std::mutex> g(mu_);
        const uint64_t a = addr & mask_;
        Line* e = find(a);
        if (e && e->state != kInvalid) { count(kRead, true); e->access_count++; return true; }
        count(kRead, false);
        device_op(kRead);
        e = obtain(a);
        e->state = kShared;
        e->tier = kL1;
        e->access_count = 1;
        return true;
    }
    bool write(uint64_t addr)
    {   // :70-106
        std::lock_guard<std::mutex> g(mu_);
        const uint64_t a = addr & mask_;
        Line* e = find(a);
        if (e && e->state == kShared) { count(kInvalidate, false); st_.invalidations_sent++; }
        count(kWrite, e != nullptr);
        device_op(kWrite);
        e = obtain(a);
        e->state = kModified;
        e->tier = kL1;
        e->access_count++;
        return true;
    }
    bool invalidate(uint64_t addr)
    {   // :108-134
        std::lock_guard<std::mutex> g(mu_);
        Line* e = find(addr & mask_);
        if (!e) return true;
        if (e->state == kModified) st_.writebacks_performed++;
        e->state = kInvalid;
        device_op(kInvalidate);
        st_.invalidations_sent++;
        return true;
    }
    bool writeback(uint64_t addr)
    {   // :136-158
        std::lock_guard<std::mutex> g(mu_);
        Line* e = find(addr & mask_);
        if (!e || e->state != kModified) return true;
        device_op(kWriteback);
        e->state = kShared;
        e->tier = kL3;
        st_.writebacks_performed++;
        return true;
    }
    bool flush()
    {   // :160-181
        std::lock_guard<std::mutex> g(mu_);
        for (Line& l : slots_)
            if (l.live && l.state == kModified) {
                device_op(kWriteback);
                l.state = kShared;
                l.tier = kL3;
                st_.writebacks_performed++;
            }
        return true;
    }
    int state(uint64_t addr)
    {
        std::lock_guard<std::mutex> g(mu_);
        const Line* e = find(addr & mask_);
        return e ? e->state : kInvalid;
    }
    int tier(uint64_t addr)
    {
        std::lock_guard<std::mutex> g(mu_);
        const Line* e = find(addr & mask_);
        return e ? e->tier : kL3;
    }
    bool promote(uint64_t addr)
    {   // :211-238 (an unknown line enters the directory as INVALID)
        std::lock_guard<std::mutex> g(mu_);
        Line* e = obtain(addr & mask_);
        if (e->tier == kL1) return true;
        device_op(kRead);
        e->tier = kL1;
        return true;
    }
    bool demote(uint64_t addr)
    {   // :240-261
        std::lock_guard<std::mutex> g(mu_);
        Line* e = find(addr & mask_);
        if (!e || e->tier == kL3) return true;
        if (e->state == kModified) { device_op(kWriteback); e->state = kShared; st_.writebacks_performed++; }
        e->tier = kL3;
        return true;
    }
    void set_tier(uint64_t addr, int t)
    {   // :263-270
        std::lock_guard<std::mutex> g(mu_);
        obtain(addr & mask_)->tier = static_cast<uint8_t>(t);
    }
    bool batch_invalidate(const uint64_t* addrs, size_t n)
    {   // :272-289: only lines in the directory get an operation; every address is counted
        std::lock_guard<std::mutex> g(mu_);
        for (size_t i = 0; i < n; ++i)
            if (Line* e = find(addrs[i] & mask_)) { e->state = kInvalid; device_op(kInvalidate); }
        st_.invalidations_sent += n;
        return true;
    }
    coherence_statistics_t stats() { std::lock_guard<std::mutex> g(mu_); return st_; }
    void reset() { std::lock_guard<std::mutex> g(mu_); std::memset(&st_, 0, sizeof(st_)); }
    size_t size() { std::lock_guard<std::mutex> g(mu_); return live_; }

private:
    // the reference's send_coherence_op_to_fpga: always completes, and is booked as a hit of its kind (:420,436-458)
    void device_op(Op op) { count(op, true); }
    void count(Op op, bool hit)
    {
        if (op == kRead) { st_.total_reads++; (hit ? st_.directory_hits : st_.directory_misses)++; }
        else if (op == kWrite) { st_.total_writes++; (hit ? st_.directory_hits : st_.directory_misses)++; }
        else st_.coherence_ops++;
    }
    size_t home(uint64_t a) const { return static_cast<size_t>((a * 0x9E3779B97F4A7C15ull) >> 17) & (slots_.size() - 1); }
    Line* find(uint64_t a)
    {
        for (size_t i = home(a);; i = (i + 1) & (slots_.size() - 1)) {
            if (!slots_[i].live) return nullptr;
            if (slots_[i].addr == a) return &slots_[i];
        }
    }
    Line* obtain(uint64_t a)
    {
        if (Line* e = find(a)) return e;
        if ((live_ + 1) * 2 > slots_.size()) {
            std::vector<Line> old(slots_.size() * 2);
            old.swap(slots_);
            for (const Line& l : old)
                if (l.live) *place(l.addr) = l;
        }
        Line* e = place(a);
        *e = Line{a, 0u, 1, kInvalid, kL3};
        ++live_;
        return e;
    }
    Line* place(uint64_t a)
    {
        size_t i = home(a);
        while (slots_[i].live) i = (i + 1) & (slots_.size() - 1);
        return &slots_[i];
    }

    const uint64_t mask_;               // addr & ~(line_size - 1), as the reference computes it for any line_size
    std::vector<Line> slots_;
    size_t live_ = 0;
    coherence_statistics_t st_{};
    std::mutex mu_;
};

Directory* dir(coherence_manager_handle_t h) { return static_cast<Directory*>(h); }

} // namespace

extern "C" {

coherence_manager_handle_t coherence_manager_create(const char* device_path, size_t cache_line_size)
{
    if (!device_path) return nullptr;
    try {
        if (std::string(device_path) != "/dev/null") {
            int n = 0;
            if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) { (void)hipGetLastError(); return nullptr; }
        }
        return new Directory(cache_line_size);
    } catch (...) {
        return nullptr;
    }
}
void coherence_manager_destroy(coherence_manager_handle_t h)
{
    if (!h) return;
    dir(h)->flush();
    delete dir(h);
}
bool coherence_manager_request_read(coherence_manager_handle_t h, uint64_t addr, void* data_out, size_t)
{
    if (!h || !data_out) return false;
    return dir(h)->read(addr);
}
bool coherence_manager_request_write(coherence_manager_handle_t h, uint64_t addr, const void* data, size_t)
{
    if (!h || !data) return false;
    return dir(h)->write(addr);
}
bool coherence_manager_invalidate(coherence_manager_handle_t h, uint64_t addr) { return h ? dir(h)->invalidate(addr) : false; }
bool coherence_manager_writeback(coherence_manager_handle_t h, uint64_t addr, const void* data, size_t)
{
    if (!h || !data) return false;
    return dir(h)->writeback(addr);
}
bool coherence_manager_flush_all(coherence_manager_handle_t h) { return h ? dir(h)->flush() : false; }
int coherence_manager_get_state(coherence_manager_handle_t h, uint64_t addr) { return h ? dir(h)->state(addr) : 0; }
int coherence_manager_get_tier(coherence_manager_handle_t h, uint64_t addr) { return h ? dir(h)->tier(addr) : 2; }
bool coherence_manager_promote_to_l1(coherence_manager_handle_t h, uint64_t addr) { return h ? dir(h)->promote(addr) : false; }
bool coherence_manager_demote_to_l3(coherence_manager_handle_t h, uint64_t addr) { return h ? dir(h)->demote(addr) : false; }
bool coherence_manager_batch_invalidate(coherence_manager_handle_t h, const uint64_t* addrs, size_t count)
{
    if (!h || !addrs) return false;
    return dir(h)->batch_invalidate(addrs, count);
}
void coherence_manager_get_statistics(coherence_manager_handle_t h, coherence_statistics_t* out)
{
    if (!h || !out) return;
    *out = dir(h)->stats();
}
void coherence_manager_reset_statistics(coherence_manager_handle_t h) { if (h) dir(h)->reset(); }
void coherence_manager_ext_update_tier(coherence_manager_handle_t h, uint64_t addr, int tier) { if (h) dir(h)->set_tier(addr, tier); }
size_t coherence_manager_ext_entry_count(coherence_manager_handle_t h) { return h ? dir(h)->size() : 0; }

} // extern "C"
