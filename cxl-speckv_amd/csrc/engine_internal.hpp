// cxl-speckv_amd/csrc/engine_internal.hpp -- what the engine_*.cpp translation units share besides the class itself:
// error / logging macros, the device scope, small host helpers.  (engine.cpp was one file of 3 200 lines until round 4; it is
// now cut along its seams: engine.cpp pool, tiers, ring, access | engine_flush.cpp look-ahead, flush, predictor |
// engine_io.cpp write / read / bulk fetch | engine_attend.cpp fused-attention planning | engine_relocate.cpp migration,
// compaction.  No behaviour changed in the cut.)
#pragma once
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
    case SPECKV_COMP_MXFP4: return kMx4RecBytes;          // 1088 B per record (3.76:1), but NOT a stride: the pool holds them tile-planar (kernels.hpp, planar_mx4)
    default: return kPageSize;
    }
}

// MXFP4 runs are tile-planar (kernels.hpp: 16 records = 16 nibble rows + 16 code rows = 136 lines): bytes of a run of `recs`
// records, and the table entry of record r of the run at `base`
bool planar_mx4(int scheme) { return scheme == SPECKV_COMP_MXFP4; }
float bits_as_float(uint32_t u) { float f; memcpy(&f, &u, sizeof(f)); return f; }
uint32_t float_bits(float f) { uint32_t u; memcpy(&u, &f, sizeof(u)); return u; }
size_t run_bytes_for(int scheme, uint32_t stride, uint64_t recs)
{
    return planar_mx4(scheme) ? static_cast<size_t>(mx4_run_bytes(recs)) : static_cast<size_t>(recs) * stride;
}
PageEntry entry_at(int scheme, uint32_t stride, uint64_t base, uint64_t r)
{
    if (planar_mx4(scheme)) return PageEntry{base + mx4_nib_off(r), 0u, bits_as_float(mx4_code_delta(r))};
    return PageEntry{base + r * stride, 0u, 1.0f};
}
// granule in which a fragmented pool serves such runs, and the records a piece of `bytes` holds
size_t run_granule_for(int scheme, uint32_t stride) { return planar_mx4(scheme) ? kMx4TileBytes : stride; }
uint64_t run_records_for(int scheme, uint32_t stride, size_t bytes) { return planar_mx4(scheme) ? bytes / kMx4TileBytes * kMx4TileRecs : bytes / stride; }

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
} // namespace speckv
