// cxl-speckv_amd/csrc/c_api.cpp -- extern "C" surface of libcxlspeckv.so.
//
// Mirrors the shape of the reference's host/src/speckv_c_api.cpp:8-121: one
// process-global engine, one mutex, every entry point validates "initialised"
// first and no C++ exception crosses the ABI.  Status codes per call are the
// reference's (SURVEY.md sect. 8b "Error conventions").
// the public headers are the export list: everything declared in them gets default visibility, the rest of the
// library is built with -fvisibility=hidden
#pragma GCC visibility push(default)
#include "tuning.hpp"
#include "../../include/speckv.h"
#include "../../include/speckv_ext.h"
#pragma GCC visibility pop
#include "engine.hpp"
#include "placement.hpp"

#include <cstring>
#include <cstdio>
#include <memory>
#include <mutex>

using speckv::Engine;

namespace {
std::unique_ptr<Engine> g_engine;
std::mutex g_mutex;
// speckv_finalize in progress: it waits until no thread is parked inside the engine with the lock released (a thread
// that waits for the GPU lets go of the mutex, Engine::wait_event) and keeps new entries out meanwhile -- they see the
// library as not initialised, which is what it is about to be.
bool g_closing = false;

template <typename F>
speckv_status_t guarded(F&& f)
{
    try {
        return static_cast<speckv_status_t>(f());
    } catch (const std::bad_alloc&) {
        return SPECKV_ERR_NOMEM;
    } catch (...) {
        return SPECKV_ERR_GENERAL;
    }
}
// every entry hands its lock to the engine, which releases it only while it waits for the GPU (engine.cpp: wait_event)
#define LOCK std::unique_lock<std::mutex> lock(g_mutex); if (g_engine && !g_closing) g_engine->enter(&lock)
#define NEED_INIT if (!g_engine || g_closing) return SPECKV_ERR_INVAL
} // namespace

extern "C" {

speckv_status_t speckv_init(const char* dev_path)
{
    std::unique_lock<std::mutex> lock(g_mutex);
    if (g_engine) return SPECKV_ERR_GENERAL;          // already initialised (speckv_c_api.cpp:16-18), also while a finalize is still draining
    return guarded([&] {
        int status = SPECKV_ERR_GENERAL;
        g_engine = Engine::open(dev_path, &status);
        return status;
    });
}

void speckv_finalize(void)
{
    std::unique_lock<std::mutex> lock(g_mutex);
    if (!g_engine || g_closing) return;                // not initialised, or another thread is already finalizing
    g_closing = true;
    g_engine->wait_idle(lock);                         // releases the mutex while it waits
    try { g_engine.reset(); } catch (...) {}
    g_closing = false;
}

speckv_status_t speckv_alloc(size_t bytes, const speckv_alloc_hint_t* hint, speckv_handle_t* out)
{
    LOCK;
    if (!g_engine || g_closing || !out) return SPECKV_ERR_INVAL;
    return guarded([&] { uint64_t h = 0; int rc = g_engine->alloc(bytes, hint, &h); if (rc == 0) *out = h; return rc; });
}

speckv_status_t speckv_free(speckv_handle_t handle)
{
    LOCK; NEED_INIT;
    return guarded([&] { return g_engine->free(handle); });
}

speckv_status_t speckv_access(speckv_handle_t handle, uint64_t offset_bytes, size_t length_bytes,
                              void** out_gpu_ptr)
{
    LOCK;
    if (!g_engine || g_closing || !out_gpu_ptr) return SPECKV_ERR_INVAL;
    return guarded([&] { return g_engine->access(handle, offset_bytes, length_bytes, out_gpu_ptr); });
}

speckv_status_t speckv_prefetch(uint32_t req_id, uint16_t layer, uint32_t cur_pos, uint32_t depth_k,
                                const int32_t* recent_tokens, uint32_t history_len)
{
    LOCK;
    if (!g_engine || g_closing || !recent_tokens || history_len == 0) return SPECKV_ERR_INVAL;
    return guarded([&] { return g_engine->prefetch(req_id, layer, cur_pos, depth_k, recent_tokens, history_len); });
}

speckv_status_t speckv_set_prefetch_depth(uint32_t depth_k)
{
    LOCK; NEED_INIT;
    return guarded([&] { return g_engine->set_prefetch_depth(depth_k); });
}

speckv_status_t speckv_set_compression_scheme(speckv_comp_scheme_t scheme)
{
    LOCK; NEED_INIT;
    return guarded([&] { return g_engine->set_scheme(static_cast<int>(scheme)); });
}

// ------------------------------------------------------------------ ext
speckv_status_t speckv_ext_set_quant_mode(speckv_quant_mode_t mode)
{
    LOCK; NEED_INIT;
    return guarded([&] { return g_engine->set_quant_mode(static_cast<int>(mode)); });
}

speckv_status_t speckv_ext_translate(speckv_handle_t handle, uint64_t offset_bytes, speckv_ext_page_info_t* out)
{
    LOCK;
    if (!g_engine || g_closing || !out) return SPECKV_ERR_INVAL;
    return guarded([&] { return g_engine->translate(handle, offset_bytes, out); });
}

speckv_status_t speckv_ext_fetch_desc(speckv_handle_t handle, uint64_t offset_bytes, speckv_dma_desc_t* out)
{
    LOCK;
    if (!g_engine || g_closing || !out) return SPECKV_ERR_INVAL;
    return guarded([&] { return g_engine->fetch_desc(handle, offset_bytes, out); });
}

speckv_status_t speckv_ext_set_layout(speckv_handle_t handle, uint32_t num_tokens, uint32_t num_layers,
                                      uint32_t num_heads, uint32_t head_dim, uint32_t bytes_per_element)
{
    LOCK; NEED_INIT;
    return guarded([&] { return g_engine->set_layout(handle, num_tokens, num_layers, num_heads, head_dim, bytes_per_element); });
}

speckv_status_t speckv_ext_write(speckv_handle_t handle, uint64_t offset_bytes, const void* src, size_t len, int src_on_device)
{
    LOCK; NEED_INIT;
    return guarded([&] { return g_engine->write(handle, offset_bytes, src, len, src_on_device != 0); });
}

speckv_status_t speckv_ext_write_strided(speckv_handle_t handle, uint64_t first_page, uint64_t page_step, uint64_t n_pages,
                                         const void* d_src, void* stream)
{
    LOCK; NEED_INIT;
    return guarded([&] { return g_engine->write_strided(handle, first_page, page_step, n_pages, d_src, static_cast<hipStream_t>(stream)); });
}

speckv_status_t speckv_ext_write_async(speckv_handle_t handle, uint64_t offset_bytes, const void* d_src, size_t len, void* stream)
{
    LOCK; NEED_INIT;
    return guarded([&] { return g_engine->write_async(handle, offset_bytes, d_src, len, static_cast<hipStream_t>(stream)); });
}

speckv_status_t speckv_ext_write_strided_batch(const speckv_handle_t* handles, const uint64_t* first_pages, const void* const* d_srcs,
                                               uint32_t n_allocations, uint64_t page_step, uint64_t n_pages_each, void* stream)
{
    LOCK; NEED_INIT;
    return guarded([&] { return g_engine->write_strided_batch(handles, first_pages, d_srcs, n_allocations, page_step, n_pages_each,
                                                              static_cast<hipStream_t>(stream)); });
}

speckv_status_t speckv_ext_write_runs(speckv_handle_t handle, const uint64_t* first_pages, const void* const* d_srcs, uint32_t n_runs,
                                      uint64_t n_pages_each, void* stream)
{
    LOCK; NEED_INIT;
    return guarded([&] { return g_engine->write_runs(handle, first_pages, d_srcs, n_runs, n_pages_each, static_cast<hipStream_t>(stream)); });
}

speckv_status_t speckv_ext_read(speckv_handle_t handle, uint64_t offset_bytes, void* dst, size_t len, int dst_on_device)
{
    LOCK; NEED_INIT;
    return guarded([&] { return g_engine->read(handle, offset_bytes, dst, len, dst_on_device != 0); });
}

speckv_status_t speckv_ext_fetch_range(speckv_handle_t handle, uint64_t first_page, uint64_t n_pages,
                                       void* d_dst, int out_f32, void* stream)
{
    LOCK; NEED_INIT;
    return guarded([&] { return g_engine->fetch_range(handle, first_page, n_pages, d_dst, out_f32 != 0, static_cast<hipStream_t>(stream), 0); });
}

speckv_status_t speckv_ext_fetch_range_engine(speckv_handle_t handle, uint64_t first_page, uint64_t n_pages,
                                              void* d_dst, int out_f32, void* stream, int engine)
{
    LOCK; NEED_INIT;
    return guarded([&] { return g_engine->fetch_range(handle, first_page, n_pages, d_dst, out_f32 != 0, static_cast<hipStream_t>(stream), engine); });
}

speckv_status_t speckv_ext_bind_request(uint32_t req_id, speckv_handle_t handle, uint32_t local_req)
{
    LOCK; NEED_INIT;
    return guarded([&] { return g_engine->bind_request(req_id, handle, local_req); });
}

speckv_status_t speckv_ext_fetch_list(speckv_handle_t handle, const uint32_t* d_pages, uint32_t n,
                                      void* d_dst, int out_f32, void* stream)
{
    LOCK; NEED_INIT;
    return guarded([&] { return g_engine->fetch_list(handle, d_pages, n, d_dst, out_f32 != 0, static_cast<hipStream_t>(stream)); });
}

speckv_status_t speckv_ext_access_batch(speckv_handle_t handle, const uint64_t* offsets, uint32_t n, void** out_ptrs)
{
    LOCK;
    if (!g_engine || g_closing || (n && (!offsets || !out_ptrs))) return SPECKV_ERR_INVAL;
    return guarded([&] { return g_engine->access_batch(handle, offsets, n, out_ptrs); });
}

speckv_status_t speckv_ext_prefetch_batch(uint32_t n, const uint32_t* req_ids, const uint16_t* layers,
                                          const uint32_t* cur_pos, const uint32_t* depth_k)
{
    LOCK;
    if (!g_engine || g_closing || (n && (!req_ids || !layers || !cur_pos))) return SPECKV_ERR_INVAL;
    return guarded([&] { return g_engine->prefetch_batch(n, req_ids, layers, cur_pos, depth_k); });
}

speckv_status_t speckv_ext_prefetch_flush(uint32_t* n_issued)
{
    LOCK; NEED_INIT;
    return guarded([&] { return g_engine->prefetch_flush(n_issued); });
}

speckv_status_t speckv_ext_prefetch_lookup(speckv_handle_t handle, uint32_t n, const uint32_t* d_req_ids,
                                           const uint32_t* d_layers, const uint32_t* d_cur_pos,
                                           const uint32_t* d_depth_k, uint32_t* d_out_pages, uint32_t cap,
                                           uint32_t* d_out_count, void* stream)
{
    LOCK;
    if (!g_engine || g_closing || !d_out_count || (n && (!d_req_ids || !d_layers || !d_cur_pos || !d_depth_k || !d_out_pages)))
        return SPECKV_ERR_INVAL;
    return guarded([&] {
        return g_engine->prefetch_lookup(handle, n, d_req_ids, d_layers, d_cur_pos, d_depth_k, d_out_pages, cap,
                                         d_out_count, static_cast<hipStream_t>(stream));
    });
}

speckv_status_t speckv_ext_prefetch_legacy_addrs(uint32_t layer, uint32_t depth_k, uint64_t* out_addrs, uint32_t* out_n)
{   // speculative_prefetcher.cpp:48,153-160 : req_id is the constant 0, position i+1
    if (!out_addrs || !out_n) return SPECKV_ERR_INVAL;
    for (uint32_t i = 0; i < depth_k; ++i)
        out_addrs[i] = (0ULL << 32) | (static_cast<uint64_t>(layer) << 16) | static_cast<uint64_t>(i + 1);
    *out_n = depth_k;
    return SPECKV_OK;
}

speckv_status_t speckv_ext_verify(uint32_t req_id, int32_t actual_token, const int32_t* predicted,
                                  uint32_t n_predicted, uint32_t* was_hit, uint32_t* new_depth)
{
    LOCK; NEED_INIT;
    return guarded([&] { return g_engine->verify(req_id, actual_token, predicted, n_predicted, was_hit, new_depth); });
}

speckv_status_t speckv_ext_verify_batch(uint32_t n, uint32_t k, const int32_t* d_actual, const int32_t* d_predicted,
                                        uint8_t* d_hit, uint32_t* d_hit_count, void* stream)
{
    if (!d_hit_count || (n && (!d_actual || !d_predicted || !d_hit))) return SPECKV_ERR_INVAL;
    if (n && (k == 0 || k > 64)) return SPECKV_ERR_INVAL;
    hipError_t e = speckv::launch_verify(n, k, d_actual, d_predicted, d_hit, d_hit_count, static_cast<hipStream_t>(stream));
    if (e != hipSuccess) {
        fprintf(stderr, "[libcxlspeckv] speckv_ext_verify_batch: %s\n", hipGetErrorString(e));
        return SPECKV_ERR_DRIVER;
    }
    return SPECKV_OK;
}

speckv_status_t speckv_ext_get_prefetch_depth(uint32_t* depth_k)
{
    LOCK;
    if (!g_engine || g_closing || !depth_k) return SPECKV_ERR_INVAL;
    *depth_k = g_engine->prefetch_depth();
    return SPECKV_OK;
}

speckv_status_t speckv_ext_poll_complete(uint32_t* done)
{
    LOCK;
    if (!g_engine || g_closing || !done) return SPECKV_ERR_INVAL;
    return guarded([&] { return g_engine->poll_complete(done); });
}

speckv_status_t speckv_ext_sync(void)
{
    LOCK; NEED_INIT;
    return guarded([&] { return g_engine->sync(); });
}

// raw codec operators: no engine state, caller-owned device buffers
static speckv_status_t codec_launch(bool compress, const speckv::CodecArgs& a, void* stream)
{
    hipError_t e = compress ? speckv::launch_compress(a, static_cast<hipStream_t>(stream))
                            : speckv::launch_decompress(a, static_cast<hipStream_t>(stream));
    if (e != hipSuccess) {
        fprintf(stderr, "[libcxlspeckv] codec launch failed: %s\n", hipGetErrorString(e));
        (void)hipGetLastError();
        return SPECKV_ERR_DRIVER;
    }
    return SPECKV_OK;
}

speckv_status_t speckv_ext_codec_compress(const void* d_src_f16, uint64_t n_blocks, void* d_recs, uint64_t rec_stride,
                                          uint32_t* d_rec_bytes, float* d_scales, int scheme, int quant_mode, void* stream)
{
    if (scheme < 0 || scheme > SPECKV_COMP_MXFP4 || quant_mode < 0 || quant_mode > 1) return SPECKV_ERR_INVAL;
    if (n_blocks && (!d_src_f16 || !d_recs || !d_rec_bytes)) return SPECKV_ERR_INVAL;
    if (rec_stride % 16) return SPECKV_ERR_INVAL;
    const uint64_t need = scheme == SPECKV_COMP_INT8 || scheme == SPECKV_COMP_FP8_E4M3 ? 2048u
                        : scheme == SPECKV_COMP_INT4_G32 ? 1152u : scheme == SPECKV_COMP_MXFP4 ? 1088u : 4096u;
    if (rec_stride < need) return SPECKV_ERR_INVAL;
    speckv::CodecArgs a{};
    a.recs = static_cast<uint8_t*>(d_recs);
    a.rec_stride = rec_stride;
    a.rec_bytes = d_rec_bytes;
    a.scales = d_scales;
    a.data = static_cast<uint8_t*>(const_cast<void*>(d_src_f16));
    a.data_stride = speckv::kPageSize;
    a.n = n_blocks;
    a.scheme = scheme;
    a.quant_mode = quant_mode;
    return codec_launch(true, a, stream);
}

speckv_status_t speckv_ext_codec_decompress(const void* d_recs, uint64_t rec_stride, const uint32_t* d_rec_bytes,
                                            const float* d_scales, uint64_t n_blocks, void* d_dst, int out_f32,
                                            int scheme, int quant_mode, void* stream)
{
    const int structured = quant_mode & SPECKV_CODEC_HINT_STRUCTURED;
    quant_mode &= ~SPECKV_CODEC_HINT_STRUCTURED;
    if (scheme < 0 || scheme > SPECKV_COMP_MXFP4 || quant_mode < 0 || quant_mode > 1) return SPECKV_ERR_INVAL;
    if (n_blocks && (!d_recs || !d_rec_bytes || !d_dst)) return SPECKV_ERR_INVAL;
    if (rec_stride % 16) return SPECKV_ERR_INVAL;
    speckv::CodecArgs a{};
    a.recs = static_cast<uint8_t*>(const_cast<void*>(d_recs));
    a.rec_stride = rec_stride;
    a.rec_bytes = const_cast<uint32_t*>(d_rec_bytes);
    a.scales = const_cast<float*>(d_scales);
    a.data = static_cast<uint8_t*>(d_dst);
    a.data_stride = out_f32 ? 2ull * speckv::kPageSize : speckv::kPageSize;
    a.n = n_blocks;
    a.scheme = scheme;
    a.quant_mode = quant_mode;
    a.out_f32 = out_f32 ? 1 : 0;
    a.structured_hint = structured ? 1 : 0;
    return codec_launch(false, a, stream);
}

// FPGACacheEngine::compress / ::decompress with the reference's own call shape: a tensor of any length, one scale, one
// delta chain, one run-length stream (cache_engine.cpp:40-116)
size_t speckv_ext_codec_tensor_workspace_bytes(uint64_t n_elems) { return speckv::tensor_compress_workspace_bytes(n_elems); }
size_t speckv_ext_codec_tensor_decode_workspace_bytes(uint64_t rle_bytes) { return speckv::tensor_decompress_workspace_bytes(rle_bytes); }

speckv_status_t speckv_ext_codec_compress_tensor(const void* d_src, uint64_t n_elems, int src_f32, void* d_rle, uint64_t* d_rle_bytes,
                                                 float* d_scale, void* d_workspace, size_t workspace_bytes, int quant_mode, void* stream)
{
    if (quant_mode < 0 || quant_mode > 1 || !d_rle_bytes || !d_scale || !d_workspace) return SPECKV_ERR_INVAL;
    if (n_elems && (!d_src || !d_rle)) return SPECKV_ERR_INVAL;
    if ((reinterpret_cast<uintptr_t>(d_rle) & 15u) || (reinterpret_cast<uintptr_t>(d_workspace) & 255u)) return SPECKV_ERR_INVAL;
    if (workspace_bytes < speckv::tensor_compress_workspace_bytes(n_elems)) return SPECKV_ERR_INVAL;
    const hipError_t e = speckv::launch_tensor_compress(d_src, n_elems, src_f32 != 0, static_cast<uint8_t*>(d_rle), d_rle_bytes, d_scale,
                                                        d_workspace, workspace_bytes, quant_mode, static_cast<hipStream_t>(stream));
    if (e != hipSuccess) {
        fprintf(stderr, "[libcxlspeckv] speckv_ext_codec_compress_tensor: %s\n", hipGetErrorString(e));
        (void)hipGetLastError();
        return SPECKV_ERR_DRIVER;
    }
    return SPECKV_OK;
}

speckv_status_t speckv_ext_codec_decompress_tensor(const void* d_rle, uint64_t rle_bytes, float scale, void* d_dst, uint64_t dst_cap_elems,
                                                   int out_f32, uint64_t* d_n_out, void* d_workspace, size_t workspace_bytes,
                                                   int quant_mode, void* stream)
{
    if (quant_mode < 0 || quant_mode > 1 || !d_workspace) return SPECKV_ERR_INVAL;
    if (rle_bytes && !d_rle) return SPECKV_ERR_INVAL;
    if (dst_cap_elems && !d_dst) return SPECKV_ERR_INVAL;
    if ((reinterpret_cast<uintptr_t>(d_dst) & 15u) || (reinterpret_cast<uintptr_t>(d_rle) & 1u) || (reinterpret_cast<uintptr_t>(d_workspace) & 255u))
        return SPECKV_ERR_INVAL;
    if (workspace_bytes < speckv::tensor_decompress_workspace_bytes(rle_bytes)) return SPECKV_ERR_INVAL;
    const hipError_t e = speckv::launch_tensor_decompress(static_cast<const uint8_t*>(d_rle), rle_bytes, scale, d_dst, dst_cap_elems, out_f32 != 0,
                                                          d_n_out, d_workspace, workspace_bytes, quant_mode, static_cast<hipStream_t>(stream));
    if (e != hipSuccess) {
        fprintf(stderr, "[libcxlspeckv] speckv_ext_codec_decompress_tensor: %s\n", hipGetErrorString(e));
        (void)hipGetLastError();
        return SPECKV_ERR_DRIVER;
    }
    return SPECKV_OK;
}

// ... and over MANY tensors per launch (the reference calls compress(data, n) per KV tile: 131 072 elements)
size_t speckv_ext_codec_tensors_workspace_bytes(uint32_t n_tensors, uint64_t max_elems) { return speckv::tensors_workspace_bytes(n_tensors, max_elems); }

speckv_status_t speckv_ext_codec_compress_tensors(uint32_t n_tensors, const speckv_ext_tensor_t* d_tensors, uint64_t max_elems, int src_f32,
                                                  uint64_t* d_rle_bytes, float* d_scales, void* d_workspace, size_t workspace_bytes, int quant_mode, void* stream)
{
    static_assert(sizeof(speckv_ext_tensor_t) == sizeof(speckv::TensorDesc), "speckv_ext_tensor_t is the kernels' descriptor");
    if (quant_mode < 0 || quant_mode > 1) return SPECKV_ERR_INVAL;
    if (n_tensors && (!d_tensors || !d_rle_bytes || !d_scales)) return SPECKV_ERR_INVAL;
    if (n_tensors && (!d_workspace || (reinterpret_cast<uintptr_t>(d_workspace) & 255u) || workspace_bytes < speckv::tensors_workspace_bytes(n_tensors, max_elems)))
        return SPECKV_ERR_INVAL;
    const hipError_t e = speckv::launch_tensors_compress(n_tensors, reinterpret_cast<const speckv::TensorDesc*>(d_tensors), max_elems, src_f32 != 0, d_rle_bytes, d_scales,
                                                         d_workspace, workspace_bytes, quant_mode, static_cast<hipStream_t>(stream));
    if (e != hipSuccess) {
        fprintf(stderr, "[libcxlspeckv] speckv_ext_codec_compress_tensors: %s\n", hipGetErrorString(e));
        (void)hipGetLastError();
        return e == hipErrorInvalidValue ? SPECKV_ERR_INVAL : SPECKV_ERR_DRIVER;
    }
    return SPECKV_OK;
}

speckv_status_t speckv_ext_codec_decompress_tensors(uint32_t n_tensors, const speckv_ext_tensor_t* d_tensors, uint64_t max_elems, const uint64_t* d_rle_bytes,
                                                    const float* d_scales, int out_f32, uint64_t* d_n_out, void* d_workspace, size_t workspace_bytes,
                                                    int quant_mode, void* stream)
{
    if (quant_mode < 0 || quant_mode > 1) return SPECKV_ERR_INVAL;
    if (n_tensors && (!d_tensors || !d_rle_bytes || !d_scales)) return SPECKV_ERR_INVAL;
    if (n_tensors && (!d_workspace || (reinterpret_cast<uintptr_t>(d_workspace) & 255u) || workspace_bytes < speckv::tensors_workspace_bytes(n_tensors, max_elems)))
        return SPECKV_ERR_INVAL;
    const hipError_t e = speckv::launch_tensors_decompress(n_tensors, reinterpret_cast<const speckv::TensorDesc*>(d_tensors), max_elems, d_rle_bytes, d_scales,
                                                           out_f32 != 0, d_n_out, d_workspace, workspace_bytes, quant_mode, static_cast<hipStream_t>(stream));
    if (e != hipSuccess) {
        fprintf(stderr, "[libcxlspeckv] speckv_ext_codec_decompress_tensors: %s\n", hipGetErrorString(e));
        (void)hipGetLastError();
        return e == hipErrorInvalidValue ? SPECKV_ERR_INVAL : SPECKV_ERR_DRIVER;
    }
    return SPECKV_OK;
}

speckv_status_t speckv_ext_qk_scores_fp8(speckv_handle_t handle, uint32_t layer, const void* d_q_f16, uint32_t g,
                                         uint32_t pos_begin, uint32_t pos_end, float* d_out, void* stream)
{
    LOCK; NEED_INIT;
    return guarded([&] {
        return g_engine->qk_scores_fp8(handle, layer, 1, d_q_f16, g, pos_begin, pos_end, d_out, static_cast<hipStream_t>(stream));
    });
}

speckv_status_t speckv_ext_qk_scores_fp8_layers(speckv_handle_t handle, uint32_t layer_begin, uint32_t n_layers,
                                                const void* d_q_f16, uint32_t g, uint32_t pos_begin, uint32_t pos_end,
                                                float* d_out, void* stream)
{
    LOCK; NEED_INIT;
    return guarded([&] {
        return g_engine->qk_scores_fp8(handle, layer_begin, n_layers, d_q_f16, g, pos_begin, pos_end, d_out,
                                       static_cast<hipStream_t>(stream));
    });
}

speckv_status_t speckv_ext_attend_fp8(speckv_handle_t handle, uint32_t layer_begin, uint32_t n_layers, const void* d_q_f16,
                                      uint32_t g, uint32_t pos_begin, uint32_t pos_end, float sm_scale, float* d_out,
                                      float* d_lse, void* stream)
{
    LOCK; NEED_INIT;
    return guarded([&] {
        return g_engine->attend_fp8(handle, layer_begin, n_layers, d_q_f16, g, pos_begin, pos_end, sm_scale, d_out, d_lse,
                                    static_cast<hipStream_t>(stream));
    });
}

speckv_status_t speckv_ext_attend_fp8_batch(uint32_t n_seq, const speckv_handle_t* handles, uint32_t layer, const void* d_q_f16,
                                            uint32_t g, const uint32_t* pos_end, float sm_scale, float* d_out, float* d_lse,
                                            void* stream)
{
    LOCK; NEED_INIT;
    return guarded([&] {
        return g_engine->attend_batch(SPECKV_COMP_FP8_E4M3, n_seq, handles, layer, d_q_f16, g, pos_end, sm_scale, d_out, d_lse,
                                      static_cast<hipStream_t>(stream));
    });
}

speckv_status_t speckv_ext_attend_int4_batch(uint32_t n_seq, const speckv_handle_t* handles, uint32_t layer, const void* d_q_f16,
                                             uint32_t g, const uint32_t* pos_end, float sm_scale, float* d_out, float* d_lse,
                                             void* stream)
{
    LOCK; NEED_INIT;
    return guarded([&] {
        return g_engine->attend_batch(SPECKV_COMP_INT4_G32, n_seq, handles, layer, d_q_f16, g, pos_end, sm_scale, d_out, d_lse,
                                      static_cast<hipStream_t>(stream));
    });
}

speckv_status_t speckv_ext_attend_batch_plan(uint32_t n_seq, const speckv_handle_t* handles, const uint32_t* pos_end,
                                             uint32_t max_pos_end, void* d_plan, size_t plan_bytes, void* stream)
{
    LOCK; NEED_INIT;
    return guarded([&] {
        return g_engine->attend_batch_plan(n_seq, handles, pos_end, max_pos_end, d_plan, plan_bytes, static_cast<hipStream_t>(stream));
    });
}

// one descriptor per sequence, then the dispatch order (one index per sequence: Engine::attend_batch_plan)
size_t speckv_ext_attend_plan_bytes(uint32_t n_seq) { return static_cast<size_t>(n_seq) * (sizeof(speckv::AttendSeq) + sizeof(uint32_t)); }

speckv_status_t speckv_ext_attend_fp8_planned(const void* d_plan, uint32_t n_seq, uint32_t layer, const void* d_q_f16, uint32_t g,
                                              uint32_t max_pos_end, float sm_scale, float* d_out, float* d_lse, void* stream)
{
    LOCK; NEED_INIT;
    return guarded([&] {
        return g_engine->attend_planned(SPECKV_COMP_FP8_E4M3, d_plan, n_seq, layer, d_q_f16, g, max_pos_end, sm_scale, d_out, d_lse,
                                        static_cast<hipStream_t>(stream));
    });
}

speckv_status_t speckv_ext_attend_int4_planned(const void* d_plan, uint32_t n_seq, uint32_t layer, const void* d_q_f16, uint32_t g,
                                               uint32_t max_pos_end, float sm_scale, float* d_out, float* d_lse, void* stream)
{
    LOCK; NEED_INIT;
    return guarded([&] {
        return g_engine->attend_planned(SPECKV_COMP_INT4_G32, d_plan, n_seq, layer, d_q_f16, g, max_pos_end, sm_scale, d_out, d_lse,
                                        static_cast<hipStream_t>(stream));
    });
}

speckv_status_t speckv_ext_attend_fold_tail(uint32_t n_rows, const uint32_t* d_rows, uint32_t heads, uint32_t g, const void* d_q_f16,
                                            const void* d_k_tail, const void* d_v_tail, uint64_t tail_stride_elems, float sm_scale,
                                            float* d_out, float* d_lse, void* stream)
{
    LOCK; NEED_INIT;
    return guarded([&] {
        return g_engine->attend_fold_tail(n_rows, d_rows, heads, g, d_q_f16, d_k_tail, d_v_tail, tail_stride_elems, sm_scale, d_out, d_lse,
                                          static_cast<hipStream_t>(stream));
    });
}

speckv_status_t speckv_ext_attend_int4(speckv_handle_t handle, uint32_t layer_begin, uint32_t n_layers, const void* d_q_f16,
                                       uint32_t g, uint32_t pos_begin, uint32_t pos_end, float sm_scale, float* d_out,
                                       float* d_lse, void* stream)
{
    LOCK; NEED_INIT;
    return guarded([&] {
        return g_engine->attend_int4(handle, layer_begin, n_layers, d_q_f16, g, pos_begin, pos_end, sm_scale, d_out, d_lse,
                                     static_cast<hipStream_t>(stream));
    });
}

speckv_status_t speckv_ext_attend_mx4(speckv_handle_t handle, uint32_t layer_begin, uint32_t n_layers, const void* d_q_f16,
                                      uint32_t g, uint32_t pos_begin, uint32_t pos_end, float sm_scale, float* d_out,
                                      float* d_lse, void* stream)
{
    LOCK; NEED_INIT;
    return guarded([&] {
        return g_engine->attend_mx4(handle, layer_begin, n_layers, d_q_f16, g, pos_begin, pos_end, sm_scale, d_out, d_lse,
                                    static_cast<hipStream_t>(stream));
    });
}

speckv_status_t speckv_ext_attend_mx4_batch(uint32_t n_seq, const speckv_handle_t* handles, uint32_t layer, const void* d_q_f16,
                                            uint32_t g, const uint32_t* pos_end, float sm_scale, float* d_out, float* d_lse,
                                            void* stream)
{
    LOCK; NEED_INIT;
    return guarded([&] {
        return g_engine->attend_batch(SPECKV_COMP_MXFP4, n_seq, handles, layer, d_q_f16, g, pos_end, sm_scale, d_out, d_lse,
                                      static_cast<hipStream_t>(stream));
    });
}

speckv_status_t speckv_ext_attend_mx4_planned(const void* d_plan, uint32_t n_seq, uint32_t layer, const void* d_q_f16, uint32_t g,
                                              uint32_t max_pos_end, float sm_scale, float* d_out, float* d_lse, void* stream)
{
    LOCK; NEED_INIT;
    return guarded([&] {
        return g_engine->attend_planned(SPECKV_COMP_MXFP4, d_plan, n_seq, layer, d_q_f16, g, max_pos_end, sm_scale, d_out, d_lse,
                                        static_cast<hipStream_t>(stream));
    });
}

speckv_status_t speckv_ext_attend_planned_tail(int scheme, const void* d_plan, uint32_t n_seq, uint32_t layer, const void* d_q_f16, uint32_t g,
                                               uint32_t max_pos_end, float sm_scale, float* d_out, float* d_lse, uint32_t n_tail,
                                               const uint32_t* d_tail_rows, const int32_t* d_tail_idx, const void* d_k_tail, const void* d_v_tail,
                                               uint64_t tail_stride_elems, void* stream)
{
    LOCK; NEED_INIT;
    if (scheme != SPECKV_COMP_FP8_E4M3 && scheme != SPECKV_COMP_INT4_G32 && scheme != SPECKV_COMP_MXFP4) return SPECKV_ERR_INVAL;
    if (n_tail > n_seq || (n_tail && n_tail < n_seq && !d_tail_rows)) return SPECKV_ERR_INVAL;
    return guarded([&] {
        const Engine::TailArgs t{n_tail, d_tail_rows, d_tail_idx, d_k_tail, d_v_tail, tail_stride_elems};
        return g_engine->attend_planned(scheme, d_plan, n_seq, layer, d_q_f16, g, max_pos_end, sm_scale, d_out, d_lse, static_cast<hipStream_t>(stream), &t);
    });
}

speckv_status_t speckv_ext_attend_planned_layers(int scheme, const void* d_plan, uint32_t n_seq, uint32_t layer_begin, uint32_t n_layers, const void* d_q_f16,
                                                 uint32_t g, uint32_t max_pos_end, float sm_scale, float* d_out, float* d_lse, uint32_t n_tail,
                                                 const uint32_t* d_tail_rows, const int32_t* d_tail_idx, const void* d_k_tail, const void* d_v_tail,
                                                 uint64_t tail_stride_elems, void* stream)
{
    LOCK; NEED_INIT;
    if (scheme != SPECKV_COMP_FP8_E4M3 && scheme != SPECKV_COMP_INT4_G32 && scheme != SPECKV_COMP_MXFP4) return SPECKV_ERR_INVAL;
    if (n_tail > n_seq || (n_tail && n_tail < n_seq && !d_tail_rows)) return SPECKV_ERR_INVAL;
    return guarded([&] {
        const Engine::TailArgs t{n_tail, d_tail_rows, d_tail_idx, d_k_tail, d_v_tail, tail_stride_elems};
        return g_engine->attend_planned_layers(scheme, d_plan, n_seq, layer_begin, n_layers, d_q_f16, g, max_pos_end, sm_scale, d_out, d_lse,
                                               static_cast<hipStream_t>(stream), n_tail ? &t : nullptr);
    });
}

speckv_status_t speckv_ext_promote_to_l1(speckv_handle_t handle, uint64_t offset_bytes)
{
    LOCK; NEED_INIT;
    return guarded([&] { return g_engine->promote_to_l1(handle, offset_bytes); });
}

speckv_status_t speckv_ext_demote_to_l3(speckv_handle_t handle, uint64_t offset_bytes)
{
    LOCK; NEED_INIT;
    return guarded([&] { return g_engine->demote_to_l3(handle, offset_bytes); });
}

speckv_status_t speckv_ext_predictor_load(const float* embedding, const float* out_weights, uint32_t vocab, int on_device)
{
    LOCK; NEED_INIT;
    return guarded([&] { return g_engine->predictor_load(embedding, out_weights, vocab, on_device != 0); });
}

speckv_status_t speckv_ext_predictor_load_lstm(const float* embedding, uint32_t vocab, uint32_t n_layers, const float* const* w_ih,
                                               const float* const* w_hh, const float* const* b_ih, const float* const* b_hh,
                                               const float* out_weights, const float* out_bias, int on_device)
{
    LOCK; NEED_INIT;
    return guarded([&] { return g_engine->predictor_load_lstm(embedding, vocab, n_layers, w_ih, w_hh, b_ih, b_hh, out_weights, out_bias, on_device != 0); });
}

speckv_status_t speckv_ext_predict_batch(uint32_t n, const int32_t* d_histories, uint32_t k, int32_t* d_tokens,
                                         float* d_conf, void* stream)
{
    LOCK; NEED_INIT;
    return guarded([&] { return g_engine->predict_batch(n, d_histories, k, d_tokens, d_conf, static_cast<hipStream_t>(stream)); });
}

speckv_status_t speckv_ext_migrate(speckv_handle_t handle, uint64_t first_page, uint64_t n_pages, uint32_t target_pool)
{
    LOCK; NEED_INIT;
    return guarded([&] { return g_engine->migrate(handle, first_page, n_pages, target_pool); });
}

speckv_status_t speckv_ext_compact(speckv_handle_t handle, uint64_t* bytes_before, uint64_t* bytes_after)
{
    LOCK; NEED_INIT;
    return guarded([&] { return g_engine->compact(handle, bytes_before, bytes_after); });
}

speckv_status_t speckv_ext_stats(speckv_ext_stats_t* out)
{
    LOCK;
    if (!g_engine || g_closing || !out) return SPECKV_ERR_INVAL;
    return guarded([&] { return g_engine->stats(out); });
}

uint32_t speckv_ext_abi_version(void) { return SPECKV_EXT_ABI_VERSION; }

speckv_status_t speckv_ext_stream_is_capturing(void* stream, int* out_capturing)
{
    if (!out_capturing) return SPECKV_ERR_INVAL;
    *out_capturing = 0;
    if (!stream) return SPECKV_OK;                              // the NULL stream cannot be captured
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(static_cast<hipStream_t>(stream), &cs) != hipSuccess) { (void)hipGetLastError(); return SPECKV_ERR_DRIVER; }
    *out_capturing = cs != hipStreamCaptureStatusNone ? 1 : 0;
    return SPECKV_OK;
}

speckv_status_t speckv_ext_set_tuning(const char* key, long long value)
{
    LOCK;
    return speckv::tuning_set(key, value) == 0 ? SPECKV_OK : SPECKV_ERR_INVAL;
}

speckv_status_t speckv_ext_stats_sized(void* out, size_t out_size, size_t* written)
{
    LOCK;
    if (!g_engine || g_closing || !out) return SPECKV_ERR_INVAL;
    return guarded([&] {
        speckv_ext_stats_t st;
        const int rc = g_engine->stats(&st);
        if (rc != SPECKV_OK) return rc;
        if (out_size < sizeof(uint64_t)) return static_cast<int>(SPECKV_ERR_INVAL);      // not even the first field
        const size_t n = out_size < sizeof(st) ? out_size : sizeof(st);
        std::memcpy(out, &st, n);
        // a caller built against a NEWER header (a larger struct) must not read uninitialised tail fields: they are zero
        if (out_size > n) std::memset(static_cast<uint8_t*>(out) + n, 0, out_size - n);
        if (written) *written = n;
        return static_cast<int>(SPECKV_OK);
    });
}

double speckv_ext_layer_compression_ratio(uint32_t layer_id)
{   // cache_engine.cpp:25-33,142-148 (80/3 = 26, 2*80/3 = 53)
    if (layer_id >= 80) return 3.2;
    if (layer_id < 80 / 3) return 3.5;
    if (layer_id > 2 * 80 / 3) return 2.75;
    return 3.2;
}

double speckv_ext_codec_model_throughput_gbps(uint32_t num_engines, double clock_mhz, uint32_t data_width_bits)
{   // cache_engine.cpp:291-296
    return (static_cast<double>(data_width_bits) / 8.0) * (clock_mhz / 1000.0) * static_cast<double>(num_engines);
}

uint64_t speckv_ext_encode_virt_page(uint32_t req_id, uint16_t layer, uint16_t head, uint32_t pos, uint8_t kind)
{   // speckv_allocator.cpp:92-103
    return (static_cast<uint64_t>(req_id) << 32) | (static_cast<uint64_t>(layer) << 16) |
           (static_cast<uint64_t>(head) << 8) | (static_cast<uint64_t>(pos) << 1) | static_cast<uint64_t>(kind);
}

uint64_t speckv_ext_rtl_prefetch_vaddr(uint32_t req_id, uint16_t layer, uint32_t pos)
{   // prefetch_core.v:92-98: {req[31:0], layer[15:0], 8'd0, pos[31:0], 1'b0} truncated to 64 bits
    return (static_cast<uint64_t>(pos) << 1) | (static_cast<uint64_t>(layer) << 41) | (static_cast<uint64_t>(req_id) << 57);
}

uint64_t speckv_ext_atu_translate(uint64_t virtual_addr)
{   // cache_engine.cpp:131-132
    return 0x4000000000ULL + (virtual_addr & 0xFFFFFFFFFFFFULL);
}

void speckv_ext_placement(uint64_t n_pages, uint32_t n_pool, uint64_t page, uint32_t* pool_index, uint64_t* record_index)
{
    (void)n_pages;
    const speckv::Placement pl = speckv::place_page(page, n_pool);
    if (pool_index) *pool_index = pl.pool;
    if (record_index) *record_index = pl.record;
}

uint64_t speckv_ext_pool_shard_pages(uint64_t n_pages, uint32_t n_pool, uint32_t pool_index)
{
    return speckv::shard_pages(n_pages, n_pool, pool_index);
}

const char* speckv_ext_backend(void) { return "hip"; }

} // extern "C"
