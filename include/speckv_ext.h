/*
 * include/speckv_ext.h -- additive entry points of libcxlspeckv.so.
 *
 * The eight functions of include/speckv.h are the drop-in surface.  The
 * reference C ABI stops at the fake pointer returned by speckv_access: it has
 * no write path (flags bit0 exists only in tests/test_dma.c:42), no way to
 * observe the page table (host/include/speckv_allocator.hpp:49-55 is private),
 * no batch form, and it never reaches its own codec
 * (src/fpga_engine/cache_engine.h:42-56) or prefetcher
 * (src/prefetcher/speculative_prefetcher.h:45-57).  The speckv_ext_* functions
 * below expose exactly those pieces, with plain pointers and sizes only, so
 * that (a) parity can be checked on logical ids and on codec bytes, (b) a
 * serving engine can drive the path in batches.  Each one cites the reference
 * interface it stands in for.
 *
 * Pointers named d_* are DEVICE pointers on the compute GPU.  `stream` is a
 * hipStream_t passed as void*: the work is ordered on it and the call returns
 * without waiting.  NULL = the engine's own stream and a SYNCHRONOUS call: it
 * first waits for the device (the caller's buffers may have been produced on
 * any stream) and returns when the result is complete.  A stream handed to the
 * library must stay alive until speckv_finalize or until the library has been
 * called on another stream after it: the engine orders later work behind the
 * stream that last wrote an allocation or last used one of its scratch buffers
 * by recording an event on that stream.
 */
#ifndef SPECKV_EXT_H
#define SPECKV_EXT_H

#include "speckv.h"

#ifdef __cplusplus
extern "C" {
#endif

#define SPECKV_PAGE_SIZE   4096u          /* speckv_allocator.cpp:18,58 */
#define SPECKV_BLOCK_ELEMS 2048u          /* fp16 elements per page */

/* ---- quantiser mode (SURVEY.md sect. 0.4) ----------------------------- */
typedef enum {
    SPECKV_QUANT_REF_EXACT = 0,  /* bit-for-bit cache_engine.cpp:186-196,275-284 */
    SPECKV_QUANT_INTENT    = 1,  /* q=clamp(round(x/s),-127,127), y=q*s (cache_engine.cpp:182, kv_compress.v:130) */
} speckv_quant_mode_t;
speckv_status_t speckv_ext_set_quant_mode(speckv_quant_mode_t mode);

/* ---- DMA descriptor: wire format of the reference (speckv_driver.hpp:9-14,
 *      driver/uapi/speckv_ioctl.h:10-15), 24 bytes -------------------------- */
typedef struct {
    uint64_t fpga_addr;   /* pool-side address (HBM of the pool GPU)            */
    uint64_t gpu_addr;    /* compute-GPU address                                */
    uint32_t bytes;
    uint32_t flags;       /* bit0 0=pool->GPU 1=GPU->pool, bit1 compressed, bit2 prefetch */
} speckv_dma_desc_t;
#define SPECKV_DMA_WRITE      0x1u
#define SPECKV_DMA_COMPRESSED 0x2u
#define SPECKV_DMA_PREFETCH   0x4u

/* ---- page table introspection (KvPageHandle, speckv_allocator.hpp:21-26) */
typedef struct {
    uint64_t virt_page_id;   /* (handle<<32)|(i<<12)          speckv_allocator.cpp:24 */
    uint64_t phys_page_id;   /* 0x4000000000+(h<<20)+(i<<12)  speckv_allocator.cpp:25 */
    uint32_t page_size;      /* 4096 */
    uint32_t flags;          /* bit0 L1, bit1 L2, bit2 compressed */
    int32_t  pool_device;    /* HIP device holding the pool copy, -1 in /dev/null mode */
    uint32_t scheme;         /* speckv_comp_scheme_t of the allocation */
    uint32_t rec_bytes;      /* bytes of the stored record (0 = never written) */
    float    scale;          /* per-block scale factor */
    uint64_t pool_addr;      /* device address of the record in the pool */
    uint64_t cache_addr;     /* device address of the decompressed copy, 0 if not resident */
    uint32_t access_count;   /* MemoryPage::access_count, cxl_memory_manager.h:34 */
    uint32_t aux_offset;     /* 0: the record's rec_bytes are contiguous at pool_addr.  MXFP4 (tile-planar pool, 16 records = 16 nibble
                              * rows + 16 code rows = 136 whole cache lines): bytes 0..1023 at pool_addr, bytes 1024..1087 (the E8M0
                              * codes) at pool_addr + aux_offset.  (The field was `reserved`, always 0, until ABI version 6.) */
} speckv_ext_page_info_t;
speckv_status_t speckv_ext_translate(speckv_handle_t handle, uint64_t offset_bytes,
                                     speckv_ext_page_info_t* out);

/* descriptor the engine would submit for a synchronous fetch of that page
 * (speckv_allocator.cpp:115-127) -- logical ids, for parity */
speckv_status_t speckv_ext_fetch_desc(speckv_handle_t handle, uint64_t offset_bytes,
                                      speckv_dma_desc_t* out);

/* ---- layout of the shim allocation (vllm_speckv_backend.py:26-43,87-100) */
speckv_status_t speckv_ext_set_layout(speckv_handle_t handle, uint32_t num_tokens,
                                      uint32_t num_layers, uint32_t num_heads,
                                      uint32_t head_dim, uint32_t bytes_per_element);

/* ---- data path --------------------------------------------------------- */
/* Populate the pool (compress with the allocation's scheme).  offset and len
 * must be multiples of 4096 (len may run to the end of the allocation).
 * Stands in for the DMA write direction the reference only sketches. */
speckv_status_t speckv_ext_write(speckv_handle_t handle, uint64_t offset_bytes,
                                 const void* src, size_t len, int src_on_device);
/* The append path of a decode loop: compress n pages first_page, first_page + page_step, ... from a contiguous device
 * buffer (n * 4096 bytes), ASYNCHRONOUSLY on `stream` (the source must stay valid until the stream gets there).  In
 * the shim layout the pages of one position pair in all (layer, kind) regions are num_tokens/2 pages apart, so one
 * call stores a sequence's new K and V rows of every layer.  Pages that are cached at the time of the call are
 * invalidated first (that case waits for the engine). */
speckv_status_t speckv_ext_write_strided(speckv_handle_t handle, uint64_t first_page, uint64_t page_step,
                                         uint64_t n_pages, const void* d_src, void* stream);
/* A contiguous page range from a device buffer, ASYNCHRONOUSLY on `stream` and without a device-wide wait (speckv_ext_write
 * with src_on_device waits for the whole device first: its source may come from any stream).  offset and len are
 * multiples of 4096.  Ordering: work the engine later does on its own stream for speckv_access / speckv_ext_read /
 * speckv_prefetch is ordered behind the write by the library; reads the caller issues on OTHER streams of its own
 * (speckv_ext_fetch_range, speckv_ext_attend_*) are the caller's to order, as with any stream. */
speckv_status_t speckv_ext_write_async(speckv_handle_t handle, uint64_t offset_bytes, const void* d_src, size_t len,
                                       void* stream);
/* The same for a batch of sequences in ONE launch: allocation handles[i] gets pages first_pages[i] + j * page_step
 * (j < n_pages_each) from d_srcs[i] + j * 4096.  All allocations must use the same compression scheme and the stream
 * must not be NULL.  (A decode step of 256 sequences appends with one call instead of 256 launches.) */
speckv_status_t speckv_ext_write_strided_batch(const speckv_handle_t* handles, const uint64_t* first_pages,
                                               const void* const* d_srcs, uint32_t n_allocations, uint64_t page_step,
                                               uint64_t n_pages_each, void* stream);
/* Several page runs of ONE allocation in one launch: run r = pages [first_pages[r], first_pages[r] + n_pages_each) from
 * d_srcs[r] (n_pages_each * 4096 contiguous bytes).  A prompt's K and V of every layer (2 * num_layers regions of the shim
 * layout) are stored with one call.  The runs must not overlap; the stream must not be NULL. */
speckv_status_t speckv_ext_write_runs(speckv_handle_t handle, const uint64_t* first_pages, const void* const* d_srcs,
                                      uint32_t n_runs, uint64_t n_pages_each, void* stream);
/* Fetch + decompress straight into a caller buffer, bypassing the tiers. */
speckv_status_t speckv_ext_read(speckv_handle_t handle, uint64_t offset_bytes,
                                void* dst, size_t len, int dst_on_device);
/* The bulk hot path: fetch + decompress pages [first_page, first_page+n) of a
 * handle into d_dst (n * 2048 elements, fp16 if out_f32==0 else fp32).
 * Asynchronous on `stream`.  One kernel launch. */
speckv_status_t speckv_ext_fetch_range(speckv_handle_t handle, uint64_t first_page,
                                       uint64_t n_pages, void* d_dst, int out_f32,
                                       void* stream);
/* The same with the fetch engine chosen by the caller: 0 = per batch (long runs on remote pools -> copy engines,
 * everything else -> fused kernel; SPECKV_REMOTE_ENGINE=kernel|copy overrides), 1 = fused peer-load + decompress
 * kernel, 2 = copy engines: the range's contiguous record runs (one per pool GPU, because pages are striped) are
 * copied with hipMemcpyPeerAsync on per-peer side streams into local staging and decompressed from there,
 * double-buffered.  Stands in for the reference's DMA path (one descriptor per page through the DMA engine:
 * host/src/speckv_allocator.cpp:115-138, hardware/rtl/dma_engine.v:150-217).  Engine 2 needs the default striped
 * placement (SPECKV_ERR_INVAL after a migration or on a fragmented allocation). */
speckv_status_t speckv_ext_fetch_range_engine(speckv_handle_t handle, uint64_t first_page,
                                              uint64_t n_pages, void* d_dst, int out_f32,
                                              void* stream, int engine);
/* Same for an arbitrary device-resident page list. */
speckv_status_t speckv_ext_fetch_list(speckv_handle_t handle, const uint32_t* d_pages,
                                      uint32_t n, void* d_dst, int out_f32, void* stream);
/* Batched speckv_access: out_ptrs[i] = device address of offsets[i]. */
speckv_status_t speckv_ext_access_batch(speckv_handle_t handle, const uint64_t* offsets,
                                        uint32_t n, void** out_ptrs);

/* ---- speculative prefetch (speculative_prefetcher.cpp:25-82, prefetch_core.v:150-241) */
/* Which allocation a request id of speckv_prefetch / speckv_ext_prefetch_batch addresses: request `req_id` is
 * request `local_req` of `handle`'s layout (one allocation per sequence: local_req 0).  handle 0 removes the binding.
 * Unbound request ids index into the most recently laid-out (else most recently allocated) allocation, as in the
 * reference's single-allocation shim (vllm_speckv_backend.py:95-100); requests that address nothing are counted in
 * speckv_ext_stats_t.prefetch_dropped. */
speckv_status_t speckv_ext_bind_request(uint32_t req_id, speckv_handle_t handle, uint32_t local_req);
/* Batched speckv_prefetch: host arrays of n requests (see speckv_ext_bind_request). */
speckv_status_t speckv_ext_prefetch_batch(uint32_t n, const uint32_t* req_ids,
                                          const uint16_t* layers, const uint32_t* cur_pos,
                                          const uint32_t* depth_k);
/* Drain queued prefetch requests now: candidates, residency filter, dedupe, ring-slot assignment and the fetch all
 * run on the device; the call only submits (no host round trip).  n_issued may be NULL; when given, the call waits
 * for the (small) assignment kernel and returns the number of pages being fetched into L2. */
speckv_status_t speckv_ext_prefetch_flush(uint32_t* n_issued);
/* The raw lookup kernel on device buffers: for request r, candidate pages of
 * positions cur_pos+1..cur_pos+depth_k (K then V), residency-filtered with the
 * handle's device flag mirror, compacted in request order into d_out_pages.
 * *d_out_count receives the total.  Deterministic. */
speckv_status_t speckv_ext_prefetch_lookup(speckv_handle_t handle, uint32_t n,
                                           const uint32_t* d_req_ids, const uint32_t* d_layers,
                                           const uint32_t* d_cur_pos, const uint32_t* d_depth_k,
                                           uint32_t* d_out_pages, uint32_t cap,
                                           uint32_t* d_out_count, void* stream);
/* Legacy address list of the reference's CPU prefetcher
 * (speculative_prefetcher.cpp:48,153-160): (0<<32)|(layer<<16)|(i+1). */
speckv_status_t speckv_ext_prefetch_legacy_addrs(uint32_t layer, uint32_t depth_k,
                                                 uint64_t* out_addrs, uint32_t* out_n);

/* Token predictor (LSTMPredictor, src/prefetcher/lstm_predictor.cpp:40-188).
 * The reference draws its weights from rand(); here the caller supplies them:
 * embedding [vocab][64], out_weights [vocab][128] (fp32, host or device).  Once loaded,
 * speckv_prefetch keeps the last 16 tokens per request, each flush predicts the next
 * tokens of the requests whose history changed (top-depth, depth <= 8) -- on a stream of the
 * engine's own, without holding the flush up: the pages a flush fetches are addressed by
 * position, the tokens only feed the hit statistics -- and
 * speckv_ext_verify(req, actual, NULL, 0) checks against that prediction (waiting for it
 * if it is still on its way).
 * speckv_ext_predict_batch is the raw operator: n histories of 16 int32 tokens
 * (device) -> top-k tokens / confidences (device, k <= 8). */
speckv_status_t speckv_ext_predictor_load(const float* embedding, const float* out_weights,
                                          uint32_t vocab, int on_device);
/* A REAL LSTM cell for the same predictor.  The reference's cell is degenerate (gates fixed at 0.5, recurrent weights never
 * read: lstm_predictor.cpp:117-146), which speckv_ext_predictor_load reproduces for parity; its semantics as a model are
 * therefore defined here: the standard LSTM with PyTorch's nn.LSTM conventions --
 *     gates = w_ih[l] x + b_ih[l] + w_hh[l] h + b_hh[l]   (rows [i | f | g | o], 4 x 128 each)
 *     c' = sigmoid(f) c + sigmoid(i) tanh(g),   h' = sigmoid(o) tanh(c'),   h_0 = c_0 = 0
 * n_layers <= 4 stacked layers (layer l > 0 is fed layer l-1's h of the same step), input = the 64-wide embedding of each of
 * the last 16 tokens, output layer logits = out_weights [vocab][128] . h_top + out_bias (may be NULL), softmax, top-k.
 * w_ih[0] is [512][64], w_ih[l > 0] and every w_hh[l] [512][128], biases [512]; fp32, host or device.  Replaces any earlier
 * predictor; speckv_ext_predictor_load switches back to the reference's cell. */
speckv_status_t speckv_ext_predictor_load_lstm(const float* embedding, uint32_t vocab, uint32_t n_layers,
                                               const float* const* w_ih, const float* const* w_hh,
                                               const float* const* b_ih, const float* const* b_hh,
                                               const float* out_weights, const float* out_bias, int on_device);
speckv_status_t speckv_ext_predict_batch(uint32_t n, const int32_t* d_histories, uint32_t k,
                                         int32_t* d_tokens, float* d_conf, void* stream);

/* Verification + adaptive depth (speculative_prefetcher.cpp:84-137).  predicted == NULL
 * verifies against the engine's own prediction for req_id (needs a loaded predictor). */
speckv_status_t speckv_ext_verify(uint32_t req_id, int32_t actual_token,
                                  const int32_t* predicted, uint32_t n_predicted,
                                  uint32_t* was_hit, uint32_t* new_depth);
/* Batched token-compare kernel: hit[r] = actual[r] in predicted[r*k .. r*k+k). */
speckv_status_t speckv_ext_verify_batch(uint32_t n, uint32_t k, const int32_t* d_actual,
                                        const int32_t* d_predicted, uint8_t* d_hit,
                                        uint32_t* d_hit_count, void* stream);
speckv_status_t speckv_ext_get_prefetch_depth(uint32_t* depth_k);

/* ---- completion queue (SpeckvDriver::poll_complete, speckv_driver.cpp:65-72;
 *      kernel side speckv_kernel_module.c:194-215): descriptors completed
 *      since the previous poll. */
speckv_status_t speckv_ext_poll_complete(uint32_t* done);
speckv_status_t speckv_ext_sync(void);

/* ---- raw codec operators on caller-owned device buffers -------------------
 * FPGACacheEngine::compress / ::decompress (cache_engine.cpp:40-116) applied
 * per 2048-element KV block.  Work without speckv_init (need a HIP device). */
speckv_status_t speckv_ext_codec_compress(const void* d_src_f16, uint64_t n_blocks,
                                          void* d_recs, uint64_t rec_stride,
                                          uint32_t* d_rec_bytes, float* d_scales,
                                          int scheme, int quant_mode, void* stream);
/* quant_mode of speckv_ext_codec_decompress may carry this flag: the records are known to be short (structured data, mean
 * record well under 512 bytes).  The launch then takes a decoder instantiation with a fast path for constant runs on
 * 8-element boundaries; the output is bit for bit the same with or without it.  (The engine sets it by itself for
 * allocations sealed by speckv_ext_compact whose packed records average under 512 bytes.) */
#define SPECKV_CODEC_HINT_STRUCTURED 0x100
speckv_status_t speckv_ext_codec_decompress(const void* d_recs, uint64_t rec_stride,
                                            const uint32_t* d_rec_bytes, const float* d_scales,
                                            uint64_t n_blocks, void* d_dst, int out_f32,
                                            int scheme, int quant_mode, void* stream);

/* The same two operators with the reference's own call shape -- a tensor of ANY length (cache_engine.cpp:40-116: n = 11
 * in the survey's known-answer vector, the RTL tile of 1024 x 128 = 131 072 elements): ONE scale over the n elements, ONE
 * int8 delta chain and ONE run-length stream; runs and their 255-element splits cross the 2048-element tiles the work is
 * cut into on the device.
 *   d_src        n_elems values, fp32 (src_f32 != 0: the reference's std::vector<float>) or fp16
 *   d_rle        the stream [value u8][count u8]...; 16-byte aligned, room for 2 * n_elems bytes rounded up to 16
 *   d_rle_bytes  (device) compressed_size;   d_scale (device) scale_factor
 *   d_workspace  256-byte aligned device scratch of speckv_ext_codec_tensor_workspace_bytes(n_elems) /
 *                speckv_ext_codec_tensor_decode_workspace_bytes(rle_bytes) bytes
 * Decompress: dst gets min(sum of counts, dst_cap_elems) elements (fp32 or fp16; 16-byte aligned), *d_n_out (device,
 * optional) that number; an odd trailing byte is dropped and a zero count emits nothing, as in the reference.
 * Pass the tensor's element count as dst_cap_elems when it is known: the library picks its decoder from rle_bytes / 2 pairs
 * against dst_cap_elems (a stream of one pair per element is decoded in one pass over the pairs, one that compresses tile by
 * tile of the output); a capacity far above the real count only costs speed, never correctness.
 * Asynchronous on `stream`; no engine needed. */
size_t speckv_ext_codec_tensor_workspace_bytes(uint64_t n_elems);
size_t speckv_ext_codec_tensor_decode_workspace_bytes(uint64_t rle_bytes);
speckv_status_t speckv_ext_codec_compress_tensor(const void* d_src, uint64_t n_elems, int src_f32, void* d_rle,
                                                 uint64_t* d_rle_bytes, float* d_scale, void* d_workspace,
                                                 size_t workspace_bytes, int quant_mode, void* stream);
speckv_status_t speckv_ext_codec_decompress_tensor(const void* d_rle, uint64_t rle_bytes, float scale, void* d_dst,
                                                   uint64_t dst_cap_elems, int out_f32, uint64_t* d_n_out,
                                                   void* d_workspace, size_t workspace_bytes, int quant_mode, void* stream);

/* The same codec over MANY tensors per launch (round 6).  The reference calls FPGACacheEngine::compress(data, n) once per KV tile
 * (cache_engine.cpp:40-82; the RTL's tile is 1024 x 128 = 131 072 elements, hardware/rtl/kv_compress.v:5-11): at that size one
 * tensor is 64 tiles of work and the single-tensor entry point above is all fixed cost (two launches, a look-back chain that has
 * barely started when it ends).  Here one launch takes thousands of tensors: every tensor its own scale, its own delta chain, its
 * own run-length stream; a tensor's workgroups (16 tiles each) find its max|x| by a rendezvous among themselves and hand their
 * chains on by look-back over the tensor's own status words; a wave takes max|x| from its own tile (fp16 tiles stay in registers for
 * the encode, fp32 tiles are read again out of the L2 / Infinity Cache).  Streams and scales are bit-identical to the single-tensor entry point's and to the reference's, per tensor.
 *   d_tensors  DEVICE array of n_tensors descriptors
 *   max_elems  the host's upper bound on the tensors' lengths (sizes the grid and the workspace; a tensor may be shorter, or empty)
 *   compress:   data = the source (fp16, or fp32 with src_f32), n = its elements, rle = where the stream goes (16-byte aligned),
 *               rle_cap >= 2 n rounded up to 16 (not checked); d_rle_bytes[i] / d_scales[i] receive the stream length / the scale
 *   decompress: data = the destination (16-byte aligned; fp16, or fp32 with out_f32), n = its room in elements (<= max_elems),
 *               rle = the stream (2 max_elems bytes at most), d_rle_bytes[i] / d_scales[i] as compress left them; d_n_out[i] (may be
 *               NULL) = elements decoded, clipped to the room
 *   d_workspace  256-byte aligned, speckv_ext_codec_tensors_workspace_bytes(n_tensors, max_elems) bytes (cleared by the call)
 * Asynchronous on `stream`; no engine needed. */
typedef struct {
    void*    data;
    uint64_t n;
    void*    rle;
    uint64_t rle_cap;
} speckv_ext_tensor_t;       /* 32 bytes */
size_t speckv_ext_codec_tensors_workspace_bytes(uint32_t n_tensors, uint64_t max_elems);
speckv_status_t speckv_ext_codec_compress_tensors(uint32_t n_tensors, const speckv_ext_tensor_t* d_tensors, uint64_t max_elems,
                                                  int src_f32, uint64_t* d_rle_bytes, float* d_scales, void* d_workspace,
                                                  size_t workspace_bytes, int quant_mode, void* stream);
speckv_status_t speckv_ext_codec_decompress_tensors(uint32_t n_tensors, const speckv_ext_tensor_t* d_tensors, uint64_t max_elems,
                                                    const uint64_t* d_rle_bytes, const float* d_scales, int out_f32,
                                                    uint64_t* d_n_out, void* d_workspace, size_t workspace_bytes,
                                                    int quant_mode, void* stream);

/* ---- 4:1 / 2:1 formats + fused dequant-matvec (BASELINE config 5; SURVEY 8a row
 *      A22: no reference counterpart, parity is against oracle/ only) -----------
 * SPECKV_COMP_INT4_G32: record 1152 B = 64 fp16 group scales + 2048 nibbles.
 * SPECKV_COMP_FP8_E4M3: record 2048 B of OCP e4m3fn + one f32 scale per block.
 * SPECKV_COMP_MXFP4:    record 1088 B = 2048 E2M1 nibbles + 64 E8M0 block scales (OCP MX v1.0; speckv_ext_attend_mx4).
 * speckv_ext_qk_scores_fp8: attention scores q.K^T for positions
 * [pos_begin, pos_end) (both even) of `layer`, computed on the matrix cores
 * directly from the FP8 records of the K region (request 0 of the shim layout;
 * needs speckv_ext_set_layout with num_heads*head_dim == 1024, head_dim 128).
 *   d_q_f16 : [num_heads][g][128] fp16 query rows (g <= 16 rows per kv head, GQA)
 *   d_out   : [num_heads][g][pos_end-pos_begin] fp32
 * The query is quantised per row to e4m3 (scale max|q|/448) on the device. */
speckv_status_t speckv_ext_qk_scores_fp8(speckv_handle_t handle, uint32_t layer,
                                         const void* d_q_f16, uint32_t g,
                                         uint32_t pos_begin, uint32_t pos_end,
                                         float* d_out, void* stream);
/* Several layers of the sequence in one launch: d_q_f16 [n_layers][num_heads][g][128],
 * d_out [n_layers][num_heads][g][pos_end-pos_begin]. */
speckv_status_t speckv_ext_qk_scores_fp8_layers(speckv_handle_t handle, uint32_t layer_begin,
                                                uint32_t n_layers, const void* d_q_f16, uint32_t g,
                                                uint32_t pos_begin, uint32_t pos_end,
                                                float* d_out, void* stream);

/* speckv_ext_attend_fp8: the whole decode attention of layers [layer_begin, layer_begin+n_layers)
 * over positions [pos_begin, pos_end) (both even), computed from the FP8 K and V records
 * without materialising fp16 KV:  out = softmax(q.K^T * sm_scale) . V  per kv head.
 *   d_q_f16 : [n_layers][num_heads][g][128] fp16      d_out : [n_layers][num_heads][g][128] fp32
 *   d_lse   : optional [n_layers][num_heads][g] fp32 log-sum-exp of the scaled scores (NULL to skip)
 * Scores on v_mfma_f32_16x16x32_fp8_fp8, weights in f16 against V widened to f16 (exact) on
 * v_mfma_f32_16x16x32_f16, fp32 accumulation; the position range is split across the chip and
 * merged by a second kernel.  Pages never written count as zeros.  Same layout requirement as
 * speckv_ext_qk_scores_fp8.  (second half; oracle: orc_attend_fp8.) */
speckv_status_t speckv_ext_attend_fp8(speckv_handle_t handle, uint32_t layer_begin, uint32_t n_layers,
                                      const void* d_q_f16, uint32_t g, uint32_t pos_begin, uint32_t pos_end,
                                      float sm_scale, float* d_out, float* d_lse, void* stream);

/* speckv_ext_attend_fp8_batch: one decode step of a batch -- the fused FP8 attention of ONE layer for n_seq sequences
 * (one allocation each, request 0 of its shim layout) over positions [0, pos_end[i]) in one launch pair.
 *   handles, pos_end : host arrays of n_seq      d_q_f16 : [n_seq][num_heads][g][128] fp16 (device)
 *   d_out : [n_seq][num_heads][g][128] fp32      d_lse : optional [n_seq][num_heads][g]
 * Every allocation must hold FP8 records with a layout whose num_tokens is a multiple of 32 (SPECKV_ERR_INVAL otherwise).
 * Any placement is served: records in one run or striped regularly over the pools take arithmetic addresses; if a member
 * of the batch has lost its regular placement (pages migrated one by one), the whole launch reads its record addresses
 * from the page tables instead (a few percent slower).  Asynchronous on `stream` (the host arrays are copied before the
 * call returns).  The split partials of all attention calls live in one scratch buffer of the engine: a call on another
 * stream than the previous one is ordered behind it by the library (an event at the old stream's tail), so callers may
 * use several streams, but such calls do not overlap on the device. */
speckv_status_t speckv_ext_attend_fp8_batch(uint32_t n_seq, const speckv_handle_t* handles, uint32_t layer,
                                            const void* d_q_f16, uint32_t g, const uint32_t* pos_end, float sm_scale,
                                            float* d_out, float* d_lse, void* stream);

/* speckv_ext_attend_int4: the same attention over SPECKV_COMP_INT4_G32 records (the 4:1 format).  K and V are
 * dequantised exactly as fetch+decompress does (fp16(q4 * group scale)), the query stays fp16, both products run
 * on v_mfma_f32_16x16x32_f16 with fp32 accumulation: the attention over the decompressed fp16 pages, without
 * writing them.  Records in one local run (the default placement of a one-pool engine) or striped regularly over the
 * pools take the arithmetic-address forms of the kernel; an allocation whose pages were migrated one by one takes the
 * same kernel with its record addresses read from the page table one tile ahead; only ranges that do not start on a
 * 32-position tile, or whose last tile would leave the layer, go through the per-wave page-table kernel.
 * (oracle: orc_attend_f16 over decompressed pages.) */
speckv_status_t speckv_ext_attend_int4(speckv_handle_t handle, uint32_t layer_begin, uint32_t n_layers,
                                       const void* d_q_f16, uint32_t g, uint32_t pos_begin, uint32_t pos_end,
                                       float sm_scale, float* d_out, float* d_lse, void* stream);

/* The batch form for INT4_G32 allocations (arguments as speckv_ext_attend_fp8_batch). */
speckv_status_t speckv_ext_attend_int4_batch(uint32_t n_seq, const speckv_handle_t* handles, uint32_t layer,
                                             const void* d_q_f16, uint32_t g, const uint32_t* pos_end, float sm_scale,
                                             float* d_out, float* d_lse, void* stream);

/* speckv_ext_attend_mx4: the same attention over SPECKV_COMP_MXFP4 records -- the 4:1 format gfx950's matrix cores read
 * natively (OCP Microscaling Formats v1.0: FP4 E2M1 elements, one E8M0 power-of-two scale per 32 elements; record = 1024 B of
 * nibbles -- byte i = element i in the low half, element 1024 + i in the high half: the two positions of a page interleaved --
 * then the 64 scale codes, one per 16 bytes: 1088 B per 4 KiB block, 3.76 : 1.  In the POOL the records of a run lie
 * tile-planar: 16 records = their 16 nibble rows followed by their 16 code rows = 136 whole cache lines, so the pool spends
 * 1088 B per page and the attention fetches nothing but record bytes; speckv_ext_translate reports a page's two pieces,
 * speckv_ext_page_info_t::aux_offset.  The raw operators speckv_ext_codec_* keep contiguous records).
 * q.K^T runs on v_mfma_scale_f32_16x16x128_f8f6f4: one instruction contracts the whole head dimension, K nibbles and scale
 * codes go in as they lie in the record, the block scales are applied by the hardware; the query is quantised to MXFP8
 * (e4m3 elements, E8M0 per 32: emax 8, saturating) on the device.  V is widened to f16 WITH its block scale by one
 * v_cvt_scalef32_pk_f16_fp4 per element pair and meets the f16 softmax weights on v_mfma_f32_16x16x32_f16, fp32 accumulation.
 * Any placement is served (one run, regular striping, page-table addresses).  g <= 16 query rows per kv head, taken in groups
 * of 8 (a second set of workgroups for rows 8 .. 15): the 16 columns of a score MFMA are 8 query rows x the 2 positions of a
 * page, so with g = 8 (GQA 8) every column is live, with g = 4 half of them are masked.
 * (oracle: orc_attend_mx4; arguments as speckv_ext_attend_int4.) */
speckv_status_t speckv_ext_attend_mx4(speckv_handle_t handle, uint32_t layer_begin, uint32_t n_layers,
                                      const void* d_q_f16, uint32_t g, uint32_t pos_begin, uint32_t pos_end,
                                      float sm_scale, float* d_out, float* d_lse, void* stream);
/* The batch and planned forms for MXFP4 allocations (arguments as speckv_ext_attend_fp8_batch / _planned). */
speckv_status_t speckv_ext_attend_mx4_batch(uint32_t n_seq, const speckv_handle_t* handles, uint32_t layer,
                                            const void* d_q_f16, uint32_t g, const uint32_t* pos_end, float sm_scale,
                                            float* d_out, float* d_lse, void* stream);
speckv_status_t speckv_ext_attend_mx4_planned(const void* d_plan, uint32_t n_seq, uint32_t layer, const void* d_q_f16,
                                              uint32_t g, uint32_t max_pos_end, float sm_scale, float* d_out,
                                              float* d_lse, void* stream);

/* ---- planned batches: the batch attention of a decode step under a HIP graph -------------------------------------
 * The batch calls above stage their per-sequence descriptors on every call and cannot be captured.  The planned form
 * splits them in two:
 *   speckv_ext_attend_batch_plan   once per decode step, OUTSIDE any capture: looks the n_seq allocations up and writes
 *       one descriptor per sequence -- valid for every layer -- into the caller's device buffer d_plan
 *       (speckv_ext_attend_plan_bytes(n_seq) bytes; the copy is ordered on `stream`), and behind them the order in which
 *       the launches take the sequences: batches whose members differ in length are dispatched by length (a long member
 *       beside a short one on every CU) -- rows of q / out / lse stay in the caller's order.  A buffer of n_seq x 64 bytes
 *       (what the size was before) still works, in the caller's order.
 *   speckv_ext_attend_{fp8,int4}_planned   one layer of the batch: kernel launches only (no look-ups, no staging,
 *       nothing allocated once the scratch is warm), so the per-layer calls of a step can be captured once and replayed
 *       for as long as every pos_end stays <= max_pos_end: grid and scratch are functions of (n_seq, max_pos_end) and of
 *       what the FIRST plan of that shape written into that buffer saw (a batch whose members differ much in length gets
 *       room for pieces per member and a merge launch; every later plan of the shape in the buffer keeps that room, so the
 *       captured launches stay valid); the lengths and piece lengths themselves are read from the plan on the device.
 *       Capture AFTER the first plan of a shape, as before.
 * Arguments as for the batch calls; max_pos_end (even) must be the value given to the plan.  `stream` must not be NULL.
 * Run each shape once outside the capture first (scratch growth during a capture is refused with SPECKV_ERR_INVAL).
 * A plan names record addresses: plan again after an allocation of the batch was freed, re-created or migrated. */
size_t speckv_ext_attend_plan_bytes(uint32_t n_seq);
speckv_status_t speckv_ext_attend_batch_plan(uint32_t n_seq, const speckv_handle_t* handles, const uint32_t* pos_end,
                                             uint32_t max_pos_end, void* d_plan, size_t plan_bytes, void* stream);
speckv_status_t speckv_ext_attend_fp8_planned(const void* d_plan, uint32_t n_seq, uint32_t layer, const void* d_q_f16,
                                              uint32_t g, uint32_t max_pos_end, float sm_scale, float* d_out,
                                              float* d_lse, void* stream);
speckv_status_t speckv_ext_attend_int4_planned(const void* d_plan, uint32_t n_seq, uint32_t layer, const void* d_q_f16,
                                               uint32_t g, uint32_t max_pos_end, float sm_scale, float* d_out,
                                               float* d_lse, void* stream);

/* A planned layer AND the position the caller still holds outside the pool, in one call (round 6: the connector used to issue
 * speckv_ext_attend_fold_tail behind every layer's attention -- eight dependent launches of a few microseconds in an 8-layer step).
 *   scheme        SPECKV_COMP_FP8_E4M3 / _INT4_G32 / _MXFP4 (the plan's)
 *   n_tail        sequences of the batch that have such a position (0: exactly speckv_ext_attend_*_planned)
 *   d_tail_rows   device array [n_tail] of their indices in the batch (NULL when n_tail == n_seq: tail i belongs to sequence i)
 *   d_tail_idx    device array [n_seq]: index of the sequence's tail rows, < 0 = none (the inverse of d_tail_rows; may be NULL when
 *                 n_tail == n_seq, or altogether -- then the fold is a launch of its own, below)
 *   d_k_tail, d_v_tail  fp16 [n_tail][layers][heads][128] (the BASE of the arrays: `layer` selects the row), tails
 *                 tail_stride_elems (a multiple of 8, >= layers * heads * 128) apart;  d_lse is required
 * MXFP4 with d_tail_idx (or n_tail == n_seq) and no empty sequence in the plan: the attention kernel folds the position in itself,
 * in the epilogue of each sequence's first split -- no launch behind it.  Otherwise one k_attend_fold_tail launch follows inside the
 * call.  Either way the result is speckv_ext_attend_*_planned followed by speckv_ext_attend_fold_tail.  Capturable. */
speckv_status_t speckv_ext_attend_planned_tail(int scheme, const void* d_plan, uint32_t n_seq, uint32_t layer, const void* d_q_f16,
                                               uint32_t g, uint32_t max_pos_end, float sm_scale, float* d_out, float* d_lse,
                                               uint32_t n_tail, const uint32_t* d_tail_rows, const int32_t* d_tail_idx,
                                               const void* d_k_tail, const void* d_v_tail, uint64_t tail_stride_elems, void* stream);

/* SEVERAL layers of a planned batch in one call, for callers that have the query rows of several layers at once (a draft model's
 * layers, a benchmark, layers whose attention inputs do not depend on each other): d_q_f16 [n_layers][n_seq][heads][g][128], d_out and
 * d_lse likewise; the tail arguments as speckv_ext_attend_planned_tail (n_tail == 0: none).  MXFP4 with a launch geometry of one split
 * per sequence (a batch that fills the chip by itself): ONE launch over layers x sequences -- the next layer's workgroups start while
 * the last of this one drain; everything else: the per-layer launches, issued from this one call.  Results equal the per-layer calls. */
speckv_status_t speckv_ext_attend_planned_layers(int scheme, const void* d_plan, uint32_t n_seq, uint32_t layer_begin, uint32_t n_layers,
                                                 const void* d_q_f16, uint32_t g, uint32_t max_pos_end, float sm_scale, float* d_out,
                                                 float* d_lse, uint32_t n_tail, const uint32_t* d_tail_rows, const int32_t* d_tail_idx,
                                                 const void* d_k_tail, const void* d_v_tail, uint64_t tail_stride_elems, void* stream);

/* speckv_ext_attend_fold_tail: one more position for rows that already hold an attention result and its log-sum-exp
 * (speckv_ext_attend_* with d_lse) -- the fp16 K / V row a decode step produced but has not stored yet (pages hold
 * position PAIRS; a connector keeps the odd position until its partner arrives):
 *     s = q.k * sm_scale;  new = logaddexp(lse, s);  out = out * exp(lse - new) + v * exp(s - new);  lse = new
 *   d_rows   : device array of n_rows sequence indices into d_q_f16 / d_out / d_lse, or NULL for 0..n_rows-1
 *   d_q_f16  : [n_seq][heads][g][128] fp16      d_out : [n_seq][heads][g][128] fp32      d_lse : [n_seq][heads][g] fp32
 *   d_k_tail, d_v_tail : fp16 [n_rows][heads][128], consecutive rows tail_stride_elems (>= heads*128, even) apart
 * A sequence without stored positions (out = 0, lse = -inf from the batch call) ends as out = v, lse = s.
 * One launch, in place, capturable. */
speckv_status_t speckv_ext_attend_fold_tail(uint32_t n_rows, const uint32_t* d_rows, uint32_t heads, uint32_t g,
                                            const void* d_q_f16, const void* d_k_tail, const void* d_v_tail,
                                            uint64_t tail_stride_elems, float sm_scale, float* d_out, float* d_lse,
                                            void* stream);

/* ---- tier manager (CXLMemoryManager, cxl_memory_manager.h:40-90) ---------- */
speckv_status_t speckv_ext_promote_to_l1(speckv_handle_t handle, uint64_t offset_bytes);
speckv_status_t speckv_ext_demote_to_l3(speckv_handle_t handle, uint64_t offset_bytes);
/* Move the pool records of pages [first_page, first_page+n) to pool GPU
 * `target_pool` (index into SPECKV_POOL_DEVICES order): hipMemcpyPeerAsync over
 * xGMI on a side stream, page table re-pointed, old slots freed.  The reference's
 * promote/demote only flip a tier tag (cxl_memory_manager.cpp:130-194). */
speckv_status_t speckv_ext_migrate(speckv_handle_t handle, uint64_t first_page,
                                   uint64_t n_pages, uint32_t target_pool);

/* Seal an allocation: pack its records back to back (128-byte aligned, page order, one extent per pool GPU) and return the
 * worst-case 4 KiB slots to the pool.  Only the variable-length scheme gains (SPECKV_COMP_INT8_DELTA_RLE; for the others the
 * call is a no-op): the pool then holds what the reference only counts (CompressedData::compressed_size, the ratio statistics
 * of cache_engine.cpp:62-78), and the copy-engine fetch moves record bytes instead of slots.  Reads are unaffected (they go
 * through the page table); a later write or migration first unpacks the allocation into slots again, so seal sequences that
 * are parked in the pool, not ones a decode loop appends to.  *bytes_before / *bytes_after (optional): pool bytes the
 * allocation holds.  Synchronous.  The packed extents are allocated before the slots are released (the records move in one
 * pass), so the pool needs room for the packed size on top of the slots for the duration of the call: SPECKV_ERR_NOMEM
 * leaves the allocation as it was. */
speckv_status_t speckv_ext_compact(speckv_handle_t handle, uint64_t* bytes_before, uint64_t* bytes_after);

/* ---- statistics (Statistics structs: cxl_memory_manager.h:73-83,
 *      speculative_prefetcher.h:59-66, cache_engine.h:65-72, memory_allocator.h:42-48) */
typedef struct {
    uint64_t l1_hits, l1_misses, l2_hits, l2_misses, l3_accesses;
    uint64_t migrations_l1_to_l3, migrations_l3_to_l1;
    uint64_t total_prefetches, successful_prefetches, mispredictions;
    uint64_t total_compressions, total_decompressions;
    uint64_t compressed_bytes, original_bytes;
    uint64_t total_allocations, total_deallocations;
    uint64_t current_allocated_bytes, peak_allocated_bytes;
    uint64_t dma_submitted, dma_completed;
    uint64_t pool_bytes_reserved, cache_bytes_reserved;
    uint32_t prefetch_depth, compression_scheme, quant_mode, n_pool_devices;
    uint64_t pool_migrated_pages;
    uint64_t prefetch_dropped;      /* speckv_prefetch requests that could not be addressed (no geometry / binding) */
    uint64_t copy_engine_runs;      /* hipMemcpyPeerAsync runs issued by the copy-engine fetch */
    uint64_t copy_engine_bytes;     /* bytes they moved into local staging */
    /* pool occupancy (cache_engine.cpp:62-78 keeps compressed_size / ratio statistics; here they are bytes of HBM):
     * capacity ratio of the live data = written_pages * 4096 / pool_bytes_in_use */
    uint64_t pool_bytes_in_use;     /* record storage held by live allocations (slots, or packed extents once sealed) */
    uint64_t written_pages;         /* pages of live allocations that hold a record */
    uint64_t sealed_allocations;    /* live allocations packed by speckv_ext_compact */
    uint64_t compactions;           /* speckv_ext_compact calls that packed an allocation */
    uint64_t flat_decoder_fetches;  /* speckv_ext_fetch_range launches that took the flat-run decoder by themselves (sealed size or length samples) */
} speckv_ext_stats_t;
/* The struct only ever grows at its end.  speckv_ext_stats() writes sizeof(speckv_ext_stats_t) of THIS header: a caller
 * compiled against an older header must use the sized form, which writes min(out_size, the library's size) bytes (fields
 * are never reordered, so a prefix is a valid older struct); bytes of the caller's buffer beyond the library's struct are
 * zero-filled (a caller built against a NEWER header reads zeros there); *written (optional) = bytes of real data; out_size < 8
 * is SPECKV_ERR_INVAL.
 * SPECKV_EXT_ABI_VERSION is bumped whenever a struct of this header grows or an entry point changes meaning;
 * speckv_ext_abi_version() returns the library's value (the Python binding refuses a mismatch). */
#define SPECKV_EXT_ABI_VERSION 6u
uint32_t        speckv_ext_abi_version(void);
speckv_status_t speckv_ext_stats(speckv_ext_stats_t* out);
speckv_status_t speckv_ext_stats_sized(void* out, size_t out_size, size_t* written);

/* ---- address encodings of the reference, pure functions ----
 * speckv_ext_encode_virt_page : SpeckvAllocator::encode_virt_page, speckv_allocator.cpp:92-103
 *                               (req<<32)|(layer<<16)|(head<<8)|(pos<<1)|kind  (fields overlap, as there)
 * speckv_ext_rtl_prefetch_vaddr: prefetch_core.v:92-98,158  low 64 bits of {req,layer,8'd0,pos,1'b0}
 * speckv_ext_atu_translate    : ATU/TLB miss mapping, cache_engine.cpp:131-132, address_translation.cpp:85-90:
 *                               0x4000000000 + (va & 0xFFFFFFFFFFFF)  (the engine itself translates through
 *                               the HBM page table; this is kept for parity of logical ids only) */
uint64_t speckv_ext_encode_virt_page(uint32_t req_id, uint16_t layer, uint16_t head, uint32_t pos, uint8_t kind);
uint64_t speckv_ext_rtl_prefetch_vaddr(uint32_t req_id, uint16_t layer, uint32_t pos);
uint64_t speckv_ext_atu_translate(uint64_t virtual_addr);

/* ---- legacy 3-tier address space (CXLMemoryManager, src/cxl_memory/cxl_memory_manager.cpp:9-322, and the
 *      cxl_access policy of src/integration/memory_allocator.cpp:105-143) -------------
 * Pure host arithmetic on logical addresses: bump allocation (virt from 0x100000000; phys L1 0x8000000000,
 * L2 0x10000000000, L3 0x20000000000), translate, tier tags with LRU / hot (> 10 touches) promotion, page states,
 * statistics -- call for call what the reference's class returns, including deallocate() forgetting only the first
 * page.  No data moves here (none moves in the reference either); the HIP engine applies the same policy to real
 * slots.  tier: 0 L1_GPU_LOCAL, 1 L2_PREFETCH, 2 L3_CXL_POOL; state: 0 INVALID, 1 SHARED, 2 EXCLUSIVE, 3 MODIFIED. */
typedef struct speckv_ext_mm speckv_ext_mm_t;
typedef struct {
    uint64_t l1_hits, l1_misses, l2_hits, l2_misses, l3_accesses;
    uint64_t migrations_l1_to_l3, migrations_l3_to_l1;
    double   l1_hit_rate, l2_hit_rate;
} speckv_ext_mm_stats_t;
speckv_ext_mm_t* speckv_ext_mm_new(uint64_t l1_gb, uint64_t l2_gb, uint64_t l3_gb, uint64_t page_size);
void     speckv_ext_mm_delete(speckv_ext_mm_t* m);
uint64_t speckv_ext_mm_allocate(speckv_ext_mm_t* m, uint64_t size_bytes, uint32_t layer_id, int preferred_tier);
void     speckv_ext_mm_deallocate(speckv_ext_mm_t* m, uint64_t virtual_addr);
uint64_t speckv_ext_mm_translate(speckv_ext_mm_t* m, uint64_t virtual_addr);
int      speckv_ext_mm_is_in_cache(speckv_ext_mm_t* m, uint64_t virtual_addr, int tier);
int      speckv_ext_mm_promote_to_l1(speckv_ext_mm_t* m, uint64_t virtual_addr);
int      speckv_ext_mm_demote_to_l3(speckv_ext_mm_t* m, uint64_t virtual_addr);
void     speckv_ext_mm_invalidate_page(speckv_ext_mm_t* m, uint64_t virtual_addr);
void     speckv_ext_mm_mark_modified(speckv_ext_mm_t* m, uint64_t virtual_addr);
int      speckv_ext_mm_get_page_state(speckv_ext_mm_t* m, uint64_t virtual_addr);
void     speckv_ext_mm_update_access_tracking(speckv_ext_mm_t* m, uint64_t virtual_addr);
int      speckv_ext_mm_is_hot_page(speckv_ext_mm_t* m, uint64_t virtual_addr);
void     speckv_ext_mm_get_statistics(speckv_ext_mm_t* m, speckv_ext_mm_stats_t* out);
uint64_t speckv_ext_mm_cxl_access(speckv_ext_mm_t* m, uint64_t base_virtual_addr, uint64_t offset);

/* ---- pool placement: the one rule the engine places records by, as pure functions -------------
 * An allocation of n_pages striped over n_pool pool GPUs: page p lives on pool p % n_pool as record p / n_pool of
 * that pool's run; pool k holds ceil((n_pages - k) / n_pool) records.  (n_pool == 0 is treated as 1.) */
void     speckv_ext_placement(uint64_t n_pages, uint32_t n_pool, uint64_t page,
                              uint32_t* pool_index, uint64_t* record_index);
uint64_t speckv_ext_pool_shard_pages(uint64_t n_pages, uint32_t n_pool, uint32_t pool_index);

/* hard-coded per-layer ratio table and analytic throughput of the reference
 * (cache_engine.cpp:25-33,142-148,286-296) */
double speckv_ext_layer_compression_ratio(uint32_t layer_id);
/* width/8 * MHz/1000 * engines (cache_engine.cpp:291-296): 51.2 for the reference's defaults */
double speckv_ext_codec_model_throughput_gbps(uint32_t num_engines, double clock_mhz, uint32_t data_width_bits);

/* Launch-form switches (tests, measurement runs): the library reads its environment ONCE, at the first speckv_init / raw codec
 * call of the process; after that a form is changed by this call only.  Keys (= the SPECKV_<KEY> environment names, lower case):
 * attend_splits, attend_tiles_per_split, attend_general, tc_multipass, tc_scan (1 one workgroup, 2 one wave), tc_no_pre,
 * tc_no_split_tiles, td_one_pass, td_expand_per_element, flush_no_small, flush_small_words, predict_batch_path, wgs_per_cu, tc_batch_one_wg (many-tensor launches: one workgroup per tensor),
 * rounds_consecutive, remote_engine (1 kernel, 2 copy engines), copy_min_run_kb, attend_stream (MXFP4, several layers of one
 * sequence: N > 0 the stream form with N workgroups, -1 never), attend_mx4_one_half (MXFP4 batches: 4-wave workgroups also where
 * the two-halves form applies), attend_order_as_given (batches of members of different lengths: the caller's dispatch order), attend_fold_launch (speckv_ext_attend_planned_tail: the fold always as a launch of its own), attend_layers_loop
 * (speckv_ext_attend_planned_layers: always per-layer launches), attend_fp8_table_regs / attend_fp8_striped_table / attend_int4_striped_wg
 * (striped and migrated pools: the earlier kernel forms, kept as A/B partners and test coverage).  0 restores the library's own rule.
 * SPECKV_ERR_INVAL for an unknown key.  Works without speckv_init.  (INTEGRATION.md lists what each one does.) */
speckv_status_t speckv_ext_set_tuning(const char* key, long long value);

/* Is `stream` (a hipStream_t) being captured into a HIP graph right now?  The batch entry points and speckv_ext_attend_batch_plan
 * refuse such a stream; a caller that plans ahead (SpeckvKVConnector.append) asks first.  *out_capturing = 0 for the NULL stream.
 * SPECKV_ERR_DRIVER when the runtime does not know the stream (destroyed by its owner).  Works without speckv_init. */
speckv_status_t speckv_ext_stream_is_capturing(void* stream, int* out_capturing);

/* library identity: "hip" when built with the HIP data path */
const char* speckv_ext_backend(void);

#ifdef __cplusplus
}
#endif
#endif /* SPECKV_EXT_H */
