/*
 * oracle/speckv_oracle.h -- CPU restatement of the FastLM/CXL-SpecKV hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library, and there only as the checker.  The shipped path
 * (cxl-speckv_amd/csrc -> libcxlspeckv.so) never links or calls it.
 *
 * Parity status: PINNED for everything that restates reference code -- ids, C ABI
 * model, INT8 / INT8_DELTA_RLE codec, memory manager, TLB, prefetcher, adaptive
 * depth, token predictor, coherence directory: validated here in the dev container
 * against the reference compiled from /root/reference by oracle/Makefile
 * (oracle/_ref/libspeckv_ref.so via oracle/ref_harness.cpp, and
 * oracle/_ref/libspeckv_ref_coh.so via oracle/coh_harness.cpp) and against the
 * golden fixtures in tests/golden/ generated from those builds by
 * tests/golden/generate_golden.py.
 * PARITY UNPINNED for the extensions that have no reference counterpart (SURVEY 8a
 * row A22): the INT4_G32 / FP8_E4M3 block formats, orc_qk_scores_fp8,
 * orc_attend_fp8, orc_attend_f16, orc_quantize_rows_e4m3 -- those are our own
 * definitions and say so where they are declared.
 *
 * All reference citations are relative to /root/reference.
 */
#ifndef SPECKV_ORACLE_H
#define SPECKV_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ */
/* fp16 helpers (IEEE binary16 <-> binary32; RNE on narrowing)         */
/* ------------------------------------------------------------------ */
float    orc_half_to_float(uint16_t h);
uint16_t orc_float_to_half(float f);

/* ------------------------------------------------------------------ */
/* Host allocator / page table ids  (host/src/speckv_allocator.cpp)    */
/* ------------------------------------------------------------------ */
#define ORC_PAGE_SIZE 4096u

/* speckv_allocator.cpp:24-25 */
uint64_t orc_virt_page_id(uint64_t handle, uint64_t page_idx);
uint64_t orc_phys_page_id(uint64_t handle, uint64_t page_idx);
/* speckv_allocator.cpp:18-19 */
uint64_t orc_num_pages(uint64_t bytes);
/* speckv_allocator.cpp:123 : gpu_addr of the sync-fetch descriptor */
uint64_t orc_desc_gpu_addr(uint64_t virt_page_id);
/* speckv_allocator.cpp:92-103 (private, never called by the reference) */
uint64_t orc_encode_virt_page(uint32_t req_id, uint16_t layer, uint16_t head,
                              uint32_t pos, uint8_t kind);

/* A model of the 8-function C ABI on the fake device ("/dev/null"):
 * host/src/speckv_c_api.cpp:13-121 + speckv_allocator.cpp:11-74.
 * Status codes are those of host/include/speckv.h:12-18. */
typedef struct orc_cabi orc_cabi_t;
orc_cabi_t* orc_cabi_new(void);
void        orc_cabi_delete(orc_cabi_t*);
int      orc_cabi_init(orc_cabi_t*, const char* dev_path);
void     orc_cabi_finalize(orc_cabi_t*);
int      orc_cabi_alloc(orc_cabi_t*, uint64_t bytes, uint64_t* out_handle);
int      orc_cabi_free(orc_cabi_t*, uint64_t handle);
int      orc_cabi_access(orc_cabi_t*, uint64_t handle, uint64_t offset,
                         uint64_t length, uint64_t* out_ptr);
int      orc_cabi_prefetch(orc_cabi_t*, uint32_t req_id, uint16_t layer,
                           uint32_t cur_pos, uint32_t depth_k,
                           const int32_t* tokens, uint32_t history_len);
int      orc_cabi_set_prefetch_depth(orc_cabi_t*, uint32_t k);
int      orc_cabi_set_compression_scheme(orc_cabi_t*, int scheme);
/* page flags after accesses (speckv_allocator.cpp:105-113,135): bit1 = L2 */
int      orc_cabi_page_flags(orc_cabi_t*, uint64_t handle, uint64_t page_idx,
                             uint32_t* out_flags);

/* Python shim offset (host/python/vllm_speckv_backend.py:87-100) */
uint64_t orc_calc_offset(uint64_t req_id, uint64_t layer, uint64_t head,
                         uint64_t pos, uint64_t kind, uint64_t entry_bytes,
                         uint64_t num_layers, uint64_t num_tokens,
                         uint64_t num_heads);
/* total bytes of one shim allocation (vllm_speckv_backend.py:39-40) */
uint64_t orc_shim_total_bytes(uint64_t num_tokens, uint64_t num_layers,
                              uint64_t num_heads, uint64_t head_dim,
                              uint64_t bytes_per_element);

/* ------------------------------------------------------------------ */
/* Codec  (src/fpga_engine/cache_engine.cpp)                           */
/* ------------------------------------------------------------------ */
enum { ORC_QUANT_REF_EXACT = 0, ORC_QUANT_INTENT = 1 };
enum { ORC_COMP_FP16 = 0, ORC_COMP_INT8 = 1, ORC_COMP_INT8_DELTA_RLE = 2 };

/* cache_engine.cpp:172-184 */
float  orc_compute_scale(const float* x, size_t n);
/* cache_engine.cpp:186-196 (REF_EXACT) / header+RTL intent (INTENT) */
void   orc_quantize(const float* x, size_t n, float scale, int mode, int8_t* q);
/* cache_engine.cpp:198-211 */
void   orc_delta_encode(const int8_t* q, size_t n, int8_t* d);
/* cache_engine.cpp:213-239 ; returns bytes written (<= 2n) */
size_t orc_rle_encode(const int8_t* d, size_t n, uint8_t* out);
/* cache_engine.cpp:241-258 ; returns elements written (<= cap) and sets
 * *total to the unclamped sum of counts */
size_t orc_rle_decode(const uint8_t* rle, size_t len, int8_t* out, size_t cap,
                      size_t* total);
/* cache_engine.cpp:260-273 */
void   orc_delta_decode(const int8_t* d, size_t n, int8_t* q);
/* cache_engine.cpp:275-284 (REF_EXACT) / q*scale (INTENT) */
void   orc_dequantize(const int8_t* q, size_t n, float scale, int mode, float* y);

/* FPGACacheEngine::compress (cache_engine.cpp:40-82): full pipeline.
 * rle must hold 2n bytes.  Returns compressed_size. */
size_t orc_compress_f32(const float* x, size_t n, int mode, float* scale,
                        uint8_t* rle);
/* FPGACacheEngine::decompress (cache_engine.cpp:84-116).  Returns elements. */
size_t orc_decompress_f32(const uint8_t* rle, size_t len, float scale, int mode,
                          float* y, size_t cap);

/* KV-block forms used by the engine: one block = n fp16 elements
 * (n = 2048 for a 4 KiB page).  fp16 -> fp32 is exact; the reference maths
 * runs in fp32; fp32 -> fp16 on the way out is one RNE rounding.
 * scheme 0: raw fp16 copy (rec = 2n bytes).
 * scheme 1: int8 quantise only (rec = n bytes).
 * scheme 2: quantise + delta + RLE (rec <= 2n bytes).
 * Returns record bytes. */
size_t orc_compress_block_f16(const uint16_t* x, size_t n, int scheme, int mode,
                              float* scale, uint8_t* rec);
size_t orc_decompress_block_f16(const uint8_t* rec, size_t len, float scale,
                                int scheme, int mode, uint16_t* y, size_t cap);
size_t orc_decompress_block_f32(const uint8_t* rec, size_t len, float scale,
                                int scheme, int mode, float* y, size_t cap);

/* The two block functions over n_blocks independent blocks on `threads` threads (test convenience: same
 * arithmetic, block b at x + b*n, recs + b*rec_stride, y + b*n; decoded blocks are zero-filled behind a short stream). */
void orc_compress_blocks_f16(const uint16_t* x, size_t n_blocks, size_t n, int scheme, int mode,
                             float* scales, uint32_t* lens, uint8_t* recs, size_t rec_stride, int threads);
void orc_decompress_blocks_f16(const uint8_t* recs, size_t rec_stride, const uint32_t* lens, const float* scales,
                               size_t n_blocks, size_t n, int scheme, int mode, uint16_t* y, int threads);

/* ---- 4:1 / 2:1 block formats of BASELINE config 5 --------
 * EXTENSION WITHOUT A REFERENCE COUNTERPART: **parity unpinned** -- there is no
 * reference code for these; this restatement is the definition the HIP kernels
 * are held to.
 *   scheme 3 INT4_G32 : record = 64 fp16 group scales (128 B) + 1024 B of
 *     nibbles (element 2i low, 2i+1 high, two's complement).  Per group of 32:
 *     s = fp16(max|x| / 7); q = clamp(roundf(x / float(s)), -7, 7) (0 when
 *     s == 0 or x is NaN); y = q * float(s).
 *   scheme 4 FP8_E4M3 : per-block f32 scale s = max|x| / 448 (1 when all zero),
 *     record = 2048 OCP e4m3fn bytes of clamp(x / s, +-448) rounded to nearest
 *     even; y = e4m3(b) * s.
 * fp16 in, fp16 (one RNE rounding) or fp32 out, like the other schemes. */
enum { ORC_COMP_INT4_G32 = 3, ORC_COMP_FP8_E4M3 = 4 };
uint8_t orc_f32_to_e4m3(float f);
float   orc_e4m3_to_f32(uint8_t b);
/* q.K^T scores straight from FP8 records (the fused dequant-matvec of config 5):
 * k_rec: n_pos rows of 128 e4m3 bytes (one kv head), k_scale[n_pos] the scale of
 * the block each row came from; q8: g rows of 128 e4m3 bytes with q_scale[g].
 * out[m*n_pos + t] = (sum_d q8[m][d]*k[t][d], fp32, d ascending) * k_scale[t] * q_scale[m]. */
void orc_qk_scores_fp8(const uint8_t* q8, const float* q_scale, size_t g,
                       const uint8_t* k_rec, const float* k_scale, size_t n_pos,
                       size_t d, float* out);
/* Decode attention of one kv head over FP8 records (own extension, parity unpinned):
 *   s[m][t]   = (sum_d q8[m][d]*k[t][d]) * k_scale[t] * q_scale[m] * sm_scale
 *   out[m][:] = sum_t softmax_t(s[m][:])[t] * v[t][:] * v_scale[t]        (double precision throughout)
 *   lse[m]    = log sum_t exp(s[m][t])   (NULL to skip);  mag[m][:] = sum_t softmax[t] * |v[t][:] * v_scale[t]|
 * mag (NULL to skip) is the magnitude the stated tolerance of the HIP kernel is relative to. */
void orc_attend_fp8(const uint8_t* q8, const float* q_scale, size_t g,
                    const uint8_t* k_rec, const float* k_scale,
                    const uint8_t* v_rec, const float* v_scale, size_t n_pos, size_t d,
                    float sm_scale, float* out, float* lse, float* mag);
/* Decode attention of one kv head over fp16 K / V rows (what fetch+decompress yields), fp16 query:
 *   out[m][:] = sum_t softmax_t(q[m].k[t] * sm_scale) * v[t][:]   in double precision;
 *   lse, mag as in orc_attend_fp8.  Checker of speckv_ext_attend_int4 (own extension, parity unpinned). */
void orc_attend_f16(const uint16_t* q16, size_t g, const uint16_t* k16, const uint16_t* v16, size_t n_pos, size_t d,
                    float sm_scale, float* out, float* lse, float* mag);
/* per-row e4m3 quantisation of the query (scale = max|q|/448, 1 if zero) */
void orc_quantize_rows_e4m3(const uint16_t* q16, size_t rows, size_t d, uint8_t* q8, float* scale);

/* ---- scheme 5 MXFP4: the 4:1 format gfx950's matrix cores read natively (BASELINE configs[4] "int4/fp8 KV compression
 * path (CDNA4 fp8 MFMA dequant), 4:1 ratio") ------------------------------------------------------------
 * EXTENSION WITHOUT A REFERENCE COUNTERPART: **parity unpinned** by the reference.  Element and scale formats and the
 * conversion are the OCP Microscaling Formats (MX) specification v1.0: blocks of 32 elements share one E8M0 scale
 * X = 2^(code - 127), elements are FP4 E2M1 (1-2-1, bias 1: 0, 0.5, 1, 1.5, 2, 3, 4, 6 and their negatives; no inf, no NaN):
 *     code = floor(log2(max|x|)) - emax_elem + 127      (emax_elem = 2 for E2M1, 8 for E4M3; max|x| over the finite part:
 *                                                        NaN elements are skipped, inf counts as 65504; 0 when max|x| == 0)
 *     q    = E2M1 of x / 2^(code - 127), rounded to nearest even, magnitudes beyond 6 clamp to 6 (NaN elements store +0)
 *     y    = e2m1(q) * 2^(code - 127)                   (exact in fp32; one RNE rounding on the way to fp16)
 * WHICH 32 elements share a scale is this format's own choice: the two halves of a block of n elements are interleaved element
 * by element first -- nibble 2i of the stream is element i, nibble 2i+1 element n/2 + i -- and the MX blocks are 32 consecutive
 * nibbles of that stream: 16 elements of the first half with the 16 matching elements of the second.  A 4 KiB KV page is two
 * positions x 1024 channels, so a block is 16 channels of BOTH positions and byte i of the record holds channel i of position 0
 * (low) and of position 1 (high) -- the 32 k values one lane feeds the block-scaled matrix instruction, and the pair of
 * positions one v_cvt_scalef32_pk_f16_fp4 widens into one f16 MFMA operand register (csrc/attend_mx4.hip).
 * Record of a 2048-element block (1088 B, 3.76 : 1): 1024 nibble bytes, then the 64 E8M0 codes (code j: bytes 16j .. 16j+15).
 * A short record decodes to zeros; code 255 (the spec's NaN) decodes to NaN.
 * Pinned independently of this file by tests/test_a22_format_pin.py (numpy float64 restatement of the spec text and torch's
 * float8_e8m0fnu bit layout). */
enum { ORC_COMP_MXFP4 = 5 };
#define ORC_MXFP4_REC_BYTES 1088u
uint8_t orc_f32_to_e2m1(float v);            /* nearest even, saturating, NaN -> 0 */
float   orc_e2m1_to_f32(uint8_t nibble);
uint8_t orc_mx_scale_code(float amax, int emax_elem);     /* E8M0 code of a block whose finite max|x| is amax */
float   orc_e8m0_to_f32(uint8_t code);                    /* 2^(code-127) (a subnormal float for code 0), NaN for 255 */
/* MXFP8 rows (the query operand of the block-scaled matrix instruction): per row, d/block blocks of `block` (<= 64) e4m3 codes
 * with one E8M0 code each (emax_elem = 8; elements = e4m3 of x / 2^(code-127), nearest even, saturating at 448).  The attention
 * kernel uses block = 16: its query operand is interleaved with zeros to meet the position-interleaved K bytes, so one
 * hardware scale block of 32 k covers 16 channels. */
void orc_quantize_rows_mxfp8(const uint16_t* q16, size_t rows, size_t d, size_t block, uint8_t* q8, uint8_t* q_codes);
/* Decode attention of one kv head over MXFP4 page rows (own extension, parity unpinned):
 *   k_rows / v_rows   : n_pos/2 page rows of d bytes (byte i = channel i of position 2r low, 2r+1 high)
 *   k_codes / v_codes : n_pos/2 rows of d/16 E8M0 codes (code j: channels 16j .. 16j+15 of both positions)
 *   q8 / q_codes      : g MXFP8 rows with blocks of q_block (orc_quantize_rows_mxfp8)
 *   s[m][t]   = (sum_d q[m][d] * k[t][d]) * sm_scale          with q, k the dequantised values
 *   out[m][:] = sum_t softmax_t(s[m][:])[t] * v[t][:]          (double precision throughout); lse, mag as in orc_attend_fp8
 * n_pos may be odd (the last page's second position is then not attended). */
void orc_attend_mx4(const uint8_t* q8, const uint8_t* q_codes, size_t q_block, size_t g,
                    const uint8_t* k_rows, const uint8_t* k_codes, const uint8_t* v_rows, const uint8_t* v_codes,
                    size_t n_pos, size_t d, float sm_scale, float* out, float* lse, float* mag);

/* cache_engine.cpp:25-33,142-148 */
double   orc_layer_compression_ratio(uint32_t layer_id);
/* cache_engine.cpp:286-296 */
double   orc_codec_throughput_gbps(size_t num_engines, double mhz, size_t width_bits);
size_t   orc_codec_pipeline_latency_cycles(void);

/* ATU / TLB (cache_engine.cpp:118-140) */
typedef struct orc_tlb orc_tlb_t;
orc_tlb_t* orc_tlb_new(size_t entries);
void       orc_tlb_delete(orc_tlb_t*);
uint64_t   orc_tlb_translate(orc_tlb_t*, uint64_t va, int* was_hit);

/* ------------------------------------------------------------------ */
/* 3-tier memory manager (src/cxl_memory/cxl_memory_manager.cpp)       */
/* ------------------------------------------------------------------ */
enum { ORC_TIER_L1 = 0, ORC_TIER_L2 = 1, ORC_TIER_L3 = 2 };
enum { ORC_STATE_INVALID = 0, ORC_STATE_SHARED = 1, ORC_STATE_EXCLUSIVE = 2,
       ORC_STATE_MODIFIED = 3 };

typedef struct orc_mm orc_mm_t;
typedef struct {
    uint64_t l1_hits, l1_misses, l2_hits, l2_misses, l3_accesses;
    uint64_t migrations_l1_to_l3, migrations_l3_to_l1;
    double   l1_hit_rate, l2_hit_rate;
} orc_mm_stats_t;

orc_mm_t* orc_mm_new(uint64_t l1_gb, uint64_t l2_gb, uint64_t l3_gb, uint64_t page_size);
void      orc_mm_delete(orc_mm_t*);
uint64_t  orc_mm_allocate(orc_mm_t*, uint64_t size_bytes, uint32_t layer_id, int tier);
void      orc_mm_deallocate(orc_mm_t*, uint64_t va);
uint64_t  orc_mm_translate(orc_mm_t*, uint64_t va);
int       orc_mm_is_in_cache(orc_mm_t*, uint64_t va, int tier);
int       orc_mm_promote_to_l1(orc_mm_t*, uint64_t va);
int       orc_mm_demote_to_l3(orc_mm_t*, uint64_t va);
void      orc_mm_invalidate_page(orc_mm_t*, uint64_t va);
void      orc_mm_mark_modified(orc_mm_t*, uint64_t va);
int       orc_mm_get_page_state(orc_mm_t*, uint64_t va);
void      orc_mm_update_access_tracking(orc_mm_t*, uint64_t va);
int       orc_mm_is_hot_page(orc_mm_t*, uint64_t va);
void      orc_mm_get_statistics(orc_mm_t*, orc_mm_stats_t*);
/* memory_allocator.cpp:105-143 access policy on top of the manager */
uint64_t  orc_mm_cxl_access(orc_mm_t*, uint64_t base_va, uint64_t offset);

/* ------------------------------------------------------------------ */
/* Speculative prefetcher (src/prefetcher/speculative_prefetcher.cpp)  */
/* ------------------------------------------------------------------ */
/* speculative_prefetcher.cpp:153-160 */
uint64_t orc_compute_kv_address(uint32_t req_id, uint32_t layer_id, uint32_t position);
/* speculative_prefetcher.cpp:25-82 : address list of one prefetch() call.
 * mm may be NULL (nothing resident).  Returns number of requests emitted. */
size_t   orc_prefetch_legacy(orc_mm_t* mm, uint32_t layer_id, size_t depth,
                             size_t n_predictions, uint64_t* out_va);

/* adaptive depth + misprediction stats (speculative_prefetcher.cpp:84-137) */
typedef struct orc_adapt orc_adapt_t;
orc_adapt_t* orc_adapt_new(size_t initial_depth);
void         orc_adapt_delete(orc_adapt_t*);
void         orc_adapt_update(orc_adapt_t*, int was_correct);
size_t       orc_adapt_depth(const orc_adapt_t*);
/* returns 1 when actual is NOT among predicted (a misprediction) */
int          orc_is_misprediction(uint32_t actual, const uint32_t* predicted, size_t n);

/* Token predictor (src/prefetcher/lstm_predictor.cpp:40-188), SURVEY 8f row N1.
 * The reference's "LSTM" is degenerate (gates fixed at 0.5, recurrent weights
 * unused, candidate g = sum_j 0.1*embedding[j]); this restates exactly that maths,
 * sequentially, in the reference's operation order.  emb: [vocab][emb_dim],
 * wout: [vocab][hidden].  history is padded with zeros at the FRONT / cut to its
 * last hist_len tokens (lstm_predictor.cpp:44-51).  Returns min(k, vocab) pairs,
 * highest probability first (ties: lower token id first; the reference's
 * std::sort leaves tie order unspecified). */
size_t orc_lstm_predict(const float* emb, const float* wout, size_t vocab, size_t emb_dim,
                        size_t hidden, size_t layers, size_t hist_len,
                        const uint32_t* history, size_t n_hist, size_t k,
                        uint32_t* out_tok, float* out_prob);
/* weights as the reference's constructor draws them (lstm_predictor.cpp:27-35):
 * srand(seed) then rand() in the order embedding, lstm (discarded here), output.
 * With seed 1 (glibc's default state) this reproduces the first predictor
 * constructed in a fresh process. */
void orc_lstm_reference_weights(unsigned seed, size_t vocab, size_t emb_dim, size_t hidden,
                                size_t layers, float* emb, float* wout);

/* RTL intent (hardware/rtl/prefetch_core.v:92-98,158): the virtual address
 * the prefetch FSM asks the ATU for at iteration idx. 64-bit truncation of
 * {req[31:0], layer[15:0], 8'd0, pos[31:0], 1'b0}. */
uint64_t orc_rtl_prefetch_vaddr(uint32_t req_id, uint16_t layer, uint32_t pos_plus);

/* Engine lookup (prefetch_core.v:150-241 intent mapped through the shim
 * layout vllm_speckv_backend.py:87-100): pages of (req, layer, kind in {K,V})
 * covering positions cur_pos+1 .. cur_pos+k (clipped to num_tokens-1), in
 * order kind-major, ascending page, duplicates removed.  `flags` (may be
 * NULL) is the per-page residency array of the allocation; pages with
 * flags&3 are filtered out (speckv_allocator.cpp:105-113).  Returns count. */
size_t orc_prefetch_pages(uint32_t req_id, uint32_t layer, uint32_t cur_pos,
                          uint32_t depth_k, uint64_t num_layers,
                          uint64_t num_tokens, uint64_t num_heads,
                          uint64_t head_dim, uint64_t bytes_per_element,
                          uint64_t alloc_pages, const uint32_t* flags,
                          uint64_t* out_pages, size_t cap);

/* ---- coherence shadow directory -----------------------------
 * CoherenceManager of the reference (src/cxl_memory/coherence_manager.cpp), restated: a per-line
 * MESI tag + tier tag + counters; every "FPGA operation" is a stub that succeeds iff the manager
 * has a driver (coherence_manager.cpp:398-424) and counts as a directory HIT of its kind
 * (:436-458), which is why one missing read adds 2 to total_reads.  States: 0 I, 1 S, 2 E, 3 M;
 * tiers: 0 L1_GPU, 1 L2_PREFETCH, 2 L3_CXL.  stats[7] = total_reads, total_writes, coherence_ops,
 * invalidations_sent, writebacks_performed, directory_hits, directory_misses. */
typedef struct orc_coh orc_coh_t;
orc_coh_t* orc_coh_new(size_t cache_line_size, int has_driver);
void orc_coh_delete(orc_coh_t* c);
int orc_coh_request_read(orc_coh_t* c, uint64_t addr);
int orc_coh_request_write(orc_coh_t* c, uint64_t addr);
int orc_coh_invalidate(orc_coh_t* c, uint64_t addr);
int orc_coh_writeback(orc_coh_t* c, uint64_t addr);
int orc_coh_flush_all(orc_coh_t* c);
int orc_coh_get_state(const orc_coh_t* c, uint64_t addr);
int orc_coh_get_tier(const orc_coh_t* c, uint64_t addr);
int orc_coh_promote_to_l1(orc_coh_t* c, uint64_t addr);
int orc_coh_demote_to_l3(orc_coh_t* c, uint64_t addr);
void orc_coh_update_tier(orc_coh_t* c, uint64_t addr, int tier);
int orc_coh_batch_invalidate(orc_coh_t* c, const uint64_t* addrs, size_t n);
void orc_coh_get_statistics(const orc_coh_t* c, uint64_t* stats7);
void orc_coh_reset_statistics(orc_coh_t* c);

#ifdef __cplusplus
}
#endif
#endif /* SPECKV_ORACLE_H */
