// cxl-speckv_amd/csrc/kernels.hpp -- launch interface of the CDNA4 kernels.
//
// One KV block = one 4 KiB page = 2048 fp16 elements.  One 64-lane wavefront
// owns one block at a time (4 wavefronts per workgroup, no workgroup barrier):
// lane l holds elements [8l + 512j, 8l + 512j + 8) for j = 0..3, so every
// global load/store instruction is one fully coalesced 1 KiB access.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace speckv {

constexpr uint32_t kPageSize = 4096;
constexpr uint32_t kBlockElems = 2048;

enum Scheme : int { kFp16 = 0, kInt8 = 1, kInt8DeltaRle = 2, kInt4G32 = 3, kFp8E4m3 = 4, kMxFp4 = 5 };
constexpr uint32_t kInt4RecBytes = 128 + 1024;   // 64 fp16 group scales + 2048 nibbles
constexpr uint32_t kMx4RecBytes = 1024 + 64;     // 2048 E2M1 nibbles + 64 E8M0 group scales (OCP MX v1.0 blocks of 32)
// MXFP4 records in the POOL are tile-planar (round 6).  A record is 8.5 cache lines: at a stride of 1088 every other record starts
// in the middle of a line (a head pair's 256 bytes then span three lines: 3.38 GB fetched for 2.85 GB of records, profiles/r05_mx4.txt),
// and in line-aligned 1152-byte slots (round 5) the ninth line of every slot was fetched for half its bytes: 1.065 x the record bytes
// from DRAM and 5.9 % of the pool's capacity (profiles/r05c_mx4_mem_pmc.json).  So 16 records of a run form a TILE of exactly
// 136 lines: the 16 nibble rows (16 x 1024 B) followed by the 16 code rows (16 x 64 B, two records per line).  Record r of a
// run (runs start on a tile boundary) has its nibbles at mx4_nib_off(r) and its codes mx4_code_delta(r) bytes behind them; the
// record BYTES are the oracle's, only their placement differs.  The raw operators (speckv_ext_codec_*) keep contiguous records.
constexpr uint32_t kMx4TileRecs = 16;
constexpr uint32_t kMx4CodePlane = kMx4TileRecs * 1024u;                 // offset of the code rows inside a tile
constexpr uint32_t kMx4TileBytes = kMx4TileRecs * kMx4RecBytes;          // 17 408 B = 136 lines
__host__ __device__ inline uint64_t mx4_nib_off(uint64_t r) { return (r >> 4) * kMx4TileBytes + (r & 15u) * 1024u; }
__host__ __device__ inline uint32_t mx4_code_delta(uint64_t r) { return kMx4CodePlane - 960u * static_cast<uint32_t>(r & 15u); }
__host__ __device__ inline uint64_t mx4_run_bytes(uint64_t recs) { return (recs + 15u) / 16u * kMx4TileBytes; }
enum QuantMode : int { kRefExact = 0, kIntent = 1 };

// Device-resident page-table entry (16 B).
struct PageEntry {
    uint64_t pool_addr;   // device address of the stored record (local or peer HBM)
    uint32_t rec_bytes;   // record length in bytes (0 = never written)
    float    scale;       // per-block scale factor; MXFP4 (no block scale): the BITS are mx4_code_delta of the record, set when
                          // the entry is pointed at its place (k_init_entries / k_retarget_entries / the host) and kept by k_compress
};

// Shim layout (vllm_speckv_backend.py:87-100) of the allocation a lookup runs on.
struct Layout {
    uint32_t num_tokens, num_layers, num_heads, head_dim, bytes_per_element;
    uint64_t alloc_pages;
};

// One row of the device allocation table: what kernels that work across allocations (the prefetch pipeline, the
// ring bookkeeping of the fetch kernel) need to know about an allocation.  A freed row has entries == nullptr.
//   d_flags / d_slot : residency in HBM (bit0 L1, bit1 L2; cache slot) -- what kernels read and keep
//   h_slot           : one word per page in pinned host memory (device-visible) -- what the host reads: the ring
//                      sequence number under which a kernel fetched the page (its only host-visible store; the host
//                      derives the L2 bit from it, Engine::l2_live), or the L1 slot of a host-promoted page
//   stamp            : per-page scratch word of the flush's first-occurrence dedupe
struct DevAlloc {
    PageEntry* entries;
    uint32_t*  d_flags;
    uint32_t*  d_slot;
    uint32_t*  h_slot;
    uint32_t*  stamp;
    Layout     layout;
};
constexpr uint32_t kNoSlot = 0xFFFFFFFFu;
constexpr uint64_t kNoOwner = ~0ull;

// One allocation's share of a grouped compress launch (CodecArgs::groups): pages first + j*page_step, j < group_n, from
// data + j*data_stride.
struct CompressGroup {
    PageEntry*     entries;
    float*         scale_tab;
    uint32_t       region_pages;
    uint32_t       scale_run;         // CodecArgs::scale_run of this allocation
    uint64_t       first;
    const uint8_t* data;
};

// Source / destination description of one codec launch.  Exactly one of
// {entries, recs, tab+alloc_list} is used as the record source.
struct CodecArgs {
    // record side
    PageEntry*      entries;      // page table (engine form), indexed by page
    uint8_t*        recs;         // raw form: record i at recs + i*rec_stride
    uint64_t        rec_stride;
    uint32_t*       rec_bytes;    // raw form
    float*          scales;       // raw form
    // block index mapping: page = page_list ? page_list[i] : first + i  (compress only: first + i*page_step when page_step != 0)
    const uint32_t* page_list;
    uint64_t        first;
    uint64_t        page_step;
    // fp16 / fp32 side: block i at data + i*data_stride_bytes, or data_list[i]
    uint8_t*        data;
    uint64_t        data_stride;
    const uint64_t* data_list;
    // count
    uint64_t        n;
    const uint32_t* n_dev;        // optional: n read from device memory (<= n)
    uint64_t        per_wave;     // set by the launcher: blocks per wave (0 = 1) ...
    uint64_t        wave_step;    // ... consecutive (0) or this many blocks apart (one per round of the grid)
    // residency mirror update on completion (decompress only): flags[page] |= set_flags (atomic)
    uint32_t*       flags;
    uint32_t        set_flags;
    int             scheme;
    int             quant_mode;
    int             out_f32;
    // records were written by k_compress (the engine's pool): INT8_DELTA_RLE streams are then known to
    // have no zero counts and a zero-padded tail, which the decoder need not re-check per pair
    int             trusted;
    // decompress, INT8_DELTA_RLE: the caller knows the records to be short (structured data: mean record well under 512 B) --
    // the launch takes the instantiation with the flat-run fast path (decode_rle_fast<.., FLAT>); results are identical
    int             structured_hint;
    // compress only: when set, the block scale of page p is also stored at scale_tab[tile order of p] -- the
    // per-tile order the fused attention reads with one 16-byte load (attend.hip); region_pages = pages of one
    // (layer, kind) region of the shim layout, a multiple of 16
    float*          scale_tab;
    uint32_t        region_pages;
    // ... and, for an allocation striped regularly over D >= 2 runs (round 6), ALSO in run order -- the scale of page p at
    // scale_tab[-D * cap + (p % D) * cap + p / D], D = scale_run & 15, cap = scale_run >> 4 (0: no such table) -- where the
    // residue-class forms of the FP8 attention find the 16 scales of a class tile in one line (scale_run_index)
    uint32_t        scale_run;
    // compress only, INT8_DELTA_RLE: every 1024th page leaves its record length in len_samples[(page >> 10) & 15] -- 16 words of
    // the allocation's host-visible memory, plain stores -- so that the host can tell, without a copy back, whether the
    // allocation holds data that compresses (mean record well under 512 B: the flat-run decoder is then chosen for its reads;
    // CodecArgs::structured_hint).  A hint only: either decoder gives the same bytes.  Null: no samples.
    uint32_t*       len_samples;
    // compress only: blocks of several allocations in one launch -- block i belongs to groups[i / group_n] (device array)
    // as its page i % group_n; entries / scale_tab / region_pages / first / data above are then unused
    const CompressGroup* groups;
    uint64_t        group_n;
    // blocks of several allocations in one launch (decompress only): block i belongs to row alloc_list[i] of tab
    const DevAlloc* tab;
    const uint32_t* alloc_list;
    // L2-ring bookkeeping done by the fetch kernel itself (decompress only, ring_owner != nullptr): block i lands in
    // cache slot slot0 + i (dst = ring_base + slot*4096); the wave evicts the slot's previous owner (clears its L2 bit
    // if it still points at this slot), records the new owner, the page's slot and its L2 bit in HBM and the slot's ring
    // sequence number in the page's host-visible word.  slot0 / seq0 come from the *_dev pointers when set, else from
    // the values.  (Synchronous misses; the device-side flush does this in its scatter kernel, one thread per page.)
    uint64_t*       ring_owner;
    uint8_t*        ring_base;
    const uint32_t* slot0_dev;
    uint32_t        slot0;
    const uint32_t* seq0_dev;     // ring sequence number of slot0 (block i: seq0 + i), stored as the page's host-visible word
    // device-side flush (list form): block i's host-visible word *host_words[i] = *seq0_dev + i is stored by the fetch
    // itself, spread over the launch, instead of by the scatter kernel in front of it (FlushArgs::final_host)
    uint32_t* const* host_words;
    uint32_t        seq0;
    uint32_t        alloc_idx;    // row of tab for every block when alloc_list == nullptr
    uint32_t*       hand_ptr;     // optional: block 0 stores new_hand = the ring sequence number after this take (host-initiated takes keep the device's current)
    uint32_t        new_hand;
    // copy-engine fetch: the records of pool (page % stripe_n) were copied into local staging, to their pool address
    // + stripe_delta[page % stripe_n]
    uint32_t        stripe_n;
    uint64_t        stripe_magic; // floor(2^35 / stripe_n) + 1: (page * magic) >> 35 == page / stripe_n for page < 2^28
    int64_t         stripe_delta[8];
    // ring form only, launches of a few waves (speckv_access on a miss): when done_flag is set, every wave ends with a
    // system-scope fence and counts itself in *done_count (device memory, zero before and after the launch); the last one
    // stores done_token to *done_flag (pinned host memory) -- the host spins on that word instead of going through the runtime's
    // completion path (profiles/tools/probe/sync_latency.hip: 7 us against 11 us for launch + wait)
    uint32_t*       done_flag;
    uint32_t*       done_count;
    uint32_t        done_token;
};

hipError_t launch_compress(const CodecArgs& a, hipStream_t s);
hipError_t launch_decompress(const CodecArgs& a, hipStream_t s);

// FPGACacheEngine::compress / ::decompress over a tensor of any length (tensor_codec.hip): one scale, one delta chain and
// one run-length stream across the whole tensor.  d_rle: 16-byte aligned, room for 2n bytes rounded up to 16; d_ws:
// 256-byte aligned scratch of tensor_*_workspace_bytes; d_n_out may be null.
size_t tensor_compress_workspace_bytes(uint64_t n);
size_t tensor_decompress_workspace_bytes(uint64_t rle_bytes);
hipError_t launch_tensor_compress(const void* d_src, uint64_t n, bool src_f32, uint8_t* d_rle, uint64_t* d_rle_bytes, float* d_scale,
                                  void* d_ws, size_t ws_bytes, int quant_mode, hipStream_t s);
hipError_t launch_tensor_decompress(const uint8_t* d_rle, uint64_t rle_bytes, float scale, void* d_dst, uint64_t dst_cap, bool out_f32,
                                    uint64_t* d_n_out, void* d_ws, size_t ws_bytes, int quant_mode, hipStream_t s);

// The same over MANY tensors in one launch each way (tensor_codec.hip: k_tcm_fused / k_tdm_fused -- several workgroups per tensor,
// tensor-local look-back, max|x| by rendezvous; tensors of at most 16 tiles: k_tcb_fused / k_tdb_fused, one workgroup each).
// TensorDesc = speckv_ext_tensor_t: {data (compress: the source; decompress: the destination), n (elements; decompress: room),
// rle (the stream, 16-byte aligned), rle_cap}.  max_elems: the host's bound on n (sizes the grid); d_ws: 256-byte aligned,
// tensors_workspace_bytes(n_tensors, max_elems) bytes.
struct TensorDesc { void* data; uint64_t n; uint8_t* rle; uint64_t rle_cap; };
size_t tensors_workspace_bytes(uint32_t n_tensors, uint64_t max_elems);
hipError_t launch_tensors_compress(uint32_t n_tensors, const TensorDesc* d_desc, uint64_t max_elems, bool src_f32, uint64_t* d_rle_bytes, float* d_scales,
                                   void* d_ws, size_t ws_bytes, int quant_mode, hipStream_t s);
hipError_t launch_tensors_decompress(uint32_t n_tensors, const TensorDesc* d_desc, uint64_t max_elems, const uint64_t* d_rle_bytes, const float* d_scales, bool out_f32,
                                     uint64_t* d_n_out, void* d_ws, size_t ws_bytes, int quant_mode, hipStream_t s);

// Prefetch lookup: 3 kernels (mask+count, scan, scatter) -> compacted page list
// in request order.  scratch must hold (2*n + 2) uint32.
hipError_t launch_prefetch_lookup(const Layout& lay, uint32_t n,
                                  const uint32_t* d_req, const uint32_t* d_layer,
                                  const uint32_t* d_pos, const uint32_t* d_k,
                                  const uint32_t* d_flags, uint32_t* d_out, uint32_t cap,
                                  uint32_t* d_count, uint32_t* d_scratch, hipStream_t s);

hipError_t launch_verify(uint32_t n, uint32_t k, const int32_t* d_actual,
                         const int32_t* d_predicted, uint8_t* d_hit,
                         uint32_t* d_hit_count, hipStream_t s);

// Host-originated residency changes applied to the device mirrors: for each update,
// d_flags[page] = (d_flags[page] & and_mask) | or_mask and, unless slot == kKeepSlot, d_slot[page] = slot.
struct MirrorUpdate { uint32_t alloc_idx, page, and_mask, or_mask, slot, pad; };
constexpr uint32_t kKeepSlot = 0xFFFFFFFEu;
hipError_t launch_apply_updates(const DevAlloc* d_tab, const MirrorUpdate* d_updates, uint32_t n, hipStream_t s);

// Device-side prefetch flush (prefetch_core.v:150-241: predict -> encode -> translate -> directory check -> issue,
// without the host).  n requests (SoA: local request index inside the allocation, layer, position, depth, table row)
// -> candidate pages in request order (fixed stride of 32*W words per request), first-occurrence dedupe through the
// per-page stamps, ordered compaction, ring-slot assignment, and the final (page, row) lists a fetch launch consumes.
struct FlushResult { uint32_t m, base, total, seq; };      // pages taken, first ring slot, distinct candidates found, ring sequence number of the first slot
struct FlushArgs {
    const DevAlloc* tab;
    uint32_t        n;            // requests
    uint32_t        W;            // candidate words per lane (max new pages of one position + 1)
    const uint32_t* req;          // [n] request index inside its allocation
    const uint32_t* layer;
    const uint32_t* pos;
    const uint32_t* depth;
    const uint32_t* row;          // [n] table row (kNoSlot = unaddressable request: contributes nothing)
    uint32_t        epoch;        // 1..255, stamps of older epochs are ignored
    uint32_t*       cand;         // [n*32*W]
    uint32_t*       wave_tot;     // [n*W/2 rounded up] + same again for the bases
    // what the fetch launch reads, in list order (block i): the page's record descriptor and its ring slot's address
    PageEntry*      final_entry;  // [max_take]
    uint64_t*       final_dst;    // [max_take]
    uint64_t*       ring_owner;   // L2 ring bookkeeping, done by the scatter kernel with one THREAD per page (see there)
    uint8_t*        ring_base;
    uint32_t        max_take;     // never more than this many pages per flush
    uint32_t        n_l2;         // ring size
    uint32_t*       hand;         // device ring hand
    FlushResult*    result_dev;   // m is read by the fetch launch (n_dev)
    FlushResult*    result_host;  // the same, stored to pinned host memory
    // null: the scatter kernel stores every page's host-visible word itself (one PCIe transaction per page, all of them in
    // front of the fetch); else it lists the words' addresses and the fetch launch stores them (CodecArgs::host_words)
    uint32_t**      final_host;   // [max_take]
};
hipError_t launch_flush_pipeline(const FlushArgs& a, hipStream_t s);
// bytes rounded up to 16: both buffers must be 16-byte aligned and padded to a multiple of 16
hipError_t launch_copy16(const void* src_host_mapped, void* dst, size_t bytes, hipStream_t s);

// entries[first+i].pool_addr = base + i*stride ; rec_bytes = 0 ; scale = 1.  stride == kPlanarMx4: entry i = record rec0 + i of
// a tile-planar MXFP4 run at base (pool_addr = base + mx4_nib_off, scale bits = mx4_code_delta)
constexpr uint64_t kPlanarMx4 = 0;
hipError_t launch_init_entries(PageEntry* d_entries, uint64_t n, uint64_t base,
                               uint64_t stride, hipStream_t s, uint64_t rec0 = 0);

// Fused dequant-matvec of BASELINE config 5: q.K^T scores straight from FP8_E4M3
// records with v_mfma_f32_16x16x32_fp8_fp8 (no fp16 K is ever materialised).
// Layout requirement: one K row (all heads of one position) = 2048 B, i.e. two
// positions per 4 KiB page (H*D = 1024 elements, e.g. 8 kv heads x 128).
//   d_q8     [H][16][128] e4m3 query rows (rows >= g are zero), d_qs [H][16] their scales
//   d_out    [H][g][n_pos] fp32, n_pos = 2 * n_pages
hipError_t launch_quantize_q_e4m3(const void* d_q_f16, uint32_t heads, uint32_t g, uint32_t d,
                                  uint8_t* d_q8, float* d_qs, hipStream_t s);
hipError_t launch_qk_scores_fp8(const PageEntry* d_entries, uint64_t first_page, uint64_t layer_page_stride,
                                uint32_t n_layers, uint32_t n_pages, uint32_t heads, uint32_t g,
                                const uint8_t* d_q8, const float* d_qs, float* d_out, hipStream_t s);

// Batch form of the linear fused attention: many sequences (allocations), one layer each, one launch.  The fields
// override their AttendArgs namesakes per sequence; part_base = index of the sequence's first (head, split) partial.
struct AttendSeq {
    const uint8_t* lin_base;          // table launches (AttendArgs::table_form: a batch member without a regular placement makes the whole
                                      // launch read its addresses from the page tables): the sequence's PageEntry array instead
    const float* scale_tab;
    uint64_t k_first, v_first;        // first K / V page of the wanted layer (position 0)
    uint32_t n_pages;                 // pages of [0, pos_end)
    uint32_t n_splits;                // ceil(tiles / tiles_per_split), <= gridDim.x
    uint32_t part_base;
    uint32_t tiles_per_split;         // this sequence's own split length (its tiles divided evenly over n_splits)
    const uint64_t* stripe_bases;     // striped launches (AttendArgs::stripe_bases set): the sequence's own run bases ...
    uint32_t layer_pages;             // pages of one layer (K + V): layer l of a planned batch starts at k_first + l * layer_pages
    uint32_t stripe_n;                // ... and pool count (1 = a single run: lin_base is bases[0])
};                                    // (64 bytes: speckv_ext_attend_plan_bytes)

// Decode attention straight from FP8_E4M3 records (attend.hip): softmax(q.K^T * sm_scale) . V per kv head,
// split over the positions, partials merged by a second kernel.  Same layout requirement as the scores.
struct AttendArgs {
    const PageEntry* entries;
    uint64_t k_first, v_first;        // first K / V page of layer_begin at pos_begin
    uint64_t layer_stride;            // pages per layer (K + V)
    uint32_t n_pages;                 // pages of the position range
    uint32_t skip_pages;              // FP8 tile forms: the first skip_pages pages of the range are masked (a range that starts inside a 32-position tile)
    uint32_t heads, g;
    uint32_t n_splits, tiles_per_split;   // tile = 16 pages = 32 positions
    const uint8_t* q8;                // [layers][heads][16][128] e4m3
    const float* qs;                  // [layers][heads][16]
    float scale_log2e;                // sm_scale * log2(e)
    const uint8_t* zero_page;         // 4 KiB of zeros: stands in for pages never written (general form)
    const uint8_t* lin_base;          // non-null: record p of the allocation sits at lin_base + p*2048 and
                                      // never-written records are zero bytes (linear, pipelined form)
    const float* scale_tab;           // linear form: page scales of the whole allocation in tile order (attend.hip)
    const uint16_t* q16;              // linear form: the fp16 query rows [layers][heads][g][128] (quantised in the kernel)
    const struct AttendSeq* seqs;     // batch form: one descriptor per sequence (blockIdx.y / (heads/4)), else null
    const uint32_t* order;            // batch form, sequences of different lengths (round 6): order[i] = the sequence whose workgroups are dispatched i-th (the
                                      // engine orders them by length so that the CUs' shares even out: attend_dispatch_order), or null = as given
    float* part_acc;                  // [layers][heads][splits][16][128]
    float* part_ml;                   // [layers][heads][splits][2][16]
    // every row has exactly ONE split (set by the engine then): the attention kernel normalises and writes the final
    // [row][g][128] output and the log-sum-exp (may be null) itself, and no merge launch follows
    float* direct_out;
    float* direct_lse;
    // planned batches (descriptors resident on the device, one set for all layers of a decode step): the layer of this
    // launch, and "decide per sequence": 1 = a sequence with one split is written directly, the others through the merge
    // launch that follows; 2 = the launch geometry allows one split at most, NO merge follows: sequences without
    // positions are zeroed by the attention kernel itself (attend_zero_rows)
    uint32_t batch_layer;
    uint32_t direct_per_seq;
    // ... and SEVERAL layers of the planned batch in one launch (speckv_ext_attend_planned_layers: callers that have the query rows of
    // several layers at once): batch_n_seq != 0 -> grid y = layer x sequence, row block y of q / out / lse = [layer][sequence], layer
    // batch_layer + y / batch_n_seq.  Only launches whose geometry has one split per sequence (direct_per_seq == 2: no merge behind).
    uint32_t batch_n_seq;
    // striped form (lin_base == nullptr, stripe_bases != nullptr): the allocation still has the regular placement over
    // 2..8 pools -- the record of page p is stripe_bases[p % stripe_n] + (p / stripe_n) * record stride (placement.hpp),
    // never-written records zero bytes -- so every address is arithmetic here too; stripe_magic = floor(2^32 / n) + 1:
    // __umulhi(p, magic) == p / n for every p < 2^28.  The 8 run bases live in a small device array of the allocation.
    // Batch form: non-null marks a striped launch, every AttendSeq then carries its own stripe_bases / stripe_n.
    const uint64_t* stripe_bases;
    uint32_t stripe_n, stripe_magic;
    // rows-first launches: grid (rows, splits) instead of (splits, rows), so that the workgroups are dispatched split by
    // split -- split 0 of every row first.  With a long split 0 and a short split 1 per row the long pieces all start at
    // once and the short ones fill the remaining workgroup slots in turns (engine_attend.cpp: int4_unequal_split)
    uint32_t rows_first;
    uint32_t rows_real;               // rows-first batch launches of the FP8 / MXFP4 kernels: grid x is padded to an ODD number of rows, so that the pieces of one member go round
                                      // the XCDs (workgroup id % 8) instead of all landing on one; rows >= rows_real are padding and return at once
    // table form of the fast kernels (single-sequence form; lin_base and stripe_bases null): an allocation whose placement
    // is no longer regular (pages migrated one by one) but whose range is tile-aligned -- every record address comes
    // from the page-table entry, looked up one tile ahead of its request; never-written pages read zero_page
    uint32_t table_form;
    // INT4, linear form, 8 kv heads: the whole-record kernel (k_attend_int4_wg8: 8 waves = 8 heads, grid (splits, layers))
    uint32_t mx4_halves;              // MXFP4, linear batch launches of single-split sequences: 8-wave workgroups, the run cut in two (k_attend_mx4<0, 2>)
    uint32_t wg8;                     // 1: the engine's choice of form; 2: workgroups of one run (8 waves) also for batches
    // ... and its STREAM form for many layers of one sequence (see k_attend_int4_wg8): n_wgs != 0 turns it on.  The rows'
    // partials then sit at (row * max_slots + slot) and the merge takes each row's count from attend_stream_count().
    uint32_t scale_run = 0;          // FP8 class forms: the allocation's run-order scale table (CodecArgs::scale_run), 0 = none (gather from the page-order table)
    uint32_t fp8_cls = 0;            // FP8 over a regularly striped pool: k_attend_fp8_dma<2>, pages by residue class (n_splits / tiles_per_split count class-major tiles)
    struct Stream { uint32_t len, rem, n_wgs, max_slots, tiles; } stream;      // tiles: per layer, 0 = ceil(n_pages / 16) (the class form of INT4_G32 over a striped pool counts by residue class)
    // planned batches, MXFP4: the position a sequence still keeps OUTSIDE the pool (the connector's odd last position, fp16 rows
    // [tail][layers][heads][128], tail_stride elements apart) is folded in by the attention kernel itself -- by the workgroup of
    // split 0, into its partial or final state, in the epilogue -- instead of by a launch of its own behind every layer's attention.
    // tail_idx[sequence] = index of its tail rows, < 0: none; null with tail_k set: sequence i has tail i.  Every sequence of such a
    // launch has at least one split (the engine sends batches with an empty member through k_attend_fold_tail as before).
    const uint16_t* tail_k;
    const uint16_t* tail_v;
    const int32_t* tail_idx;
    uint64_t tail_stride;
};
// The stream partition: `total` tiles in layer-major order cut into n_wgs contiguous pieces, the first `rem` one longer
// (len = total / n_wgs >= 1, rem = total % n_wgs).  begin(w) = first tile of piece w; wg_of(G) = the piece tile G is in.
__host__ __device__ inline uint64_t attend_stream_begin(uint32_t w, uint32_t len, uint32_t rem)
{
    return static_cast<uint64_t>(w) * len + (w < rem ? w : rem);
}
__host__ __device__ inline uint32_t attend_stream_wg_of(uint64_t G, uint32_t len, uint32_t rem)
{
    const uint64_t cut = static_cast<uint64_t>(rem) * (len + 1u);
    return G < cut ? static_cast<uint32_t>(G / (len + 1u)) : rem + static_cast<uint32_t>((G - cut) / len);
}
// partials of layer `layer` (n_tiles tiles per layer): pieces wg_of(first tile) .. wg_of(last tile)
__host__ __device__ inline uint32_t attend_stream_count(uint32_t layer, uint32_t n_tiles, uint32_t len, uint32_t rem)
{
    const uint64_t g0 = static_cast<uint64_t>(layer) * n_tiles;
    return attend_stream_wg_of(g0 + n_tiles - 1u, len, rem) - attend_stream_wg_of(g0, len, rem) + 1u;
}
// MXFP4 over a striped pool (k_attend_mx4 form 1): the positions of the range are taken CLASS by class -- class c = the pages j
// of the range with j % stripe_n == c, which are consecutive records of ONE run for K and of one run for V -- each class in
// tiles of 16 pages; every class gets the tile count of the largest (trailing tiles of the others are masked).
__host__ __device__ inline uint32_t mx4_class_tiles(uint32_t n_pages, uint32_t stripe_n)
{
    return ((n_pages + stripe_n - 1u) / stripe_n + 15u) / 16u;
}
// tiles of a range in that form (stripe_n = 1, a single run: the plain count)
__host__ __device__ inline uint32_t mx4_striped_tiles(uint32_t n_pages, uint32_t stripe_n)
{
    const uint32_t n = stripe_n ? stripe_n : 1u;
    return n_pages ? n * mx4_class_tiles(n_pages, n) : 0u;
}
#if defined(__HIPCC__)
// record address of page p in the striped form; `bases` = the allocation's run bases copied to LDS
__device__ __forceinline__ const uint8_t* attend_stripe_rec(const uint64_t* bases, uint32_t p, uint32_t n, uint32_t magic, uint32_t stride)
{
    const uint32_t q = n == 1u ? p : __umulhi(p, magic);              // (n is uniform: batches may mix single-run and striped sequences)
    return reinterpret_cast<const uint8_t*>(bases[p - q * n]) + static_cast<uint64_t>(q) * stride;
}
#endif
#if defined(__HIPCC__)
// rows of a sequence without positions in a batch launch that has no merge behind it (one wave per kv head).
// (Plain values, not the AttendArgs: a reference to the kernel's argument block makes the compiler keep it in memory,
// and the LDS-DMA kernels need its pointers in scalar registers.)
__device__ __forceinline__ void attend_zero_rows(float* direct_out, float* direct_lse, uint32_t g, uint64_t row, uint32_t lane)
{
    float* dst = direct_out + row * g * 128u;
    for (uint32_t i = lane; i < g * 128u; i += 64u) dst[i] = 0.0f;
    if (direct_lse && lane < g) direct_lse[row * g + lane] = -__builtin_inff();
}
#endif
// out / lse of rows d_rows[i] (null: i) += the position whose fp16 K / V rows are d_k_tail / d_v_tail [i][heads][128]
// (consecutive i tail_stride_elems apart); see k_attend_fold_tail
hipError_t launch_attend_fold_tail(uint32_t n_rows, const uint32_t* d_rows, uint32_t heads, uint32_t g, const void* d_q_f16,
                                   const void* d_k_tail, const void* d_v_tail, uint64_t tail_stride_elems, float sm_scale,
                                   float* d_out, float* d_lse, hipStream_t s, uint32_t n_layers = 1, uint32_t layer_rows = 0);
// (n_layers > 1: the same rows for n_layers consecutive layers in one launch -- layer l's q / out / lse row blocks start l * layer_rows
//  row blocks further on, its tail rows l * heads * 128 elements further on)
// a.lin_base set: linear form (a.scale_tab, a.q16); else page-table form (a.q8 / a.qs from launch_quantize_q_e4m3)
hipError_t launch_attend_fp8(const AttendArgs& a, uint32_t n_layers, float* d_out, float* d_lse, hipStream_t s);
// scale_tab[tile order of p] = entries[p].rec_bytes >= 2048 ? entries[p].scale : 0 for every page (set_layout time)
hipError_t launch_build_scale_tab(const PageEntry* d_entries, uint64_t n_pages, uint32_t region_pages, float* d_scale_tab, hipStream_t s, uint32_t scale_run = 0);
// the run-order part of a scale table (CodecArgs::scale_run): index of page p relative to scale_tab
__host__ __device__ inline int64_t scale_run_index(uint64_t p, uint32_t scale_run)
{
    const uint32_t D = scale_run & 15u, cap = scale_run >> 4;
    return -static_cast<int64_t>(D) * cap + static_cast<int64_t>(p % D) * cap + static_cast<int64_t>(p / D);
}
// position of page-in-tile j (0..15) in the tile order [kb][r]: pages 2kb, 2kb+1, 8+2kb, 9+2kb of lane group kb
__host__ __device__ inline uint32_t attend_tile_slot(uint32_t j) { return j < 8u ? ((j >> 1) << 2) + (j & 1u) : (((j - 8u) >> 1) << 2) + 2u + (j & 1u); }
// INT4_G32 records (attend_int4.hip; a.lin_base set: linear form, else page-table form with a.entries / a.zero_page;
// a.q8 = the fp16 query rows
// [layers][heads][g][128]); writes the split partials, launch_attend_combine merges them
hipError_t launch_attend_int4(const AttendArgs& a, uint32_t n_layers, hipStream_t s);
hipError_t launch_attend_combine(const AttendArgs& a, uint32_t n_layers, float* d_out, float* d_lse, hipStream_t s);
// MXFP4 records (attend_mx4.hip): a.lin_base | a.stripe_bases | a.table_form choose the address form, a.q16 = the fp16 query rows,
// a.seqs the batch descriptors; n_rows = layers or sequences.  Launches the merge itself when the rows are not final.
hipError_t launch_attend_mx4(const AttendArgs& a, uint32_t n_rows, float* d_out, float* d_lse, hipStream_t s);
// scores only, linear form (a.lin_base, a.scale_tab, a.q16, a.tiles_per_split): d_out [layers][heads][g][2*n_pages]
hipError_t launch_qk_scores_fp8_linear(const AttendArgs& a, uint32_t n_layers, float* d_out, hipStream_t s);
// batch form: a.seqs (device) holds n_seq descriptors, a.q16 = [n_seq][heads][g][128], a.n_splits = the largest
// per-sequence split count; d_out [n_seq][heads][g][128]
hipError_t launch_attend_fp8_batch(const AttendArgs& a, uint32_t n_seq, float* d_out, float* d_lse, hipStream_t s);

// every page's record (rec_bytes rounded up to 16) copied to d_new_addr[page], then entries[page].pool_addr = d_new_addr[page]
hipError_t launch_repack(PageEntry* d_entries, const uint64_t* d_new_addr, uint64_t n, hipStream_t s);
// entries[i].pool_addr = base + i*stride (record bytes / scale untouched): after a migration.  stride == kPlanarMx4: record
// rec0 + i of the tile-planar run at base (address and code delta, record bytes untouched)
hipError_t launch_retarget_entries(PageEntry* d_entries, uint64_t n, uint64_t base, uint64_t stride, hipStream_t s, uint64_t rec0 = 0);


// Token predictor (lstm_predictor.cpp:40-188): n histories of 16 tokens -> top-k (k <= 8) tokens and
// confidences.  d_emb [vocab][64], d_wout [vocab][128]; d_hid (n*128 floats) and d_logits (n*vocab
// floats) are scratch.
// lstm != nullptr && lstm->layers > 0: a real LSTM cell (PyTorch nn.LSTM conventions, k_lstm_cell) instead of the reference's
// degenerate one; bias[l] = b_ih[l] + b_hh[l]; out_bias may be null
struct LstmParams { const float* w_ih_t[4]; const float* w_hh_t[4]; const float* bias[4]; const float* out_bias; uint32_t layers; };   // weights arranged: lstm_arranged_index
// Where k_lstm_cell wants the weight (gate row, column) of a 512 x cols matrix.  The kernel numbers the gate rows
// v = 4 * unit + gate (the four gates of a hidden unit next to each other; nn.LSTM's row is 128 * gate + unit).  Thread
// tid = 8 * (v / 8) + s keeps the column slice s = col / (cols / 8) of the eight rows v / 8 * 8 + 0..7, as register
// q = i * (cols / 8) + col % (cols / 8) with i = (v % 8) ^ s (the xor makes the reduction over the eight slices free of
// selects); the array is [q][512 threads], read coalesced.
inline uint32_t lstm_cell_row(uint32_t row) { return 4u * (row % 128u) + row / 128u; }
inline size_t lstm_arranged_index(uint32_t row, uint32_t col, uint32_t cols)
{
    const uint32_t v = lstm_cell_row(row), cs = cols / 8u, s = col / cs, c = col % cs, i = (v & 7u) ^ s, tid = (v & ~7u) + s;
    return static_cast<size_t>(i * cs + c) * 512u + tid;
}
hipError_t launch_predict(uint32_t n, const int32_t* d_hist, const float* d_emb, const float* d_wout, uint32_t vocab,
                          uint32_t layers, uint32_t k, float* d_hid, float* d_logits, void* d_ws, int32_t* d_tok, float* d_conf,
                          hipStream_t s, const LstmParams* lstm = nullptr);
// d_wout: the output layer's weights ARRANGED by launch_arrange_wout (arranged_wout_bytes(vocab) bytes) from the row-major [vocab][128]
size_t arranged_wout_bytes(uint32_t vocab);
hipError_t launch_arrange_wout(const float* d_src, float* d_dst, uint32_t vocab, hipStream_t s);
// d_ws: predict_ws_bytes(n, vocab) bytes of scratch (the parts of the split top-k: 4096 logits each, at most 64 -- larger
// vocabularies take the one-workgroup kernel and need none)
constexpr uint32_t kPredictTopkSpan = 4096, kPredictTopkMaxParts = 64, kPredictWsPerPart = 8 + 8 * 8;
inline uint32_t predict_topk_parts(uint32_t vocab)
{
    const uint32_t parts = (vocab + kPredictTopkSpan - 1u) / kPredictTopkSpan;
    return parts <= kPredictTopkMaxParts ? parts : 0u;
}
// a handful of requests take k_predict_small: parts of 128 rows, at most kPredictSmallMaxParts of them (vocabularies up to 32 768)
constexpr uint32_t kPredictSmallN = 4, kPredictSmallMaxParts = 256;
inline size_t predict_ws_bytes(uint32_t n, uint32_t vocab)
{
    const size_t batch = static_cast<size_t>(n) * predict_topk_parts(vocab) * kPredictWsPerPart;
    const size_t small = static_cast<size_t>(kPredictSmallN) * kPredictSmallMaxParts * kPredictWsPerPart;
    return (batch > small ? batch : small) + 64;
}

} // namespace speckv
