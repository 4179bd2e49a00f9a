// cxl-speckv_amd/csrc/engine_attend.cpp -- fused decode attention: launch planning for the FP8 / INT4 kernels, single sequences and batches (Engine members)
#include "engine_internal.hpp"
#include "tuning.hpp"

namespace speckv {

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
    // NULL = the engine's stream and a synchronous call: the query may have been produced on any stream of the caller
    if (!s) HIP_TRY(hipDeviceSynchronize());
    hipStream_t st = s ? s : stream_;
    {   // linear form (records in one run, scale table, tile-aligned range inside the layer's region): direct loads
        const uint32_t n_tiles = (n_pages + 15u) / 16u;
        const bool fits = pos_begin % 32u == 0u && a->d_scale_tab && a->linear_base && !tuning().attend_general &&
                          static_cast<uint64_t>(pos_begin) + static_cast<uint64_t>(n_tiles) * 32u <= L.num_tokens;
        if (fits) {
            AttendArgs k{};
            k.k_first = first_page;
            k.layer_stride = layer_stride;
            k.n_pages = n_pages;
            k.heads = L.num_heads;
            k.g = g;
            k.tiles_per_split = 16;
            k.lin_base = a->linear_base;
            k.scale_tab = a->d_scale_tab;
            k.q16 = static_cast<const uint16_t*>(d_q_f16);
            HIP_TRY(launch_qk_scores_fp8_linear(k, n_layers, d_out, st));
            note_use(a, s);
            if (!s) RC_TRY(wait_stream());
            return SPECKV_OK;
        }
    }
    const size_t rows = static_cast<size_t>(n_layers) * L.num_heads * 16;
    uint8_t* q8 = static_cast<uint8_t*>(scratch(s_req_, rows * 128 + rows * sizeof(float), s));
    if (!q8) return SPECKV_ERR_NOMEM;
    float* qs = reinterpret_cast<float*>(q8 + rows * 128);
    HIP_TRY(launch_quantize_q_e4m3(d_q_f16, n_layers * L.num_heads, g, L.head_dim, q8, qs, st));
    HIP_TRY(launch_qk_scores_fp8(a->d_entries, first_page, layer_stride, n_layers, n_pages, L.num_heads, g, q8, qs, d_out, st));
    note_use(a, s);
    if (!s) RC_TRY(wait_stream());
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
    DeviceScope device_scope(device_);
    // NULL = the engine's stream and a synchronous call: the query may have been produced on any stream of the caller
    if (!s) HIP_TRY(hipDeviceSynchronize());
    hipStream_t st = s ? s : stream_;
    const size_t out_elems = static_cast<size_t>(n_layers) * L.num_heads * g * 128;
    if (pos_end == pos_begin) {  // empty range: softmax over nothing -> zeros (and -inf lse is left to the caller)
        HIP_TRY(hipMemsetAsync(d_out, 0, out_elems * sizeof(float), st));
        if (!s) HIP_TRY(hipStreamSynchronize(stream_));
        return SPECKV_OK;
    }
    // The scale table is laid out in tiles of 32 positions from the start of a region: a range that starts inside a tile is
    // attended from the tile's start with its leading positions masked (AttendArgs::skip_pages), so every range of a layout
    // with a scale table takes the tile forms (linear / striped / table) -- the per-wave page-table kernel is left with the
    // layouts that have none (num_tokens not a multiple of 32).
    const bool has_tab = a->d_scale_tab != nullptr;
    const uint32_t begin_al = has_tab ? (pos_begin & ~31u) : pos_begin;
    const uint32_t skip_pages = (pos_begin - begin_al) / 2;
    const uint32_t n_pages = (pos_end - begin_al) / 2;
    pos_begin = begin_al;
    // shim layout [req 0][layer][kind][pos][head]: K pages of a layer, then its V pages
    const uint64_t k_first = (static_cast<uint64_t>(layer) * 2 * L.num_tokens + pos_begin) / 2;
    const uint64_t v_first = k_first + L.num_tokens / 2;
    const uint64_t layer_stride = static_cast<uint64_t>(L.num_tokens);
    if (v_first + (n_layers - 1) * layer_stride + n_pages > a->n_pages) return SPECKV_ERR_GENERAL;
    if (!d_zero_page_) {
        if (is_capturing(s)) return SPECKV_ERR_INVAL;        // first call must run outside a capture (see scratch())
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&d_zero_page_), kPageSize));
        HIP_TRY(hipMemset(d_zero_page_, 0, kPageSize));
    }
    // splits: ~20 waves per CU over the launch (the LDS-DMA kernel keeps 8 resident; measured at 70B-shaped, 80 layers:
    // 8 splits/row 0.72 of HBM peak at 32k and 0.63 at 8k, 16 splits 0.705 / 0.61, 4 splits 0.71 / 0.63, 2: 0.63 / 0.58)
    // A launch that already has 128+ workgroup columns (layers x head quads) is best left unsplit: each workgroup then
    // streams one long run, the rows are final (no partials, no merge launch) -- 80 layers: 1 split 0.73 / 0.70 / 0.64 of
    // HBM peak at 32k / 8k / 2k context against 0.71 / 0.62 / 0.48 with 8 splits.
    uint32_t n_tiles = (n_pages + 15u) / 16u;
    const uint32_t rows = n_layers * L.num_heads;
    // linear form: records in one run, scale table present, tiles aligned with the table's (pos_begin a multiple of 32),
    // and the last (possibly ragged) 32-position tile must not read past the K / V region of its layer
    // (with the range aligned as above and num_tokens a multiple of 32 the tiles never leave the region)
    const bool fits = has_tab && static_cast<uint64_t>(pos_begin) + static_cast<uint64_t>(n_tiles) * 32u <= L.num_tokens;
    const bool general_env = tuning().attend_general != 0;
    const uint8_t* lin_base = (general_env || !fits) ? nullptr : a->linear_base;
    // regular striping over several pools: the same kernel with computed record addresses (no page-table chase)
    const bool striped = !lin_base && fits && a->stripe_n >= 2 && !general_env;
    // no regular placement (pages migrated one by one), or SPECKV_ATTEND_GENERAL set (measurements, tests): the fast kernel
    // with its record addresses from the page table, looked up one request ahead
    const bool table = fits && !lin_base && !striped;
    // (the page-table form has nothing to gain from whole rows: it hides its look-ups behind other waves and always
    // goes through the merge -- 80 layers x 8k: one split 0.13 of HBM peak, eight 0.18+)
    // (striped / moved placements run the same DMA pipeline with their addresses from the page table, k_attend_fp8_dma<TABLE>: the
    //  same rule; the register-staged kernels of rounds 2-5 -- a range that starts inside a tile, or on request -- want the splits)
    const bool dma_table = (striped || table) && skip_pages == 0 && tuning().attend_fp8_table_regs == 0;
    // striped regularly: the linear pipeline over the range's pages by residue class (k_attend_fp8_dma<2>): the tiles are then counted per class
    const bool cls = striped && dma_table && a->stripe_n <= 8 && tuning().attend_fp8_striped_table <= 0;
    if (cls) n_tiles = mx4_striped_tiles(n_pages, a->stripe_n);
    uint32_t want = ((lin_base || dma_table) && rows / 4u >= 128u && n_tiles < 768u) ? 1u : (5120u + rows - 1u) / rows;     // (32k and beyond: 8 splits, below)
    // per-layer calls are latency-bound: short contexts want short splits (measured best: 2 tiles per split at 2k
    // context, 4 at 8k, 8 at 32k), long multi-layer launches are bounded by `want` above
    const uint32_t min_tiles = std::min(8u, std::max(2u, n_tiles / 64u));
    want = std::min(want, std::max(1u, n_tiles / min_tiles));
    // ... and never more than ONE round of workgroups (round 6, profiles/r06_layers_by_context.txt: one layer x 128k took 512 splits = 1024
    // workgroups 0.55 of the roofline, 128 splits = 256 workgroups 0.68; 4 layers x 32k 0.65 -> 0.74, 8 x 32k 0.59 -> 0.68, 12 x 16k 0.56 -> 0.65)
    if (lin_base || dma_table) want = std::min(want, std::max(1u, cus() / std::max(1u, rows / 4u)));
    // Launches of many rows (several layers of one sequence): every workgroup resident at once and the CUs evenly loaded counts for
    // more than the split length -- 80 layers = 160 workgroup rows: 3 splits = 480 workgroups (15 of 16 CUs hold two) 0.777 of the
    // roofline at 32k and 0.738 at 8k, 5 splits = 800 (some CUs four, some three) 0.67, 8 = 1280 (a second round) 0.765, one split
    // (the DMA kernel, 160 of 256 CUs busy) 0.73 / 0.70; 32 layers x 32k: 4 splits = 256 workgroups 0.734, 20 splits 0.72.
    // So: the split count whose workgroups fill whole rounds of the CUs best, one round at most, splits of 64 tiles or more.
    // (A stream form as the INT4 / MXFP4 kernels have it -- the launch's rows x tiles in one equal piece per resident workgroup --
    //  was built for k_attend_fp8_linear and dropped: at the kernel's 128 registers the row-boundary path spilled, and the extra
    //  partials cost the short contexts more than the balance gave: 32k x 80 0.7635 against 0.7745, 8k 0.66 / 0.74, 128k 0.794 / 0.783.)
    if ((lin_base || cls) && rows / 4u >= 32u) {
        const uint32_t wg_rows = rows / 4u, n_cu = cus();
        uint32_t best_s = 1;
        double best = -1.0;
        for (uint32_t sp = 1; sp <= 16u; ++sp) {
            // (pieces under 64 tiles -- never under 16 -- only while the launch has not filled one round: 32 layers x 2k were 64 workgroups 0.36 of the
            //  roofline, 8 pieces 0.56; 4k: 128 workgroups 0.64, 8 pieces 0.685)
            if (sp > 1u && (n_tiles / sp < 16u || (n_tiles / sp < 64u && static_cast<uint64_t>(wg_rows) * (sp - 1u) >= n_cu))) break;
            const uint64_t wgs = static_cast<uint64_t>(wg_rows) * sp, cap = static_cast<uint64_t>(sp == 1u ? 2u : 4u) * n_cu;      // (resident: DMA kernel 2 per CU, register-staged 4)
            if (wgs > cap && sp > 1u) break;
            const double per_cu = static_cast<double>(wgs) / n_cu, score = per_cu / std::ceil(per_cu);
            if (score > best + 1e-9) { best = score; best_s = sp; }
        }
        want = best_s;
    }
    if (tuning().attend_splits > 0) want = static_cast<uint32_t>(tuning().attend_splits);
    const EvenSplit es = even_split(n_tiles, std::max(1u, std::min(want, 2048u)));
    const uint32_t n_splits = es.n_splits, tiles_per_split = es.tiles_per_split;
    const size_t q_bytes = static_cast<size_t>(rows) * 16 * 128, qs_bytes = static_cast<size_t>(rows) * 16 * sizeof(float);
    const size_t acc_bytes = static_cast<size_t>(rows) * n_splits * 16 * 128 * sizeof(float);
    const size_t ml_bytes = static_cast<size_t>(rows) * n_splits * 32 * sizeof(float);
    uint8_t* buf = static_cast<uint8_t*>(scratch(s_attn_, q_bytes + qs_bytes + acc_bytes + ml_bytes, s));
    if (!buf) return SPECKV_ERR_NOMEM;
    AttendArgs k{};
    k.entries = a->d_entries;
    k.k_first = k_first;
    k.v_first = v_first;
    k.layer_stride = layer_stride;
    k.n_pages = n_pages;
    k.skip_pages = skip_pages;
    k.heads = L.num_heads;
    k.g = g;
    k.n_splits = n_splits;
    k.tiles_per_split = tiles_per_split;
    k.q8 = buf;
    k.qs = reinterpret_cast<float*>(buf + q_bytes);
    k.scale_log2e = sm_scale * 1.4426950408889634f;
    k.zero_page = d_zero_page_;
    k.scale_tab = a->d_scale_tab;
    k.q16 = static_cast<const uint16_t*>(d_q_f16);
    k.lin_base = lin_base;
    if (striped) {
        k.stripe_bases = a->d_stripe;
        k.stripe_n = a->stripe_n;
        k.stripe_magic = static_cast<uint32_t>((1ull << 32) / a->stripe_n + 1u);
    }
    k.part_acc = reinterpret_cast<float*>(buf + q_bytes + qs_bytes);
    k.part_ml = reinterpret_cast<float*>(buf + q_bytes + qs_bytes + acc_bytes);
    if (cls) { k.fp8_cls = 1u; k.scale_run = a->scale_run; }
    if (table) { k.table_form = 1u; k.lin_base = nullptr; k.stripe_bases = nullptr; }
    if (!k.lin_base && !striped && !table)         // the linear / striped / table forms quantise the query in their own prologue
        HIP_TRY(launch_quantize_q_e4m3(d_q_f16, rows, g, L.head_dim, buf, reinterpret_cast<float*>(buf + q_bytes), st));
    if (n_splits == 1u) { k.direct_out = d_out; k.direct_lse = d_lse; }      // no merge launch (linear / striped form)
    HIP_TRY(launch_attend_fp8(k, n_layers, d_out, d_lse, st));
    note_use(a, s);
    if (!s) RC_TRY(wait_stream());
    return SPECKV_OK;
}

// One decode step of a batch: the fused attention of ONE layer for many sequences (allocations) in one launch
// (BASELINE configs[3] shape: 256 sequences).  Every allocation must qualify for the linear form.
// Not capturable into a HIP graph: the per-call descriptors travel through a pinned slot that later calls reuse, so a
// replay would read other calls' descriptors -- the call refuses to run on a capturing stream (the per-sequence
// entry points speckv_ext_attend_fp8 / _int4 are capturable).
// Split length of a batch launch (see the measurements quoted in attend_batch).  seqs[i].n_splits holds the tile count of
// sequence i (null: n_seq sequences of uniform_tiles each, the bound a plan is sized for).
// INT4 batch launches between half a machine and a whole one of workgroup columns: every long sequence in a long and a short
// piece, dispatched rows-first (ring_rule.hpp: int4_unequal_fraction / unequal_pieces).  The environment switches are for
// measurement runs.
using UnequalSplit = UnequalFraction;
static UnequalSplit int4_unequal_split(uint32_t n_seq, uint32_t hq, uint32_t tiles_max)
{
    if (tuning().attend_tiles_per_split > 0) return {false, 1.0};          // (a forced split length: plain even splits)
    return int4_unequal_fraction(n_seq * hq, tiles_max);
}

static uint32_t batch_tiles_per_split(bool fp8, uint32_t n_seq, uint32_t heads, uint64_t total_tiles, const AttendSeq* seqs,
                                      uint32_t uniform_tiles)
{
    if (tuning().attend_tiles_per_split > 0) return static_cast<uint32_t>(tuning().attend_tiles_per_split);
    const uint32_t hq = heads / 4u;
    if (fp8) {
        // FP8: the busiest-CU cost rule of ring_rule.hpp (48 sequences x 16k: 288 workgroups 0.50 of HBM peak, 192: 0.64,
        // 768: 0.71; 32 x 32k: 256 workgroups 0.79, 512: 0.76, 384: 0.63; 128 x 2k: unsplit 0.73, two splits 0.56)
        std::vector<uint32_t> tiles;
        if (seqs) { tiles.resize(n_seq); for (uint32_t i = 0; i < n_seq; ++i) tiles[i] = seqs[i].n_splits; }
        return fp8_batch_tiles_per_split(seqs ? tiles.data() : nullptr, n_seq, uniform_tiles, hq, 256u, 8u);
    }
    const uint64_t wg_target = 768u;
    uint32_t tps = static_cast<uint32_t>(std::max<uint64_t>(8, (total_tiles * hq + wg_target - 1u) / wg_target));
    tps = (static_cast<uint64_t>(n_seq) * hq >= 384u) ? 256u : std::min(tps, 256u);   // enough columns: whole sequences
    return tps;
}

// INT4 batches on the whole-record kernel (k_attend_int4_wg8<2>: workgroups = sequences x splits, one 16-wave workgroup per
// CU resident, its two halves merged in LDS): one round of resident workgroups when the batch is smaller than that, whole
// sequences otherwise (a whole sequence is final: no partials, no merge launch); never under 8 tiles a split.
// More sequences than CUs: workgroups of one run (8 waves, two resident per CU) -- a finishing workgroup's successor starts
// under its neighbour's stream, where a second round of 16-wave workgroups would wait for the whole CU (512 x 1k 0.52 -> 0.54,
// 1024 x 1k 0.56 -> 0.595: profiles/r04_batch_short.txt); AttendArgs::wg8 = 2.
static uint32_t int4_wg8_form(uint32_t n_seq, uint32_t cus) { return n_seq > cus ? 2u : 1u; }
static uint32_t int4_wg8_batch_tps(uint32_t n_seq, uint32_t tiles_max, uint32_t cus)
{
    if (tuning().attend_tiles_per_split > 0) return static_cast<uint32_t>(tuning().attend_tiles_per_split);      // (tests, measurement runs)
    const uint32_t resident = cus;                                        // 16-wave workgroups (two halves each), one per CU
    const uint32_t splits = std::max(1u, resident / std::max(1u, n_seq));
    // more sequences than CUs: the pieces that balance the last round (ring_rule.hpp balanced_tiles_per_piece; 260 x 8k 0.45 -> 0.61
    // of the HBM roofline, 300 0.52 -> 0.67, 340 0.58 -> 0.70, 384 0.63 -> 0.71)
    if (n_seq > resident && tiles_max >= 64u) return balanced_tiles_per_piece(nullptr, n_seq, tiles_max, 1u, resident, kPiecesInt4Wg8);
    // between half a machine and a whole one (the 16-wave form): 130 x 8k 0.43 -> 0.57, 160 0.53 -> 0.64, 200 and up stay whole
    if (2u * n_seq > resident && tiles_max >= 64u) return balanced_tiles_per_piece(nullptr, n_seq, tiles_max, 1u, resident, kPiecesInt4Halves);
    // (the floor was 32 tiles until round 6: with few sequences that left most of the machine idle -- 8 x 8k: 64 workgroups 0.16 of the HBM
    //  roofline, 256 workgroups of 8 tiles 0.34; 16 x 8k 0.32 -> 0.49; 4 x 8k 0.08 -> 0.21)
    return std::max(8u, (tiles_max + splits - 1u) / splits);
}

// MXFP4 batches (k_attend_mx4: one workgroup of 4 waves = the 8 kv heads per (sequence, split), three tiles deep in LDS: ONE
// workgroup resident per CU): one round of resident workgroups -- measured at 256 sequences x 8k: whole sequences (256
// workgroups) 0.77 of the HBM roofline, two splits each 0.73 (profiles/r05_mx4.txt) -- never under 8 tiles a split; a whole
// sequence is final (no partials, no merge launch).
//
// More sequences than half the CUs (round 6): the pieces per sequence that balance the last round of workgroups (ring_rule.hpp
// balanced_tiles_per_piece: 260 x 8k 0.57 -> 0.69 of the HBM roofline, 300 0.65 -> 0.74, 340 0.72 -> 0.78; 360 and up stay whole).
static uint32_t mx4_batch_tps(uint32_t n_seq, uint32_t tiles_max, uint32_t cus)
{
    if (tuning().attend_tiles_per_split > 0) return static_cast<uint32_t>(tuning().attend_tiles_per_split);      // (tests, measurement runs)
    const uint32_t resident = cus;
    const uint32_t splits = std::max(1u, resident / std::max(1u, n_seq));
    // (up to CUs sequences whole ones run on the 8-wave halves form: pieces pay from 8k context -- 130 x 8k 0.63 -> 0.65, 160 0.71 -> 0.76; 4k: 0.63 -> 0.61, 0.73 -> 0.70)
    if ((n_seq > resident && tiles_max >= 64u) || (2u * n_seq > resident && tiles_max >= 256u)) return balanced_tiles_per_piece(nullptr, n_seq, tiles_max, 1u, resident, kPiecesMx4);
    return std::max(8u, (tiles_max + splits - 1u) / splits);
}

// Sequences of different lengths (round 6, profiles/r06_ragged_batches.txt): the order their workgroups are dispatched in decides how evenly the
// CUs are loaded -- a CU receives workgroups i, i + CUs, i + 2 CUs, ... of the launch.  Kernels that keep several workgroups resident per
// CU (FP8: 4; INT4 on one-run workgroups: 2) get the sequences sorted by length and laid out as a serpentine over rounds of `round`
// sequences (one round = the sequences whose workgroups cover the CUs once): a CU then holds a long one with a short one -- 256 sequences
// of 1k .. 16k, FP8: 0.54 of the HBM roofline as given, 0.63 sorted, 0.78 as a serpentine; 512: 0.535 -> 0.79, INT4 0.48 -> 0.67.
// round = 0 (MXFP4, one workgroup per CU): longest first, the short ones fill the tail (512 sequences 0.62 -> 0.84).
// Returns false (order as given) when the lengths do not differ by more than a tile in eight.
static bool attend_dispatch_order(const AttendSeq* seqs, uint32_t n_seq, uint32_t round, uint32_t* order)
{
    std::vector<uint32_t> len(n_seq);
    for (uint32_t i = 0; i < n_seq; ++i) len[i] = seqs[i].n_pages;
    return dispatch_order_by_length(len.data(), n_seq, round, order);           // ring_rule.hpp
}
// sequences per round of the CUs for the kernel a batch of this format runs on (0: longest first)
static uint32_t attend_order_round(bool fp8, bool mx4, uint32_t heads, uint32_t n_seq, uint32_t cus)
{
    if (mx4) return 0u;
    if (fp8) return std::max(1u, cus / std::max(1u, heads / 4u));
    if (heads == 8u) return cus;                               // whole-record kernel (one column per sequence)
    return std::max(1u, cus / std::max(1u, heads / 4u));
}

int Engine::attend_batch(int scheme, uint32_t n_seq, const uint64_t* handles, uint32_t layer, const void* d_q_f16, uint32_t g,
                         const uint32_t* pos_end, float sm_scale, float* d_out, float* d_lse, hipStream_t s)
{
    const bool fp8 = scheme == SPECKV_COMP_FP8_E4M3, mx4 = scheme == SPECKV_COMP_MXFP4;
    if (null_) return no_data_path("speckv_ext_attend_*_batch");
    if (n_seq == 0) return SPECKV_OK;
    if (is_capturing(s)) {
        SPECKV_ERR("speckv_ext_attend_*_batch cannot be captured into a HIP graph (its descriptors are staged per call); "
                   "capture the per-sequence speckv_ext_attend_fp8 / _int4 calls instead");
        return SPECKV_ERR_INVAL;
    }
    if (!handles || !pos_end || !d_q_f16 || !d_out || g == 0 || g > 16) return SPECKV_ERR_INVAL;
    std::vector<AttendSeq> seqs(n_seq);
    uint64_t total_tiles = 0;
    uint32_t heads = 0;
    bool any_striped = false;                 // then the whole launch takes the striped kernels (a single run is "striped over 1")
    bool any_table = false;                   // ... or, with a member that has no regular placement, the table forms
    std::vector<const PageEntry*> ents(n_seq);
    for (uint32_t i = 0; i < n_seq; ++i) {
        Allocation* a = find(handles[i]);
        if (!a) return SPECKV_ERR_GENERAL;
        if (!a->has_layout || a->scheme != scheme) return SPECKV_ERR_INVAL;
        const Layout& L = a->layout;
        if (L.head_dim != 128 || L.bytes_per_element != 2 || L.num_heads * L.head_dim != 1024 || L.num_tokens % 2) return SPECKV_ERR_INVAL;
        if (layer >= L.num_layers || pos_end[i] % 2 || pos_end[i] > L.num_tokens) return SPECKV_ERR_INVAL;
        const uint32_t n_pages = pos_end[i] / 2, n_tiles = (n_pages + 15u) / 16u;
        if ((fp8 && !a->d_scale_tab) || static_cast<uint64_t>(n_tiles) * 32u > L.num_tokens) {
            SPECKV_ERR("speckv_ext_attend_*_batch: sequence %u does not qualify for the tile-aligned forms (pos_end rounded up "
                       "to 32 inside the layer%s)", i, fp8 ? ", layout with num_tokens %% 32 == 0" : "");
            return SPECKV_ERR_INVAL;
        }
        heads = L.num_heads;
        note_use(a, s);
        any_table = any_table || !a->stripe_n;                 // no regular placement (migrated pages): the launch reads addresses from the page tables
        ents[i] = a->d_entries;
        any_striped = any_striped || !a->linear_base;
        seqs[i].stripe_bases = a->d_stripe;
        seqs[i].stripe_n = a->stripe_n;
        seqs[i].lin_base = a->linear_base;
        seqs[i].scale_tab = a->d_scale_tab;
        seqs[i].k_first = static_cast<uint64_t>(layer) * L.num_tokens;       // (layer*2*T)/2
        seqs[i].v_first = seqs[i].k_first + L.num_tokens / 2;
        seqs[i].n_pages = n_pages;
        seqs[i].layer_pages = L.num_tokens;                                   // K + V pages of one layer
        seqs[i].n_splits = n_tiles;                                           // tiles for now, splits below
        total_tiles += n_tiles;
    }
    DeviceScope device_scope(device_);
    // NULL = the engine's stream and a synchronous call: the query may have been produced on any stream of the caller
    if (!s) HIP_TRY(hipDeviceSynchronize());
    hipStream_t st = s ? s : stream_;
    const size_t out_elems = static_cast<size_t>(n_seq) * heads * g * 128;
    if (total_tiles == 0) {
        HIP_TRY(hipMemsetAsync(d_out, 0, out_elems * sizeof(float), st));
        if (!s) HIP_TRY(hipStreamSynchronize(stream_));
        return SPECKV_OK;
    }
    // one split length for the whole batch.  A batch brings its own parallelism: the fewer, longer splits the better, down
    // to about one round of resident workgroups (256 sequences x 8k context, one layer, FP8: 8 tiles per split 0.50 of
    // HBM peak, 32: 0.59, 64: 0.67, 128: 0.72, 256 = no split: 0.74; INT4: 64..128 best, 0.59; at 2k context both
    // formats want no split at all).  INT4 target: 768 workgroups, never under 8 tiles per split; FP8: the cost rule of
    // batch_tiles_per_split.
    // INT4 (arithmetic-bound kernel): splits longer than 256 tiles stop paying (256 sequences x 32k: 256 tiles per split
    // 0.67, 512: 0.65, 1024 = no split: 0.60), shorter sequences are best left whole (8k 0.63 against 0.59 in two
    // splits, 4k 0.60 / 0.52, 2k 0.58 / 0.43: single-split rows are final, no partials and no merge).
    // FP8 over striped pools, every member placed regularly: the register-staged kernel by residue classes (k_attend_fp8_linear<.., CLS>);
    // on request (tests, A/B) the DMA pipeline with its addresses from the page tables (k_attend_fp8_dma<1>), in page order
    // (measured: batches of 256 x 8k over 7 runs 0.70-0.71 by residue classes against 0.71-0.73 through the page tables -- the table form
    //  stays the default for batches; attend_fp8_striped_table = -1 takes the class form: tests, A/B)
    const bool fp8_cls = fp8 && any_striped && !any_table && tuning().attend_fp8_table_regs == 0 && tuning().attend_fp8_striped_table < 0;
    if (fp8 && any_striped && tuning().attend_fp8_table_regs == 0 && !fp8_cls) any_table = true;
    if (any_table)                                           // (AttendSeq::lin_base carries the page table in table launches)
        for (uint32_t i = 0; i < n_seq; ++i) seqs[i].lin_base = reinterpret_cast<const uint8_t*>(ents[i]);
    // INT4_G32, 8 kv heads, every member placed regularly (one run = "striped over 1"): the whole-record kernel by residue classes
    const bool int4_cls = !fp8 && !mx4 && any_striped && !any_table && heads == 8u && tuning().attend_int4_striped_wg == 0;
    if ((mx4 || int4_cls || fp8_cls) && any_striped && !any_table) {    // the striped forms of k_attend_mx4 / k_attend_int4_wg8 / k_attend_fp8_linear count their tiles by residue class
        total_tiles = 0;
        for (uint32_t i = 0; i < n_seq; ++i) { seqs[i].n_splits = mx4_striped_tiles(seqs[i].n_pages, seqs[i].stripe_n); total_tiles += seqs[i].n_splits; }
    }
    uint32_t tiles_max = 0;
    for (uint32_t i = 0; i < n_seq; ++i) tiles_max = std::max(tiles_max, seqs[i].n_splits);
    const bool wg8 = !fp8 && !mx4 && (!any_striped || int4_cls) && !any_table && heads == 8u;
    uint32_t tps = mx4 ? mx4_batch_tps(n_seq, tiles_max, cus()) : wg8 ? int4_wg8_batch_tps(n_seq, tiles_max, cus()) : batch_tiles_per_split(fp8, n_seq, heads, total_tiles, seqs.data(), 0);
    // members of different lengths: pieces on account of the lengths (ring_rule.hpp ragged_tiles_per_piece), dispatched by length and rows first (below)
    if ((fp8 || mx4 || wg8) && tuning().attend_tiles_per_split <= 0 && tuning().attend_order_as_given == 0) {
        std::vector<uint32_t> tl(n_seq);
        for (uint32_t i = 0; i < n_seq; ++i) tl[i] = seqs[i].n_splits;
        const uint32_t r = ragged_tiles_per_piece(tl.data(), n_seq, cus(), fp8 ? heads / 4u : 1u, fp8 ? 4u : 1u);
        if (r && r < tps) tps = r;
    }
    const UnequalSplit unequal = (fp8 || mx4 || wg8) ? UnequalSplit{false, 1.0} : int4_unequal_split(n_seq, heads / 4u, tiles_max);
    uint32_t max_splits = 0;
    uint64_t parts = 0;
    for (uint32_t i = 0; i < n_seq; ++i) {
        const uint32_t n_tiles = seqs[i].n_splits;
        if ((n_tiles + tps - 1u) / tps > 2048u) return SPECKV_ERR_INVAL;
        // the sequence's tiles divided evenly over its splits (171 + 85 tiles instead of 128 + 128 cost 15 %)
        EvenSplit es = even_split(n_tiles, (n_tiles + tps - 1u) / tps);
        if (unequal.on) es = unequal_pieces(unequal, n_tiles);
        seqs[i].tiles_per_split = n_tiles ? es.tiles_per_split : tps;
        seqs[i].n_splits = es.n_splits;
        seqs[i].part_base = static_cast<uint32_t>(parts);
        parts += static_cast<uint64_t>(heads) * seqs[i].n_splits;
        max_splits = std::max(max_splits, seqs[i].n_splits);
    }
    const size_t acc_bytes = static_cast<size_t>(parts) * 16 * 128 * sizeof(float), ml_bytes = static_cast<size_t>(parts) * 32 * sizeof(float);
    uint8_t* buf = static_cast<uint8_t*>(scratch(s_attn_, acc_bytes + ml_bytes, s));
    if (!buf) return SPECKV_ERR_NOMEM;
    // The kernels read the descriptors IN PLACE from a pinned slot: each workgroup fetches its own 64 bytes over the host link when
    // it starts (one round trip, all workgroups at once).  Copying the slot to the device first was a copy-engine operation in
    // front of every launch, 7-8 us that the kernels waited for: 256 x 1k MXFP4 59.5 -> 53 us per call, FP8 96 -> 88
    // (profiles/r05_mx4.txt; the planned form never had it).  A slot is reused once the launches that read it have finished.
    const size_t desc_bytes = seqs.size() * sizeof(AttendSeq), seq_bytes = desc_bytes + seqs.size() * sizeof(uint32_t);      // descriptors, then the dispatch order
    if (seq_ring_.slot_bytes < seq_bytes) {
        if (seq_ring_.base) { HIP_TRY(hipDeviceSynchronize()); (void)hipHostFree(seq_ring_.base); seq_ring_.base = nullptr; }
        seq_ring_.slot_bytes = std::max<size_t>(seq_bytes * 2, 16384);
        HIP_TRY(hipHostMalloc(&seq_ring_.base, seq_ring_.slot_bytes * kSeqRingSlots, hipHostMallocDefault));
        for (auto& ev : seq_ring_.ev)
            if (!ev) HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    }
    const int slot = seq_ring_.next;
    seq_ring_.next = (slot + 1) % kSeqRingSlots;
    HIP_TRY(hipEventSynchronize(seq_ring_.ev[slot]));          // whatever last used this slot (a launch that read it in place, a plan's copy) has finished
    void* staged = static_cast<uint8_t*>(seq_ring_.base) + static_cast<size_t>(slot) * seq_ring_.slot_bytes;
    memcpy(staged, seqs.data(), desc_bytes);
    const bool ordered = tuning().attend_order_as_given == 0 &&
                         attend_dispatch_order(seqs.data(), n_seq, attend_order_round(fp8, mx4, heads, n_seq, cus()), reinterpret_cast<uint32_t*>(static_cast<uint8_t*>(staged) + desc_bytes));
    AttendSeq* d_seqs = nullptr;
    HIP_TRY(hipHostGetDevicePointer(reinterpret_cast<void**>(&d_seqs), staged, 0));
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
    k.lin_base = (any_striped || any_table) ? nullptr : seqs[0].lin_base;           // (overridden per sequence)
    if (any_striped) k.stripe_bases = seqs[0].stripe_bases;          // marks a striped launch (each sequence brings its own table)
    if (fp8_cls) k.fp8_cls = 1u;
    if (any_table) {
        k.table_form = 1u; k.lin_base = nullptr; k.stripe_bases = nullptr;
        if (!d_zero_page_) {
            HIP_TRY(hipMalloc(reinterpret_cast<void**>(&d_zero_page_), kPageSize));
            HIP_TRY(hipMemset(d_zero_page_, 0, kPageSize));
        }
        k.zero_page = d_zero_page_;
    }
    k.seqs = d_seqs;
    if (ordered) k.order = reinterpret_cast<const uint32_t*>(d_seqs + n_seq);
    if (ordered && max_splits > 1u) k.rows_first = 1u;                 // (members of different lengths with pieces: see launch_attend_fp8_batch)
    k.part_acc = reinterpret_cast<float*>(buf);
    k.part_ml = reinterpret_cast<float*>(buf + acc_bytes);
    bool one_split_each = true;                           // then the attention kernel writes the final rows itself
    for (uint32_t i = 0; i < n_seq; ++i) one_split_each = one_split_each && seqs[i].n_splits == 1u;
    if (one_split_each) { k.direct_out = d_out; k.direct_lse = d_lse; }
    // MXFP4, whole sequences in single runs and a CU to each: 8-wave workgroups, the run cut in two (k_attend_mx4<0, 2>)
    if (mx4 && one_split_each && !any_striped && !any_table && static_cast<uint64_t>(n_seq) * ((g + 7u) / 8u) <= cus() && tuning().attend_mx4_one_half == 0) k.mx4_halves = 1u;
    if (unequal.on) k.rows_first = 1u;
    if (wg8) k.wg8 = int4_wg8_form(n_seq, cus());
    // The kernels read the slot's descriptors in place: whatever was launched must have passed before the slot is written again.
    // The guard event is recorded on EVERY way out from here on (a merge launch that fails behind an attention launch
    // that went through used to leave the slot unguarded under a kernel still fetching from it).
    struct SlotGuard {
        hipEvent_t ev; hipStream_t st;
        ~SlotGuard() { if (hipEventRecord(ev, st) != hipSuccess) (void)hipGetLastError(); }
    } slot_guard{seq_ring_.ev[slot], st};
    if (fp8) {
        HIP_TRY(launch_attend_fp8_batch(k, n_seq, d_out, d_lse, st));
    } else if (mx4) {
        HIP_TRY(launch_attend_mx4(k, n_seq, d_out, d_lse, st));
    } else {
        HIP_TRY(launch_attend_int4(k, n_seq, st));        // grid y = sequences x head groups, as for layers
        if (!k.direct_out) HIP_TRY(launch_attend_combine(k, n_seq, d_out, d_lse, st));
    }
    if (!s) HIP_TRY(hipStreamSynchronize(stream_));
    return SPECKV_OK;
}

// ---- planned batches: descriptors resident on the device, the launches capturable ---------------------------------------
// A decode step under a HIP graph replays the same launches with new sequence lengths.  speckv_ext_attend_batch_plan
// (outside the graph, once per step) writes one descriptor per sequence -- valid for every layer -- into a device buffer
// of the caller; speckv_ext_attend_*_planned is kernel launches only: no handle look-ups, no staging, grid and scratch
// sized from max_pos_end alone, so a captured launch stays valid for as long as the lengths stay within that bound.
struct PlanGeometry { uint32_t tps, max_splits; uint64_t parts_bound; UnequalSplit unequal; };
static PlanGeometry plan_geometry(bool fp8, uint32_t n_seq, uint32_t heads, uint32_t max_pos_end, uint32_t cus, bool mx4 = false, uint32_t mx4_stripe_n_max = 0)
{
    // (MXFP4 -- and INT4_G32 on the whole-record kernel -- over striped pools count tiles by residue class: at most ceil(pages / 16) +
    //  runs + 1 of them, whatever a member's run count; mx4_stripe_n_max is the largest run count of such a plan, 0 otherwise)
    const uint32_t tiles_max = (max_pos_end / 2u + 15u) / 16u + (mx4_stripe_n_max >= 2u ? mx4_stripe_n_max + 1u : 0u);
    PlanGeometry g{};
    if (mx4) {
        g.unequal = UnequalSplit{false, 1.0};
        g.tps = mx4_batch_tps(n_seq, tiles_max, cus);
        g.max_splits = std::max(1u, (tiles_max + g.tps - 1u) / g.tps);
        g.parts_bound = static_cast<uint64_t>(n_seq) * heads * g.max_splits;
        return g;
    }
    if (!fp8 && heads == 8u) {            // the whole-record kernel's geometry (a striped / table launch runs it on the 4-head kernels)
        g.unequal = UnequalSplit{false, 1.0};
        g.tps = int4_wg8_batch_tps(n_seq, tiles_max, cus);
        g.max_splits = std::max(1u, (tiles_max + g.tps - 1u) / g.tps);
        g.parts_bound = static_cast<uint64_t>(n_seq) * heads * g.max_splits;
        return g;
    }
    g.unequal = fp8 ? UnequalSplit{false, 1.0} : int4_unequal_split(n_seq, heads / 4u, tiles_max);
    if (g.unequal.on) {                                       // (the rule depends on the plan's bound only: plan and launch agree)
        g.tps = tiles_max;
        g.max_splits = 2u;
        g.parts_bound = static_cast<uint64_t>(n_seq) * heads * 2u;
        return g;
    }
    g.tps = batch_tiles_per_split(fp8, n_seq, heads, static_cast<uint64_t>(tiles_max) * n_seq, nullptr, tiles_max);
    g.max_splits = std::max(1u, (tiles_max + g.tps - 1u) / g.tps);
    g.parts_bound = static_cast<uint64_t>(n_seq) * heads * g.max_splits;
    return g;
}

int Engine::attend_batch_plan(uint32_t n_seq, const uint64_t* handles, const uint32_t* pos_end, uint32_t max_pos_end,
                              void* d_plan, size_t plan_bytes, hipStream_t s)
{
    if (null_) return no_data_path("speckv_ext_attend_batch_plan");
    if (n_seq == 0) return SPECKV_OK;
    if (!handles || !pos_end || !d_plan || !s || max_pos_end % 2 || plan_bytes < n_seq * sizeof(AttendSeq)) return SPECKV_ERR_INVAL;
    if (is_capturing(s)) return SPECKV_ERR_INVAL;            // the plan is what changes between replays: it stays outside the graph
    std::vector<AttendSeq> seqs(n_seq);
    int scheme = -1;
    bool any_striped = false, any_table = false;
    std::vector<const PageEntry*> ents(n_seq);
    uint32_t heads = 0, min_layers = UINT32_MAX;
    for (uint32_t i = 0; i < n_seq; ++i) {
        Allocation* a = find(handles[i]);
        if (!a) return SPECKV_ERR_GENERAL;
        if (scheme < 0) scheme = a->scheme;
        if (!a->has_layout || a->scheme != scheme || (scheme != SPECKV_COMP_FP8_E4M3 && scheme != SPECKV_COMP_INT4_G32 && scheme != SPECKV_COMP_MXFP4)) return SPECKV_ERR_INVAL;
        const Layout& L = a->layout;
        if (L.head_dim != 128 || L.bytes_per_element != 2 || L.num_heads * L.head_dim != 1024 || L.num_tokens % 2) return SPECKV_ERR_INVAL;
        if (pos_end[i] % 2 || pos_end[i] > L.num_tokens || pos_end[i] > max_pos_end) return SPECKV_ERR_INVAL;
        const uint32_t n_pages = pos_end[i] / 2, n_tiles = (n_pages + 15u) / 16u;
        const bool fp8 = scheme == SPECKV_COMP_FP8_E4M3;
        if ((fp8 && !a->d_scale_tab) || static_cast<uint64_t>(n_tiles) * 32u > L.num_tokens) return SPECKV_ERR_INVAL;
        min_layers = std::min(min_layers, L.num_layers);
        heads = L.num_heads;
        note_use(a, s);
        any_table = any_table || !a->stripe_n;
        ents[i] = a->d_entries;
        any_striped = any_striped || !a->linear_base;
        seqs[i].stripe_bases = a->d_stripe;
        seqs[i].stripe_n = a->stripe_n;
        seqs[i].lin_base = a->linear_base;
        seqs[i].scale_tab = a->d_scale_tab;
        seqs[i].k_first = 0;                                   // layer 0; the launch adds layer * layer_pages
        seqs[i].v_first = L.num_tokens / 2;
        seqs[i].layer_pages = L.num_tokens;
        seqs[i].n_pages = n_pages;
        seqs[i].n_splits = n_tiles;
    }
    const bool fp8_cls = scheme == SPECKV_COMP_FP8_E4M3 && any_striped && !any_table && tuning().attend_fp8_table_regs == 0 && tuning().attend_fp8_striped_table < 0;      // (see attend_batch)
    if (scheme == SPECKV_COMP_FP8_E4M3 && any_striped && tuning().attend_fp8_table_regs == 0 && !fp8_cls) any_table = true;
    uint32_t stripe_n_max = 0;
    const bool int4_cls = scheme == SPECKV_COMP_INT4_G32 && any_striped && !any_table && heads == 8u && tuning().attend_int4_striped_wg == 0;
    if ((scheme == SPECKV_COMP_MXFP4 || int4_cls || fp8_cls) && any_striped && !any_table)
        for (uint32_t i = 0; i < n_seq; ++i) {
            seqs[i].n_splits = mx4_striped_tiles(seqs[i].n_pages, seqs[i].stripe_n);
            stripe_n_max = std::max(stripe_n_max, seqs[i].stripe_n);
        }
    PlanGeometry g = plan_geometry(scheme == SPECKV_COMP_FP8_E4M3, n_seq, heads, max_pos_end, cus(), scheme == SPECKV_COMP_MXFP4, stripe_n_max);
    if (g.max_splits > 2048u) return SPECKV_ERR_INVAL;
    const uint32_t rule_tps = g.tps, rule_splits = g.max_splits;
    // The launches of a plan may sit in a captured graph: their grid (pieces per member at most) and the presence of the merge launch must not change under
    // it.  So the FIRST plan of a shape (members, format, bound) in a buffer fixes them -- from the lengths it sees: members of different lengths get room
    // for pieces (ragged_tiles_per_piece) and the rows-first grid -- and every later plan of that shape in that buffer keeps them (a later batch that wants
    // more pieces than there is room for gets longer ones).  A new shape (the caller captures anew for it anyway) decides anew.
    uint32_t plan_tps = g.tps;
    bool plan_rows_first = g.unequal.on, plan_sticky = false;
    {
        const std::array<uint64_t, 4> room_key{reinterpret_cast<uintptr_t>(d_plan), n_seq | (static_cast<uint64_t>(scheme) << 32), max_pos_end | (static_cast<uint64_t>(g.tps) << 32), g.max_splits};
        const auto old = plan_rooms_.find(room_key);               // (... the rule for equal lengths is part of the shape: the tuning keys move it)
        const bool same_shape = old != plan_rooms_.end();
        uint32_t r = 0, n_max = 0;
        // (INT4: on the whole-record kernel only, as in attend_batch)
        const bool int4_wg8 = scheme == SPECKV_COMP_INT4_G32 && heads == 8u && (!any_striped || int4_cls) && !any_table;
        if (!g.unequal.on && tuning().attend_order_as_given == 0 && tuning().attend_tiles_per_split <= 0 && (scheme != SPECKV_COMP_INT4_G32 || int4_wg8)) {
            std::vector<uint32_t> tl(n_seq);
            for (uint32_t i = 0; i < n_seq; ++i) { tl[i] = seqs[i].n_splits; n_max = std::max(n_max, tl[i]); }
            r = ragged_tiles_per_piece(tl.data(), n_seq, cus(), scheme == SPECKV_COMP_FP8_E4M3 ? heads / 4u : 1u, scheme == SPECKV_COMP_FP8_E4M3 ? 4u : 1u);
        }
        if (same_shape) {
            plan_sticky = true;
            g.max_splits = old->second.first;
            plan_rows_first = old->second.second;
        } else if (r && r < g.tps) {
            // (the launches' scratch is sized for members x heads x room: 32 768 partials = 270 MB at most, 8 pieces at least)
            const uint32_t room = std::max(8u, 32768u / std::max(1u, n_seq * heads));
            g.max_splits = std::max(g.max_splits, std::min(room, (n_max + r - 1u) / r));
            plan_rows_first = true;
        }
        if (r && r < g.tps && g.max_splits > 1u) plan_tps = std::max(r, (n_max + g.max_splits - 1u) / g.max_splits);      // (pieces on account of the lengths, as many as there is room for)
    }
    if (plans_.size() >= 64 && !plans_.count(d_plan)) plans_.clear();        // (buffers of long-gone steps)
    if (any_table)
        for (uint32_t i = 0; i < n_seq; ++i) seqs[i].lin_base = reinterpret_cast<const uint8_t*>(ents[i]);
    if (any_table && !d_zero_page_) {
        DeviceScope zero_scope(device_);
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&d_zero_page_), kPageSize));
        HIP_TRY(hipMemset(d_zero_page_, 0, kPageSize));
    }
    bool any_empty = false;
    for (uint32_t i = 0; i < n_seq; ++i) any_empty = any_empty || pos_end[i] == 0u;
    // the dispatch order behind the descriptors, where the caller's buffer has the room (speckv_ext_attend_plan_bytes says so since round 6)
    std::vector<uint32_t> order(n_seq);
    // (always written when there is room -- the order as given for sequences of one length: a graph captured over this plan keeps reading it as the lengths change)
    const bool ordered = plan_bytes >= n_seq * (sizeof(AttendSeq) + sizeof(uint32_t));
    const bool by_length = ordered && tuning().attend_order_as_given == 0 &&
                           attend_dispatch_order(seqs.data(), n_seq, attend_order_round(scheme == SPECKV_COMP_FP8_E4M3, scheme == SPECKV_COMP_MXFP4, heads, n_seq, cus()), order.data());
    if (ordered && !by_length)
        for (uint32_t i = 0; i < n_seq; ++i) order[i] = i;
    // (a first plan whose members differ in length and have pieces -- by whichever rule -- takes the rows-first grid, as the batch entry does)
    if (!plan_sticky && by_length && g.max_splits > 1u) plan_rows_first = true;
    if (!plan_sticky) {
        if (plan_rooms_.size() >= 4096) plan_rooms_.clear();   // (graphs captured over plans older than this must be captured anew: not a working set anyone has)
        plan_rooms_[{reinterpret_cast<uintptr_t>(d_plan), n_seq | (static_cast<uint64_t>(scheme) << 32), max_pos_end | (static_cast<uint64_t>(rule_tps) << 32), rule_splits}] = {g.max_splits, plan_rows_first};
    }
    plans_[d_plan] = PlanInfo{n_seq, scheme, min_layers, max_pos_end, any_striped, any_table, stripe_n_max, any_empty, ordered, g.max_splits, plan_rows_first, rule_tps, rule_splits};
    uint64_t parts = 0;
    for (uint32_t i = 0; i < n_seq; ++i) {
        const uint32_t n_tiles = seqs[i].n_splits;
        const EvenSplit es = g.unequal.on ? unequal_pieces(g.unequal, n_tiles) : even_split(n_tiles, (n_tiles + plan_tps - 1u) / plan_tps);
        seqs[i].tiles_per_split = n_tiles ? es.tiles_per_split : plan_tps;
        seqs[i].n_splits = es.n_splits;
        seqs[i].part_base = static_cast<uint32_t>(parts);
        parts += static_cast<uint64_t>(heads) * seqs[i].n_splits;
    }
    DeviceScope device_scope(device_);
    const size_t desc_bytes = seqs.size() * sizeof(AttendSeq), seq_bytes = desc_bytes + (ordered ? seqs.size() * sizeof(uint32_t) : 0u);
    if (seq_ring_.slot_bytes < seq_bytes) {
        if (seq_ring_.base) { HIP_TRY(hipDeviceSynchronize()); (void)hipHostFree(seq_ring_.base); seq_ring_.base = nullptr; }
        seq_ring_.slot_bytes = std::max<size_t>(seq_bytes * 2, 16384);
        HIP_TRY(hipHostMalloc(&seq_ring_.base, seq_ring_.slot_bytes * kSeqRingSlots, hipHostMallocDefault));
        for (auto& ev : seq_ring_.ev)
            if (!ev) HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    }
    const int slot = seq_ring_.next;
    seq_ring_.next = (slot + 1) % kSeqRingSlots;
    HIP_TRY(hipEventSynchronize(seq_ring_.ev[slot]));
    void* staged = static_cast<uint8_t*>(seq_ring_.base) + static_cast<size_t>(slot) * seq_ring_.slot_bytes;
    memcpy(staged, seqs.data(), desc_bytes);
    if (ordered) memcpy(static_cast<uint8_t*>(staged) + desc_bytes, order.data(), n_seq * sizeof(uint32_t));
    HIP_TRY(hipMemcpyAsync(d_plan, staged, seq_bytes, hipMemcpyHostToDevice, s));
    HIP_TRY(hipEventRecord(seq_ring_.ev[slot], s));
    return SPECKV_OK;
}

// Several layers of a planned batch in one call (speckv_ext_attend_planned_layers): d_q_f16 [n_layers][n_seq][heads][g][128], d_out /
// d_lse likewise.  MXFP4 with a geometry of one split per sequence: ONE launch over layers x sequences (the workgroups of the next
// layer start while the last of this one drain: a batch of 256 x 2k is eight launches of 92 us, each a third fill and drain --
// or one of 0.68 ms); anything else: the per-layer launches, from one call.
int Engine::attend_planned_layers(int scheme, const void* d_plan, uint32_t n_seq, uint32_t layer_begin, uint32_t n_layers, const void* d_q_f16, uint32_t g,
                                  uint32_t max_pos_end, float sm_scale, float* d_out, float* d_lse, hipStream_t s, const TailArgs* tail)
{
    if (n_layers == 0 || n_seq == 0) return SPECKV_OK;
    if (n_layers == 1) return attend_planned(scheme, d_plan, n_seq, layer_begin, d_q_f16, g, max_pos_end, sm_scale, d_out, d_lse, s, tail);
    if (g == 0 || g > 16) return SPECKV_ERR_INVAL;
    const auto plan = plans_.find(d_plan);
    const bool one_launch = scheme == SPECKV_COMP_MXFP4 && plan != plans_.end() && layer_begin + n_layers <= plan->second.n_layers &&
                            static_cast<uint64_t>(n_seq) * n_layers <= 65535u && tuning().attend_layers_loop == 0 &&
                            plan->second.max_splits == 1u;
    if (one_launch) return attend_planned(scheme, d_plan, n_seq, layer_begin, d_q_f16, g, max_pos_end, sm_scale, d_out, d_lse, s, tail, n_layers);
    // layer by layer; the position the caller still holds outside the pool is folded into ALL layers by one launch at the end (the
    // attention launches of the step then stand back to back: 8 layers of FP8, 256 x 2k, 7 fold launches and their gaps less)
    const size_t rows = static_cast<size_t>(n_seq) * 8u * g;
    const bool have_tail = tail && tail->n_tail != 0u;
    const bool fold_once = have_tail && scheme != SPECKV_COMP_MXFP4 && d_lse && tail->d_k_tail && tail->d_v_tail && tuning().attend_fold_launch == 0;     // (MXFP4 folds inside its kernel)
    if (have_tail && fold_once && (tail->stride_elems % 8u || tail->stride_elems < static_cast<uint64_t>(layer_begin + n_layers) * 8u * 128u)) return SPECKV_ERR_INVAL;
    for (uint32_t l = 0; l < n_layers; ++l)
        RC_TRY(attend_planned(scheme, d_plan, n_seq, layer_begin + l, static_cast<const uint16_t*>(d_q_f16) + l * rows * 128u, g, max_pos_end, sm_scale,
                              d_out + l * rows * 128u, d_lse ? d_lse + l * rows : nullptr, s, fold_once ? nullptr : tail));
    if (fold_once) {
        DeviceScope device_scope(device_);
        const uint64_t off = static_cast<uint64_t>(layer_begin) * 8u * 128u;
        HIP_TRY(launch_attend_fold_tail(tail->n_tail, tail->n_tail == n_seq ? nullptr : tail->d_tail_rows, 8u, g, d_q_f16,
                                        static_cast<const uint16_t*>(tail->d_k_tail) + off, static_cast<const uint16_t*>(tail->d_v_tail) + off, tail->stride_elems,
                                        sm_scale, d_out, d_lse, s, n_layers, n_seq));
    }
    return SPECKV_OK;
}

int Engine::attend_planned(int scheme, const void* d_plan, uint32_t n_seq, uint32_t layer, const void* d_q_f16, uint32_t g,
                           uint32_t max_pos_end, float sm_scale, float* d_out, float* d_lse, hipStream_t s, const TailArgs* tail, uint32_t n_layers)
{
    const bool fp8 = scheme == SPECKV_COMP_FP8_E4M3, mx4 = scheme == SPECKV_COMP_MXFP4;
    if (null_) return no_data_path("speckv_ext_attend_*_planned");
    if (n_seq == 0) return SPECKV_OK;
    if (!d_plan || !d_q_f16 || !d_out || !s || g == 0 || g > 16 || max_pos_end % 2 || max_pos_end == 0) return SPECKV_ERR_INVAL;
    const uint32_t heads = 8;                                  // the page-wise layout: 8 kv heads x 128
    const auto plan = plans_.find(d_plan);                     // what speckv_ext_attend_batch_plan last wrote there
    if (plan == plans_.end() || plan->second.n_seq != n_seq || plan->second.scheme != scheme || plan->second.max_pos_end != max_pos_end ||
        layer >= plan->second.n_layers || n_layers == 0 || n_layers > plan->second.n_layers - layer) {
        SPECKV_ERR("speckv_ext_attend_*_planned: no plan of this shape at %p (n_seq, format and max_pos_end as planned, layer inside every layout)", d_plan);
        return SPECKV_ERR_INVAL;
    }
    PlanGeometry pg = plan_geometry(fp8, n_seq, heads, max_pos_end, cus(), mx4, plan->second.mx4_stripe_n_max);
    pg.max_splits = plan->second.max_splits;                   // (what the first plan of this shape in this buffer fixed: attend_batch_plan)
    pg.parts_bound = static_cast<uint64_t>(n_seq) * heads * pg.max_splits;
    DeviceScope device_scope(device_);
    const size_t acc_bytes = static_cast<size_t>(pg.parts_bound) * 16 * 128 * sizeof(float), ml_bytes = static_cast<size_t>(pg.parts_bound) * 32 * sizeof(float);
    uint8_t* buf = static_cast<uint8_t*>(scratch(s_attn_, acc_bytes + ml_bytes, s));      // (growth during a capture is refused: warm up once)
    if (!buf) return is_capturing(s) ? SPECKV_ERR_INVAL : SPECKV_ERR_NOMEM;
    AttendArgs k{};
    k.heads = heads;
    k.g = g;
    k.n_splits = pg.max_splits;
    k.tiles_per_split = pg.tps;
    k.layer_stride = 0;
    k.q16 = static_cast<const uint16_t*>(d_q_f16);
    k.q8 = static_cast<const uint8_t*>(d_q_f16);
    k.scale_log2e = sm_scale * 1.4426950408889634f;
    if (plan->second.table) { k.table_form = 1u; k.zero_page = d_zero_page_; }           // a member without a regular placement: addresses from the page tables
    else if (plan->second.striped) {                           // striped launch: every descriptor brings its table
        k.stripe_bases = reinterpret_cast<const uint64_t*>(1);
        if (!fp8 && !mx4 && plan->second.mx4_stripe_n_max) k.wg8 = int4_wg8_form(n_seq, cus());      // (planned by residue classes: the whole-record kernel)
        if (fp8 && plan->second.mx4_stripe_n_max) k.fp8_cls = 1u;                                   // (... the register-staged kernel by residue classes)
    }
    else { k.lin_base = reinterpret_cast<const uint8_t*>(1); if (!fp8 && !mx4) k.wg8 = int4_wg8_form(n_seq, cus()); }     // non-null: linear form (the real base comes from the descriptor)
    k.seqs = static_cast<const AttendSeq*>(d_plan);
    if (plan->second.ordered) k.order = reinterpret_cast<const uint32_t*>(static_cast<const AttendSeq*>(d_plan) + n_seq);
    k.part_acc = reinterpret_cast<float*>(buf);
    k.part_ml = reinterpret_cast<float*>(buf + acc_bytes);
    k.batch_layer = layer;
    k.direct_out = d_out;                                      // sequences with a single split are written directly ...
    k.direct_lse = d_lse;
    // ... decided per sequence on the device, the merge skips those; a geometry of one split at most needs no merge at all
    k.direct_per_seq = pg.max_splits == 1u ? 2u : 1u;
    if (mx4 && pg.max_splits == 1u && !plan->second.striped && !plan->second.table && static_cast<uint64_t>(n_seq) * ((g + 7u) / 8u) <= cus() &&
        tuning().attend_mx4_one_half == 0)
        k.mx4_halves = 1u;                                     // (see attend_batch)
    if (pg.unequal.on || plan->second.rows_first) k.rows_first = 1u;
    // The position the caller still holds outside the pool (TailArgs: a connector's odd last position).  MXFP4: folded in by the
    // attention kernel's own epilogue (split 0 of every sequence) -- needs a lse to be of use to nobody else, a batch without an
    // empty member (an empty sequence has no split to fold into) and the tail index by sequence; otherwise, and for the other
    // formats, one k_attend_fold_tail launch behind the attention, as the connector used to issue itself.
    const bool have_tail = tail && tail->n_tail != 0u;
    if (have_tail && (!tail->d_k_tail || !tail->d_v_tail || !d_lse || tail->stride_elems % 8u || tail->stride_elems < static_cast<uint64_t>(layer + 1u) * heads * 128u))
        return SPECKV_ERR_INVAL;
    const bool fold_in_kernel = have_tail && mx4 && !plan->second.any_empty && !plan->second.table && (tail->d_tail_idx || tail->n_tail == n_seq) && tuning().attend_fold_launch == 0;
    if (n_layers > 1u) {                                       // (attend_planned_layers checked the geometry: MXFP4, one split per sequence)
        if (!mx4 || pg.max_splits != 1u) return SPECKV_ERR_INVAL;
        k.batch_n_seq = n_seq;
    }
    if (fold_in_kernel) {
        k.tail_k = static_cast<const uint16_t*>(tail->d_k_tail);
        k.tail_v = static_cast<const uint16_t*>(tail->d_v_tail);
        k.tail_idx = tail->n_tail == n_seq && !tail->d_tail_idx ? nullptr : tail->d_tail_idx;
        k.tail_stride = tail->stride_elems;
    }
    if (fp8) {
        HIP_TRY(launch_attend_fp8_batch(k, n_seq, d_out, d_lse, s));
    } else if (mx4) {
        HIP_TRY(launch_attend_mx4(k, n_seq * n_layers, d_out, d_lse, s));
    } else {
        HIP_TRY(launch_attend_int4(k, n_seq, s));
        if (k.direct_per_seq != 2u) HIP_TRY(launch_attend_combine(k, n_seq, d_out, d_lse, s));
    }
    if (have_tail && !fold_in_kernel) {
        const size_t rows = static_cast<size_t>(n_seq) * heads * g;
        for (uint32_t l = 0; l < n_layers; ++l) {
            const uint64_t off = static_cast<uint64_t>(layer + l) * heads * 128u;
            HIP_TRY(launch_attend_fold_tail(tail->n_tail, tail->n_tail == n_seq ? nullptr : tail->d_tail_rows, heads, g, static_cast<const uint16_t*>(d_q_f16) + l * rows * 128u,
                                            static_cast<const uint16_t*>(tail->d_k_tail) + off, static_cast<const uint16_t*>(tail->d_v_tail) + off, tail->stride_elems,
                                            sm_scale, d_out + l * rows * 128u, d_lse + l * rows, s));
        }
    }
    return SPECKV_OK;
}

int Engine::attend_fold_tail(uint32_t n_rows, const uint32_t* d_rows, uint32_t heads, uint32_t g, const void* d_q_f16, const void* d_k_tail,
                             const void* d_v_tail, uint64_t tail_stride_elems, float sm_scale, float* d_out, float* d_lse, hipStream_t s)
{
    if (null_) return no_data_path("speckv_ext_attend_fold_tail");
    if (n_rows == 0) return SPECKV_OK;
    if (!d_q_f16 || !d_k_tail || !d_v_tail || !d_out || !d_lse || heads == 0 || g == 0 || g > 16 || tail_stride_elems % 2 ||
        tail_stride_elems < static_cast<uint64_t>(heads) * 128u)
        return SPECKV_ERR_INVAL;
    DeviceScope device_scope(device_);
    if (!s) HIP_TRY(hipDeviceSynchronize());                   // NULL: the engine's stream, synchronous (include/speckv_ext.h)
    HIP_TRY(launch_attend_fold_tail(n_rows, d_rows, heads, g, d_q_f16, d_k_tail, d_v_tail, tail_stride_elems, sm_scale, d_out, d_lse,
                                    s ? s : stream_));
    if (!s) HIP_TRY(hipStreamSynchronize(stream_));
    return SPECKV_OK;
}

// Launch geometry of the whole-record INT4 kernel (k_attend_int4_wg8; 512-thread workgroups, two resident per CU); `cus` = the
// compute units of the ENGINE's device (Engine::cus()).
// Stream form (many layers of one sequence): the launch's n_layers x n_tiles tiles, layer-major, in as many equal pieces as
// workgroups are resident at once -- one pipeline fill per workgroup, no partial last round, few partials per layer.  Worth
// it when a piece is long enough to amortise its fill (>= 16 tiles); *max_slots = most pieces any layer is cut into.
constexpr uint32_t kStreamMinTiles = 896;            // context (tiles of 32 positions) from which several layers of one sequence take the stream form
static bool int4_wg8_stream(uint32_t n_layers, uint32_t n_tiles, uint32_t cus, AttendArgs::Stream* out)
{
    const uint64_t total = static_cast<uint64_t>(n_layers) * n_tiles;
    uint64_t wgs = 2ull * static_cast<uint64_t>(cus);
    // (round 6: from 28k context only.  Below it the fixed grid of ONE round -- layers x splits <= CUs, whole-layer rows final or merged --
    //  is faster: 80 layers x 4k 0.515 (stream) against 0.567, 8k 0.60 / 0.63, 16k 0.64 / 0.66, 24k 0.667 / 0.66, 32k 0.70 / 0.67, 64k 0.71 / 0.69;
    //  70 layers x 4k 0.46 / 0.55; profiles/r06_layers_by_context.txt)
    if (n_layers < 2 || total < 16u * wgs || n_tiles < kStreamMinTiles) return false;
    out->n_wgs = static_cast<uint32_t>(wgs);
    out->len = static_cast<uint32_t>(total / wgs);
    out->rem = static_cast<uint32_t>(total % wgs);
    out->max_slots = 1;
    for (uint32_t l = 0; l < n_layers; ++l) out->max_slots = std::max(out->max_slots, attend_stream_count(l, n_tiles, out->len, out->rem));
    return true;
}
// Fixed grid (per-layer calls, short launches): splits x layers workgroups, in whole rounds of the resident set when the
// launch is that long, else as many 8-tile pieces as there are.
static uint32_t int4_wg8_splits(uint32_t n_layers, uint32_t n_tiles, uint32_t cus)
{
    const uint64_t resident = static_cast<uint64_t>(cus);                 // (16-wave workgroups, two halves each: one per CU)
    const uint64_t total = static_cast<uint64_t>(n_layers) * n_tiles;
    uint64_t wgs = std::max<uint64_t>(1, total / 64u);
    if (wgs >= resident && n_layers >= 2u) wgs = resident;                        // several layers: ONE round (80 layers x 16k: 3 splits 0.65, 6: 0.63, 10: 0.54)
    else if (wgs >= resident) wgs = (wgs + resident / 2u) / resident * resident;  // whole rounds
    else wgs = std::min<uint64_t>(resident, std::max<uint64_t>(wgs, total / 8u));
    const uint64_t per_layer = std::max<uint64_t>(1, n_layers >= 2u ? wgs / n_layers : (wgs + n_layers / 2u) / n_layers);
    return static_cast<uint32_t>(std::min<uint64_t>(per_layer, std::max<uint32_t>(1u, n_tiles / 4u)));
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
    // NULL = the engine's stream and a synchronous call: the query may have been produced on any stream of the caller
    if (!s) HIP_TRY(hipDeviceSynchronize());
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
    uint32_t n_tiles = (n_pages + 15u) / 16u;
    // linear form: records in one local run and every 32-position tile inside the layer's K / V region; otherwise the
    // page-table form of the same kernel
    const bool fits = static_cast<uint64_t>(pos_begin) + static_cast<uint64_t>(n_tiles) * 32u <= L.num_tokens;
    const bool general_env = tuning().attend_general != 0;
    const bool linear = a->linear_base && fits && !general_env;
    const bool striped = !linear && a->stripe_n >= 2 && fits && !general_env;
    // a pool striped over 2..8 runs, 8 kv heads: the whole-record kernel over the range's pages by residue class (every tile 16
    // consecutive records of one run: the linear form's fetch; k_attend_int4_wg8<.., CLS>)
    const bool cls = striped && L.num_heads == 8 && a->stripe_n <= 8 && tuning().attend_int4_striped_wg == 0;
    if (cls) n_tiles = mx4_striped_tiles(n_pages, a->stripe_n);
    // everything else -- no regular placement, a last tile that would leave the region, SPECKV_ATTEND_GENERAL (measurements,
    // tests) -- takes the workgroup kernel with its record addresses from the page table: its look-ups are clamped to the
    // range, so a ragged last tile never reads a record it has no business with.  (The per-wave page-table kernel of rounds
    // 1-3, 0.37 of HBM peak, is gone.)
    const bool table = !linear && !striped;
    if (!linear && !d_zero_page_) {
        if (is_capturing(s)) return SPECKV_ERR_INVAL;
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&d_zero_page_), kPageSize));
        HIP_TRY(hipMemset(d_zero_page_, 0, kPageSize));
    }
    const uint32_t rows = n_layers * L.num_heads;
    uint32_t want = (5120u + rows - 1u) / rows;      // VALU-bound kernel: fewer, longer splits measured best
    const uint32_t min_tiles = std::min(8u, std::max(2u, n_tiles / 64u));      // per-layer calls: see attend_fp8
    want = std::min(want, std::max(1u, n_tiles / min_tiles));
    // Workgroups go to the 8 XCDs round-robin by linear id = split + n_splits * (layer, head quad): with a split count that
    // is a multiple of 8 the two workgroups that share a page's scale line (head quads 0 and 1) run on the same XCD, next
    // to each other (measured at 32k x 80 layers: 8 splits 0.598, 10 or 12 splits 0.56, 16 splits 0.595)
    if (want > 8u) want &= ~7u;
    // whole-record kernel (8 waves = 8 heads, one workgroup per CU): workgroups = splits x layers, in whole rounds of the CUs
    const bool wg8 = (linear || cls) && L.num_heads == 8;
    if (wg8) want = int4_wg8_splits(n_layers, n_tiles, cus());
    const bool forced_splits = tuning().attend_splits > 0;
    if (forced_splits) want = static_cast<uint32_t>(tuning().attend_splits);
    EvenSplit es = even_split(n_tiles, std::max(1u, std::min(want, 2048u)));
    if (!wg8 && es.n_splits > 8u && (es.n_splits & 7u) && !forced_splits)      // the rounding can fall off a multiple of 8
        es = even_split(n_tiles, es.n_splits & ~7u);
    AttendArgs k{};
    const bool stream = wg8 && !forced_splits && int4_wg8_stream(n_layers, n_tiles, cus(), &k.stream);
    if (stream && cls) k.stream.tiles = n_tiles;                   // (the merge counts a layer's partials from the same tile count)
    const uint32_t n_splits = stream ? k.stream.max_slots : es.n_splits, tiles_per_split = es.tiles_per_split;      // (stream: slots per row)
    const size_t acc_bytes = static_cast<size_t>(rows) * n_splits * 16 * 128 * sizeof(float);
    const size_t ml_bytes = static_cast<size_t>(rows) * n_splits * 32 * sizeof(float);
    uint8_t* buf = static_cast<uint8_t*>(scratch(s_attn_, acc_bytes + ml_bytes, s));
    if (!buf) return SPECKV_ERR_NOMEM;
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
    if (table) k.table_form = 1u;
    if (wg8) k.wg8 = 1u;
    if (striped) {
        k.stripe_bases = a->d_stripe;
        k.stripe_n = a->stripe_n;
        k.stripe_magic = static_cast<uint32_t>((1ull << 32) / a->stripe_n + 1u);
    }
    k.zero_page = d_zero_page_;
    k.part_acc = reinterpret_cast<float*>(buf);
    k.part_ml = reinterpret_cast<float*>(buf + acc_bytes);
    if (n_splits == 1u && !stream) { k.direct_out = d_out; k.direct_lse = d_lse; }      // no merge launch
    HIP_TRY(launch_attend_int4(k, n_layers, st));
    if (!k.direct_out) HIP_TRY(launch_attend_combine(k, n_layers, d_out, d_lse, st));
    note_use(a, s);
    if (!s) RC_TRY(wait_stream());
    return SPECKV_OK;
}

// Fused decode attention over MXFP4 K and V records (attend_mx4.hip): any placement.
int Engine::attend_mx4(uint64_t handle, uint32_t layer, uint32_t n_layers, const void* d_q_f16, uint32_t g,
                       uint32_t pos_begin, uint32_t pos_end, float sm_scale, float* d_out, float* d_lse, hipStream_t s)
{
    if (null_) return no_data_path("speckv_ext_attend_mx4");
    Allocation* a = find(handle);
    if (!a) return SPECKV_ERR_GENERAL;
    if (!a->has_layout || a->scheme != SPECKV_COMP_MXFP4) return SPECKV_ERR_INVAL;
    const Layout& L = a->layout;
    if (L.head_dim != 128 || L.bytes_per_element != 2 || L.num_heads * L.head_dim != 1024 || L.num_tokens % 2) return SPECKV_ERR_INVAL;
    if (n_layers == 0 || layer >= L.num_layers || n_layers > L.num_layers - layer || pos_begin % 2 || pos_begin > pos_end ||
        pos_end > L.num_tokens || pos_end % 2)
        return SPECKV_ERR_INVAL;
    if (g == 0 || g > 16 || !d_q_f16 || !d_out) return SPECKV_ERR_INVAL;
    const uint32_t n_pages = (pos_end - pos_begin) / 2;
    DeviceScope device_scope(device_);
    // NULL = the engine's stream and a synchronous call: the query may have been produced on any stream of the caller
    if (!s) HIP_TRY(hipDeviceSynchronize());
    hipStream_t st = s ? s : stream_;
    const size_t out_elems = static_cast<size_t>(n_layers) * L.num_heads * g * 128;
    if (n_pages == 0) {
        HIP_TRY(hipMemsetAsync(d_out, 0, out_elems * sizeof(float), st));
        if (!s) HIP_TRY(hipStreamSynchronize(stream_));
        return SPECKV_OK;
    }
    // tiles of 32 positions count from pos_begin (the scales travel inside the records: no table to stay aligned with)
    const uint64_t k_first = (static_cast<uint64_t>(layer) * 2 * L.num_tokens + pos_begin) / 2;
    const uint64_t v_first = k_first + L.num_tokens / 2;
    const uint64_t layer_stride = static_cast<uint64_t>(L.num_tokens);
    if (v_first + (n_layers - 1) * layer_stride + n_pages > a->n_pages) return SPECKV_ERR_GENERAL;
    uint32_t n_tiles = (n_pages + 15u) / 16u;
    // arithmetic addresses need every 32-position tile inside the layer's K / V region (the last one may be ragged); otherwise,
    // or without a regular placement, the page-table form: its look-ups are clamped to the range
    const bool fits = static_cast<uint64_t>(pos_begin) + static_cast<uint64_t>(n_tiles) * 32u <= L.num_tokens;
    const bool general_env = tuning().attend_general != 0;
    const bool linear = a->linear_base && fits && !general_env;
    // a pool striped over 2..8 runs: the range's pages by residue class of the page index (every tile 16 consecutive records of one
    // run: the linear form's fetch), whatever the range -- rows past a class's end are masked, nothing needs to "fit"
    const bool striped = !linear && a->stripe_n >= 2 && a->stripe_n <= 8 && !general_env;
    const bool table = !linear && !striped;
    if (table && !d_zero_page_) {
        if (is_capturing(s)) return SPECKV_ERR_INVAL;
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&d_zero_page_), kPageSize));
        HIP_TRY(hipMemset(d_zero_page_, 0, kPageSize));
    }
    if (striped) n_tiles = mx4_striped_tiles(n_pages, a->stripe_n);
    // workgroups = splits x layers x query-row groups (each covers the 8 kv heads; one resident per CU): one round of the CUs,
    // rounded down -- 80 layers at 32k: 3 splits (240 workgroups) 0.765 of the HBM roofline, 6 splits 0.72-0.75 (profiles/r05_mx4.txt)
    const uint32_t rows = n_layers * L.num_heads;
    const uint32_t columns = n_layers * ((g + 7u) / 8u);
    uint32_t want = std::max(1u, (table ? 2u : 1u) * cus() / columns);          // (the page-table form: two workgroups per CU)
    const uint32_t min_tiles = std::min(8u, std::max(2u, n_tiles / 64u));      // per-layer calls: see attend_fp8
    want = std::min(want, std::max(1u, n_tiles / min_tiles));
    if (tuning().attend_splits > 0) want = static_cast<uint32_t>(tuning().attend_splits);
    const EvenSplit es = even_split(n_tiles, std::max(1u, std::min(want, 2048u)));
    AttendArgs k{};
    // several layers of one long sequence: the stream form -- all tiles of the call in layer-major order cut into one equal piece
    // per CU (80 layers x 3 splits of the fixed grid occupy 240 CUs of 256; profiles/r05_mx4.txt) -- from 28k context (round 6: below it
    // the fixed grid is ahead, 80 layers x 2k 0.48 (stream) against 0.55, 4k 0.61 / 0.66, 12k 0.77 / 0.83, 24k 0.81 / 0.82, 32k 0.83 / 0.80,
    // 64k 0.84 / 0.75; 70 layers x 4k 0.56 / 0.66, 100 layers x 4k 0.65 / 0.72; profiles/r06_layers_by_context.txt)
    const uint32_t zgroups = (g + 7u) / 8u;
    const uint64_t total_tiles = static_cast<uint64_t>(n_layers) * n_tiles;
    const int32_t stream_knob = tuning().attend_stream;                  // N > 0: that many pieces (tests cut small calls oddly), -1: never
    const uint32_t stream_wgs = stream_knob > 0 ? static_cast<uint32_t>(stream_knob) : cus() / zgroups;
    const bool stream = linear && n_layers >= 2u && (n_pages & 15u) == 0u && tuning().attend_splits <= 0 && stream_wgs >= 1u &&
                        total_tiles >= stream_wgs && total_tiles / stream_wgs <= 0xFFFFFFFFull &&
                        (stream_knob > 0 || (stream_knob == 0 && total_tiles >= 16ull * stream_wgs && es.n_splits > 1u && n_tiles >= kStreamMinTiles));      // (a fixed grid of whole layers writes final rows: no partials, no merge)
    if (stream) {
        k.stream.n_wgs = stream_wgs;
        k.stream.len = static_cast<uint32_t>(total_tiles / stream_wgs);
        k.stream.rem = static_cast<uint32_t>(total_tiles % stream_wgs);
        k.stream.max_slots = 1;
        for (uint32_t l = 0; l < n_layers; ++l) k.stream.max_slots = std::max(k.stream.max_slots, attend_stream_count(l, n_tiles, k.stream.len, k.stream.rem));
    }
    const uint32_t n_splits = stream ? k.stream.max_slots : es.n_splits, tiles_per_split = es.tiles_per_split;       // (stream: slots per row)
    const size_t acc_bytes = static_cast<size_t>(rows) * n_splits * 16 * 128 * sizeof(float);
    const size_t ml_bytes = static_cast<size_t>(rows) * n_splits * 32 * sizeof(float);
    uint8_t* buf = static_cast<uint8_t*>(scratch(s_attn_, acc_bytes + ml_bytes, s));
    if (!buf) return SPECKV_ERR_NOMEM;
    k.entries = a->d_entries;
    k.k_first = k_first;
    k.v_first = v_first;
    k.layer_stride = layer_stride;
    k.n_pages = n_pages;
    k.heads = L.num_heads;
    k.g = g;
    k.n_splits = n_splits;
    k.tiles_per_split = tiles_per_split;
    k.q16 = static_cast<const uint16_t*>(d_q_f16);
    k.scale_log2e = sm_scale * 1.4426950408889634f;
    k.lin_base = linear ? a->linear_base : nullptr;
    if (table) k.table_form = 1u;
    if (striped) {
        k.stripe_bases = a->d_stripe;
        k.stripe_n = a->stripe_n;
        k.stripe_magic = static_cast<uint32_t>((1ull << 32) / a->stripe_n + 1u);
    }
    k.zero_page = d_zero_page_;
    k.part_acc = reinterpret_cast<float*>(buf);
    k.part_ml = reinterpret_cast<float*>(buf + acc_bytes);
    if (n_splits == 1u && !stream) { k.direct_out = d_out; k.direct_lse = d_lse; }      // no merge launch
    HIP_TRY(launch_attend_mx4(k, n_layers, d_out, d_lse, st));
    note_use(a, s);
    if (!s) RC_TRY(wait_stream());
    return SPECKV_OK;
}

} // namespace speckv
