// cxl-speckv_amd/csrc/attend_mx4.hip -- decode attention straight from MXFP4 pool records (scheme 5: OCP MX v1.0 elements and
// scales, E2M1 with one E8M0 scale per 32; BASELINE configs[4] "int4/fp8 KV compression path (CDNA4 fp8 MFMA dequant), 4:1 ratio";
// no reference counterpart, parity is against oracle/orc_attend_mx4).
//
// The record (oracle/speckv_oracle.h): a 4 KiB page is two positions x 8 kv heads x 128 channels; byte i of the 1024 nibble bytes
// holds channel i of position 0 (low nibble) and of position 1 (high), one E8M0 code per 16 bytes.  A head's row of a page is
// therefore 128 contiguous bytes [channel][position] + 8 codes, and an MX block = 16 channels of both positions = 16 bytes.
//
// q.K^T runs on the instruction the format was made for: v_mfma_scale_f32_16x16x128_f8f6f4 takes 32 k values per lane and one
// E8M0 scale per lane and operand -- one 16-byte block of the record and its code, exactly as they lie there.  Its k index
// therefore runs over (channel, position) pairs; the query operand meets it interleaved with zeros:
//     A row  = page r of the tile's 16,  k = 2 * channel + position          (two instructions: channels 0..63, 64..127)
//     B col  = (query row q, which position w),  B[k] = q[channel] if position == w else 0     (e4m3, MXFP8 blocks of 16 channels)
//     D[r][(q, w)] = score of query row q against position 2r + w
// With 8 query rows per kv head (GQA 8) all 16 columns are live and every lane ends up with the scores of its OWN query row:
// lane (c = 8w + q, kb = lane / 16) holds pages 4kb .. 4kb+3 -- the softmax never crosses lanes except for its row maximum / sum.
//
// p.V contracts over positions, which the block-scaled instruction cannot do from this layout; it runs on v_mfma_f32_16x16x32_f16.
// V is widened by v_cvt_scalef32_pk_f16_fp4: ONE instruction turns a byte -- channel d of both positions of a page -- into the
// f16 pair (d @ position 0, d @ position 1) times the block's scale, which IS an operand register of that MFMA (two consecutive
// k slots of one row).  No shuffles, no separate scaling.  The weights meet it zero-padded the same way as the query:
//     A row  = channel 8c + s of MFMA s (lane c takes bytes 8c .. 8c+7 of the head's row), k slot (2i, 2i+1) = page 4kb + i, position 0 | 1
//     B col  = (q, w): slot pair (p, 0) for w = 0, (0, p) for w = 1 -- the lane's own four weights
//     D      = the sum over the positions of parity w; the two parities are added once, in the epilogue.
//
// Operand maps as measured on the hardware (profiles/probes/mxprobe.hip, mxprobe2.hip; lane = (c = lane%16, kb = lane/16)):
//   e2m1 operand  row/col c, k = 32 kb + nibble (low nibble of byte 0 first), registers 0..3
//   e4m3 operand  row/col c, bytes 0..15 -> k = 16 kb + i, bytes 16..31 -> k = 64 + 16 kb + (i-16)
//   scale operand byte 0 of lane (c, kb) scales (row/col c, k block kb), factor 2^(code-127)
//   D             lane (c, kb) register r = D[row 4 kb + r][col c]
//
// Memory: the texture path of a CU handles one wave-level load instruction per ~25-30 cycles whatever its width (measured on the
// first version of this kernel, which kept its tiles in registers: 33 narrow loads per 8.7 KB tile, texture addresser 75 % busy
// at 0.47 of the HBM roofline, profiles/r05_mx4.txt).  So every byte arrives by LDS-DMA in full 1 KiB instructions: per wave and
// 32-position tile of its two heads 4 for K, 4 for V, 1 + 1 for the codes -- 10 instructions for 8704 bytes -- two tiles deep per
// wave; the operands are then read from LDS.  All vector-memory traffic of the loop is inline assembly with explicit counters.
#include "kernels.hpp"
#include <atomic>
#include <type_traits>

#ifndef MX4_CODES_POLICY
#define MX4_CODES_POLICY ""        // the 64 bytes of codes of a record are read by all four waves of the workgroup: default cache policy (the rows: nt)
#endif

namespace speckv {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef int v8i __attribute__((ext_vector_type(8)));

namespace {

#define MX_GP(T, p) ((const T __attribute__((address_space(1)))*)(reinterpret_cast<uintptr_t>(p)))
// Records in the pool are tile-planar (kernels.hpp): 16 records of a run = 16 nibble rows (1024 B each) + 16 code rows (64 B each)
// = 136 whole cache lines.  A 16-page tile of the kernel that starts at record R of its run reads rows R .. R+15; with R a multiple
// of 16 (every tile-aligned range of a layout whose regions are multiples of 16 pages) that is ONE storage tile: the ten 1 KiB
// requests below touch exactly its 136 lines, and nothing else (round 5's 1152-byte slots: 144).  Any other phase R % 16 works
// too: rows past the storage tile's 16th continue in the next one (lane offsets + one tile's codes, tile_offsets).
constexpr uint32_t kWavesPerWg = 4;                          // two heads per wave: a workgroup covers the 8 kv heads -- the four waves that read a
                                                             // record's 64 bytes of codes then sit on one CU (one L2): measured with two workgroups per
                                                             // record, each code line came from HBM up to four times (profiles/r05_mx4.txt)
// per wave and stage: K rows [16 pages][256 B] | V rows [16 pages][256 B] | K codes [16][16 B] | V codes [16][16 B]
constexpr uint32_t kStK = 0, kStV = 4096, kStKC = 8192, kStVC = 8448, kStage = 8704;
#ifndef MX4_STAGES
#define MX4_STAGES 3          // (2: two workgroups per CU fit; measured 0.71-0.75 of the HBM roofline against 0.73-0.77 with 3 and one workgroup per CU)
#endif
constexpr uint32_t kStagesDefault = MX4_STAGES;              // tiles a wave keeps in LDS: the one it works on and kStages - 1 on their way
constexpr uint32_t kTabEnt = 2u * 512u;                      // page-table form: the 32 entries (16 K pages, 16 V pages) of a tile, for the two tiles ahead

template <typename T> __device__ __forceinline__ T* uniform_ptr(T* p)
{
    const uint64_t v = reinterpret_cast<uint64_t>(p);
    const uint32_t lo = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(v)), hi = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(v >> 32));
    return reinterpret_cast<T*>((static_cast<uint64_t>(hi) << 32) | lo);
}
// LDS-DMA: 16 (4) bytes per lane from base + voff to LDS address lds_dst + 16 (4) * lane
__device__ __forceinline__ void dma16(uint32_t lds_dst, const uint8_t* base, uint32_t voff)
{
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3 nt\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(lds_dst), "v"(voff), "s"(base) : "memory");
}
__device__ __forceinline__ void dma4(uint32_t lds_dst, const uint8_t* base, uint32_t voff)
{
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %2, %3 nt\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(lds_dst), "v"(voff), "s"(base) : "memory");
}
// the same with a full address per lane (page-table form: every row of a tile may lie anywhere)
__device__ __forceinline__ void dma16v(uint32_t lds_dst, const uint8_t* addr)
{
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, off nt\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(lds_dst), "v"(addr) : "memory");
}
__device__ __forceinline__ void dma4v(uint32_t lds_dst, const uint8_t* addr)
{
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %2, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(lds_dst), "v"(addr) : "memory");
}
// a region's share of a tile in one statement: four 1 KiB pieces of rows + the codes; M0 saved and restored once
__device__ __forceinline__ void dma_region(uint32_t lds_rows, uint32_t lds_codes, const uint8_t* base, const uint32_t (&goff)[4], uint32_t goffc)
{
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\t"
                 "s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, %8 nt\n\t"
                 "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %4, %8 nt\n\t"
                 "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %5, %8 nt\n\t"
                 "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %6, %8 nt\n\t"
                 "s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %7, %8" MX4_CODES_POLICY "\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(lds_rows), "s"(lds_codes), "v"(goff[0]), "v"(goff[1]), "v"(goff[2]), "v"(goff[3]), "v"(goffc), "s"(base) : "memory", "scc");
}
__device__ __forceinline__ float max_over_kb(float v)
{
    const uint32_t u = __float_as_uint(v);
    const auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    const uint32_t m = __float_as_uint(fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1])));
    const auto b = __builtin_amdgcn_permlane32_swap(m, m, false, false);
    return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}
__device__ __forceinline__ float sum_over_kb(float v)
{
    const uint32_t u = __float_as_uint(v);
    const auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    const uint32_t m = __float_as_uint(__uint_as_float(a[0]) + __uint_as_float(a[1]));
    const auto b = __builtin_amdgcn_permlane32_swap(m, m, false, false);
    return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}
template <uint32_t N> __device__ __forceinline__ void wait_all_but()
{
    static_assert(N == 15u || N == 25u, "vmcnt immediate");
    if (N == 15u) asm volatile("s_waitcnt vmcnt(15)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(25)" ::: "memory");
}
// the lane 8 further on in its row of 16 (the other parity of the same query row): DPP row_ror:8
__device__ __forceinline__ float other_parity(float v)
{
    return __uint_as_float(static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(__float_as_uint(v)), 0x128, 0xF, 0xF, false)));
}
// an fp16 query element as the MX conversion sees it: NaN counts as 0, inf as 65504 (oracle: orc_quantize_rows_mxfp8)
__device__ __forceinline__ float q_clean(uint32_t half_bits)
{
    const float x = static_cast<float>(__builtin_bit_cast(_Float16, static_cast<uint16_t>(half_bits)));
    return (x == x) ? fminf(fmaxf(x, -65504.0f), 65504.0f) : 0.0f;
}

} // namespace

// FORM 0: records in one run (record p = record p of the tile-planar run at lin_base; never-written records are zero bytes = zeros with code 0)
// FORM 1: striped regularly over 1..8 runs (AttendArgs::stripe_bases; the record of page p = record p / n of the run bases[p % n]).  The
//         range's pages are taken by residue CLASS (mx4_class_tiles): class c = the pages j of the range with j % n == c, every
//         class in tiles of 16 -- a tile = 16 pages n apart = 16 CONSECUTIVE records of one run for K and of one run for V, so
//         it is fetched exactly as in form 0 (one scalar base, the same ten instructions): attention does not care in which order
//         it meets the positions.  (The first striped form took the pages in order and asked seven runs for two or three records
//         each, per tile and region, with an address per lane: 0.64 of the roofline against 0.77 for form 0; this one 0.74:
//         profiles/r05_mx4.txt.)  Every class gets the tile count of the largest; rows and tiles past a class's end are masked
//         and fetch the class's last record again.
// FORM 2: no regular placement, or a last tile that would leave the layer's region: every record address from its page-table
//         entry, clamped to the range (never-written pages read the zero page); staged through registers, one tile at a time
// FORM 3: FORM 0 as a STREAM (AttendArgs::stream, many layers of one sequence; see k_attend_int4_wg8): the launch's tiles in
//         layer-major order are cut into n_wgs equal pieces, one per workgroup = one per CU.  80 layers x 3 splits of the fixed
//         grid fill 240 of 256 CUs; 256 pieces of 320 tiles fill them all.  Across a layer boundary the DMA pipeline goes on, the
//         finished layer's partial is stored and the next layer's query rows are loaded.  grid (n_wgs, 1, query-row groups).
// grid (splits, rows [, query-row groups of 8]); a workgroup = 4 waves = the 8 kv heads.
// HALVES = 2 (form 0, batches whose sequences are one split each and no more than the CUs): 8 waves -- the workgroup's run of tiles cut
//         in two, waves 4..7 take the second half with stages of their own (two per wave: 136 KiB), and at the end their (m, l,
//         accumulators) cross over through the LDS the stages no longer need and waves 0..3 store the merged rows.  A sequence of a
//         batch then has two pipelines running and two fills overlapping instead of one -- what a short context is made of
//         (256 x 1k: a third of the launch is fill and drain) -- and still needs no partials and no merge launch.
template <int FORM, int HALVES = 1>
__global__ __launch_bounds__(64 * kWavesPerWg * HALVES) __attribute__((amdgpu_waves_per_eu((FORM == 2 || HALVES == 2) ? 2 : kStagesDefault > 2u ? 1 : 2, (FORM == 2 || HALVES == 2) ? 2 : kStagesDefault > 2u ? 1 : 2))) void k_attend_mx4(AttendArgs a)
{
    constexpr uint32_t kStages = HALVES == 2 ? 2u : kStagesDefault;
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_dyn[];                  // kWavesPerWg x 2 stages (68 KiB: beyond the static limit)
    // (the page-table form: two stages per wave + two small buffers of page-table entries, two workgroups per CU: below)
    constexpr uint32_t kWaveLds = FORM == 2 ? 2u * kStage + kTabEnt : kStages * kStage;
    uint8_t (*lds)[kWaveLds] = reinterpret_cast<uint8_t (*)[kWaveLds]>(lds_dyn);
    __shared__ uint64_t s_bases[8];
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave8 = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t wave = wave8 & 3u, half = wave8 >> 2;                 // (half: HALVES == 2 only)
    const uint32_t c = lane & 15u, kb = lane >> 4;
    const uint32_t w = c >> 3, ql = c & 7u;                              // this lane's column: position parity w, query row ql of the group
    const uint32_t q = blockIdx.z * 8u + ql;                             // query row inside the kv head's g rows
    constexpr bool STREAM = FORM == 3, CLS = FORM == 1;
    const uint32_t split = a.rows_first ? blockIdx.y : blockIdx.x;       // (rows_first: batches of members of different lengths, see launch_attend_fp8_batch)
    uint32_t layer = a.rows_first ? blockIdx.x : blockIdx.y;             // batch form: the sequence index
    if (a.rows_first && layer >= a.rows_real) return;                    // (padding row: see AttendArgs::rows_real)
    if (a.seqs && a.order) {                                             // (workgroup-uniform; several layers in one launch: y = layer x sequence)
        const uint32_t li = a.batch_n_seq ? layer / a.batch_n_seq : 0u;
        layer = li * a.batch_n_seq + a.order[layer - li * a.batch_n_seq];
    }
    // stream form: this workgroup's piece = `count` tiles from tile `ct` of layer `layer` on; its partial of that layer is the
    // layer's slot-th (slots count from the piece that holds the layer's first tile)
    uint32_t ct = 0u, count = 0u, slot = 0u;
    if (STREAM) {
        const uint32_t nt = (a.n_pages + 15u) / 16u;
        const uint64_t g0 = attend_stream_begin(blockIdx.x, a.stream.len, a.stream.rem), g1 = attend_stream_begin(blockIdx.x + 1u, a.stream.len, a.stream.rem);
        layer = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(g0 / nt));                 // (the divisions run in the vector ALU)
        ct = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(g0 - static_cast<uint64_t>(layer) * nt));
        count = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(g1 - g0));
        slot = __builtin_amdgcn_readfirstlane(blockIdx.x - attend_stream_wg_of(static_cast<uint64_t>(layer) * nt, a.stream.len, a.stream.rem));
    }
    const uint32_t h0 = wave * 2u;                                       // first of this wave's two kv heads
    uint64_t row0 = static_cast<uint64_t>(layer) * a.heads + h0;         // query / output row block of head 0 (head 1: + 1)
    uint64_t part0 = row0 * a.n_splits + split, part_step = a.n_splits;  // partials of head 0 | head 1: part0, part0 + part_step
    uint32_t my_splits = a.n_splits;
    if (STREAM) { part0 = row0 * a.stream.max_slots + slot; part_step = a.stream.max_slots; my_splits = 0u; }      // (never the direct output)
    int32_t tail = -1;                                                   // this sequence's row in the tail arrays (split 0 folds it in)
    if (a.seqs) {                                                        // workgroup-uniform: per-sequence geometry
        uint32_t seq = layer;
        if (a.batch_n_seq) {                                             // several layers in one launch: y = layer x sequence
            const uint32_t li = layer / a.batch_n_seq;
            seq = layer - li * a.batch_n_seq;
            a.batch_layer += li;
        }
        // (not the page-table form: it sits at the register limit of two waves per SIMD; the engine sends its tails through k_attend_fold_tail)
        if (FORM != 2 && a.tail_k && split == 0u) tail = a.tail_idx ? a.tail_idx[seq] : static_cast<int32_t>(seq);
        const AttendSeq sq = a.seqs[seq];
        if (split >= sq.n_splits) {
            if (sq.n_splits == 0u && split == 0u && blockIdx.z == 0u && a.direct_out && a.direct_per_seq == 2u) {
                attend_zero_rows(a.direct_out, a.direct_lse, a.g, row0, lane);
                attend_zero_rows(a.direct_out, a.direct_lse, a.g, row0 + 1u, lane);
            }
            return;
        }
        a.lin_base = uniform_ptr(sq.lin_base);
        a.k_first = sq.k_first + static_cast<uint64_t>(a.batch_layer) * sq.layer_pages;
        a.v_first = sq.v_first + static_cast<uint64_t>(a.batch_layer) * sq.layer_pages;
        a.n_pages = sq.n_pages;
        a.tiles_per_split = sq.tiles_per_split;
        my_splits = sq.n_splits;
        part0 = sq.part_base + static_cast<uint64_t>(h0) * sq.n_splits + split;
        part_step = sq.n_splits;
        if (FORM == 2) a.entries = reinterpret_cast<const PageEntry*>(sq.lin_base);
        if (FORM == 1) {
            a.stripe_bases = sq.stripe_bases;
            a.stripe_n = sq.stripe_n;
            a.stripe_magic = sq.stripe_n > 1u ? static_cast<uint32_t>((1ull << 32) / sq.stripe_n + 1u) : 0u;
        }
    } else {
        a.k_first += static_cast<uint64_t>(layer) * a.layer_stride;
        a.v_first += static_cast<uint64_t>(layer) * a.layer_stride;
    }
    if (CLS) {
        if (threadIdx.x < 8u) s_bases[threadIdx.x] = a.stripe_bases[threadIdx.x];
        __syncthreads();
    }

    const uint32_t cls_n = CLS ? a.stripe_n : 1u, cls_magic = CLS ? static_cast<uint32_t>(a.stripe_magic) : 0u, cls_pages = a.n_pages;
    const uint32_t cls_m = CLS ? mx4_class_tiles(cls_pages, cls_n) : 0u;           // tiles per class
    const uint32_t n_tiles = CLS ? cls_n * cls_m : (a.n_pages + 15u) / 16u;
    uint32_t t0 = STREAM ? 0u : split * a.tiles_per_split;               // (stream: positions in the piece, 0 .. count)
    uint32_t t1 = STREAM ? count : min(t0 + a.tiles_per_split, n_tiles);
    if (HALVES == 2) {                                                   // waves 0..3: the first half of the run, waves 4..7: the second
        const uint32_t tm = t0 + (t1 - t0 + 1u) / 2u;
        if (half) t0 = tm; else t1 = tm;
    }
    float m_run[2] = {-INFINITY, -INFINITY}, l_run[2] = {0.0f, 0.0f};
    f32x4 acc[2][8];
#pragma unroll
    for (int hh = 0; hh < 2; ++hh)
#pragma unroll
        for (int s = 0; s < 8; ++s) acc[hh][s] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};

    // the tail position (AttendArgs::tail_k): per head its score in the kernel's log2 units and the lane's 32 V values (channels
    // 32 kb .. + 31 of the head, fp16 pairs) -- fetched beside the query rows, so that the epilogue waits for nothing
    float tail_sc[2] = {-INFINITY, -INFINITY};
    u32x4 tail_v[2][4];
#pragma unroll
    for (int hh = 0; hh < 2; ++hh)
#pragma unroll
        for (int j = 0; j < 4; ++j) tail_v[hh][j] = u32x4{0u, 0u, 0u, 0u};
    // ---- the two parities of a query row are added; the lanes of parity 0 write the partial (or, single split: the final) result
    auto store_rows = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            const float l_half = sum_over_kb(l_run[hh]);
            float l_tot = l_half + other_parity(l_half);
            f32x4 o[8];
#pragma unroll
            for (int s = 0; s < 8; ++s)
#pragma unroll
                for (int r = 0; r < 4; ++r) o[s][r] = acc[hh][s][r] + other_parity(acc[hh][s][r]);
            if (tail >= 0) {                                             // (workgroup-uniform) one more position: its score and V row were fetched beside the query
                const float m_new = fmaxf(m_run[hh], tail_sc[hh]);
                const float fa = (m_run[hh] == -INFINITY) ? 0.0f : __builtin_amdgcn_exp2f(m_run[hh] - m_new), fb = __builtin_amdgcn_exp2f(tail_sc[hh] - m_new);
                m_run[hh] = m_new;
                l_tot = l_tot * fa + fb;
#pragma unroll
                for (int s = 0; s < 8; ++s)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        // o[s][r] is output channel 32 kb + 8 r + s: half (8 r + s) of the lane's 32 V values = word r of piece r' ... packed two per dword
                        const uint32_t wd = tail_v[hh][r][s >> 1];
                        const float vch = static_cast<float>(__builtin_bit_cast(_Float16, static_cast<uint16_t>((s & 1) ? (wd >> 16) : (wd & 0xFFFFu))));
                        o[s][r] = o[s][r] * fa + fb * vch;
                    }
            }
            if (w != 0u || q >= a.g) continue;
            const uint64_t row = row0 + hh;
            if (a.direct_out && (!a.direct_per_seq || my_splits == 1u)) {
                const float ws = l_tot > 0.0f ? 1.0f / l_tot : 0.0f;
                float* dst = a.direct_out + (row * a.g + q) * 128u + 32u * kb;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    *reinterpret_cast<f32x4*>(dst + 8 * r) = f32x4{o[0][r], o[1][r], o[2][r], o[3][r]} * ws;
                    *reinterpret_cast<f32x4*>(dst + 8 * r + 4) = f32x4{o[4][r], o[5][r], o[6][r], o[7][r]} * ws;
                }
                if (a.direct_lse && kb == 0)
                    a.direct_lse[row * a.g + q] = l_tot > 0.0f ? (m_run[hh] + log2f(l_tot)) * 0.6931471805599453f : -INFINITY;
                continue;
            }
            const uint64_t part = part0 + hh * part_step;
            if (kb == 0) {
                a.part_ml[part * 32u + q] = m_run[hh];
                a.part_ml[part * 32u + 16u + q] = l_tot;
            }
            float* dst = a.part_acc + (part * 16u + q) * 128u + 32u * kb;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                *reinterpret_cast<f32x4*>(dst + 8 * r) = f32x4{o[0][r], o[1][r], o[2][r], o[3][r]};
                *reinterpret_cast<f32x4*>(dst + 8 * r + 4) = f32x4{o[4][r], o[5][r], o[6][r], o[7][r]};
            }
        }
    };

    if (t0 < t1) {                                                       // wave-uniform
        const uint32_t lbase = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(reinterpret_cast<uintptr_t>(&lds[HALVES == 2 ? wave8 : wave][0])));
        const uint8_t* const lptr = &lds[HALVES == 2 ? wave8 : wave][0];
        // ---- staging: DMA instruction i (0..3) of a region fills page rows 4i .. 4i+3: lane l -> row 4i + l/16, 16-byte slot l%16,
        // which holds piece (slot ^ row) of the two heads' 256 bytes -- the XOR spreads the operand reads below over the banks
        const uint32_t srow = lane >> 4, sslot = lane & 15u;
        // byte offset of the lane's piece from the nibble row of the tile's FIRST record (scalar base), for a tile that starts on a
        // storage-tile boundary (phase 0): row r at r * 1024; its codes at 16384 + r * 64
        uint32_t goff[4];
#pragma unroll
        for (uint32_t i = 0; i < 4; ++i) goff[i] = (4u * i + srow) * 1024u + h0 * 128u + ((sslot ^ ((4u * i + srow) & 15u)) * 16u);
        const uint32_t goffc = kMx4CodePlane + (lane >> 2) * 64u + h0 * 8u + (lane & 3u) * 4u;      // codes: lane l -> page l/4, dword l%4 of the two heads' 16
        // ... and for a tile whose first record is slot `ph` of its storage tile, rows clamped to imax (masked rows fetch a live one):
        // rows with ph + r >= 16 lie in the next storage tile, 1024 bytes further on than r * 1024 says; a code row lies
        // 16384 - 960 * slot behind its nibble row (mx4_code_delta)
        auto issue = [&](uint32_t drows, uint32_t dcodes, const uint8_t* rt, uint32_t ph, uint32_t imax) __attribute__((always_inline)) {
            if (ph == 0u && imax == 15u) {                                       // (wave-uniform)
                dma_region(drows, dcodes, rt, goff, goffc);
                return;
            }
            uint32_t gc[4];
#pragma unroll
            for (uint32_t i = 0; i < 4; ++i) {
                const uint32_t r = min(4u * i + srow, imax);
                gc[i] = r * 1024u + (ph + r >= 16u ? 1024u : 0u) + h0 * 128u + ((sslot ^ ((4u * i + srow) & 15u)) * 16u);
            }
            const uint32_t rc = min(lane >> 2, imax);
            const uint32_t gcc = (kMx4CodePlane - 960u * ph) + rc * 64u + (ph + rc >= 16u ? kMx4CodePlane : 0u) + h0 * 8u + (lane & 3u) * 4u;
            dma_region(drows, dcodes, rt, gc, gcc);
        };
        const uint32_t kfirst = static_cast<uint32_t>(a.k_first), vfirst = static_cast<uint32_t>(a.v_first), last_pg = a.n_pages - 1u;
        const uint32_t last = t1 - 1u;
        // FORM 2 (page table): the 32 entries of a tile -- lane l < 16: K page l of the tile, lanes 16 .. 31: V page l - 16 -- are fetched
        // two tiles ahead with one load per wave, parked in LDS, and read back from there by the lanes that need them when the tile's
        // ten LDS-DMAs go out (one tile ahead, a full address per lane).  The entry load is inline assembly and waited for by hand
        // at the END of a tile's arithmetic, together with the next tile's DMAs: nothing of this is the compiler's to count.
        const uint32_t ent_lds = lbase + 2u * kStage;
        auto ent_fetch = [&](uint32_t tt) -> u32x4 {
            const uint32_t idx = lane & 31u, pg = min(16u * min(tt, last) + (idx & 15u), last_pg);
            const PageEntry* ep = a.entries + ((idx < 16u ? kfirst : vfirst) + pg);
            u32x4 e;
            asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(e) : "v"(ep) : "memory");
            return e;
        };
        // (parked RESOLVED -- {nibble row or the zero page, distance to its code row} -- so that the requests below need nothing but the
        //  address: never-written pages read zeros)
        auto ent_store = [&](uint32_t eb, const u32x4 e) {                                 // (lanes 32 .. 63 repeat lanes 0 .. 31: same bytes)
            const bool ok = e.z >= kMx4RecBytes;
            const uint64_t zp = reinterpret_cast<uint64_t>(a.zero_page);
            u32x4 r;
            r.x = ok ? e.x : static_cast<uint32_t>(zp); r.y = ok ? e.y : static_cast<uint32_t>(zp >> 32); r.z = ok ? e.w : 1024u; r.w = 0u;
            asm volatile("ds_write_b128 %0, %1" :: "v"(ent_lds + eb * 512u + (lane & 31u) * 16u), "v"(r) : "memory");
        };
        auto issue_table_tile = [&](uint32_t sbuf, uint32_t eb) __attribute__((always_inline)) {
            const uint32_t ebase = ent_lds + eb * 512u;
            const uint32_t dst = lbase + sbuf * kStage;
            // One region at a time: its five entries (the lane's four nibble rows, its code row) are read TOGETHER, one wait, five requests.
            // (Ten read -> wait -> request pairs one after the other left a hundred cycles of LDS latency in front of every request, with
            //  one wave per SIMD to hide nothing behind: the page-table form ran at 0.75 of the roofline against the linear form's 0.83.
            //  Rolled over the two regions: the kernel sits near the 256 registers of two waves per SIMD.)
#pragma unroll 1
            for (uint32_t rg = 0; rg < 2u; ++rg) {
                typedef uint32_t u32x2e __attribute__((ext_vector_type(2)));
                u32x2e ra[4];
                u32x4 ec;
                const uint32_t eaddr = ebase + (rg * 16u + srow) * 16u, caddr = ebase + (rg * 16u + (lane >> 2)) * 16u;
                asm volatile("ds_read_b64 %0, %5\n\tds_read_b64 %1, %5 offset:64\n\tds_read_b64 %2, %5 offset:128\n\tds_read_b64 %3, %5 offset:192\n\t"
                             "ds_read_b128 %4, %6\n\ts_waitcnt lgkmcnt(0)"
                             : "=&v"(ra[0]), "=&v"(ra[1]), "=&v"(ra[2]), "=&v"(ra[3]), "=&v"(ec) : "v"(eaddr), "v"(caddr) : "memory");
#pragma unroll
                for (uint32_t i = 0; i < 4u; ++i) {
                    const uint8_t* r = reinterpret_cast<const uint8_t*>(static_cast<uint64_t>(ra[i].x) | (static_cast<uint64_t>(ra[i].y) << 32));
                    dma16v(dst + (rg ? kStV : kStK) + 1024u * i, r + h0 * 128u + ((sslot ^ ((4u * i + srow) & 15u)) * 16u));
                }
                const uint8_t* rc = reinterpret_cast<const uint8_t*>(static_cast<uint64_t>(ec.x) | (static_cast<uint64_t>(ec.y) << 32));
                dma4v(dst + (rg ? kStVC : kStKC), rc + ec.z + h0 * 8u + (lane & 3u) * 4u);
            }
        };
        // stream form: the next tile each region (K, V) will ask for -- its first page, its tile number in the layer, tiles left
        // in the piece; the requests run ahead of the arithmetic across layer boundaries, a request past the piece repeats the last
        uint32_t rq_page[2] = {kfirst + 16u * ct, vfirst + 16u * ct}, rq_ct[2] = {ct, ct}, rq_left[2] = {count, count};
        const uint32_t layer_gap = static_cast<uint32_t>(a.layer_stride) - 16u * n_tiles;   // from the end of one layer's region to the next one's start
        // class form: the next tile each region will ask for = (class, tile of the class), tiles left in the split
        uint32_t cls0 = 0u, cq_cls[2] = {0u, 0u}, cq_m[2] = {0u, 0u}, cq_left[2] = {t1 - t0, t1 - t0};
        const uint32_t jq = CLS ? cls_pages / cls_n : 0u, jr = CLS ? cls_pages - jq * cls_n : 0u;      // class c holds jq + (c < jr) pages
        // ... and what the class a region is in resolves to -- its run's base, its first record there, its page count -- looked up when the
        // region ENTERS a class (147 tiles apart at 32k x 7 runs), not per request: an LDS read of the run base, the wait the compiler
        // puts behind it (which also waits for the tile's operand reads) and a handful of vector-to-scalar moves in front of every
        // request cost the class form 3-4 points of the roofline with one wave per SIMD to hide nothing behind (round 6)
        const uint8_t* cq_base[2] = {nullptr, nullptr};
        uint32_t cq_rec0[2] = {0u, 0u}, cq_cnt[2] = {1u, 1u};
        uint32_t cq_g[2][4], cq_gc[2];                                           // the lane's request offsets at the class's phase (whole tiles: imax = 15)
        auto cls_enter = [&](uint32_t rg) __attribute__((always_inline)) {
            uint32_t cls = min(cq_cls[rg], cls_n - 1u);
            uint32_t cnt = jq + (cls < jr ? 1u : 0u);                             // pages of the class
            if (cnt == 0u) { cls = 0u; cnt = 1u; }                                // (a range of fewer pages than runs: an empty class fetches the range's first record, all masked)
            const uint32_t pg = (rg ? vfirst : kfirst) + cls;                     // the class's first page: pool pg % n, record pg / n
            const uint32_t rec0 = cls_n == 1u ? pg : __builtin_amdgcn_readfirstlane(__umulhi(pg, cls_magic)), pool = pg - rec0 * cls_n;     // (all wave-uniform)
            cq_base[rg] = uniform_ptr(reinterpret_cast<const uint8_t*>(s_bases[pool]));
            cq_rec0[rg] = rec0;
            cq_cnt[rg] = __builtin_amdgcn_readfirstlane(cnt);
            const uint32_t ph = rec0 & 15u;                                       // every tile of the class starts in this slot of its storage tile
#pragma unroll
            for (uint32_t i = 0; i < 4; ++i) {
                const uint32_t r = 4u * i + srow;
                cq_g[rg][i] = r * 1024u + (ph + r >= 16u ? 1024u : 0u) + h0 * 128u + ((sslot ^ (r & 15u)) * 16u);
            }
            const uint32_t rc = lane >> 2;
            cq_gc[rg] = (kMx4CodePlane - 960u * ph) + rc * 64u + (ph + rc >= 16u ? kMx4CodePlane : 0u) + h0 * 8u + (lane & 3u) * 4u;
        };
        if (CLS) {
            cls0 = __builtin_amdgcn_readfirstlane(t0 / cls_m);
            cq_cls[0] = cq_cls[1] = cls0;
            cq_m[0] = cq_m[1] = t0 - cls0 * cls_m;
            cls_enter(0u); cls_enter(1u);
        }
        // one region's share of a tile (rg = 0: K rows + K codes, 1: V rows + V codes) into stage `buf`: 5 DMA instructions
        auto stage = [&](uint32_t tt, uint32_t buf, uint32_t rg) __attribute__((always_inline)) {
            const uint32_t tc = min(tt, last);
            const uint32_t dst = lbase + buf * kStage;
            const uint32_t first = rg ? vfirst : kfirst;
            const uint32_t drows = dst + (rg ? kStV : kStK), dcodes = dst + (rg ? kStVC : kStKC);
            if (CLS) {
                const uint32_t cnt = cq_cnt[rg];
                const uint32_t mm = min(cq_m[rg], (cnt - 1u) >> 4);               // a tile past the class's end fetches its last one (all masked)
                const uint32_t imax = min(15u, cnt - 1u - 16u * mm);              // rows past the class's end fetch its last record
                const uint32_t R = cq_rec0[rg] + 16u * mm;                        // the tile's first record in its run
                const uint8_t* rt = cq_base[rg] + mx4_nib_off(R);                 // (scalar: the cursor is)
                if (imax == 15u) dma_region(__builtin_amdgcn_readfirstlane(drows), __builtin_amdgcn_readfirstlane(dcodes), rt, cq_g[rg], cq_gc[rg]);      // (wave-uniform)
                else issue(__builtin_amdgcn_readfirstlane(drows), __builtin_amdgcn_readfirstlane(dcodes), rt, R & 15u, imax);
                if (cq_left[rg] > 1u) {
                    --cq_left[rg];
                    if (++cq_m[rg] == cls_m) { cq_m[rg] = 0u; ++cq_cls[rg]; cls_enter(rg); }
                }
            } else if (STREAM) {
                const uint8_t* rt = uniform_ptr(a.lin_base + mx4_nib_off(rq_page[rg]));    // (wave-uniform by construction)
                issue(drows, dcodes, rt, rq_page[rg] & 15u, 15u);
                if (rq_left[rg] > 1u) {
                    --rq_left[rg];
                    rq_page[rg] += 16u;
                    if (++rq_ct[rg] == n_tiles) { rq_ct[rg] = 0u; rq_page[rg] += layer_gap; }
                }
            } else if (FORM == 0) {
                const uint64_t R = static_cast<uint64_t>(first) + 16ull * tc;
                const uint8_t* rt = a.lin_base + mx4_nib_off(R);                  // (scalar)
                issue(drows, dcodes, rt, first & 15u, 15u);
            }
        };
        // request order K(t), V(t), K(t+1), V(t+1), ...: five instructions each, so "all but the 15 youngest" is "this region has landed"
        if (FORM != 2) {
            stage(t0, 0u, 0u);
            stage(t0, 0u, 1u);
#pragma unroll
            for (uint32_t sg = 1; sg < kStages; ++sg) { stage(t0 + sg, sg, 0u); stage(t0 + sg, sg, 1u); }
        } else {
            u32x4 e0 = ent_fetch(t0), e1 = ent_fetch(t0 + 1u);
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(e0), "+v"(e1) :: "memory");
            ent_store(0u, e0);
            ent_store(1u, e1);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            issue_table_tile(0u, 0u);
        }

        // ---- query operands (MXFP8, blocks of 16 channels, zero-interleaved): lane (c = 8w + q, kb) takes channels 8kb + 32 gq + 0..7
        // (gq = 0..3) of query row q of both heads; operand [head][half hf] bytes 0..15 <- gq = 2 hf, bytes 16..31 <- gq = 2 hf + 1
        v8i QB[2][2];
        int q_code[2][2];
        auto load_query = [&]() __attribute__((always_inline)) {
            const bool live = q < a.g;
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                const uint16_t* qrow = a.q16 + ((row0 + hh) * a.g + min(q, a.g - 1u)) * 128u;
                float x[4][8];
                uint32_t code[4];
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    const u32x4 wv = *MX_GP(u32x4, qrow + 8u * kb + 32u * gq);
                    const uint32_t ws[4] = {wv.x, wv.y, wv.z, wv.w};
                    float amax = 0.0f;
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        x[gq][k] = q_clean((ws[k >> 1] >> (16 * (k & 1))) & 0xFFFFu);
                        amax = fmaxf(amax, fabsf(x[gq][k]));
                    }
                    // the block of 16 channels = this lane's 8 and lane ^ 16's 8 (kb ^ 1)
                    const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(amax), __float_as_uint(amax), false, false);
                    amax = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
                    code[gq] = (live && amax > 0.0f) ? (__float_as_uint(amax) >> 23) - 8u : 0u;            // floor(log2 amax) - 8 + 127
                }
                uint32_t qd[16];
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    const float mul = __uint_as_float((254u - code[gq]) << 23);                             // 2^(127 - code), exact
                    float v[8];
#pragma unroll
                    for (int k = 0; k < 8; ++k) v[k] = live ? fminf(fmaxf(x[gq][k] * mul, -448.0f), 448.0f) : 0.0f;
                    int p0 = __builtin_amdgcn_cvt_pk_fp8_f32(v[0], v[1], 0, false);
                    p0 = __builtin_amdgcn_cvt_pk_fp8_f32(v[2], v[3], p0, true);
                    int p1 = __builtin_amdgcn_cvt_pk_fp8_f32(v[4], v[5], 0, false);
                    p1 = __builtin_amdgcn_cvt_pk_fp8_f32(v[6], v[7], p1, true);
                    // bytes b0..b7 -> (b0, 0, b1, 0, ...) for parity 0, (0, b0, 0, b1, ...) for parity 1   (selector 0x0c = a zero byte)
                    const uint32_t sl = w ? 0x010C000Cu : 0x0C010C00u, sh = w ? 0x030C020Cu : 0x0C030C02u;
                    qd[4 * gq + 0] = __builtin_amdgcn_perm(0u, static_cast<uint32_t>(p0), sl);
                    qd[4 * gq + 1] = __builtin_amdgcn_perm(0u, static_cast<uint32_t>(p0), sh);
                    qd[4 * gq + 2] = __builtin_amdgcn_perm(0u, static_cast<uint32_t>(p1), sl);
                    qd[4 * gq + 3] = __builtin_amdgcn_perm(0u, static_cast<uint32_t>(p1), sh);
                }
#pragma unroll
                for (int hf = 0; hf < 2; ++hf) {
                    QB[hh][hf] = v8i{static_cast<int>(qd[8 * hf + 0]), static_cast<int>(qd[8 * hf + 1]), static_cast<int>(qd[8 * hf + 2]), static_cast<int>(qd[8 * hf + 3]),
                                     static_cast<int>(qd[8 * hf + 4]), static_cast<int>(qd[8 * hf + 5]), static_cast<int>(qd[8 * hf + 6]), static_cast<int>(qd[8 * hf + 7])};
                    // the lane's scale operand: hardware block kb of this instruction = channels 64 hf + 16 kb .. + 15 = the block lanes
                    // (kb' = 2 (kb & 1), 2 (kb & 1) + 1) computed as their group gq = 2 hf + (kb >> 1)
                    const uint32_t src = c + 32u * (kb & 1u);
                    const uint32_t s0 = __shfl(code[2 * hf], src), s1 = __shfl(code[2 * hf + 1], src);
                    q_code[hh][hf] = static_cast<int>((kb >> 1) ? s1 : s0);
                }
            }
        };
        load_query();
        if (tail >= 0) {                                                     // (workgroup-uniform; split 0 of a sequence that keeps a position outside the pool)
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                // the lane holds output channels 32 kb + 8 r + s of query row ql: the same 32 channels of q and k make its share of q.k
                const uint64_t trow = static_cast<uint64_t>(tail) * a.tail_stride + (static_cast<uint64_t>(a.batch_layer) * a.heads + h0 + hh) * 128u + 32u * kb;
                const uint16_t* qp = a.q16 + ((row0 + hh) * a.g + min(q, a.g - 1u)) * 128u + 32u * kb;
                float dot = 0.0f;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const u32x4 qw = *MX_GP(u32x4, qp + 8 * j), kw = *MX_GP(u32x4, a.tail_k + trow + 8 * j);
                    tail_v[hh][j] = *MX_GP(u32x4, a.tail_v + trow + 8 * j);
                    const uint32_t qq[4] = {qw.x, qw.y, qw.z, qw.w}, kk[4] = {kw.x, kw.y, kw.z, kw.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const f16x2 q2 = __builtin_bit_cast(f16x2, qq[e]), k2 = __builtin_bit_cast(f16x2, kk[e]);
                        dot = __builtin_fmaf(static_cast<float>(q2.x), static_cast<float>(k2.x), dot);
                        dot = __builtin_fmaf(static_cast<float>(q2.y), static_cast<float>(k2.y), dot);
                    }
                }
                tail_sc[hh] = sum_over_kb(dot) * a.scale_log2e;
            }
        }
        // ---- where the lane reads its operands in a stage
        uint32_t rk[2][2], rkc[2][2], rv[2][4], rvc[2][4];
#pragma unroll
        for (uint32_t hh = 0; hh < 2; ++hh) {
#pragma unroll
            for (uint32_t hf = 0; hf < 2; ++hf) {
                rk[hh][hf] = kStK + c * 256u + (((hh * 8u + hf * 4u + kb) ^ c) * 16u);                  // row = page c: one 16-byte MX block
                rkc[hh][hf] = kStKC + c * 16u + hh * 8u + hf * 4u + kb;
            }
#pragma unroll
            for (uint32_t i = 0; i < 4; ++i) {
                const uint32_t pg = 4u * kb + i;
                rv[hh][i] = kStV + pg * 256u + (((hh * 8u + (c >> 1)) ^ pg) * 16u) + (c & 1u) * 8u;      // bytes 8c .. 8c+7 of the head's row
                rvc[hh][i] = kStVC + pg * 16u + hh * 8u + (c >> 1);
            }
        }
        const bool ragged = (a.n_pages & 15u) != 0u;
        const uint32_t n_pos = 2u * a.n_pages, skip_pos = 2u * a.skip_pages;
        uint32_t cc_cls = cls0, cc_m = t0 - cls0 * cls_m;                    // class form: the tile the arithmetic is at
        const uint32_t wshift = 16u * w;
        // one tile out of stage BUF (a compile-time constant: the LDS reads then carry the stage as an immediate offset)
        auto tile_body = [&](uint32_t tile, auto buf_c) __attribute__((always_inline)) {
            const uint32_t buf = buf_c;                                   // (an integral_constant: folded; the page-table form passes a variable)
            const uint8_t* st = lptr + buf * kStage;
            // ---- K of this tile has landed (younger requests: V of this tile, K and V of the next): both heads' blocks and codes
            // to registers, and the region goes straight back to the DMA for the next-but-one tile
            if (FORM != 2) wait_all_but<10u * kStages - 5u>();
            u32x4 kx[2][2];
            uint32_t kc[2][2];
#pragma unroll
            for (int hh = 0; hh < 2; ++hh)
#pragma unroll
                for (int hf = 0; hf < 2; ++hf) {
                    kx[hh][hf] = *reinterpret_cast<const u32x4*>(st + rk[hh][hf]);
                    kc[hh][hf] = st[rkc[hh][hf]];
                }
            if (FORM != 2) {
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(kx[0][0]), "+v"(kx[0][1]), "+v"(kx[1][0]), "+v"(kx[1][1]), "+v"(kc[0][0]), "+v"(kc[0][1]), "+v"(kc[1][0]), "+v"(kc[1][1]) :: "memory");
                stage(tile + kStages, buf, 0u);
            }
            f16x8 P[2];
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                // ---- scores: two block-scaled MFMAs (channels 0..63, 64..127) over the tile's 16 pages
                f32x4 s = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                for (int hf = 0; hf < 2; ++hf) {
                    const v8i A = {static_cast<int>(kx[hh][hf].x), static_cast<int>(kx[hh][hf].y), static_cast<int>(kx[hh][hf].z), static_cast<int>(kx[hh][hf].w), 0, 0, 0, 0};
                    s = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(A, QB[hh][hf], s, 4 /* A: e2m1 */, 0 /* B: e4m3 */, 0, static_cast<int>(kc[hh][hf]), 0, q_code[hh][hf]);
                }
                float sc[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) sc[r] = s[r] * a.scale_log2e;
                if (CLS) {
                    if (cc_m + 1u == cls_m) {                                                   // wave-uniform: the class may end inside its last tile
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (cc_cls + cls_n * (16u * cc_m + 4u * kb + r) >= cls_pages) sc[r] = -INFINITY;
                    }
                } else if ((ragged && tile + 1u == n_tiles) || (a.skip_pages && tile == 0u)) {  // wave-uniform: positions beyond / in front of the range
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const uint32_t pos = tile * 32u + 2u * (4u * kb + r) + w;
                        if (pos >= n_pos || pos < skip_pos) sc[r] = -INFINITY;
                    }
                }
                // ---- online softmax of the lane's query row (its positions sit in the 8 lanes {c, c ^ 8} x kb).  The reference
                // maximum only moves when a tile's maximum passes it by more than 2^8 (weights <= 256: no concern in f16, the
                // accumulators are fp32): the running maximum of a decode step settles within the first tiles of a split, and the
                // 32 accumulator registers per head are then left alone.
                float mx = fmaxf(fmaxf(sc[0], sc[1]), fmaxf(sc[2], sc[3]));
                mx = max_over_kb(mx);
                mx = fmaxf(mx, other_parity(mx));
                if (__builtin_expect(__ballot(mx > m_run[hh] + 8.0f) != 0ull, 0)) {         // (wave-uniform; -inf + 8 = -inf: the first tile comes here)
                    const float m_new = (mx > m_run[hh] + 8.0f) ? mx : m_run[hh];
                    const float f = (m_new == -INFINITY) ? 1.0f : __builtin_amdgcn_exp2f(m_run[hh] - m_new);
                    m_run[hh] = m_new;
                    l_run[hh] *= f;
#pragma unroll
                    for (int t = 0; t < 8; ++t) acc[hh][t] *= f;
                }
                const float m_use = (m_run[hh] == -INFINITY) ? 0.0f : m_run[hh];
                float p[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) { p[r] = __builtin_amdgcn_exp2f(sc[r] - m_use); l_run[hh] += p[r]; }
                // weights as the B operand: slot pair i = (p_i, 0) for parity 0, (0, p_i) for parity 1
                u32x4 pw;
#pragma unroll
                for (int r = 0; r < 4; ++r) pw[r] = static_cast<uint32_t>(__builtin_bit_cast(uint16_t, static_cast<_Float16>(p[r]))) << wshift;
                P[hh] = __builtin_bit_cast(f16x8, pw);
            }
            // ---- V of this tile has landed (younger: K and V of the next tile, K of the one after): both heads' pieces and codes
            if (FORM != 2) wait_all_but<10u * kStages - 5u>();
            u32x2 vx[2][4];
            uint32_t vcode[2][4];
#pragma unroll
            for (int hh = 0; hh < 2; ++hh)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    vx[hh][i] = *reinterpret_cast<const u32x2*>(st + rv[hh][i]);
                    vcode[hh][i] = st[rvc[hh][i]];
                }
            if (FORM != 2) {
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(vx[0][0]), "+v"(vx[0][1]), "+v"(vx[0][2]), "+v"(vx[0][3]), "+v"(vx[1][0]), "+v"(vx[1][1]), "+v"(vx[1][2]), "+v"(vx[1][3]),
                             "+v"(vcode[0][0]), "+v"(vcode[0][1]), "+v"(vcode[0][2]), "+v"(vcode[0][3]), "+v"(vcode[1][0]), "+v"(vcode[1][1]), "+v"(vcode[1][2]), "+v"(vcode[1][3]) :: "memory");
                stage(tile + kStages, buf, 1u);
            }
            // ---- out^T += V^T . P^T: a byte of the record = channel d of both positions of a page -> one operand register
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                float vsc[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) vsc[i] = __uint_as_float(vcode[hh][i] << 23);          // the block's E8M0 code as a float's exponent field
#define MX_PV(S, WORD, SEL)                                                                                                          \
    {                                                                                                                                \
        const u32x4 vw = {__builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_f16_fp4(vx[hh][0].WORD, vsc[0], SEL)),        \
                          __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_f16_fp4(vx[hh][1].WORD, vsc[1], SEL)),        \
                          __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_f16_fp4(vx[hh][2].WORD, vsc[2], SEL)),        \
                          __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_f16_fp4(vx[hh][3].WORD, vsc[3], SEL))};       \
        acc[hh][S] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, vw), P[hh], acc[hh][S], 0, 0, 0);               \
    }
                MX_PV(0, x, 0) MX_PV(1, x, 1) MX_PV(2, x, 2) MX_PV(3, x, 3) MX_PV(4, y, 0) MX_PV(5, y, 1) MX_PV(6, y, 2) MX_PV(7, y, 3)
#undef MX_PV
            }
            if (CLS && ++cc_m == cls_m) { cc_m = 0u; ++cc_cls; }
        };
        if (FORM == 2) {
            // tile in stage buf: its DMAs have landed; the next tile's go out (its entries are in LDS), the entries of the tile after that
            // are asked for; the arithmetic; then ONE wait for both, and the entries move into the LDS buffer this tile's had.
            // (ONE copy of the tile's arithmetic with the stage as a variable: two copies with constant stages left the scheduler room to
            //  overlap them, and the kernel -- at the 256 registers of two waves per SIMD -- spilled)
            uint32_t tbuf = 0u;
#pragma unroll 1
            for (uint32_t tile = t0; tile < t1; ++tile) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (tile + 1u < t1) issue_table_tile(tbuf ^ 1u, tbuf ^ 1u);
                u32x4 e2 = ent_fetch(tile + 2u);
                tile_body(tile, tbuf);
                asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)" : "+v"(e2) :: "memory");
                ent_store(tbuf, e2);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                tbuf ^= 1u;
            }
        } else if (STREAM) {
            // the piece, tile by tile; at the end of a layer its partial goes out, the state starts over and the next layer's query
            // rows come in (their loads are the compiler's: it waits for everything in flight once, a pipeline fill per boundary)
            uint32_t done = 0u;
            auto next = [&]() __attribute__((always_inline)) -> bool {
                if (++done == count) return true;
                if (++ct == n_tiles) {
                    store_rows();
                    ct = 0u;
                    ++layer;
                    row0 += a.heads;
                    part0 = row0 * a.stream.max_slots;                    // this piece holds the new layer's first tile: slot 0
#pragma unroll
                    for (int hh = 0; hh < 2; ++hh) {
                        m_run[hh] = -INFINITY;
                        l_run[hh] = 0.0f;
#pragma unroll
                        for (int t = 0; t < 8; ++t) acc[hh][t] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
                    }
                    load_query();
                }
                return false;
            };
#pragma unroll 1
            for (;;) {
                tile_body(ct, std::integral_constant<uint32_t, 0u>{});
                if (next()) break;
                tile_body(ct, std::integral_constant<uint32_t, 1u>{});
                if (next()) break;
                if (kStages > 2u) {
                    tile_body(ct, std::integral_constant<uint32_t, (kStages > 2u ? 2u : 0u)>{});
                    if (next()) break;
                }
            }
        } else {
#pragma unroll 1
            for (uint32_t tile = t0; tile < t1; tile += kStages) {
                tile_body(tile, std::integral_constant<uint32_t, 0u>{});
                if (tile + 1u >= t1) break;
                tile_body(tile + 1u, std::integral_constant<uint32_t, 1u>{});
                if (kStages > 2u) {
                    if (tile + 2u >= t1) break;
                    tile_body(tile + 2u, std::integral_constant<uint32_t, (kStages > 2u ? 2u : 0u)>{});
                }
            }
        }
        if (FORM != 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the re-requested tail tiles: nothing may land after the wave ends
    }
    if (HALVES == 2) {
        // the second half's state crosses over: 68 registers per lane = exactly the 17 408 bytes of a wave's two stages, [register][lane]
        float* xw = reinterpret_cast<float*>(&lds[4u + wave][0]);
        __syncthreads();                                                 // (every wave's DMAs have landed: vmcnt(0) above)
        if (half) {
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                xw[(2 * hh) * 64 + lane] = m_run[hh];
                xw[(2 * hh + 1) * 64 + lane] = l_run[hh];
#pragma unroll
                for (int t = 0; t < 8; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) xw[(4 + 32 * hh + 4 * t + r) * 64 + lane] = acc[hh][t][r];
            }
        }
        __syncthreads();
        if (half) return;
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            const float mb = xw[(2 * hh) * 64 + lane], lb = xw[(2 * hh + 1) * 64 + lane];
            const float mn = fmaxf(m_run[hh], mb);
            const float fa = (mn == -INFINITY) ? 1.0f : __builtin_amdgcn_exp2f(m_run[hh] - mn);      // (-inf - finite: 0)
            const float fb = (mn == -INFINITY) ? 1.0f : __builtin_amdgcn_exp2f(mb - mn);
            m_run[hh] = mn;
            l_run[hh] = l_run[hh] * fa + lb * fb;
#pragma unroll
            for (int t = 0; t < 8; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[hh][t][r] = acc[hh][t][r] * fa + xw[(4 + 32 * hh + 4 * t + r) * 64 + lane] * fb;
        }
    }
    store_rows();
}

// a.lin_base set: linear form; else a.stripe_bases: striped; else a.table_form: page-table form.  Writes the final rows itself
// when a.direct_out allows it, else the split partials followed by launch_attend_combine.
hipError_t launch_attend_mx4(const AttendArgs& a_in, uint32_t n_rows, float* d_out, float* d_lse, hipStream_t s)
{
    AttendArgs a = a_in;
    a.rows_real = n_rows;
    if (n_rows == 0 || a.n_splits == 0 || a.heads != 8u) return a.heads != 8u ? hipErrorInvalidValue : hipSuccess;
    if (!a.seqs && a.n_pages == 0) return hipSuccess;
    const int form = a.lin_base ? (a.stream.n_wgs ? 3 : 0) : a.stripe_bases ? 1 : a.table_form ? 2 : -1;
    if (form == 1 && (a.skip_pages || (!a.seqs && (a.stripe_n < 1u || a.stripe_n > 8u)))) return hipErrorInvalidValue;
    if (form < 0 || (a.stream.n_wgs && (form != 3 || a.seqs || (a.n_pages & 15u) || a.skip_pages))) return hipErrorInvalidValue;
    const dim3 grid = form == 3 ? dim3(a.stream.n_wgs, 1u, (a.g + 7u) / 8u) : a.rows_first ? dim3(n_rows | 1u, a.n_splits, (a.g + 7u) / 8u) : dim3(a.n_splits, n_rows, (a.g + 7u) / 8u);
    const dim3 block(64 * kWavesPerWg);
    constexpr size_t lds_bytes = static_cast<size_t>(kWavesPerWg) * kStagesDefault * kStage;
    // more than 64 KiB of dynamic LDS has to be allowed per function AND per device: once for every device this process launches on
    static std::atomic<uint64_t> allowed{0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev > 63) return hipErrorInvalidDevice;
    if (!(allowed.load(std::memory_order_acquire) >> dev & 1ull)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_attend_mx4<0>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_attend_mx4<1>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_attend_mx4<2>), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<size_t>(kWavesPerWg) * (2u * kStage + kTabEnt));
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_attend_mx4<3>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_attend_mx4<0, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<size_t>(2 * kWavesPerWg) * 2 * kStage);
        if (e != hipSuccess) return e;
        allowed.fetch_or(1ull << dev, std::memory_order_release);
    }
    if (form == 0 && a.mx4_halves) hipLaunchKernelGGL((k_attend_mx4<0, 2>), grid, dim3(128 * kWavesPerWg), static_cast<size_t>(2 * kWavesPerWg) * 2 * kStage, s, a);
    else if (form == 0) hipLaunchKernelGGL(k_attend_mx4<0>, grid, block, lds_bytes, s, a);
    else if (form == 1) hipLaunchKernelGGL(k_attend_mx4<1>, grid, block, lds_bytes, s, a);
    else if (form == 3) hipLaunchKernelGGL(k_attend_mx4<3>, grid, block, lds_bytes, s, a);
    else hipLaunchKernelGGL(k_attend_mx4<2>, grid, block, static_cast<size_t>(kWavesPerWg) * (2u * kStage + kTabEnt), s, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    const bool all_final = a.direct_out && (!a.direct_per_seq || a.direct_per_seq == 2u);
    if (all_final) return hipSuccess;
    return launch_attend_combine(a, n_rows, d_out, d_lse, s);
}

} // namespace speckv
