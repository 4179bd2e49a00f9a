// cxl-speckv_amd/csrc/ring_rule.hpp -- the L2 prefetch ring's arithmetic, shared by the host (Engine::take_l2_run,
// Engine::l2_live), the device (k_flush_assign) and the CPU property tests (tests/csrc/host_rules_test.cpp): the two
// sides must agree exactly, because the host derives a page's residency from the sequence number a kernel stored for it.
//
// The ring has n slots.  A sequence number counts every slot the hand has passed -- slots skipped at the end of a lap
// included -- so slot = seq % n.  A run of m slots never wraps: if it does not fit before the end of the lap, the rest of
// the lap is skipped.  (Reference: the FIFO prefetch buffer of cxl_memory_manager.cpp:196-221 / prefetch_core.v:210-216,
// which moves nothing; here slots hold decompressed pages.)
#pragma once
#include <algorithm>
#include <cstdint>
#ifndef __host__
#define __host__
#define __device__
#endif

namespace speckv {

struct RingRun { uint32_t seq; uint32_t slot; uint32_t next; };      // first sequence number, its slot, the hand afterwards

// take m <= n slots at hand position `seq`
__host__ __device__ inline RingRun ring_take(uint32_t seq, uint32_t m, uint32_t n)
{
    uint32_t slot = seq % n;
    if (slot + m > n) { seq += n - slot; slot = 0; }
    return RingRun{seq, slot, seq + m};
}

// A page stored under sequence number q is intact until the hand has passed q + n: with the hand at `seq`,
// live <=> 0 < seq - q <= n  (32-bit differences: the engine renumbers long before a wrap could alias).
__host__ __device__ inline bool ring_live(uint32_t seq, uint32_t q, uint32_t n)
{
    return static_cast<uint32_t>(seq - q) - 1u < n;
}

// n_tiles divided evenly over at most `want` splits: {tiles per split, number of splits}, every split non-empty
struct EvenSplit { uint32_t tiles_per_split; uint32_t n_splits; };
__host__ __device__ inline EvenSplit even_split(uint32_t n_tiles, uint32_t want)
{
    if (n_tiles == 0) return EvenSplit{want ? want : 1u, 0u};
    if (want == 0) want = 1;
    if (want > n_tiles) want = n_tiles;
    const uint32_t tps = (n_tiles + want - 1u) / want;
    return EvenSplit{tps, (n_tiles + tps - 1u) / tps};
}

// INT4 batch attention launches with between half a machine and a whole one of workgroup columns (sequences x head groups in
// [384, 672]; 768 workgroups are resident, three per CU -- profiles/tools/probe/occupancy_lds.hip): unsplit they leave the
// CUs a third or more empty for the whole launch, and the kernel is bound by instruction issue, so occupancy is speed (256 x 8k:
// 512 columns 0.64 of HBM peak; the same columns at 384 / 768 sequences 0.664 / 0.692).  Every long sequence then goes in TWO
// pieces, a long one (fraction a of its tiles) and a short one, dispatched rows-first (AttendArgs::rows_first): the long pieces
// all start at once, the short ones take the remaining slots in turns.  a from a sweep on the MI355X (8k context,
// profiles/r03_int4_batch_split_sweep.txt: a = 0.5 .. 0.85):
//   384 columns: 0.545 whole, 0.63-0.64 for a <= 0.65;   448: 0.594 whole, 0.645 at a = 0.75, 0.62 at 0.8, no gain below 0.7;
//   512: 0.638 whole, 0.656 at 0.65, 0.669 at 0.8;         640: 0.607 whole, 0.64 at 0.5 and at 0.8
// -> near-equal halves up to 416 columns (the split launch is then about one round of workgroups), 0.8 beyond.  (171 + 85 tiles in
// the splits-first order of the other launches measured 0.52: the order is what makes it work; the merge is k_attend_combine_small.)
constexpr uint32_t kUnequalMinTiles = 192;      // 6k positions (256 x 4k: the split costs 4 %)
struct UnequalFraction { bool on; double a; };
inline UnequalFraction int4_unequal_fraction(uint32_t columns, uint32_t tiles_max)
{
    if (columns < 384u || columns > 672u || tiles_max < kUnequalMinTiles) return {false, 1.0};
    return {true, columns <= 416u ? 0.55 : 0.8};
}
// the two pieces of one sequence: {tiles of the first piece, number of pieces}; sequences shorter than kUnequalMinTiles stay whole
inline EvenSplit unequal_pieces(const UnequalFraction& u, uint32_t n_tiles)
{
    if (n_tiles < kUnequalMinTiles) return EvenSplit{n_tiles ? n_tiles : 1u, n_tiles ? 1u : 0u};
    uint32_t first = static_cast<uint32_t>(u.a * n_tiles + 0.999);
    if (first > n_tiles - 1u) first = n_tiles - 1u;
    if (first < (n_tiles + 1u) / 2u) first = (n_tiles + 1u) / 2u;     // the first piece is the long one
    return EvenSplit{first, 2u};
}

// Batch attention launches of MORE workgroup columns than the machine takes in whole rounds (round 6; profiles/r06_batch_over_cus.txt).
// Whole sequences make W = columns workgroups: F = W / CUs full rounds and a last one of the fraction r = (W % CUs) / CUs, and the
// last round is not cheap -- a CU that holds one workgroup more than its neighbours runs all of them slower (FP8, INT4: the CU's
// rate is shared), a workgroup alone on a CU is only about twice as fast as one among 256 (MXFP4) -- 260 sequences x 8k ran at
// 0.59 / 0.45 / 0.57 of the HBM roofline (FP8 / INT4 / MXFP4) where 256 run at 0.80 / 0.69 / 0.85.  Cut into s pieces the
// sequences make s times the workgroups of 1/s the length, and the launch costs, in tiles of one workgroup column,
//     (F + last(r)) x tiles per piece x pieces(s),    last(r) = max(last_min, last_base + last_slope r) for r > 0,
//                                                     pieces(s) = piece_base + piece_step s for s > 1 (merge launch, ramps), 1 whole
// -- fitted per kernel to sweeps of s = 1 .. 8 at 260 / 300 / 340 / 384 / 448 sequences x 8k (within 3 % of every point).  The
// cheapest s of 1 .. 8 is taken if it beats whole sequences by 3 % or more.  Returns the tiles per piece.
// The same holds between half a machine and a whole one (130 x 8k whole: FP8 0.46, INT4 0.43, MXFP4 0.63 of the roofline; 160: 0.56 /
// 0.53 / 0.71): the rule is applied from CUs / 2 sequences (FP8: more than CUs columns) up -- 130 -> 0.64 / 0.57 / 0.66, 160 -> 0.72 / 0.64 / 0.76.
struct PieceModel { double last_min, last_base, last_slope, piece_base, piece_step; uint32_t min_tiles; };
constexpr PieceModel kPiecesMx4{0.5, 0.48, 0.36, 1.04, 0.022, 24u};         // k_attend_mx4: one workgroup column per sequence, one workgroup per CU
constexpr PieceModel kPiecesFp8{0.75, 0.75, 0.25, 0.98, 0.02, 32u};       // k_attend_fp8_*: kv heads / 4 columns per sequence, up to four workgroups per CU
constexpr PieceModel kPiecesInt4Wg8{0.55, 0.3, 0.6, 1.0, 0.008, 32u};     // k_attend_int4_wg8<1>: one column per sequence, two workgroups per CU
constexpr PieceModel kPiecesInt4Halves{0.8, 0.62, 0.31, 1.0, 0.012, 32u}; // k_attend_int4_wg8<2> (batches of at most CUs sequences): 16-wave workgroups, one per CU -- and
                                                                         // a workgroup alone on its CU is hardly faster than one among 256 (the kernel is bound by instruction issue)
inline uint32_t balanced_tiles_per_piece(const uint32_t* tiles, uint32_t n_seq, uint32_t uniform_tiles, uint32_t columns_per_seq,
                                         uint32_t n_cus, const PieceModel& m)
{
    uint32_t n_max = tiles ? 0u : uniform_tiles;
    if (tiles) for (uint32_t i = 0; i < n_seq; ++i) n_max = tiles[i] > n_max ? tiles[i] : n_max;
    if (n_max == 0 || n_seq == 0 || columns_per_seq == 0 || n_cus == 0) return n_max ? n_max : 8u;
    double whole_cost = 0.0, best_cost = 0.0;
    uint32_t best = n_max;
    for (uint32_t sp = 1u; sp <= 8u; ++sp) {
        const uint32_t tps = (n_max + sp - 1u) / sp;
        if (sp > 1u && tps < m.min_tiles) break;
        uint64_t w = 0;
        if (tiles) for (uint32_t i = 0; i < n_seq; ++i) w += (tiles[i] + tps - 1u) / tps;
        else w = static_cast<uint64_t>(n_seq) * ((uniform_tiles + tps - 1u) / tps);
        w *= columns_per_seq;
        const double full = static_cast<double>(w / n_cus), rest = static_cast<double>(w % n_cus) / n_cus;
        const double last = rest > 0.0 ? (m.last_base + m.last_slope * rest > m.last_min ? m.last_base + m.last_slope * rest : m.last_min) : 0.0;
        const double cost = (full + last) * tps * (sp == 1u ? 1.0 : m.piece_base + m.piece_step * sp);
        if (sp == 1u) { whole_cost = best_cost = cost; continue; }
        if (cost < best_cost && cost <= 0.97 * whole_cost) { best_cost = cost; best = tps; }
    }
    return best;
}

// Split length (tiles) of an FP8 batch attention launch.  tiles[i] = tiles of sequence i (null: n_seq sequences of
// uniform_tiles each); columns_per_seq = workgroup columns one sequence contributes (kv heads / 4).  The four workgroups a
// CU can hold share its rate, so a launch takes about  ceil(workgroups / 256) x (tiles per split + 3)  tile times, plus
// about 16 for the merge launch if anything is split (8 with at most 8 splits per row: the one-wave merge kernel; measured
// 100 x 4k: whole 0.54 of HBM peak, two splits 0.61): the rule prices whole sequences and the split counts that just fill
// 1 .. 4 workgroups per CU, and takes the cheapest.  Beyond 1024 workgroups the launch runs in waves of 1024.
inline uint32_t fp8_batch_tiles_per_split(const uint32_t* tiles, uint32_t n_seq, uint32_t uniform_tiles, uint32_t columns_per_seq,
                                          uint32_t n_cus = 256u, uint32_t merge_cost_small = 8u)
{
    uint32_t n_max = tiles ? 0u : uniform_tiles;
    if (tiles) for (uint32_t i = 0; i < n_seq; ++i) n_max = tiles[i] > n_max ? tiles[i] : n_max;
    if (n_max == 0 || n_seq == 0 || columns_per_seq == 0) return 8u;
    const uint64_t columns = static_cast<uint64_t>(n_seq) * columns_per_seq;
    if (columns > n_cus && n_max >= 128u)                        // more than one round of whole sequences, 4k context and up (at 2k pieces gave nothing:
                                                                 // 300 x 2k whole 0.66, two pieces 0.62): the pieces that balance the last round (above)
        return balanced_tiles_per_piece(nullptr, n_seq, n_max, columns_per_seq, n_cus, kPiecesFp8);      // (priced on the longest member: equal pieces; the engine's dispatch order evens out the rest)
    uint64_t best_cost = UINT64_MAX;
    uint32_t best = n_max;
    for (uint32_t r = 0; r <= 4u; ++r) {
        uint64_t sp = r == 0 ? 1u : static_cast<uint64_t>(n_cus) * r / columns;
        if (sp > 2048u) sp = 2048u;
        if (sp == 0 || (r > 0 && sp == 1u)) continue;
        const uint32_t tps = static_cast<uint32_t>((n_max + sp - 1u) / sp);
        if (sp > 1u && tps < 8u) continue;                     // splits under 8 tiles cost more than they spread
        uint64_t wgs = 0;
        if (tiles) for (uint32_t i = 0; i < n_seq; ++i) wgs += (tiles[i] + tps - 1u) / tps;
        else wgs = static_cast<uint64_t>(n_seq) * ((uniform_tiles + tps - 1u) / tps);
        wgs *= columns_per_seq;
        const uint64_t resident = 4ull * n_cus;
        const uint64_t slots = wgs <= resident ? (wgs + n_cus - 1u) / n_cus : 4u * ((wgs + resident - 1u) / resident);
        const uint64_t cost = slots * (tps + (sp > 1u ? 3u : 0u)) + (sp > 1u ? (sp <= 8u ? merge_cost_small : 16u) : 0u);
        if (cost < best_cost) { best_cost = cost; best = tps; }
    }
    return best;
}

// Members of DIFFERENT lengths: whole sequences make the longest member the launch -- 256 members of 1k .. 16k on the one-workgroup-per-CU kernels: INT4 0.48,
// MXFP4 0.66 of the HBM roofline where 256 x 8k run at 0.69 / 0.85; with a heavy tail (one member in 16 at 32k, the others 1k .. 4k) every format collapses: FP8 0.26,
// INT4 0.12, MXFP4 0.20.  Dispatched by length (dispatch_order_by_length) and ROWS FIRST (every member's first piece, then the second pieces ...: with the
// splits-first grid a batch in which few members have several pieces puts its real workgroups on one or two XCDs -- linear id = piece + pieces x member) pieces
// pay as soon as the longest member exceeds a CU's fair share of the launch, share = tiles x workgroup columns per member / CUs (profiles/r06_ragged_batches.txt):
//   256 x 1k .. 16k   share 272 (one column) -- INT4 0.48 -> 0.60, MXFP4 0.66 -> 0.71 with pieces of 128;  FP8 (two columns: share 544 >= the longest) whole 0.78, pieces 0.69-0.76
//   512 x 1k .. 16k   share 534: whole sequences in order 0.70 / 0.80, pieces 0.68 / 0.73
//   256, heavy tail   share 139 / 278 -- FP8 0.26 -> 0.53-0.58 (pieces of 128 / 64), INT4 0.12 -> 0.41 / 0.38, MXFP4 0.20 -> 0.49 / 0.44
// -> pieces of a share's worth of tiles, 128 at most (32 at least, half the longest member at most) when the longest member is over 1.25 x share; 0 = no pieces on
//    account of the lengths.
// slots_per_cu: workgroups the kernel needs resident per CU to run at its rate (FP8's register-staged batch kernel: 4 -- one alone on a CU reads at a third of its
// share of the machine's rate: 64 members with a heavy tail, 300 workgroups of 70 tiles: 293 us for 0.57 GB; the LDS-DMA kernels: 1): the piece length aims at that many.
inline uint32_t ragged_tiles_per_piece(const uint32_t* tiles, uint32_t n_seq, uint32_t n_cus, uint32_t columns_per_seq, uint32_t slots_per_cu = 1u)
{
    uint64_t total = 0;
    uint32_t n_max = 0;
    for (uint32_t i = 0; i < n_seq; ++i) { total += tiles[i]; n_max = tiles[i] > n_max ? tiles[i] : n_max; }
    if (n_seq < 2u || n_cus == 0u || total == 0u || n_max < 64u) return 0u;
    const double share = static_cast<double>(total) * columns_per_seq / n_cus;
    if (n_max <= 1.25 * share) return 0u;
    if (n_max <= 1.25 * (static_cast<double>(total) / n_seq)) return 0u;           // members of (nearly) one length: the rules for equal lengths (few of them are over a CU's share too)
    const double per_slot = share / (slots_per_cu ? slots_per_cu : 1u);
    uint32_t tps = per_slot >= 128.0 ? 128u : static_cast<uint32_t>(per_slot + 0.5);      // (a slot's worth, 128 tiles at most: 192 and 163 measured 10 % behind 128)
    if (tps > n_max / 2u) tps = n_max / 2u;
    const uint32_t floor_tiles = slots_per_cu > 1u ? 16u : 32u;
    if (tps < floor_tiles) tps = floor_tiles;
    return (n_max + tps - 1u) / tps > 2048u ? (n_max + 2047u) / 2048u : tps;
}

// The order in which a batch's sequences are dispatched (Engine::attend_batch / attend_batch_plan; AttendArgs::order): by length, longest
// first, and -- round != 0 -- as a serpentine over rounds of `round` sequences (every second round reversed), so that the workgroups a
// CU receives from consecutive rounds are a long one and a short one.  False (nothing written) when the lengths differ by no more
// than an eighth of the longest: the order as given.
inline bool dispatch_order_by_length(const uint32_t* len, uint32_t n, uint32_t round, uint32_t* order)
{
    uint32_t lo = UINT32_MAX, hi = 0;
    for (uint32_t i = 0; i < n; ++i) { lo = len[i] < lo ? len[i] : lo; hi = len[i] > hi ? len[i] : hi; }
    if (n < 2u || hi - lo <= hi / 8u) return false;
    for (uint32_t i = 0; i < n; ++i) order[i] = i;
    std::stable_sort(order, order + n, [len](uint32_t x, uint32_t y) { return len[x] > len[y]; });
    if (round)
        for (uint32_t k = round; k < n; k += 2u * round) {
            std::reverse(order + k, order + (k + round < n ? k + round : n));
        }
    return true;
}

} // namespace speckv
