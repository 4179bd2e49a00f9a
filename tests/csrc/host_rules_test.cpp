// tests/csrc/host_rules_test.cpp -- TEST-ONLY C wrappers around the header-only host/device rules of the engine
// (ring_rule.hpp: ring run / liveness arithmetic, even splits), for the CPU property tests.
#include "../../cxl-speckv_amd/csrc/ring_rule.hpp"

extern "C" {
void rules_ring_take(uint32_t seq, uint32_t m, uint32_t n, uint32_t* out3)
{
    const speckv::RingRun r = speckv::ring_take(seq, m, n);
    out3[0] = r.seq; out3[1] = r.slot; out3[2] = r.next;
}
int rules_ring_live(uint32_t seq, uint32_t q, uint32_t n) { return speckv::ring_live(seq, q, n) ? 1 : 0; }
void rules_even_split(uint32_t n_tiles, uint32_t want, uint32_t* out2)
{
    const speckv::EvenSplit e = speckv::even_split(n_tiles, want);
    out2[0] = e.tiles_per_split; out2[1] = e.n_splits;
}
uint32_t rules_fp8_batch_tiles_per_split(const uint32_t* tiles, uint32_t n_seq, uint32_t uniform_tiles, uint32_t columns_per_seq)
{
    return speckv::fp8_batch_tiles_per_split(tiles, n_seq, uniform_tiles, columns_per_seq);
}
// model: 0 = MXFP4 (k_attend_mx4), 1 = FP8, 2 = INT4 on the whole-record kernel (one-run workgroups), 3 = the same, 16-wave form
uint32_t rules_balanced_tiles_per_piece(const uint32_t* tiles, uint32_t n_seq, uint32_t uniform_tiles, uint32_t columns_per_seq, uint32_t n_cus, uint32_t model)
{
    const speckv::PieceModel& m = model == 0u ? speckv::kPiecesMx4 : model == 1u ? speckv::kPiecesFp8 : model == 2u ? speckv::kPiecesInt4Wg8 : speckv::kPiecesInt4Halves;
    return speckv::balanced_tiles_per_piece(tiles, n_seq, uniform_tiles, columns_per_seq, n_cus, m);
}
uint32_t rules_ragged_tiles_per_piece(const uint32_t* tiles, uint32_t n_seq, uint32_t n_cus, uint32_t model)
{
    return speckv::ragged_tiles_per_piece(tiles, n_seq, n_cus, model == 1u ? 2u : 1u, model == 1u ? 4u : 1u);          // (model 1 = FP8 with 8 kv heads: two workgroup columns per member)
}
int rules_dispatch_order(const uint32_t* len, uint32_t n, uint32_t round, uint32_t* order) { return speckv::dispatch_order_by_length(len, n, round, order) ? 1 : 0; }
// {on, first piece, pieces} of a sequence of n_tiles in an INT4 batch of `columns` workgroup columns whose longest member has tiles_max
void rules_int4_unequal(uint32_t columns, uint32_t tiles_max, uint32_t n_tiles, uint32_t* out3)
{
    const speckv::UnequalFraction u = speckv::int4_unequal_fraction(columns, tiles_max);
    const speckv::EvenSplit e = u.on ? speckv::unequal_pieces(u, n_tiles) : speckv::EvenSplit{n_tiles ? n_tiles : 1u, n_tiles ? 1u : 0u};
    out3[0] = u.on ? 1u : 0u; out3[1] = e.tiles_per_split; out3[2] = e.n_splits;
}
}
