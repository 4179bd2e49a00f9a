// cxl-speckv_amd/csrc/tuning.hpp -- every switch of the library that is not part of an entry point's arguments, in ONE place.
// The environment is read exactly once (first use: the first speckv_init or raw codec call of the process); after that the
// launch paths read plain fields -- no getenv behind speckv_ext_attend_*, fetch_range or the codec operators (VERDICT r4 #8;
// tests/test_build_guards.py greps for it).  Tests and measurement runs that flip a form between two calls use
// speckv_ext_set_tuning(key, value) (key = the field name); INTEGRATION.md lists the keys.
#pragma once
#include <cstdint>

namespace speckv {

struct Tuning {
    // launch geometry of the fused attention (0 = the library's own rule)
    int32_t attend_splits = 0;              // SPECKV_ATTEND_SPLITS            single-sequence calls: splits per row
    int32_t attend_tiles_per_split = 0;     // SPECKV_ATTEND_TILES_PER_SPLIT   batch calls: tiles (32 positions) per split
    int32_t attend_stream = 0;              // SPECKV_ATTEND_STREAM            MXFP4, several layers of one sequence: N > 0 the stream form with N workgroups, -1 never, 0 by size
    int32_t attend_mx4_one_half = 0;        // SPECKV_ATTEND_MX4_ONE_HALF      MXFP4 batches: 4-wave workgroups also where the two-halves form applies (A/B, tests)
    int32_t attend_order_as_given = 0;      // SPECKV_ATTEND_ORDER_AS_GIVEN    batches of different lengths: dispatch the sequences in the caller's order (A/B, tests)
    int32_t attend_general = 0;             // SPECKV_ATTEND_GENERAL           1: page-table forms even where an arithmetic form applies (tests)
    int32_t attend_fp8_table_regs = 0;      // SPECKV_ATTEND_FP8_TABLE_REGS    FP8 over striped / moved placements: the register-staged kernels of rounds 2-5 instead of the DMA pipeline (tests, A/B)
    int32_t attend_fp8_dma = 0;             // SPECKV_ATTEND_FP8_DMA           FP8, one sequence: 1 = the LDS-DMA kernel whatever the split count, -1 = the register-staged one (tests, A/B)
    int32_t attend_fp8_striped_table = 0;   // SPECKV_ATTEND_FP8_STRIPED_TABLE FP8 over regularly striped pools: 1 = single-sequence calls through the page table (k_attend_fp8_dma<1>) instead of residue classes; -1 = batches / plans by residue classes instead of the page tables (tests, A/B)
    int32_t attend_int4_striped_wg = 0;     // SPECKV_ATTEND_INT4_STRIPED_WG   INT4_G32 over striped pools: the 4-head kernel with an address per lane (rounds 2-5) instead of the whole-record kernel by residue classes (tests, A/B)
    int32_t attend_layers_loop = 0;         // SPECKV_ATTEND_LAYERS_LOOP       planned_layers: per-layer launches also where one launch could take them all (tests, A/B)
    int32_t attend_fold_launch = 0;         // SPECKV_ATTEND_FOLD_LAUNCH       planned_tail: the tail position by a launch of its own also where the kernel could fold it (tests, A/B)
    // whole-tensor codec: the multi-launch forms the one-pass kernels replaced, kept as cross-checks of each other (tests)
    int32_t tc_multipass = 0;               // SPECKV_TC_MULTIPASS
    int32_t tc_scan = 0;                    // SPECKV_TC_SCAN                  0 grids of waves, 1 ("wg") one workgroup, 2 ("serial") one wave
    int32_t tc_no_pre = 0;                  // SPECKV_TC_NO_PRE                summary and emit as two plain passes
    int32_t tc_no_split_tiles = 0;          // SPECKV_TC_NO_SPLIT_TILES        element-wise loop for long stretches
    int32_t td_one_pass = 0;                // SPECKV_TD_ONE_PASS              one-pass decoder also for streams of few pairs
    int32_t td_expand_per_element = 0;      // SPECKV_TD_EXPAND_PER_ELEMENT    the expand loop of rounds 2-3
    int32_t tc_batch_one_wg = 0;            // SPECKV_TC_BATCH_ONE_WG          many-tensor launches: one workgroup per tensor whatever its size (chains in LDS; tests, A/B)
    // prefetch flush / predictor forms (tests compare them)
    int32_t flush_no_small = 0;             // SPECKV_FLUSH_NO_SMALL           always the four-launch pipeline
    int32_t flush_small_words = 0;          // SPECKV_FLUSH_SMALL_WORDS        candidate words up to which one workgroup flushes (0: default)
    int32_t predict_batch_path = 0;         // SPECKV_PREDICT_BATCH_PATH       the batch kernels also for a handful of requests
    // block codec launches
    int32_t wgs_per_cu = 0;                 // SPECKV_WGS_PER_CU               grid cap of the block codec (0: default)
    int32_t rounds_consecutive = 0;         // SPECKV_ROUNDS=consecutive       a wave's blocks consecutive instead of one grid apart
    // remote fetch engine: 0 per batch, 1 fused peer-load kernel, 2 copy engines (SPECKV_REMOTE_ENGINE=kernel|copy)
    int32_t remote_engine = 0;
    int32_t copy_min_run_kb = 1024;         // SPECKV_COPY_MIN_RUN_KB          shortest per-pool run the copy engines take by themselves
};

Tuning& tuning();                                        // process-wide; filled from the environment on first use
int tuning_set(const char* key, long long value);        // 0 = set, -1 = no such key

} // namespace speckv
