// cxl-speckv_amd/csrc/tuning.cpp -- see tuning.hpp
#include "tuning.hpp"

#include <cstdlib>
#include <cstring>
#include <mutex>

namespace speckv {

namespace {
struct Key { const char* name; const char* env; int32_t Tuning::* field; };
const Key kKeys[] = {
    {"attend_splits", "SPECKV_ATTEND_SPLITS", &Tuning::attend_splits},
    {"attend_tiles_per_split", "SPECKV_ATTEND_TILES_PER_SPLIT", &Tuning::attend_tiles_per_split},
    {"attend_general", "SPECKV_ATTEND_GENERAL", &Tuning::attend_general},
    {"attend_fold_launch", "SPECKV_ATTEND_FOLD_LAUNCH", &Tuning::attend_fold_launch},
    {"attend_layers_loop", "SPECKV_ATTEND_LAYERS_LOOP", &Tuning::attend_layers_loop},
    {"attend_fp8_dma", "SPECKV_ATTEND_FP8_DMA", &Tuning::attend_fp8_dma},
    {"attend_fp8_striped_table", "SPECKV_ATTEND_FP8_STRIPED_TABLE", &Tuning::attend_fp8_striped_table},
    {"attend_int4_striped_wg", "SPECKV_ATTEND_INT4_STRIPED_WG", &Tuning::attend_int4_striped_wg},
    {"attend_fp8_table_regs", "SPECKV_ATTEND_FP8_TABLE_REGS", &Tuning::attend_fp8_table_regs},
    {"attend_stream", "SPECKV_ATTEND_STREAM", &Tuning::attend_stream},
    {"attend_mx4_one_half", "SPECKV_ATTEND_MX4_ONE_HALF", &Tuning::attend_mx4_one_half},
    {"attend_order_as_given", "SPECKV_ATTEND_ORDER_AS_GIVEN", &Tuning::attend_order_as_given},
    {"tc_multipass", "SPECKV_TC_MULTIPASS", &Tuning::tc_multipass},
    {"tc_scan", "SPECKV_TC_SCAN", &Tuning::tc_scan},
    {"tc_no_pre", "SPECKV_TC_NO_PRE", &Tuning::tc_no_pre},
    {"tc_no_split_tiles", "SPECKV_TC_NO_SPLIT_TILES", &Tuning::tc_no_split_tiles},
    {"td_one_pass", "SPECKV_TD_ONE_PASS", &Tuning::td_one_pass},
    {"td_expand_per_element", "SPECKV_TD_EXPAND_PER_ELEMENT", &Tuning::td_expand_per_element},
    {"tc_batch_one_wg", "SPECKV_TC_BATCH_ONE_WG", &Tuning::tc_batch_one_wg},
    {"flush_no_small", "SPECKV_FLUSH_NO_SMALL", &Tuning::flush_no_small},
    {"flush_small_words", "SPECKV_FLUSH_SMALL_WORDS", &Tuning::flush_small_words},
    {"predict_batch_path", "SPECKV_PREDICT_BATCH_PATH", &Tuning::predict_batch_path},
    {"wgs_per_cu", "SPECKV_WGS_PER_CU", &Tuning::wgs_per_cu},
    {"rounds_consecutive", "SPECKV_ROUNDS", &Tuning::rounds_consecutive},
    {"remote_engine", "SPECKV_REMOTE_ENGINE", &Tuning::remote_engine},
    {"copy_min_run_kb", "SPECKV_COPY_MIN_RUN_KB", &Tuning::copy_min_run_kb},
};
// a few of the environment's values are words, as they always were
int32_t parse(const char* env_name, const char* v)
{
    if (!strcmp(env_name, "SPECKV_TC_SCAN")) return !strcmp(v, "wg") ? 1 : !strcmp(v, "serial") ? 2 : atoi(v);
    if (!strcmp(env_name, "SPECKV_REMOTE_ENGINE")) return !strcmp(v, "kernel") ? 1 : !strcmp(v, "copy") ? 2 : atoi(v);
    if (!strcmp(env_name, "SPECKV_ROUNDS")) return !strcmp(v, "consecutive") ? 1 : atoi(v);
    if (!*v) return 1;                                   // set but empty: "on"
    const int32_t n = static_cast<int32_t>(atoi(v));
    return (n == 0 && v[0] != '0') ? 1 : n;              // a non-numeric word: "on"
}
Tuning g_tuning;
std::once_flag g_once;
} // namespace

Tuning& tuning()
{
    std::call_once(g_once, [] {
        for (const Key& k : kKeys)
            if (const char* v = getenv(k.env)) g_tuning.*(k.field) = parse(k.env, v);
    });
    return g_tuning;
}

int tuning_set(const char* key, long long value)
{
    if (!key) return -1;
    Tuning& t = tuning();
    for (const Key& k : kKeys)
        if (!strcmp(k.name, key)) { t.*(k.field) = static_cast<int32_t>(value); return 0; }
    return -1;
}

} // namespace speckv
