/* speckv_coherence.h -- the coherence shadow directory of libcxlspeckv.so.
 *
 * Same 14 exported functions, names, argument order and return conventions as the reference's
 * C wrapper src/cxl_memory/coherence_c_api.cpp:33-209 (which its Python binding
 * host/python/cxlspeckv_coherence.py:70-120 loads from libcxlspeckv.so), so that binding works
 * against this library unchanged.  Behaviour follows CoherenceManager
 * (src/cxl_memory/coherence_manager.cpp) call for call, including its counting rules (a read that
 * misses adds 2 to total_reads: the miss and the fetch operation, :52-56 and :420), because callers
 * read the statistics.  The reference moves no data in any of these calls (its device operations
 * are stubs, :398-434); neither does this directory: `data_out` is left untouched and `data` is not
 * read.  Data movement between HBM tiers is the job of speckv_access / speckv_ext_* (speckv_ext.h).
 *
 * Differences a caller can see: nothing is printed by flush_all / destroy (the reference writes two
 * lines to stdout, :163,177); the table is a flat open-addressing map (no allocation per line).
 * Thread-safe: one mutex per manager. */
#ifndef SPECKV_COHERENCE_H
#define SPECKV_COHERENCE_H
#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef void* coherence_manager_handle_t;               /* coherence_c_api.cpp:17 */

typedef struct {                                         /* coherence_c_api.cpp:20-28 */
    uint64_t total_reads, total_writes, coherence_ops, invalidations_sent, writebacks_performed,
             directory_hits, directory_misses;
} coherence_statistics_t;

/* coherence states (coherence_manager.h:30-35): 0 INVALID, 1 SHARED, 2 EXCLUSIVE, 3 MODIFIED
 * memory tiers     (coherence_manager.h:38-42): 0 L1_GPU, 1 L2_PREFETCH, 2 L3_CXL */

/* device_path NULL -> NULL.  "/dev/null" (the reference's fake device) always works; any other path
 * needs a visible HIP device (it names the engine's GPU as in speckv_init), else NULL -- the
 * reference returns NULL when its driver cannot be constructed (:37-44). */
coherence_manager_handle_t coherence_manager_create(const char* device_path, size_t cache_line_size);
void coherence_manager_destroy(coherence_manager_handle_t handle);                 /* flushes first (:27-30) */
bool coherence_manager_request_read(coherence_manager_handle_t handle, uint64_t addr, void* data_out, size_t size);
bool coherence_manager_request_write(coherence_manager_handle_t handle, uint64_t addr, const void* data, size_t size);
bool coherence_manager_invalidate(coherence_manager_handle_t handle, uint64_t addr);
bool coherence_manager_writeback(coherence_manager_handle_t handle, uint64_t addr, const void* data, size_t size);
bool coherence_manager_flush_all(coherence_manager_handle_t handle);
int  coherence_manager_get_state(coherence_manager_handle_t handle, uint64_t addr);   /* NULL handle -> 0 */
int  coherence_manager_get_tier(coherence_manager_handle_t handle, uint64_t addr);    /* NULL handle -> 2 */
bool coherence_manager_promote_to_l1(coherence_manager_handle_t handle, uint64_t addr);
bool coherence_manager_demote_to_l3(coherence_manager_handle_t handle, uint64_t addr);
bool coherence_manager_batch_invalidate(coherence_manager_handle_t handle, const uint64_t* addrs, size_t count);
void coherence_manager_get_statistics(coherence_manager_handle_t handle, coherence_statistics_t* stats_out);
void coherence_manager_reset_statistics(coherence_manager_handle_t handle);

/* additive (CoherenceManager methods the reference's C wrapper does not export) */
void   coherence_manager_ext_update_tier(coherence_manager_handle_t handle, uint64_t addr, int tier);   /* update_tier, :263-270 */
size_t coherence_manager_ext_entry_count(coherence_manager_handle_t handle);                            /* directory size */

#ifdef __cplusplus
}
#endif
#endif
