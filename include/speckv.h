/*
 * include/speckv.h -- the drop-in C ABI of libcxlspeckv.so (MI355X build).
 *
 * These are exactly the eight entry points, types and status codes that the
 * reference's FFI for this path binds (reference: host/include/speckv.h:12-66,
 * implemented there by host/src/speckv_c_api.cpp:13-121; bound from Python by
 * host/python/speckv_ctypes.py:9-62).  Signatures are unchanged so the
 * existing ctypes shim / vLLM backend loads this library as is.  What is
 * behind them is new: a HIP slab pool in HBM, HIP fetch + decompress kernels
 * and a batched prefetch lookup; no kernel module, no ioctl.
 *
 * dev_path (speckv_init):
 *   "/dev/null"      the reference's fake device (SURVEY.md sect. 0.3): page
 *                    table and indexing only, no device memory, no data path.
 *                    Every status code and every returned logical address is
 *                    bit-identical to the reference on "/dev/null".
 *   "hip:N" | "/dev/speckvN" | anything else
 *                    the MI355X engine on HIP device N (default: the current
 *                    device).  Fails with SPECKV_ERR_GENERAL when no HIP device
 *                    is usable -- like the reference when /dev/speckv0 cannot
 *                    be opened.  There is no CPU fallback for the data path.
 */
#ifndef SPECKV_H
#define SPECKV_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* reference host/include/speckv.h:12-18 */
typedef enum {
    SPECKV_OK          = 0,
    SPECKV_ERR_GENERAL = -1,
    SPECKV_ERR_DRIVER  = -2,
    SPECKV_ERR_NOMEM   = -3,
    SPECKV_ERR_INVAL   = -4,
} speckv_status_t;

/* reference host/include/speckv.h:21-24 ; preferred_node selects the pool
 * device the allocation is placed on (0 = let the engine stripe). */
typedef struct {
    uint32_t preferred_node;
    uint32_t reserved;
} speckv_alloc_hint_t;

/* reference host/include/speckv.h:27 */
typedef uint64_t speckv_handle_t;

/* reference host/include/speckv.h:30-31 / speckv_c_api.cpp:13-39 */
speckv_status_t speckv_init(const char* dev_path);
void            speckv_finalize(void);

/* reference host/include/speckv.h:34-36 / speckv_c_api.cpp:41-53,
 * speckv_allocator.cpp:11-38.  Handles count from 1; pages are 4 KiB. */
speckv_status_t speckv_alloc(size_t bytes,
                             const speckv_alloc_hint_t* hint,
                             speckv_handle_t* out_handle);

/* reference host/include/speckv.h:39 / speckv_c_api.cpp:55-64 ; unknown or
 * already freed handles return SPECKV_OK, as the reference does. */
speckv_status_t speckv_free(speckv_handle_t handle);

/* reference host/include/speckv.h:44-47 / speckv_c_api.cpp:66-83,
 * speckv_allocator.cpp:54-74.  Ensures the page(s) covering
 * [offset_bytes, offset_bytes+length_bytes) are resident in compute-GPU HBM
 * (synchronous fetch + decompress on a miss) and returns the device address
 * of offset_bytes.  In "/dev/null" mode returns the reference's logical
 * address phys_page_id + offset%4096. */
speckv_status_t speckv_access(speckv_handle_t handle,
                              uint64_t offset_bytes,
                              size_t   length_bytes,
                              void**   out_gpu_ptr);

/* reference host/include/speckv.h:51-56 / speckv_c_api.cpp:85-99.
 * Queues a speculative look-ahead for positions cur_pos+1..cur_pos+depth_k of
 * (req_id, layer); requests are drained in batches by one lookup kernel. */
speckv_status_t speckv_prefetch(uint32_t       req_id,
                                uint16_t       layer,
                                uint32_t       cur_pos,
                                uint32_t       depth_k,
                                const int32_t* recent_tokens,
                                uint32_t       history_len);

/* reference host/include/speckv.h:59-63 */
typedef enum {
    SPECKV_COMP_FP16           = 0,
    SPECKV_COMP_INT8           = 1,
    SPECKV_COMP_INT8_DELTA_RLE = 2,
    /* additive values (no reference counterpart; BASELINE config 5, see speckv_ext.h) */
    SPECKV_COMP_INT4_G32       = 3,
    SPECKV_COMP_FP8_E4M3       = 4,
    SPECKV_COMP_MXFP4          = 5,   /* OCP MX v1.0: E2M1 elements, one E8M0 scale per 32 (the format gfx950's matrix cores read) */
} speckv_comp_scheme_t;

/* reference host/include/speckv.h:65-66 / speckv_c_api.cpp:101-121.
 * The scheme applies to allocations made after the call. */
speckv_status_t speckv_set_prefetch_depth(uint32_t depth_k);
speckv_status_t speckv_set_compression_scheme(speckv_comp_scheme_t scheme);

#ifdef __cplusplus
}
#endif
#endif /* SPECKV_H */
