"""ctypes binding of libcxlspeckv.so.

Same class, constructor, attributes and methods as the reference's binding
(reference host/python/speckv_ctypes.py:7-98) so callers written against it run
unchanged; the ``ext_*`` methods bind the additive entry points of
include/speckv_ext.h when the loaded library exports them.
"""
import ctypes
from ctypes import (c_char_p, c_float, c_int, c_int32, c_size_t, c_uint8, c_uint16, c_uint32, c_uint64, c_void_p)


class SpeckvError(RuntimeError):
    """RuntimeError carrying the speckv_status_t code."""

    def __init__(self, what, status):
        super().__init__(f"{what} failed: {status}")
        self.status = status


class PageInfo(ctypes.Structure):
    """speckv_ext_page_info_t (include/speckv_ext.h)."""
    _fields_ = [("virt_page_id", c_uint64), ("phys_page_id", c_uint64), ("page_size", c_uint32),
                ("flags", c_uint32), ("pool_device", c_int32), ("scheme", c_uint32), ("rec_bytes", c_uint32),
                ("scale", c_float), ("pool_addr", c_uint64), ("cache_addr", c_uint64),
                ("access_count", c_uint32), ("aux_offset", c_uint32)]


class DmaDesc(ctypes.Structure):
    """speckv_dma_desc_t == reference SpeckvDmaDesc (host/include/speckv_driver.hpp:9-14)."""
    _fields_ = [("fpga_addr", c_uint64), ("gpu_addr", c_uint64), ("bytes", c_uint32), ("flags", c_uint32)]


class Stats(ctypes.Structure):
    """speckv_ext_stats_t (include/speckv_ext.h)."""
    _fields_ = [(n, c_uint64) for n in (
        "l1_hits", "l1_misses", "l2_hits", "l2_misses", "l3_accesses", "migrations_l1_to_l3",
        "migrations_l3_to_l1", "total_prefetches", "successful_prefetches", "mispredictions",
        "total_compressions", "total_decompressions", "compressed_bytes", "original_bytes",
        "total_allocations", "total_deallocations", "current_allocated_bytes", "peak_allocated_bytes",
        "dma_submitted", "dma_completed", "pool_bytes_reserved", "cache_bytes_reserved")] + \
        [(n, c_uint32) for n in ("prefetch_depth", "compression_scheme", "quant_mode", "n_pool_devices")] + \
        [("pool_migrated_pages", c_uint64), ("prefetch_dropped", c_uint64), ("copy_engine_runs", c_uint64),
         ("copy_engine_bytes", c_uint64), ("pool_bytes_in_use", c_uint64), ("written_pages", c_uint64),
         ("sealed_allocations", c_uint64), ("compactions", c_uint64), ("flat_decoder_fetches", c_uint64)]


_u32p = ctypes.POINTER(c_uint32)
_u64p = ctypes.POINTER(c_uint64)

_EXT_SIGNATURES = {
    "speckv_ext_stream_is_capturing": [c_void_p, ctypes.POINTER(c_int)],
    "speckv_ext_set_quant_mode": [c_int],
    "speckv_ext_translate": [c_uint64, c_uint64, ctypes.POINTER(PageInfo)],
    "speckv_ext_fetch_desc": [c_uint64, c_uint64, ctypes.POINTER(DmaDesc)],
    "speckv_ext_set_layout": [c_uint64, c_uint32, c_uint32, c_uint32, c_uint32, c_uint32],
    "speckv_ext_write": [c_uint64, c_uint64, c_void_p, c_size_t, c_int],
    "speckv_ext_read": [c_uint64, c_uint64, c_void_p, c_size_t, c_int],
    "speckv_ext_write_strided": [c_uint64, c_uint64, c_uint64, c_uint64, c_void_p, c_void_p],
    "speckv_ext_write_strided_batch": [c_void_p, c_void_p, c_void_p, c_uint32, c_uint64, c_uint64, c_void_p],
    "speckv_ext_write_async": [c_uint64, c_uint64, c_void_p, c_size_t, c_void_p],
    "speckv_ext_write_runs": [c_uint64, c_void_p, c_void_p, c_uint32, c_uint64, c_void_p],
    "speckv_ext_fetch_range": [c_uint64, c_uint64, c_uint64, c_void_p, c_int, c_void_p],
    "speckv_ext_fetch_range_engine": [c_uint64, c_uint64, c_uint64, c_void_p, c_int, c_void_p, c_int],
    "speckv_ext_bind_request": [c_uint32, c_uint64, c_uint32],
    "speckv_ext_fetch_list": [c_uint64, c_void_p, c_uint32, c_void_p, c_int, c_void_p],
    "speckv_ext_access_batch": [c_uint64, _u64p, c_uint32, ctypes.POINTER(c_void_p)],
    "speckv_ext_prefetch_batch": [c_uint32, _u32p, ctypes.POINTER(c_uint16), _u32p, _u32p],
    "speckv_ext_prefetch_flush": [_u32p],
    "speckv_ext_prefetch_lookup": [c_uint64, c_uint32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_uint32, c_void_p, c_void_p],
    "speckv_ext_prefetch_legacy_addrs": [c_uint32, c_uint32, _u64p, _u32p],
    "speckv_ext_verify": [c_uint32, c_int32, ctypes.POINTER(c_int32), c_uint32, _u32p, _u32p],
    "speckv_ext_verify_batch": [c_uint32, c_uint32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p],
    "speckv_ext_get_prefetch_depth": [_u32p],
    "speckv_ext_poll_complete": [_u32p],
    "speckv_ext_sync": [],
    "speckv_ext_codec_compress": [c_void_p, c_uint64, c_void_p, c_uint64, c_void_p, c_void_p, c_int, c_int, c_void_p],
    "speckv_ext_codec_decompress": [c_void_p, c_uint64, c_void_p, c_void_p, c_uint64, c_void_p, c_int, c_int, c_int, c_void_p],
    "speckv_ext_codec_compress_tensor": [c_void_p, c_uint64, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_int, c_void_p],
    "speckv_ext_codec_decompress_tensor": [c_void_p, c_uint64, ctypes.c_float, c_void_p, c_uint64, c_int, c_void_p, c_void_p, c_size_t, c_int, c_void_p],
    "speckv_ext_qk_scores_fp8": [c_uint64, c_uint32, c_void_p, c_uint32, c_uint32, c_uint32, c_void_p, c_void_p],
    "speckv_ext_qk_scores_fp8_layers": [c_uint64, c_uint32, c_uint32, c_void_p, c_uint32, c_uint32, c_uint32, c_void_p, c_void_p],
    "speckv_ext_attend_fp8": [c_uint64, c_uint32, c_uint32, c_void_p, c_uint32, c_uint32, c_uint32, ctypes.c_float, c_void_p, c_void_p, c_void_p],
    "speckv_ext_attend_fp8_batch": [c_uint32, ctypes.POINTER(c_uint64), c_uint32, c_void_p, c_uint32, _u32p, ctypes.c_float, c_void_p, c_void_p, c_void_p],
    "speckv_ext_attend_int4_batch": [c_uint32, ctypes.POINTER(c_uint64), c_uint32, c_void_p, c_uint32, _u32p, ctypes.c_float, c_void_p, c_void_p, c_void_p],
    "speckv_ext_attend_batch_plan": [c_uint32, ctypes.POINTER(c_uint64), _u32p, c_uint32, c_void_p, c_size_t, c_void_p],
    "speckv_ext_attend_fp8_planned": [c_void_p, c_uint32, c_uint32, c_void_p, c_uint32, c_uint32, ctypes.c_float, c_void_p, c_void_p, c_void_p],
    "speckv_ext_attend_int4_planned": [c_void_p, c_uint32, c_uint32, c_void_p, c_uint32, c_uint32, ctypes.c_float, c_void_p, c_void_p, c_void_p],
    "speckv_ext_attend_fold_tail": [c_uint32, c_void_p, c_uint32, c_uint32, c_void_p, c_void_p, c_void_p, c_uint64, ctypes.c_float, c_void_p, c_void_p, c_void_p],
    "speckv_ext_attend_int4": [c_uint64, c_uint32, c_uint32, c_void_p, c_uint32, c_uint32, c_uint32, ctypes.c_float, c_void_p, c_void_p, c_void_p],
    "speckv_ext_attend_mx4": [c_uint64, c_uint32, c_uint32, c_void_p, c_uint32, c_uint32, c_uint32, ctypes.c_float, c_void_p, c_void_p, c_void_p],
    "speckv_ext_attend_mx4_batch": [c_uint32, ctypes.POINTER(c_uint64), c_uint32, c_void_p, c_uint32, _u32p, ctypes.c_float, c_void_p, c_void_p, c_void_p],
    "speckv_ext_attend_mx4_planned": [c_void_p, c_uint32, c_uint32, c_void_p, c_uint32, c_uint32, ctypes.c_float, c_void_p, c_void_p, c_void_p],
    "speckv_ext_attend_planned_layers": [c_int, c_void_p, c_uint32, c_uint32, c_uint32, c_void_p, c_uint32, c_uint32, ctypes.c_float, c_void_p, c_void_p, c_uint32,
                                         c_void_p, c_void_p, c_void_p, c_void_p, c_uint64, c_void_p],
    "speckv_ext_attend_planned_tail": [c_int, c_void_p, c_uint32, c_uint32, c_void_p, c_uint32, c_uint32, ctypes.c_float, c_void_p, c_void_p, c_uint32,
                                       c_void_p, c_void_p, c_void_p, c_void_p, c_uint64, c_void_p],
    "speckv_ext_promote_to_l1": [c_uint64, c_uint64],
    "speckv_ext_demote_to_l3": [c_uint64, c_uint64],
    "speckv_ext_migrate": [c_uint64, c_uint64, c_uint64, c_uint32],
    "speckv_ext_compact": [c_uint64, _u64p, _u64p],
    "speckv_ext_predictor_load": [c_void_p, c_void_p, c_uint32, c_int],
    "speckv_ext_predict_batch": [c_uint32, c_void_p, c_uint32, c_void_p, c_void_p, c_void_p],
    "speckv_ext_predictor_load_lstm": [c_void_p, c_uint32, c_uint32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int],
    "speckv_ext_stats": [ctypes.POINTER(Stats)],
    "speckv_ext_stats_sized": [c_void_p, c_size_t, ctypes.POINTER(c_size_t)],
}


EXT_ABI_VERSION = 6        # SPECKV_EXT_ABI_VERSION of include/speckv_ext.h


def bind_ext(lib):
    """Attach argtypes/restype of every speckv_ext_* symbol the library exports."""
    found = []
    for name, args in _EXT_SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            continue
        fn.argtypes, fn.restype = args, c_int
        found.append(name)
    for name, res, args in (("speckv_ext_layer_compression_ratio", ctypes.c_double, [c_uint32]), ("speckv_ext_backend", c_char_p, []),
                            ("speckv_ext_attend_plan_bytes", c_size_t, [c_uint32]),
                            ("speckv_ext_codec_tensor_workspace_bytes", c_size_t, [c_uint64]),
                            ("speckv_ext_codec_tensor_decode_workspace_bytes", c_size_t, [c_uint64])):
        try:
            fn = getattr(lib, name)
        except AttributeError:
            continue
        fn.restype = res
        fn.argtypes = args
        found.append(name)
    # the structs of include/speckv_ext.h are mirrored by hand in this file: a library built from another header revision
    # would write past (or short of) them
    try:
        ver = lib.speckv_ext_abi_version
    except AttributeError:
        ver = None
    if ver is not None:
        ver.restype, ver.argtypes = c_uint32, []
        if ver() != EXT_ABI_VERSION:
            raise RuntimeError(f"libcxlspeckv.so speaks speckv_ext ABI {ver()}, this binding {EXT_ABI_VERSION}: rebuild the library")
        found.append("speckv_ext_abi_version")
    return found


class SpeckvLib:
    def __init__(self, path: str, dev_path: str = "/dev/speckv0"):
        from .build import load_library
        self.lib = load_library(path)

        # typedef uint64_t speckv_handle_t;
        self.handle_t = c_uint64

        self.lib.speckv_init.argtypes = [c_char_p]
        self.lib.speckv_init.restype = c_int
        self.lib.speckv_finalize.argtypes = []
        self.lib.speckv_finalize.restype = None

        class AllocHint(ctypes.Structure):
            _fields_ = [("preferred_node", c_uint32), ("reserved", c_uint32)]
        self.AllocHint = AllocHint

        self.lib.speckv_alloc.argtypes = [c_size_t, ctypes.POINTER(AllocHint), ctypes.POINTER(self.handle_t)]
        self.lib.speckv_alloc.restype = c_int
        self.lib.speckv_free.argtypes = [self.handle_t]
        self.lib.speckv_free.restype = c_int
        self.lib.speckv_access.argtypes = [self.handle_t, c_uint64, c_size_t, ctypes.POINTER(c_void_p)]
        self.lib.speckv_access.restype = c_int
        self.lib.speckv_prefetch.argtypes = [c_uint32, c_uint16, c_uint32, c_uint32, ctypes.POINTER(c_int32), c_uint32]
        self.lib.speckv_prefetch.restype = c_int
        self.lib.speckv_set_prefetch_depth.argtypes = [c_uint32]
        self.lib.speckv_set_prefetch_depth.restype = c_int
        self.lib.speckv_set_compression_scheme.argtypes = [c_int]
        self.lib.speckv_set_compression_scheme.restype = c_int
        self.ext = bind_ext(self.lib)

        ret = self.lib.speckv_init(dev_path.encode("ascii"))
        if ret != 0:
            raise SpeckvError("speckv_init", ret)

    # ---- the reference surface (speckv_ctypes.py:64-98) -------------
    def alloc(self, bytes_needed, preferred_node=0):
        hint = self.AllocHint(preferred_node, 0)
        handle = self.handle_t()
        ret = self.lib.speckv_alloc(bytes_needed, ctypes.byref(hint), ctypes.byref(handle))
        if ret != 0:
            raise SpeckvError("speckv_alloc", ret)
        return handle.value

    def free(self, handle):
        ret = self.lib.speckv_free(handle)
        if ret != 0:
            raise SpeckvError("speckv_free", ret)

    def access(self, handle, offset, length):
        gpu_ptr = c_void_p()
        ret = self.lib.speckv_access(handle, offset, length, ctypes.byref(gpu_ptr))
        if ret != 0:
            raise SpeckvError("speckv_access", ret)
        return gpu_ptr.value

    def prefetch(self, req_id, layer, cur_pos, depth_k, tokens):
        arr = (c_int32 * len(tokens))(*tokens)
        ret = self.lib.speckv_prefetch(req_id, layer, cur_pos, depth_k, arr, len(tokens))
        if ret != 0:
            raise SpeckvError("speckv_prefetch", ret)

    def set_prefetch_depth(self, depth_k):
        ret = self.lib.speckv_set_prefetch_depth(depth_k)
        if ret != 0:
            raise SpeckvError("speckv_set_prefetch_depth", ret)

    def set_compression_scheme(self, scheme):
        ret = self.lib.speckv_set_compression_scheme(scheme)
        if ret != 0:
            raise SpeckvError("speckv_set_compression_scheme", ret)

    # ---- additions ---------------------------------------------------
    def finalize(self):
        self.lib.speckv_finalize()

    def _ext(self, name, *args):
        ret = getattr(self.lib, name)(*args)
        if ret != 0:
            raise SpeckvError(name, ret)

    def translate(self, handle, offset):
        info = PageInfo()
        self._ext("speckv_ext_translate", handle, offset, ctypes.byref(info))
        return info

    def fetch_desc(self, handle, offset):
        d = DmaDesc()
        self._ext("speckv_ext_fetch_desc", handle, offset, ctypes.byref(d))
        return d

    def set_layout(self, handle, num_tokens, num_layers, num_heads, head_dim, bytes_per_element):
        self._ext("speckv_ext_set_layout", handle, num_tokens, num_layers, num_heads, head_dim, bytes_per_element)

    def set_quant_mode(self, mode):
        self._ext("speckv_ext_set_quant_mode", mode)

    def write(self, handle, offset, src_ptr, nbytes, on_device):
        self._ext("speckv_ext_write", handle, offset, c_void_p(src_ptr), nbytes, int(on_device))

    def write_strided(self, handle, first_page, page_step, n_pages, d_src, stream=None):
        self._ext("speckv_ext_write_strided", handle, first_page, page_step, n_pages, c_void_p(d_src), c_void_p(stream or 0))

    def write_strided_batch(self, handles, first_pages, d_srcs, page_step, n_pages_each, stream):
        """One launch for a batch of allocations: handles[i] gets pages first_pages[i] + j*page_step from d_srcs[i] + j*4096."""
        n = len(handles)
        as_arr = lambda v, t: v if isinstance(v, ctypes.Array) else (c_void_p(v.ctypes.data) if hasattr(v, "ctypes") else (t * n)(*v))
        hs, fs, ps = as_arr(handles, c_uint64), as_arr(first_pages, c_uint64), as_arr(d_srcs, c_void_p)     # numpy uint64 arrays pass as they are
        self._ext("speckv_ext_write_strided_batch", hs, fs, ps, n, page_step, n_pages_each, c_void_p(stream))

    def write_async(self, handle, offset, d_src, nbytes, stream):
        """A contiguous page range from a device buffer, asynchronously on `stream` (no device-wide wait)."""
        self._ext("speckv_ext_write_async", handle, offset, c_void_p(d_src), nbytes, c_void_p(stream or 0))

    def write_runs(self, handle, first_pages, d_srcs, n_pages_each, stream):
        """Several page runs of one allocation in ONE launch: run r = n_pages_each pages from first_pages[r], read from d_srcs[r]."""
        n = len(first_pages)
        self._ext("speckv_ext_write_runs", handle, (c_uint64 * n)(*first_pages), (c_void_p * n)(*d_srcs), n, n_pages_each,
                  c_void_p(stream))

    def read(self, handle, offset, dst_ptr, nbytes, on_device):
        self._ext("speckv_ext_read", handle, offset, c_void_p(dst_ptr), nbytes, int(on_device))

    ENGINE_AUTO, ENGINE_KERNEL, ENGINE_COPY = 0, 1, 2

    def fetch_range(self, handle, first_page, n_pages, d_dst, out_f32=False, stream=None, engine=None):
        """engine: None/0 = chosen per batch, 1 = fused peer-load + decompress kernel, 2 = copy engines + local decompress."""
        if engine:
            self._ext("speckv_ext_fetch_range_engine", handle, first_page, n_pages, c_void_p(d_dst), int(out_f32),
                      c_void_p(stream or 0), int(engine))
        else:
            self._ext("speckv_ext_fetch_range", handle, first_page, n_pages, c_void_p(d_dst), int(out_f32), c_void_p(stream or 0))

    def bind_request(self, req_id, handle, local_req=0):
        self._ext("speckv_ext_bind_request", req_id, handle, local_req)

    def fetch_list(self, handle, d_pages, n, d_dst, out_f32=False, stream=None):
        self._ext("speckv_ext_fetch_list", handle, c_void_p(d_pages), n, c_void_p(d_dst), int(out_f32), c_void_p(stream or 0))

    def access_batch(self, handle, offsets):
        n = len(offsets)
        offs = (c_uint64 * n)(*offsets)
        out = (c_void_p * n)()
        self._ext("speckv_ext_access_batch", handle, offs, n, out)
        return [p or 0 for p in out]

    def prefetch_batch(self, req_ids, layers, cur_pos, depth_k=None):
        """Sequences of ints, or contiguous numpy arrays (uint32 / uint16 / uint32 / uint32) passed without a copy."""
        n = len(req_ids)
        if hasattr(req_ids, "ctypes"):
            as_p = lambda a, t: a.ctypes.data_as(ctypes.POINTER(t))
            self._ext("speckv_ext_prefetch_batch", n, as_p(req_ids, c_uint32), as_p(layers, c_uint16), as_p(cur_pos, c_uint32),
                      as_p(depth_k, c_uint32) if depth_k is not None else None)
            return
        k = (c_uint32 * n)(*depth_k) if depth_k is not None else None
        self._ext("speckv_ext_prefetch_batch", n, (c_uint32 * n)(*req_ids), (c_uint16 * n)(*layers),
                  (c_uint32 * n)(*cur_pos), k)

    def prefetch_flush(self, want_count=True):
        """Submit the queued look-ahead requests (device-side pipeline).  want_count=False only submits; True also
        waits for the page count (not for the data)."""
        if not want_count:
            self._ext("speckv_ext_prefetch_flush", None)
            return None
        n = c_uint32()
        self._ext("speckv_ext_prefetch_flush", ctypes.byref(n))
        return n.value

    def verify(self, req_id, actual_token, predicted=None):
        """predicted=None: verify against the engine's own prediction for req_id."""
        hit, depth = c_uint32(), c_uint32()
        if predicted is None:
            self._ext("speckv_ext_verify", req_id, actual_token, None, 0, ctypes.byref(hit), ctypes.byref(depth))
        else:
            arr = (c_int32 * max(len(predicted), 1))(*predicted)
            self._ext("speckv_ext_verify", req_id, actual_token, arr, len(predicted), ctypes.byref(hit), ctypes.byref(depth))
        return bool(hit.value), depth.value

    def predictor_load(self, emb_ptr, wout_ptr, vocab, on_device):
        self._ext("speckv_ext_predictor_load", c_void_p(emb_ptr), c_void_p(wout_ptr), vocab, int(on_device))

    def predictor_load_lstm(self, emb_ptr, vocab, w_ih, w_hh, b_ih, b_hh, wout_ptr, out_bias_ptr, on_device):
        """A real LSTM cell (PyTorch nn.LSTM layout): w_ih / w_hh / b_ih / b_hh are lists of pointers, one per layer."""
        n = len(w_ih)
        arr = lambda ps: (c_void_p * n)(*ps)
        self._ext("speckv_ext_predictor_load_lstm", c_void_p(emb_ptr), vocab, n, arr(w_ih), arr(w_hh), arr(b_ih), arr(b_hh),
                  c_void_p(wout_ptr), c_void_p(out_bias_ptr or 0), int(on_device))

    def predict_batch(self, n, d_hist, k, d_tok, d_conf, stream=None):
        self._ext("speckv_ext_predict_batch", n, c_void_p(d_hist), k, c_void_p(d_tok), c_void_p(d_conf), c_void_p(stream or 0))

    def prefetch_depth(self):
        d = c_uint32()
        self._ext("speckv_ext_get_prefetch_depth", ctypes.byref(d))
        return d.value

    def poll_complete(self):
        d = c_uint32()
        self._ext("speckv_ext_poll_complete", ctypes.byref(d))
        return d.value

    def sync(self):
        self._ext("speckv_ext_sync")

    def qk_scores_fp8(self, handle, layer, d_q_f16, g, pos_begin, pos_end, d_out, stream=None):
        self._ext("speckv_ext_qk_scores_fp8", handle, layer, c_void_p(d_q_f16), g, pos_begin, pos_end,
                  c_void_p(d_out), c_void_p(stream or 0))

    def qk_scores_fp8_layers(self, handle, layer_begin, n_layers, d_q_f16, g, pos_begin, pos_end, d_out, stream=None):
        self._ext("speckv_ext_qk_scores_fp8_layers", handle, layer_begin, n_layers, c_void_p(d_q_f16), g, pos_begin,
                  pos_end, c_void_p(d_out), c_void_p(stream or 0))

    def attend_fp8(self, handle, layer_begin, n_layers, d_q_f16, g, pos_begin, pos_end, sm_scale, d_out, d_lse=None,
                   stream=None):
        self._ext("speckv_ext_attend_fp8", handle, layer_begin, n_layers, c_void_p(d_q_f16), g, pos_begin, pos_end,
                  ctypes.c_float(sm_scale), c_void_p(d_out), c_void_p(d_lse or 0), c_void_p(stream or 0))

    def attend_fp8_batch(self, handles, layer, d_q_f16, g, pos_end, sm_scale, d_out, d_lse=None, stream=None):
        """handles / pos_end: sequences of ints, or ready ctypes arrays (c_uint64 / c_uint32) that a caller reuses over
        the layers of a decode step."""
        n = len(handles)
        hs = handles if isinstance(handles, ctypes.Array) else (c_uint64 * n)(*handles)
        pe = pos_end if isinstance(pos_end, ctypes.Array) else (c_uint32 * n)(*pos_end)
        self._ext("speckv_ext_attend_fp8_batch", n, hs, layer, c_void_p(d_q_f16), g, pe, ctypes.c_float(sm_scale),
                  c_void_p(d_out), c_void_p(d_lse or 0), c_void_p(stream or 0))

    def attend_int4_batch(self, handles, layer, d_q_f16, g, pos_end, sm_scale, d_out, d_lse=None, stream=None):
        """handles / pos_end: sequences of ints, or ready ctypes arrays (c_uint64 / c_uint32) that a caller reuses over
        the layers of a decode step."""
        n = len(handles)
        hs = handles if isinstance(handles, ctypes.Array) else (c_uint64 * n)(*handles)
        pe = pos_end if isinstance(pos_end, ctypes.Array) else (c_uint32 * n)(*pos_end)
        self._ext("speckv_ext_attend_int4_batch", n, hs, layer, c_void_p(d_q_f16), g, pe, ctypes.c_float(sm_scale),
                  c_void_p(d_out), c_void_p(d_lse or 0), c_void_p(stream or 0))

    def attend_fold_tail(self, n_rows, d_rows, heads, g, d_q_f16, d_k_tail, d_v_tail, tail_stride_elems, sm_scale, d_out, d_lse,
                         stream=None):
        """Fold the not-yet-stored position (fp16 K / V tails) into out / lse of rows d_rows (0 / None: all, in order)."""
        self._ext("speckv_ext_attend_fold_tail", n_rows, c_void_p(d_rows or 0), heads, g, c_void_p(d_q_f16), c_void_p(d_k_tail),
                  c_void_p(d_v_tail), tail_stride_elems, ctypes.c_float(sm_scale), c_void_p(d_out), c_void_p(d_lse), c_void_p(stream or 0))

    def attend_plan_bytes(self, n_seq):
        return int(self.lib.speckv_ext_attend_plan_bytes(n_seq))

    def attend_batch_plan(self, handles, pos_end, max_pos_end, d_plan, plan_bytes, stream):
        """Once per decode step, outside any graph capture: the per-sequence descriptors of the step (valid for every
        layer) written to the device buffer d_plan."""
        n = len(handles)
        hs = handles if isinstance(handles, ctypes.Array) else (c_uint64 * n)(*handles)
        pe = pos_end if isinstance(pos_end, ctypes.Array) else (c_uint32 * n)(*pos_end)
        self._ext("speckv_ext_attend_batch_plan", n, hs, pe, max_pos_end, c_void_p(d_plan), plan_bytes, c_void_p(stream))

    def attend_planned(self, scheme, d_plan, n_seq, layer, d_q_f16, g, max_pos_end, sm_scale, d_out, d_lse, stream):
        """One layer of a planned batch: kernel launches only (capturable).  scheme: 4 (FP8_E4M3) or 3 (INT4_G32)."""
        name = {4: "speckv_ext_attend_fp8_planned", 3: "speckv_ext_attend_int4_planned", 5: "speckv_ext_attend_mx4_planned"}[scheme]
        self._ext(name, c_void_p(d_plan), n_seq, layer, c_void_p(d_q_f16), g, max_pos_end, ctypes.c_float(sm_scale),
                  c_void_p(d_out), c_void_p(d_lse or 0), c_void_p(stream))

    def attend_planned_layers(self, scheme, d_plan, n_seq, layer_begin, n_layers, d_q_f16, g, max_pos_end, sm_scale, d_out, d_lse, stream, n_tail=0,
                              d_tail_rows=0, d_tail_idx=0, d_k_tail=0, d_v_tail=0, tail_stride_elems=0):
        """Several layers of a planned batch in one call: q [n_layers][n_seq][heads][g][128], out / lse likewise."""
        self._ext("speckv_ext_attend_planned_layers", scheme, c_void_p(d_plan), n_seq, layer_begin, n_layers, c_void_p(d_q_f16), g, max_pos_end,
                  ctypes.c_float(sm_scale), c_void_p(d_out), c_void_p(d_lse or 0), n_tail, c_void_p(d_tail_rows or 0), c_void_p(d_tail_idx or 0),
                  c_void_p(d_k_tail or 0), c_void_p(d_v_tail or 0), tail_stride_elems, c_void_p(stream))

    def attend_planned_tail(self, scheme, d_plan, n_seq, layer, d_q_f16, g, max_pos_end, sm_scale, d_out, d_lse, n_tail, d_tail_rows, d_tail_idx,
                            d_k_tail, d_v_tail, tail_stride_elems, stream):
        """attend_planned + the position still held outside the pool (fp16 rows [n_tail][layers][heads][128]) in one call: folded in by
        the MXFP4 attention kernel itself, by one fold launch inside the call for the other formats."""
        self._ext("speckv_ext_attend_planned_tail", scheme, c_void_p(d_plan), n_seq, layer, c_void_p(d_q_f16), g, max_pos_end, ctypes.c_float(sm_scale),
                  c_void_p(d_out), c_void_p(d_lse), n_tail, c_void_p(d_tail_rows or 0), c_void_p(d_tail_idx or 0), c_void_p(d_k_tail or 0),
                  c_void_p(d_v_tail or 0), tail_stride_elems, c_void_p(stream))

    def attend_int4(self, handle, layer_begin, n_layers, d_q_f16, g, pos_begin, pos_end, sm_scale, d_out, d_lse=None,
                    stream=None):
        self._ext("speckv_ext_attend_int4", handle, layer_begin, n_layers, c_void_p(d_q_f16), g, pos_begin, pos_end,
                  ctypes.c_float(sm_scale), c_void_p(d_out), c_void_p(d_lse or 0), c_void_p(stream or 0))

    def attend_mx4(self, handle, layer_begin, n_layers, d_q_f16, g, pos_begin, pos_end, sm_scale, d_out, d_lse=None,
                   stream=None):
        self._ext("speckv_ext_attend_mx4", handle, layer_begin, n_layers, c_void_p(d_q_f16), g, pos_begin, pos_end,
                  ctypes.c_float(sm_scale), c_void_p(d_out), c_void_p(d_lse or 0), c_void_p(stream or 0))

    def attend_mx4_batch(self, handles, layer, d_q_f16, g, pos_end, sm_scale, d_out, d_lse=None, stream=None):
        n = len(handles)
        hs = handles if isinstance(handles, ctypes.Array) else (c_uint64 * n)(*handles)
        pe = pos_end if isinstance(pos_end, ctypes.Array) else (c_uint32 * n)(*pos_end)
        self._ext("speckv_ext_attend_mx4_batch", n, hs, layer, c_void_p(d_q_f16), g, pe, ctypes.c_float(sm_scale),
                  c_void_p(d_out), c_void_p(d_lse or 0), c_void_p(stream or 0))

    def stream_is_capturing(self, stream):
        """True while `stream` (a hipStream_t value) is being captured into a HIP graph"""
        out = ctypes.c_int(0)
        self._ext("speckv_ext_stream_is_capturing", c_void_p(stream or 0), ctypes.byref(out))
        return bool(out.value)

    def set_tuning(self, key, value):
        """A launch-form switch of the library (speckv_ext_set_tuning: the environment is read once per process)."""
        self.lib.speckv_ext_set_tuning.argtypes = [ctypes.c_char_p, ctypes.c_longlong]
        self.lib.speckv_ext_set_tuning.restype = ctypes.c_int
        if self.lib.speckv_ext_set_tuning(key.encode(), int(value)) != 0:
            raise ValueError(f"speckv_ext_set_tuning: no such key {key!r}")

    def promote_to_l1(self, handle, offset):
        return self.lib.speckv_ext_promote_to_l1(handle, offset) == 0

    def demote_to_l3(self, handle, offset):
        return self.lib.speckv_ext_demote_to_l3(handle, offset) == 0

    def migrate(self, handle, first_page, n_pages, target_pool):
        self._ext("speckv_ext_migrate", handle, first_page, n_pages, target_pool)

    def compact(self, handle):
        """Seal an allocation (pack its INT8_DELTA_RLE records, free the slots).  Returns (pool bytes before, after)."""
        before, after = c_uint64(), c_uint64()
        self._ext("speckv_ext_compact", handle, ctypes.byref(before), ctypes.byref(after))
        return before.value, after.value

    def stats(self):
        s = Stats()
        n = c_size_t()
        self._ext("speckv_ext_stats_sized", ctypes.byref(s), ctypes.sizeof(s), ctypes.byref(n))
        if n.value != ctypes.sizeof(s):
            raise RuntimeError(f"speckv_ext_stats_sized wrote {n.value} bytes, the binding's struct has {ctypes.sizeof(s)}")
        return s
