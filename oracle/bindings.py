"""ctypes bindings for the CHECKER libraries (test infrastructure only).

* ``Oracle``    -> oracle/_build/libspeckv_oracle.so  (our C restatement)
* ``Reference`` -> oracle/_ref/libspeckv_ref.so       (the reference itself,
  compiled in the dev container from /root/reference by oracle/Makefile)

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this module.  The product package never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(_HERE, "_build", "libspeckv_oracle.so")
REF_SO = os.path.join(_HERE, "_ref", "libspeckv_ref.so")

u8p = C.POINTER(C.c_uint8)
i8p = C.POINTER(C.c_int8)
u16p = C.POINTER(C.c_uint16)
u32p = C.POINTER(C.c_uint32)
i32p = C.POINTER(C.c_int32)
u64p = C.POINTER(C.c_uint64)
f32p = C.POINTER(C.c_float)
f64p = C.POINTER(C.c_double)
szp = C.POINTER(C.c_size_t)


def build_oracle(force=False):
    """Compile the C restatement (and, when /root/reference exists, the
    reference checker).  Building the checker is not using it."""
    if force or not os.path.exists(ORACLE_SO) or (
            os.path.getmtime(ORACLE_SO) < os.path.getmtime(os.path.join(_HERE, "speckv_oracle.c"))):
        subprocess.check_call(["make", "-s", "-C", _HERE, "oracle"])
    if os.path.isdir("/root/reference"):
        subprocess.check_call(["make", "-s", "-C", _HERE, "ref"])
    return ORACLE_SO


def _ptr(a, t):
    return a.ctypes.data_as(t)


class Oracle:
    """Thin numpy front-end over libspeckv_oracle.so."""

    REF_EXACT, INTENT = 0, 1
    FP16, INT8, INT8_DELTA_RLE, INT4_G32, FP8_E4M3, MXFP4 = 0, 1, 2, 3, 4, 5

    def __init__(self, path=None):
        path = path or ORACLE_SO
        if not os.path.exists(path):
            build_oracle()
        self.lib = L = C.CDLL(path)
        sig = {
            "orc_half_to_float": (C.c_float, [C.c_uint16]),
            "orc_float_to_half": (C.c_uint16, [C.c_float]),
            "orc_virt_page_id": (C.c_uint64, [C.c_uint64, C.c_uint64]),
            "orc_phys_page_id": (C.c_uint64, [C.c_uint64, C.c_uint64]),
            "orc_num_pages": (C.c_uint64, [C.c_uint64]),
            "orc_desc_gpu_addr": (C.c_uint64, [C.c_uint64]),
            "orc_encode_virt_page": (C.c_uint64, [C.c_uint32, C.c_uint16, C.c_uint16, C.c_uint32, C.c_uint8]),
            "orc_cabi_new": (C.c_void_p, []),
            "orc_cabi_delete": (None, [C.c_void_p]),
            "orc_cabi_init": (C.c_int, [C.c_void_p, C.c_char_p]),
            "orc_cabi_finalize": (None, [C.c_void_p]),
            "orc_cabi_alloc": (C.c_int, [C.c_void_p, C.c_uint64, u64p]),
            "orc_cabi_free": (C.c_int, [C.c_void_p, C.c_uint64]),
            "orc_cabi_access": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64, u64p]),
            "orc_cabi_prefetch": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint16, C.c_uint32, C.c_uint32, i32p, C.c_uint32]),
            "orc_cabi_set_prefetch_depth": (C.c_int, [C.c_void_p, C.c_uint32]),
            "orc_cabi_set_compression_scheme": (C.c_int, [C.c_void_p, C.c_int]),
            "orc_cabi_page_flags": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint64, u32p]),
            "orc_calc_offset": (C.c_uint64, [C.c_uint64] * 9),
            "orc_shim_total_bytes": (C.c_uint64, [C.c_uint64] * 5),
            "orc_compute_scale": (C.c_float, [f32p, C.c_size_t]),
            "orc_quantize": (None, [f32p, C.c_size_t, C.c_float, C.c_int, i8p]),
            "orc_delta_encode": (None, [i8p, C.c_size_t, i8p]),
            "orc_rle_encode": (C.c_size_t, [i8p, C.c_size_t, u8p]),
            "orc_rle_decode": (C.c_size_t, [u8p, C.c_size_t, i8p, C.c_size_t, szp]),
            "orc_delta_decode": (None, [i8p, C.c_size_t, i8p]),
            "orc_dequantize": (None, [i8p, C.c_size_t, C.c_float, C.c_int, f32p]),
            "orc_compress_f32": (C.c_size_t, [f32p, C.c_size_t, C.c_int, f32p, u8p]),
            "orc_decompress_f32": (C.c_size_t, [u8p, C.c_size_t, C.c_float, C.c_int, f32p, C.c_size_t]),
            "orc_compress_block_f16": (C.c_size_t, [u16p, C.c_size_t, C.c_int, C.c_int, f32p, u8p]),
            "orc_decompress_block_f16": (C.c_size_t, [u8p, C.c_size_t, C.c_float, C.c_int, C.c_int, u16p, C.c_size_t]),
            "orc_decompress_block_f32": (C.c_size_t, [u8p, C.c_size_t, C.c_float, C.c_int, C.c_int, f32p, C.c_size_t]),
            "orc_compress_blocks_f16": (None, [u16p, C.c_size_t, C.c_size_t, C.c_int, C.c_int, f32p, u32p, u8p, C.c_size_t, C.c_int]),
            "orc_decompress_blocks_f16": (None, [u8p, C.c_size_t, u32p, f32p, C.c_size_t, C.c_size_t, C.c_int, C.c_int, u16p, C.c_int]),
            "orc_f32_to_e4m3": (C.c_uint8, [C.c_float]),
            "orc_e4m3_to_f32": (C.c_float, [C.c_uint8]),
            "orc_qk_scores_fp8": (None, [u8p, f32p, C.c_size_t, u8p, f32p, C.c_size_t, C.c_size_t, f32p]),
            "orc_attend_fp8": (None, [u8p, f32p, C.c_size_t, u8p, f32p, u8p, f32p, C.c_size_t, C.c_size_t, C.c_float, f32p, f32p, f32p]),
            "orc_attend_f16": (None, [u16p, C.c_size_t, u16p, u16p, C.c_size_t, C.c_size_t, C.c_float, f32p, f32p, f32p]),
            "orc_coh_new": (C.c_void_p, [C.c_size_t, C.c_int]),
            "orc_coh_delete": (None, [C.c_void_p]),
            "orc_coh_request_read": (C.c_int, [C.c_void_p, C.c_uint64]),
            "orc_coh_request_write": (C.c_int, [C.c_void_p, C.c_uint64]),
            "orc_coh_invalidate": (C.c_int, [C.c_void_p, C.c_uint64]),
            "orc_coh_writeback": (C.c_int, [C.c_void_p, C.c_uint64]),
            "orc_coh_flush_all": (C.c_int, [C.c_void_p]),
            "orc_coh_get_state": (C.c_int, [C.c_void_p, C.c_uint64]),
            "orc_coh_get_tier": (C.c_int, [C.c_void_p, C.c_uint64]),
            "orc_coh_promote_to_l1": (C.c_int, [C.c_void_p, C.c_uint64]),
            "orc_coh_demote_to_l3": (C.c_int, [C.c_void_p, C.c_uint64]),
            "orc_coh_update_tier": (None, [C.c_void_p, C.c_uint64, C.c_int]),
            "orc_coh_batch_invalidate": (C.c_int, [C.c_void_p, u64p, C.c_size_t]),
            "orc_coh_get_statistics": (None, [C.c_void_p, u64p]),
            "orc_coh_reset_statistics": (None, [C.c_void_p]),
            "orc_quantize_rows_e4m3": (None, [u16p, C.c_size_t, C.c_size_t, u8p, f32p]),
            "orc_f32_to_e2m1": (C.c_uint8, [C.c_float]),
            "orc_e2m1_to_f32": (C.c_float, [C.c_uint8]),
            "orc_mx_scale_code": (C.c_uint8, [C.c_float, C.c_int]),
            "orc_e8m0_to_f32": (C.c_float, [C.c_uint8]),
            "orc_quantize_rows_mxfp8": (None, [u16p, C.c_size_t, C.c_size_t, C.c_size_t, u8p, u8p]),
            "orc_attend_mx4": (None, [u8p, u8p, C.c_size_t, C.c_size_t, u8p, u8p, u8p, u8p, C.c_size_t, C.c_size_t, C.c_float, f32p, f32p, f32p]),
            "orc_layer_compression_ratio": (C.c_double, [C.c_uint32]),
            "orc_codec_throughput_gbps": (C.c_double, [C.c_size_t, C.c_double, C.c_size_t]),
            "orc_codec_pipeline_latency_cycles": (C.c_size_t, []),
            "orc_tlb_new": (C.c_void_p, [C.c_size_t]),
            "orc_tlb_delete": (None, [C.c_void_p]),
            "orc_tlb_translate": (C.c_uint64, [C.c_void_p, C.c_uint64, C.POINTER(C.c_int)]),
            "orc_mm_new": (C.c_void_p, [C.c_uint64] * 4),
            "orc_mm_delete": (None, [C.c_void_p]),
            "orc_mm_allocate": (C.c_uint64, [C.c_void_p, C.c_uint64, C.c_uint32, C.c_int]),
            "orc_mm_deallocate": (None, [C.c_void_p, C.c_uint64]),
            "orc_mm_translate": (C.c_uint64, [C.c_void_p, C.c_uint64]),
            "orc_mm_is_in_cache": (C.c_int, [C.c_void_p, C.c_uint64, C.c_int]),
            "orc_mm_promote_to_l1": (C.c_int, [C.c_void_p, C.c_uint64]),
            "orc_mm_demote_to_l3": (C.c_int, [C.c_void_p, C.c_uint64]),
            "orc_mm_invalidate_page": (None, [C.c_void_p, C.c_uint64]),
            "orc_mm_mark_modified": (None, [C.c_void_p, C.c_uint64]),
            "orc_mm_get_page_state": (C.c_int, [C.c_void_p, C.c_uint64]),
            "orc_mm_update_access_tracking": (None, [C.c_void_p, C.c_uint64]),
            "orc_mm_is_hot_page": (C.c_int, [C.c_void_p, C.c_uint64]),
            "orc_mm_get_statistics": (None, [C.c_void_p, C.c_void_p]),
            "orc_mm_cxl_access": (C.c_uint64, [C.c_void_p, C.c_uint64, C.c_uint64]),
            "orc_compute_kv_address": (C.c_uint64, [C.c_uint32, C.c_uint32, C.c_uint32]),
            "orc_prefetch_legacy": (C.c_size_t, [C.c_void_p, C.c_uint32, C.c_size_t, C.c_size_t, u64p]),
            "orc_adapt_new": (C.c_void_p, [C.c_size_t]),
            "orc_adapt_delete": (None, [C.c_void_p]),
            "orc_adapt_update": (None, [C.c_void_p, C.c_int]),
            "orc_adapt_depth": (C.c_size_t, [C.c_void_p]),
            "orc_is_misprediction": (C.c_int, [C.c_uint32, u32p, C.c_size_t]),
            "orc_rtl_prefetch_vaddr": (C.c_uint64, [C.c_uint32, C.c_uint16, C.c_uint32]),
            "orc_lstm_predict": (C.c_size_t, [f32p, f32p] + [C.c_size_t] * 5 + [u32p, C.c_size_t, C.c_size_t, u32p, f32p]),
            "orc_lstm_reference_weights": (None, [C.c_uint, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t, f32p, f32p]),
            "orc_prefetch_pages": (C.c_size_t, [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32] + [C.c_uint64] * 6 + [u32p, u64p, C.c_size_t]),
        }
        for name, (res, args) in sig.items():
            fn = getattr(L, name)
            fn.restype, fn.argtypes = res, args

    # ---- codec, numpy in / numpy out -------------------------------
    def compress_f32(self, x, mode=0):
        x = np.ascontiguousarray(x, dtype=np.float32)
        rle = np.empty(max(2 * x.size, 2), dtype=np.uint8)
        scale = C.c_float()
        n = self.lib.orc_compress_f32(_ptr(x, f32p), x.size, mode, C.byref(scale), _ptr(rle, u8p))
        return np.float32(scale.value), rle[:n].copy()

    def decompress_f32(self, rle, scale, mode=0, cap=None):
        rle = np.ascontiguousarray(rle, dtype=np.uint8)
        if cap is None:
            cap = int(rle[1::2].astype(np.int64).sum()) if rle.size >= 2 else 0
        y = np.empty(max(cap, 1), dtype=np.float32)
        n = self.lib.orc_decompress_f32(_ptr(rle, u8p), rle.size, C.c_float(float(scale)), mode, _ptr(y, f32p), cap)
        return y[:n].copy()

    def compress_block_f16(self, x, scheme=2, mode=0):
        """x: uint16 view of fp16 data, one block.  Returns (scale, record)."""
        x = np.ascontiguousarray(x).view(np.uint16).ravel()
        rec = np.empty(max(2 * x.size, 2), dtype=np.uint8)
        scale = C.c_float()
        n = self.lib.orc_compress_block_f16(_ptr(x, u16p), x.size, scheme, mode, C.byref(scale), _ptr(rec, u8p))
        return np.float32(scale.value), rec[:n].copy()

    def decompress_block_f16(self, rec, scale, scheme=2, mode=0, cap=2048):
        rec = np.ascontiguousarray(rec, dtype=np.uint8)
        y = np.zeros(cap, dtype=np.uint16)
        n = self.lib.orc_decompress_block_f16(_ptr(rec, u8p), rec.size, C.c_float(float(scale)), scheme, mode, _ptr(y, u16p), cap)
        return y[:n].view(np.float16).copy()

    def decompress_block_f32(self, rec, scale, scheme=2, mode=0, cap=2048):
        rec = np.ascontiguousarray(rec, dtype=np.uint8)
        y = np.zeros(cap, dtype=np.float32)
        n = self.lib.orc_decompress_block_f32(_ptr(rec, u8p), rec.size, C.c_float(float(scale)), scheme, mode, _ptr(y, f32p), cap)
        return y[:n].copy()

    def compress_blocks_f16(self, x, scheme=2, mode=0, n=2048, threads=None):
        """x: (B, n) fp16.  Returns (scales f32[B], lens u32[B], recs u8[B, 2n]).  Blocks are independent: the C
        batch driver spreads them over `threads` threads (default: the host's cores, at most 32)."""
        x = np.ascontiguousarray(x).view(np.uint16).reshape(-1, n)
        B = x.shape[0]
        recs = np.zeros((B, 2 * n), dtype=np.uint8)
        scales = np.zeros(B, dtype=np.float32)
        lens = np.zeros(B, dtype=np.uint32)
        if B:
            self.lib.orc_compress_blocks_f16(_ptr(x, u16p), B, n, scheme, mode, _ptr(scales, f32p), _ptr(lens, u32p),
                                             _ptr(recs, u8p), 2 * n, self._threads(threads, B))
        return scales, lens, recs

    def decompress_blocks_f16(self, recs, lens, scales, scheme=2, mode=0, n=2048, threads=None):
        recs = np.ascontiguousarray(recs, dtype=np.uint8)
        lens = np.ascontiguousarray(lens, dtype=np.uint32)
        scales = np.ascontiguousarray(scales, dtype=np.float32)
        B = len(lens)
        y = np.zeros((B, n), dtype=np.uint16)
        if B:
            self.lib.orc_decompress_blocks_f16(_ptr(recs, u8p), recs.shape[1], _ptr(lens, u32p), _ptr(scales, f32p), B, n,
                                               scheme, mode, _ptr(y, u16p), self._threads(threads, B))
        return y.view(np.float16)

    @staticmethod
    def _threads(threads, n_blocks):
        if threads is None:
            threads = min(32, os.cpu_count() or 1)
        return max(1, min(int(threads), (n_blocks + 63) // 64))

    def lstm_predict(self, emb, wout, history, k, hist_len=16, layers=2):
        emb = np.ascontiguousarray(emb, np.float32); wout = np.ascontiguousarray(wout, np.float32)
        h = np.ascontiguousarray(history, np.uint32)
        tok = np.zeros(k, np.uint32); prob = np.zeros(k, np.float32)
        n = self.lib.orc_lstm_predict(_ptr(emb, f32p), _ptr(wout, f32p), emb.shape[0], emb.shape[1], wout.shape[1], layers,
                                      hist_len, _ptr(h, u32p), h.size, k, _ptr(tok, u32p), _ptr(prob, f32p))
        return tok[:n], prob[:n]

    def lstm_reference_weights(self, seed=1, vocab=32000, emb_dim=64, hidden=128, layers=2):
        emb = np.zeros((vocab, emb_dim), np.float32); wout = np.zeros((vocab, hidden), np.float32)
        self.lib.orc_lstm_reference_weights(seed, vocab, emb_dim, hidden, layers, _ptr(emb, f32p), _ptr(wout, f32p))
        return emb, wout

    def prefetch_pages(self, req, layer, cur_pos, k, L, T, H, D, bpe, alloc_pages, flags=None, cap=64):
        out = np.zeros(cap, dtype=np.uint64)
        fp = _ptr(np.ascontiguousarray(flags, dtype=np.uint32), u32p) if flags is not None else None
        n = self.lib.orc_prefetch_pages(req, layer, cur_pos, k, L, T, H, D, bpe, alloc_pages, fp, _ptr(out, u64p), cap)
        return out[:n].copy()


class MMStats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("l1_hits", "l1_misses", "l2_hits", "l2_misses", "l3_accesses",
                                          "migrations_l1_to_l3", "migrations_l3_to_l1")] + \
               [("l1_hit_rate", C.c_double), ("l2_hit_rate", C.c_double)]


def have_reference():
    return os.path.exists(REF_SO)


class Reference:
    """The reference's own code (oracle/_ref/libspeckv_ref.so)."""

    def __init__(self, path=None):
        path = path or REF_SO
        self.lib = L = C.CDLL(path)
        sig = {
            "ref_engine_new": (C.c_void_p, []),
            "ref_engine_delete": (None, [C.c_void_p]),
            "ref_engine_compress": (C.c_size_t, [C.c_void_p, f32p, C.c_size_t, C.c_uint32, f32p, u8p, C.c_size_t, szp]),
            "ref_engine_decompress": (C.c_size_t, [C.c_void_p, u8p, C.c_size_t, C.c_float, f32p, C.c_size_t]),
            "ref_engine_compress_blocks": (C.c_size_t, [C.c_void_p, f32p, C.c_size_t, C.c_size_t, f32p, u8p, C.c_size_t, u32p]),
            "ref_engine_decompress_blocks": (C.c_size_t, [C.c_void_p, u8p, C.c_size_t, u32p, f32p, C.c_size_t, f32p, C.c_size_t]),
            "ref_engine_ratio": (C.c_double, [C.c_void_p, C.c_uint32]),
            "ref_engine_throughput": (C.c_double, [C.c_void_p]),
            "ref_engine_translate": (C.c_uint64, [C.c_void_p, C.c_uint64]),
            "ref_mm_new": (C.c_void_p, [C.c_size_t] * 3),
            "ref_mm_delete": (None, [C.c_void_p]),
            "ref_mm_allocate": (C.c_uint64, [C.c_void_p, C.c_size_t, C.c_uint32, C.c_int]),
            "ref_mm_deallocate": (None, [C.c_void_p, C.c_uint64]),
            "ref_mm_translate": (C.c_uint64, [C.c_void_p, C.c_uint64]),
            "ref_mm_is_in_cache": (C.c_int, [C.c_void_p, C.c_uint64, C.c_int]),
            "ref_mm_promote_to_l1": (C.c_int, [C.c_void_p, C.c_uint64]),
            "ref_mm_demote_to_l3": (C.c_int, [C.c_void_p, C.c_uint64]),
            "ref_mm_invalidate_page": (None, [C.c_void_p, C.c_uint64]),
            "ref_mm_mark_modified": (None, [C.c_void_p, C.c_uint64]),
            "ref_mm_get_page_state": (C.c_int, [C.c_void_p, C.c_uint64]),
            "ref_mm_update_access_tracking": (None, [C.c_void_p, C.c_uint64]),
            "ref_mm_is_hot_page": (C.c_int, [C.c_void_p, C.c_uint64]),
            "ref_mm_get_statistics": (None, [C.c_void_p, u64p, f64p]),
            "ref_pf_new": (C.c_void_p, [C.c_void_p, C.c_size_t, C.c_size_t]),
            "ref_pf_delete": (None, [C.c_void_p]),
            "ref_pf_prefetch": (C.c_size_t, [C.c_void_p, u32p, C.c_size_t, C.c_uint32, C.c_size_t, u64p, u32p, f32p, C.c_size_t]),
            "ref_pf_update_accuracy": (None, [C.c_void_p, C.c_uint32, C.c_int]),
            "ref_pf_adaptive_depth": (C.c_size_t, [C.c_void_p]),
            "ref_pf_handle_misprediction": (C.c_size_t, [C.c_void_p, C.c_uint32, u32p, C.c_size_t]),
            "ref_ca_new": (C.c_void_p, [C.c_size_t] * 3),
            "ref_ca_delete": (None, [C.c_void_p]),
            "ref_ca_malloc": (C.c_uint64, [C.c_void_p, C.c_size_t, C.c_uint32]),
            "ref_ca_free": (None, [C.c_void_p, C.c_uint64]),
            "ref_ca_access": (C.c_uint64, [C.c_void_p, C.c_uint64, C.c_size_t, C.c_size_t]),
            "ref_ca_stats": (None, [C.c_void_p, u64p]),
            # the reference's own C ABI (host/include/speckv.h)
            "speckv_init": (C.c_int, [C.c_char_p]),
            "speckv_finalize": (None, []),
            "speckv_alloc": (C.c_int, [C.c_size_t, C.c_void_p, u64p]),
            "speckv_free": (C.c_int, [C.c_uint64]),
            "speckv_access": (C.c_int, [C.c_uint64, C.c_uint64, C.c_size_t, C.POINTER(C.c_void_p)]),
            "speckv_prefetch": (C.c_int, [C.c_uint32, C.c_uint16, C.c_uint32, C.c_uint32, i32p, C.c_uint32]),
            "speckv_set_prefetch_depth": (C.c_int, [C.c_uint32]),
            "speckv_set_compression_scheme": (C.c_int, [C.c_int]),
        }
        for name, (res, args) in sig.items():
            fn = getattr(L, name)
            fn.restype, fn.argtypes = res, args
        self.engine = L.ref_engine_new()

    def compress_f32(self, x, layer=0):
        x = np.ascontiguousarray(x, dtype=np.float32)
        rle = np.empty(max(2 * x.size, 2), dtype=np.uint8)
        scale = C.c_float()
        orig = C.c_size_t()
        n = self.lib.ref_engine_compress(self.engine, _ptr(x, f32p), x.size, layer, C.byref(scale),
                                         _ptr(rle, u8p), rle.size, C.byref(orig))
        return np.float32(scale.value), rle[:n].copy()

    def decompress_f32(self, rle, scale, cap=None):
        rle = np.ascontiguousarray(rle, dtype=np.uint8)
        if cap is None:
            cap = int(rle[1::2].astype(np.int64).sum()) if rle.size >= 2 else 0
        y = np.empty(max(cap, 1), dtype=np.float32)
        n = self.lib.ref_engine_decompress(self.engine, _ptr(rle, u8p), rle.size, C.c_float(float(scale)), _ptr(y, f32p), cap)
        return y[:min(n, cap)].copy()


# ---- coherence directory: one driver for the three implementations ------------------------------
REF_COH_SO = os.path.join(_HERE, "_ref", "libspeckv_ref_coh.so")
COH_OPS = ("read", "write", "invalidate", "writeback", "flush_all", "promote_to_l1", "demote_to_l3", "update_tier",
           "batch_invalidate", "get_state", "get_tier", "reset_statistics")


def have_reference_coherence():
    return os.path.exists(REF_COH_SO)


class _CohBase:
    """op(name, addr_or_list, arg) -> int result; stats() -> 7 counters.  Subclasses bind a backend."""

    def run(self, ops):
        """ops: list of (name, addr | [addrs], tier).  Returns [result, ...] + final (states, stats)."""
        return [self.op(*o) for o in ops]


class OracleCoherence(_CohBase):
    def __init__(self, oracle, line=64, has_driver=1):
        self.L = oracle.lib
        self.h = self.L.orc_coh_new(line, has_driver)

    def close(self):
        if self.h: self.L.orc_coh_delete(self.h); self.h = None

    def op(self, name, a=0, t=0):
        L, h = self.L, self.h
        if name == "read": return L.orc_coh_request_read(h, a)
        if name == "write": return L.orc_coh_request_write(h, a)
        if name == "flush_all": return L.orc_coh_flush_all(h)
        if name == "update_tier": L.orc_coh_update_tier(h, a, t); return 0
        if name == "reset_statistics": L.orc_coh_reset_statistics(h); return 0
        if name == "batch_invalidate":
            arr = np.ascontiguousarray(a, np.uint64)
            return L.orc_coh_batch_invalidate(h, _ptr(arr, u64p), arr.size)
        return getattr(L, "orc_coh_" + name)(h, a)

    def stats(self):
        st = np.zeros(7, np.uint64)
        self.L.orc_coh_get_statistics(self.h, _ptr(st, u64p))
        return [int(v) for v in st]


class ReferenceCoherence(_CohBase):
    """The reference's CoherenceManager itself (oracle/_ref/libspeckv_ref_coh.so, oracle/coh_harness.cpp)."""

    def __init__(self, line=64, has_driver=1):
        self.L = L = C.CDLL(REF_COH_SO)
        L.refcoh_new.restype = C.c_void_p; L.refcoh_new.argtypes = [C.c_size_t, C.c_int]
        L.refcoh_delete.argtypes = [C.c_void_p]
        for f in ("request_read", "request_write", "writeback"):
            getattr(L, "refcoh_" + f).argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_size_t]
        for f in ("invalidate", "get_state", "get_tier", "promote_to_l1", "demote_to_l3"):
            getattr(L, "refcoh_" + f).argtypes = [C.c_void_p, C.c_uint64]
        L.refcoh_flush_all.argtypes = [C.c_void_p]; L.refcoh_reset_statistics.argtypes = [C.c_void_p]
        L.refcoh_update_tier.argtypes = [C.c_void_p, C.c_uint64, C.c_int]
        L.refcoh_batch_invalidate.argtypes = [C.c_void_p, u64p, C.c_size_t]
        L.refcoh_get_statistics.argtypes = [C.c_void_p, u64p]
        self.h = L.refcoh_new(line, has_driver)
        self.buf = C.create_string_buffer(256)

    def close(self):
        if self.h: self.L.refcoh_delete(self.h); self.h = None

    def op(self, name, a=0, t=0):
        L, h = self.L, self.h
        if name == "read": return L.refcoh_request_read(h, a, self.buf, 64)
        if name == "write": return L.refcoh_request_write(h, a, self.buf, 64)
        if name == "writeback": return L.refcoh_writeback(h, a, self.buf, 64)
        if name == "flush_all": return L.refcoh_flush_all(h)
        if name == "update_tier": L.refcoh_update_tier(h, a, t); return 0
        if name == "reset_statistics": L.refcoh_reset_statistics(h); return 0
        if name == "batch_invalidate":
            arr = np.ascontiguousarray(a, np.uint64)
            return L.refcoh_batch_invalidate(h, _ptr(arr, u64p), arr.size)
        return getattr(L, "refcoh_" + name)(h, a)

    def stats(self):
        st = np.zeros(7, np.uint64)
        self.L.refcoh_get_statistics(self.h, _ptr(st, u64p))
        return [int(v) for v in st]
