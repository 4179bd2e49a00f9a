"""KV allocator shim for a vLLM-style decode loop, on top of libcxlspeckv.so.

Public surface = the reference's shim (reference host/python/vllm_speckv_backend.py:8-100):
``CxlSpeckvKVAllocator(lib_path, dev_path, page_size)`` with ``allocate``,
``get_kv_ptr``, ``prefetch_step`` and the ``_calc_offset`` layout rule
``[req][layer][kind(K/V)][pos][head]``.  The reference file cannot be imported (its
trailing example is a SyntaxError at line 104), so this is our own implementation
of that surface; the batched helpers at the bottom use include/speckv_ext.h when
the loaded library exports it (they are no-ops against the reference's library).
"""
import ctypes
from typing import List, NamedTuple, Optional, Sequence

from .speckv_ctypes import SpeckvLib

K_KIND, V_KIND = 0, 1


class KVGeometry(NamedTuple):
    """Shape of one request's KV region (what ``allocate`` was called with)."""
    num_tokens: int
    num_layers: int
    num_heads: int
    head_dim: int
    bytes_per_element: int

    @property
    def total_bytes(self) -> int:
        # K and V, every layer, every position (reference vllm_speckv_backend.py:39-40)
        return 2 * self.num_tokens * self.num_layers * self.num_heads * self.head_dim * self.bytes_per_element

    def linear_entry(self, req_id: int, layer: int, head: int, pos: int, kind: int) -> int:
        """Index of a (head_dim)-sized entry in the layout [req][layer][kind][pos][head]."""
        row = (req_id * self.num_layers + layer) * 2 + kind
        return (row * self.num_tokens + pos) * self.num_heads + head


class CxlSpeckvKVAllocator:
    def __init__(self, lib_path: str, dev_path: str = "/dev/speckv0", page_size: int = 4096):
        self._speckv = SpeckvLib(lib_path, dev_path)
        self._page_size = page_size
        self._handle: Optional[int] = None
        self._geom = KVGeometry(0, 0, 0, 0, 0)

    # ---- the reference surface ------------------------------------------------
    def allocate(self, num_tokens: int, num_layers: int, num_heads: int, head_dim: int, bytes_per_element: int):
        """Reserve the KV region of one request; returns the library handle.
        A second call replaces the live handle (single-request shim, as the reference)."""
        self._geom = KVGeometry(num_tokens, num_layers, num_heads, head_dim, bytes_per_element)
        self._handle = self._speckv.alloc(self._geom.total_bytes, preferred_node=0)
        if "speckv_ext_set_layout" in self._speckv.ext:
            # tells the engine how (req, layer, pos) maps to pages, for speckv_prefetch
            self._speckv.set_layout(self._handle, *self._geom)
        return self._handle

    def get_kv_ptr(self, req_id: int, layer: int, head: int, pos: int, kind: int, entry_bytes: int) -> int:
        """Device address of one KV entry; the covering page is resident on return."""
        out = ctypes.c_void_p()
        status = self._speckv.lib.speckv_access(
            self._handle, self._calc_offset(req_id, layer, head, pos, kind, entry_bytes), entry_bytes, ctypes.byref(out))
        if status != 0:
            raise RuntimeError(f"speckv_access failed: {status}")
        return out.value

    def prefetch_step(self, req_id: int, layer: int, cur_pos: int, recent_tokens: List[int], depth_k: int = 4):
        """Queue the speculative look-ahead of one (request, layer)."""
        history = (ctypes.c_int32 * len(recent_tokens))(*recent_tokens)
        status = self._speckv.lib.speckv_prefetch(req_id, layer, cur_pos, depth_k, history, len(recent_tokens))
        if status != 0:
            raise RuntimeError(f"speckv_prefetch failed: {status}")

    def _calc_offset(self, req_id: int, layer: int, head: int, pos: int, kind: int, entry_bytes: int) -> int:
        """Byte offset of an entry: layout [req][layer][kind][pos][head] x entry_bytes
        (reference vllm_speckv_backend.py:95-100)."""
        return self._geom.linear_entry(req_id, layer, head, pos, kind) * entry_bytes

    # kept for callers that read the reference's private fields
    _num_tokens = property(lambda self: self._geom.num_tokens)
    _num_layers = property(lambda self: self._geom.num_layers)
    _num_heads = property(lambda self: self._geom.num_heads)
    _head_dim = property(lambda self: self._geom.head_dim)
    _bytes_per_element = property(lambda self: self._geom.bytes_per_element)

    # ---- batched helpers (own additions, need speckv_ext_*) -------------------
    @property
    def handle(self) -> Optional[int]:
        return self._handle

    @property
    def lib(self) -> SpeckvLib:
        return self._speckv

    @property
    def geometry(self) -> KVGeometry:
        return self._geom

    def prefetch_decode_step(self, req_ids: Sequence[int], cur_pos: Sequence[int], depth_k: int = 0):
        """One decode step of a batch: every (request, layer) look-ahead in one call,
        drained by one lookup kernel and one fetch kernel.  Returns pages fetched."""
        reqs, layers, pos = [], [], []
        for r, p in zip(req_ids, cur_pos):
            reqs += [r] * self._geom.num_layers
            layers += list(range(self._geom.num_layers))
            pos += [p] * self._geom.num_layers
        self._speckv.prefetch_batch(reqs, layers, pos, [depth_k] * len(reqs))
        return self._speckv.prefetch_flush()

    def block_table(self, req_id: int, layer: int, kind: int, positions: Sequence[int]) -> List[int]:
        """Device addresses of the [pos] rows (all heads) of one layer: a paged-attention block table."""
        entry = self._geom.head_dim * self._geom.bytes_per_element
        return self._speckv.access_batch(
            self._handle, [self._calc_offset(req_id, layer, 0, p, kind, entry) for p in positions])

    def kv_rows(self, req_id: int, layer: int, kind: int, pos_begin: int, pos_end: int):
        """The [pos_begin, pos_end) rows of one layer's K (kind 0) or V (kind 1) as ONE torch fp16 tensor
        [positions][num_heads][head_dim] that aliases the engine's cache slots (no copy): the tensor view over
        ``out_gpu_ptr`` the reference's shim only mentions (vllm_speckv_backend.py:64).  The covering pages are made
        resident by one ``speckv_access`` over the span, which returns them contiguous.  The view stays valid until
        those slots are recycled (ring order) or the handle is freed -- consume it within the step."""
        import torch
        g = self._geom
        row_bytes = g.num_heads * g.head_dim * g.bytes_per_element
        off = self._calc_offset(req_id, layer, 0, pos_begin, kind, g.head_dim * g.bytes_per_element)
        nbytes = (pos_end - pos_begin) * row_bytes
        ptr = self._speckv.access(self._handle, off, nbytes)

        class _Span:                       # minimal __cuda_array_interface__ carrier (works on ROCm builds of torch)
            __cuda_array_interface__ = {"shape": (pos_end - pos_begin, g.num_heads, g.head_dim), "typestr": "<f2",
                                        "data": (ptr, False), "version": 2, "strides": None}
        return torch.as_tensor(_Span(), device="cuda")

    def close(self):
        self._speckv.finalize()
