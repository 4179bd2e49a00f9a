"""KV allocator shim for a vLLM-style decode loop.

Same class name, constructor and methods as the reference's shim (reference
host/python/vllm_speckv_backend.py:8-100): ``allocate`` / ``get_kv_ptr`` /
``prefetch_step`` / ``_calc_offset`` behave identically and drive the same
C ABI.  The reference file itself cannot be imported (SyntaxError in its example
at line 104) -- this is our own code with the same surface, plus batched helpers
that use include/speckv_ext.h when the loaded library has them.
"""
import ctypes
from typing import Any, Dict, List, Optional, Sequence

from .speckv_ctypes import SpeckvLib


class CxlSpeckvKVAllocator:
    def __init__(self, lib_path: str, dev_path: str = "/dev/speckv0", page_size: int = 4096):
        self._speckv = SpeckvLib(lib_path, dev_path)
        self._page_size = page_size
        self._handle: Optional[int] = None
        self._req_id_counter = 1
        self._req_state: Dict[int, Dict[str, Any]] = {}

        # model configuration (set by allocate)
        self._num_layers = 0
        self._num_heads = 0
        self._num_tokens = 0
        self._head_dim = 0
        self._bytes_per_element = 0

    def allocate(self, num_tokens: int, num_layers: int, num_heads: int, head_dim: int, bytes_per_element: int):
        """Allocate the KV region of one request (K and V, all layers)."""
        self._num_tokens = num_tokens
        self._num_layers = num_layers
        self._num_heads = num_heads
        self._head_dim = head_dim
        self._bytes_per_element = bytes_per_element

        total_bytes = num_tokens * num_layers * num_heads * head_dim * bytes_per_element * 2  # K+V
        self._handle = self._speckv.alloc(total_bytes, preferred_node=0)
        if "speckv_ext_set_layout" in self._speckv.ext:
            # lets the engine map (req, layer, pos) to pages for speckv_prefetch
            self._speckv.set_layout(self._handle, num_tokens, num_layers, num_heads, head_dim, bytes_per_element)
        return self._handle

    def get_kv_ptr(self, req_id: int, layer: int, head: int, pos: int, kind: int, entry_bytes: int) -> int:
        """Device address of one KV entry (resident after the call)."""
        offset = self._calc_offset(req_id, layer, head, pos, kind, entry_bytes)
        gpu_ptr = ctypes.c_void_p()
        ret = self._speckv.lib.speckv_access(self._handle, offset, entry_bytes, ctypes.byref(gpu_ptr))
        if ret != 0:
            raise RuntimeError(f"speckv_access failed: {ret}")
        return gpu_ptr.value

    def prefetch_step(self, req_id: int, layer: int, cur_pos: int, recent_tokens: List[int], depth_k: int = 4):
        """Queue the look-ahead of one (request, layer) for the next tokens."""
        hist_len = len(recent_tokens)
        arr = (ctypes.c_int32 * hist_len)(*recent_tokens)
        ret = self._speckv.lib.speckv_prefetch(req_id, layer, cur_pos, depth_k, arr, hist_len)
        if ret != 0:
            raise RuntimeError(f"speckv_prefetch failed: {ret}")

    def _calc_offset(self, req_id: int, layer: int, head: int, pos: int, kind: int, entry_bytes: int) -> int:
        """Layout [req][layer][kind(K/V)][pos][head] (reference vllm_speckv_backend.py:95-100)."""
        return (
            (((req_id * self._num_layers + layer) * 2 + kind)
             * self._num_tokens + pos) * self._num_heads + head
        ) * entry_bytes

    # ---- batched helpers (own additions) ------------------------------
    @property
    def handle(self) -> Optional[int]:
        return self._handle

    @property
    def lib(self) -> SpeckvLib:
        return self._speckv

    def prefetch_decode_step(self, req_ids: Sequence[int], cur_pos: Sequence[int], depth_k: int = 0):
        """One decode step of a batch: every (request, layer) look-ahead in one call,
        drained by one lookup kernel + one fetch kernel."""
        reqs, layers, pos = [], [], []
        for r, p in zip(req_ids, cur_pos):
            for layer in range(self._num_layers):
                reqs.append(r); layers.append(layer); pos.append(p)
        self._speckv.prefetch_batch(reqs, layers, pos, [depth_k] * len(reqs))
        return self._speckv.prefetch_flush()

    def block_table(self, req_id: int, layer: int, kind: int, positions: Sequence[int]) -> List[int]:
        """Device addresses of the [pos] rows (all heads) of one layer: a paged-attention block table."""
        eb = self._head_dim * self._bytes_per_element
        offs = [self._calc_offset(req_id, layer, 0, p, kind, eb) for p in positions]
        return self._speckv.access_batch(self._handle, offs)

    def close(self):
        self._speckv.finalize()
