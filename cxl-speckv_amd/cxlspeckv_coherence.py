"""Coherence shadow directory: Python surface over libcxlspeckv.so's coherence_manager_* exports.

Names and behaviour of the reference's binding (host/python/cxlspeckv_coherence.py:41-338:
``CoherenceState``, ``MemoryTier``, ``CoherenceManager`` and its methods), written from the C
interface in include/speckv_coherence.h.  Unlike the reference module this one does not load the
library at import time (the library is located through cxl_speckv_amd.library_path(), or the
``lib_path`` argument), and ``request_read`` hands back the zero-filled buffer the reference hands
back -- no call of this directory moves data (see the header).
"""
import ctypes as _C
from enum import IntEnum
from typing import Dict, List, Optional

from .build import load_library

_STAT_FIELDS = ("total_reads", "total_writes", "coherence_ops", "invalidations_sent", "writebacks_performed",
                "directory_hits", "directory_misses")


class CoherenceState(IntEnum):
    INVALID = 0
    SHARED = 1
    EXCLUSIVE = 2
    MODIFIED = 3


class MemoryTier(IntEnum):
    L1_GPU = 0
    L2_PREFETCH = 1
    L3_CXL = 2


class _Stats(_C.Structure):
    _fields_ = [(name, _C.c_uint64) for name in _STAT_FIELDS]


def bind_coherence(lib):
    """Declare the C signatures (include/speckv_coherence.h) on a loaded CDLL; returns it."""
    H, U64, SZ, VP = _C.c_void_p, _C.c_uint64, _C.c_size_t, _C.c_void_p
    table = {
        "coherence_manager_create": (H, [_C.c_char_p, SZ]),
        "coherence_manager_destroy": (None, [H]),
        "coherence_manager_request_read": (_C.c_bool, [H, U64, VP, SZ]),
        "coherence_manager_request_write": (_C.c_bool, [H, U64, VP, SZ]),
        "coherence_manager_invalidate": (_C.c_bool, [H, U64]),
        "coherence_manager_writeback": (_C.c_bool, [H, U64, VP, SZ]),
        "coherence_manager_flush_all": (_C.c_bool, [H]),
        "coherence_manager_get_state": (_C.c_int, [H, U64]),
        "coherence_manager_get_tier": (_C.c_int, [H, U64]),
        "coherence_manager_promote_to_l1": (_C.c_bool, [H, U64]),
        "coherence_manager_demote_to_l3": (_C.c_bool, [H, U64]),
        "coherence_manager_batch_invalidate": (_C.c_bool, [H, _C.POINTER(U64), SZ]),
        "coherence_manager_get_statistics": (None, [H, _C.POINTER(_Stats)]),
        "coherence_manager_reset_statistics": (None, [H]),
        "coherence_manager_ext_update_tier": (None, [H, U64, _C.c_int]),
        "coherence_manager_ext_entry_count": (SZ, [H]),
    }
    for name, (res, args) in table.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    return lib


class CoherenceManager:
    def __init__(self, device_path: str = "/dev/speckv0", cache_line_size: int = 64, lib_path: Optional[str] = None):
        self._lib = bind_coherence(load_library(lib_path))
        self._handle = self._lib.coherence_manager_create(device_path.encode("utf-8"), cache_line_size)
        if not self._handle:
            raise RuntimeError("Failed to create CoherenceManager")

    def close(self):
        if getattr(self, "_handle", None):
            self._lib.coherence_manager_destroy(self._handle)
            self._handle = None

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, exc_type, exc_val, exc_tb):
        self.flush_all()
        return False

    # ---- operations -------------------------------------------------------
    def request_read(self, addr: int, size: int) -> Optional[bytes]:
        buf = _C.create_string_buffer(size)
        return bytes(buf) if self._lib.coherence_manager_request_read(self._handle, addr, buf, size) else None

    def request_write(self, addr: int, data: bytes) -> bool:
        return bool(self._lib.coherence_manager_request_write(self._handle, addr, data, len(data)))

    def invalidate(self, addr: int) -> bool:
        return bool(self._lib.coherence_manager_invalidate(self._handle, addr))

    def writeback(self, addr: int, data: bytes) -> bool:
        return bool(self._lib.coherence_manager_writeback(self._handle, addr, data, len(data)))

    def flush_all(self) -> bool:
        return bool(self._lib.coherence_manager_flush_all(self._handle))

    def batch_invalidate(self, addrs: List[int]) -> bool:
        arr = (_C.c_uint64 * len(addrs))(*addrs)
        return bool(self._lib.coherence_manager_batch_invalidate(self._handle, arr, len(addrs)))

    def promote_to_l1(self, addr: int) -> bool:
        return bool(self._lib.coherence_manager_promote_to_l1(self._handle, addr))

    def demote_to_l3(self, addr: int) -> bool:
        return bool(self._lib.coherence_manager_demote_to_l3(self._handle, addr))

    def update_tier(self, addr: int, tier: MemoryTier):          # additive (CoherenceManager::update_tier)
        self._lib.coherence_manager_ext_update_tier(self._handle, addr, int(tier))

    # ---- queries ----------------------------------------------------------
    def get_state(self, addr: int) -> CoherenceState:
        return CoherenceState(self._lib.coherence_manager_get_state(self._handle, addr))

    def get_tier(self, addr: int) -> MemoryTier:
        return MemoryTier(self._lib.coherence_manager_get_tier(self._handle, addr))

    def is_valid(self, addr: int) -> bool:
        return self.get_state(addr) != CoherenceState.INVALID

    def is_modified(self, addr: int) -> bool:
        return self.get_state(addr) == CoherenceState.MODIFIED

    def entry_count(self) -> int:                                   # additive
        return int(self._lib.coherence_manager_ext_entry_count(self._handle))

    def get_statistics(self) -> Dict[str, float]:
        st = _Stats()
        self._lib.coherence_manager_get_statistics(self._handle, _C.byref(st))
        out = {name: int(getattr(st, name)) for name in _STAT_FIELDS}
        looked_up = out["directory_hits"] + out["directory_misses"]
        out["hit_rate"] = out["directory_hits"] / looked_up if looked_up else 0.0
        return out

    def reset_statistics(self):
        self._lib.coherence_manager_reset_statistics(self._handle)
