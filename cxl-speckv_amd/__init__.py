"""cxl-speckv_amd -- MI355X-native speculative KV-cache engine.

Holds only what the hot path needs:
  csrc/                   HIP kernels + engine + the C ABI (-> lib/libcxlspeckv.so)
  speckv_ctypes.py        ctypes binding, same surface as the reference's
                          host/python/speckv_ctypes.py:7-98
  vllm_speckv_backend.py  CxlSpeckvKVAllocator, same surface as the reference's
                          host/python/vllm_speckv_backend.py:8-100
  build.py                hipcc driver (gfx950)

There is no CPU implementation of the data path in this package: without the
built HIP library every constructor raises.
"""
from .build import LIB_PATH, build_library, library_path, load_library  # noqa: F401
from .speckv_ctypes import SpeckvLib, SpeckvError  # noqa: F401
from .vllm_speckv_backend import CxlSpeckvKVAllocator  # noqa: F401

__all__ = ["SpeckvLib", "SpeckvError", "CxlSpeckvKVAllocator", "build_library", "library_path", "load_library", "LIB_PATH"]
