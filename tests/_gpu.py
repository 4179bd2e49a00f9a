"""Helpers for the -m gpu tests: drive libcxlspeckv.so through its C ABI with
torch tensors as plain device buffers (torch is plumbing only)."""
import contextlib
import ctypes as C
import gc

import numpy as np

import cxl_speckv_amd as pkg
from cxl_speckv_amd.speckv_ctypes import bind_ext

N = 2048


def load_raw_lib():
    """The library without speckv_init: raw codec / verify operators only."""
    lib = pkg.load_library()
    bind_ext(lib)
    return lib


def load_debug_lib():
    """tests/_build/libspeckv_debug.so: TEST-ONLY self-check kernels over the product's device helpers
    (tests/csrc/debug_kernels.hip); built on demand, never part of libcxlspeckv.so."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call(["make", "-s", "-C", os.path.join(root, "tests", "csrc")])
    pkg.load_library()                      # one HIP runtime in the process (torch's), mapped first
    return C.CDLL(os.path.join(root, "tests", "_build", "libspeckv_debug.so"))


_TUNE_WORDS = {"wg": 1, "serial": 2, "kernel": 1, "copy": 2}


def set_tuning(key, value):
    """speckv_ext_set_tuning: the library reads its environment once; tests that flip a launch form between two calls say so
    through the C ABI (include/speckv_ext.h).  Words of the old environment values ("wg", "serial", ...) are accepted."""
    lib = pkg.load_library()
    lib.speckv_ext_set_tuning.argtypes = [C.c_char_p, C.c_longlong]
    lib.speckv_ext_set_tuning.restype = C.c_int
    rc = lib.speckv_ext_set_tuning(key.encode(), int(_TUNE_WORDS.get(value, value)))
    assert rc == 0, (key, value, rc)


def torch_mod():
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch


@contextlib.contextmanager
def graph_capture(graph, stream):
    """torch.cuda.graph with the cyclic garbage collector held off: a collection that runs inside the capture may
    destroy a graph or stream left in a reference cycle by an EARLIER test (pytest.raises keeps frames alive), and HIP
    calls of that kind abort a global-mode capture."""
    torch = torch_mod()
    gc.collect()
    was_enabled = gc.isenabled()
    gc.disable()
    try:
        with torch.cuda.graph(graph, stream=stream):
            yield
    finally:
        if was_enabled:
            gc.enable()


def stream_ptr():
    torch = torch_mod()
    return torch.cuda.current_stream().cuda_stream


def gpu_compress(lib, x16, scheme, mode, rec_stride=4096):
    """x16: (B, 2048) float16 numpy.  Returns scales f32[B], lens u32[B], recs u8[B, stride]."""
    torch = torch_mod()
    x16 = np.ascontiguousarray(x16, dtype=np.float16).reshape(-1, N)
    B = x16.shape[0]
    d_x = torch.from_numpy(x16.view(np.int16)).cuda()
    d_recs = torch.full((B, rec_stride), 0xA5, dtype=torch.uint8, device="cuda")
    d_len = torch.zeros(B, dtype=torch.int32, device="cuda")
    d_scale = torch.zeros(B, dtype=torch.float32, device="cuda")
    rc = lib.speckv_ext_codec_compress(d_x.data_ptr(), B, d_recs.data_ptr(), rec_stride, d_len.data_ptr(),
                                       d_scale.data_ptr(), scheme, mode, stream_ptr())
    assert rc == 0, rc
    torch.cuda.synchronize()
    return (d_scale.cpu().numpy(), d_len.cpu().numpy().view(np.uint32), d_recs.cpu().numpy())


def gpu_decompress(lib, recs, lens, scales, scheme, mode, out_f32=False):
    torch = torch_mod()
    recs = np.ascontiguousarray(recs, dtype=np.uint8)
    B, stride = recs.shape
    d_recs = torch.from_numpy(recs).cuda()
    d_len = torch.from_numpy(np.ascontiguousarray(lens, dtype=np.uint32).view(np.int32)).cuda()
    d_scale = torch.from_numpy(np.ascontiguousarray(scales, dtype=np.float32)).cuda()
    if out_f32:
        d_y = torch.full((B, N), float("nan"), dtype=torch.float32, device="cuda")
    else:
        d_y = torch.full((B, N), 0x7E00, dtype=torch.int16, device="cuda")
    rc = lib.speckv_ext_codec_decompress(d_recs.data_ptr(), stride, d_len.data_ptr(), d_scale.data_ptr(), B,
                                         d_y.data_ptr(), int(out_f32), scheme, mode, stream_ptr())
    assert rc == 0, rc
    torch.cuda.synchronize()
    y = d_y.cpu().numpy()
    return y if out_f32 else y.view(np.float16)


def assert_same_float_bits(a, b, what=""):
    """Bit-exact except that NaNs only have to be NaNs on both sides."""
    a = np.asarray(a); b = np.asarray(b)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    na, nb = np.isnan(a), np.isnan(b)
    assert np.array_equal(na, nb), f"{what}: NaN positions differ"
    ua = a.view(np.uint16 if a.dtype == np.float16 else np.uint32)
    ub = b.view(np.uint16 if b.dtype == np.float16 else np.uint32)
    bad = (ua != ub) & ~na
    assert not bad.any(), f"{what}: {int(bad.sum())} elements differ, first at {np.argwhere(bad)[0]}"


def _hip_runtime():
    """The one HIP runtime already mapped in this process (torch's bundled copy)."""
    import os
    torch = torch_mod()
    path = os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so")
    rt = C.CDLL(path if os.path.exists(path) else "libamdhip64.so")
    rt.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    rt.hipMemcpy.restype = C.c_int
    return rt


def dev_to_host(ptr, nbytes):
    """Copy nbytes from a raw device address (as returned by speckv_access)."""
    torch_mod().cuda.synchronize()
    buf = np.empty(nbytes, np.uint8)
    rc = _hip_runtime().hipMemcpy(buf.ctypes.data, C.c_void_p(ptr), nbytes, 2)   # hipMemcpyDeviceToHost
    assert rc == 0, rc
    return buf


def stored_record(info, nbytes):
    """The first nbytes of a page's pool record as speckv_ext_translate describes it: contiguous at pool_addr, or -- tile-planar
    MXFP4, aux_offset != 0 -- bytes 0..1023 there and the 64 codes aux_offset further on (include/speckv_ext.h)."""
    nbytes = int(nbytes)
    if not info.aux_offset or nbytes <= 1024:
        return dev_to_host(info.pool_addr, nbytes)
    return np.concatenate([dev_to_host(info.pool_addr, 1024), dev_to_host(info.pool_addr + info.aux_offset, nbytes - 1024)])
