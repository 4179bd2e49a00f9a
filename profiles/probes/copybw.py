import ctypes, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from importlib import import_module
b = import_module("cxl-speckv_amd.build"); b._preload_torch_hip_runtime()
lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libcopybw.so"))
nbytes = 512 << 20
src = torch.randint(0, 255, (nbytes,), dtype=torch.uint8, device="cuda"); dst = torch.empty_like(src)
s = torch.cuda.current_stream().cuda_stream
V = ctypes.c_void_p
def timeit(fn, reps=20):
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.05:
        for _ in range(10): fn()
        torch.cuda.synchronize()
    a, c = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    c.record(); torch.cuda.synchronize()
    return a.elapsed_time(c) / reps
for unr in (1, 4, 8, 16):
    for nt in (0, 1):
        for wgs in (2048, 8192, 32768):
            ms = timeit(lambda: lib.probe_copy(V(src.data_ptr()), V(dst.data_ptr()), ctypes.c_uint64(nbytes), unr, nt, wgs, V(s)))
            print(f"copy unr={unr:2d} nt={nt} wgs={wgs:5d}: {ms*1e3:.1f} us {2*nbytes/ms/1e9:.2f} TB/s ({2*nbytes/ms/1e9/8:.3f})", flush=True)
