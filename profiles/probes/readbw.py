import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from importlib import import_module
b = import_module("cxl-speckv_amd.build")
b._preload_torch_hip_runtime()
lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libreadbw.so"))
T, L = 32768, 80
nbytes = T * L * 2048
buf = torch.randint(0, 255, (nbytes,), dtype=torch.uint8, device="cuda")
out = torch.zeros(4, dtype=torch.int32, device="cuda")
s = torch.cuda.current_stream().cuda_stream
def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    a, c = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    c.record(); torch.cuda.synchronize()
    return a.elapsed_time(c) / reps
V = ctypes.c_void_p
for unr in (1, 4, 8):
    for nt in (0, 1):
        for wgs in (1024, 4096, 16384):
            ms = timeit(lambda: lib.probe_linear(V(buf.data_ptr()), ctypes.c_uint64(nbytes), unr, nt, wgs, V(out.data_ptr()), V(s)))
            print(f"linear unr={unr} nt={nt} wgs={wgs}: {ms:.3f} ms {nbytes/ms/1e9:.2f} TB/s")
for splits in (8, 16, 64):
    for rot in (0, 1):
        ms = timeit(lambda: lib.probe_heads(V(buf.data_ptr()), T, L, splits, rot, V(out.data_ptr()), V(s)))
        print(f"heads splits={splits} rot={rot}: {ms:.3f} ms {nbytes/ms/1e9:.2f} TB/s")
