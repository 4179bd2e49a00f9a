"""speckv_access on a miss, call by call: python profiles/tools/access_miss.py [n]   (run it under rocprofv3 --kernel-trace --stats
for the kernel's own duration).  Prints the median wall time of a miss, of a hit, and of speckv_ext_sync right after a miss."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import cxl_speckv_amd as pkg
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
kv = pkg.CxlSpeckvKVAllocator(pkg.library_path(), "hip:0")
lib = kv.lib
lib.set_compression_scheme(2)
T, L = 4096, 8
h = kv.allocate(T, L, 8, 128, 2)
n_pages = T * L * 8 * 128 * 2 * 2 // 4096
x = np.random.default_rng(1).standard_normal((n_pages, 2048)).astype(np.float16)
lib.write(h, 0, x.ctypes.data, x.nbytes, False)
lib.sync()
miss, hit = [], []
pages = np.random.default_rng(2).permutation(n_pages)[:n]
for p in pages[:50]:
    lib.access(h, int(p) * 4096, 1)
for p in pages[50:]:
    t0 = time.perf_counter_ns(); lib.access(h, int(p) * 4096, 1); t1 = time.perf_counter_ns()
    lib.access(h, int(p) * 4096, 1); t2 = time.perf_counter_ns()
    miss.append(t1 - t0); hit.append(t2 - t1)
q = lambda v, f: float(np.percentile(np.array(v) / 1e3, f))
print(f"access miss: median {q(miss, 50):.2f} us  p10 {q(miss, 10):.2f}  p90 {q(miss, 90):.2f}   (python ctypes call included, ~0.8 us)")
print(f"access hit : median {q(hit, 50):.2f} us  p10 {q(hit, 10):.2f}  p90 {q(hit, 90):.2f}")
st = lib.stats()
print("l2_misses", st.l2_misses, "l2_hits", st.l2_hits)
kv.close()
