import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import cxl_speckv_amd as pkg
n_seq, Lyr, T = 256, 80, 128
os.environ["SPECKV_L2_MB"] = "2048"
kv = pkg.CxlSpeckvKVAllocator(pkg.library_path(), "hip:0")
lib = kv.lib
lib.set_compression_scheme(2)
for s_ in range(n_seq):
    h = lib.alloc(2 * T * Lyr * 8 * 128 * 2)
    lib.set_layout(h, T, Lyr, 8, 128, 2)
    lib.bind_request(s_, h, 0)
n_req = n_seq * Lyr
reqs = np.repeat(np.arange(n_seq, dtype=np.uint32), Lyr)
layers = np.tile(np.arange(Lyr, dtype=np.uint16), n_seq)
depth = np.full(n_req, 4, np.uint32)
for rep in range(8):
    pos = np.full(n_req, 8 * rep, np.uint32)
    lib.prefetch_batch(reqs, layers, pos, depth)
    t0 = time.perf_counter()
    lib.prefetch_flush(want_count=False)
    t1 = time.perf_counter()
    lib.sync()
    t2 = time.perf_counter()
    print("flush", rep, "submit_us", round((t1 - t0) * 1e6, 1), "total_us", round((t2 - t0) * 1e6, 1), flush=True)
