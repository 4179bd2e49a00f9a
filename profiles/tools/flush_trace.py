import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import cxl_speckv_amd as pkg
T, Lyr = 4096, 32
kv = pkg.CxlSpeckvKVAllocator(pkg.library_path(), "hip:0")
lib = kv.lib
lib.set_compression_scheme(2)
h = kv.allocate(T, Lyr, 8, 128, 2)
n_blocks = T * Lyr * 8 * 128 * 2 * 2 // 4096
src = torch.randn((n_blocks, 2048), device="cuda").to(torch.float16)
lib.write(h, 0, src.data_ptr(), src.numel() * 2, on_device=True)
rng = np.random.default_rng(5)
n_req = 256 * Lyr
reqs = np.zeros(n_req, np.uint32)
layers = (np.arange(n_req) % Lyr).astype(np.uint16)
depth = np.full(n_req, 4, np.uint32)
for rep in range(12):
    pos = rng.integers(0, T - 8, n_req).astype(np.uint32)
    lib.prefetch_batch(reqs, layers, pos, depth)
    t0 = time.perf_counter()
    lib.prefetch_flush(want_count=False)
    t1 = time.perf_counter()
    lib.sync()
    t2 = time.perf_counter()
    print("flush", rep, "submit_us", round((t1 - t0) * 1e6, 1), "total_us", round((t2 - t0) * 1e6, 1), flush=True)

# reference: the same number of random pages through fetch_list (no ring bookkeeping), back to back
pages = torch.from_numpy(rng.permutation(n_blocks)[:19000].astype(np.int32)).cuda()
dst = torch.empty((19000, 2048), dtype=torch.float16, device="cuda")
st = torch.cuda.Stream()
torch.cuda.synchronize()
for rep in range(6):
    lib.fetch_list(h, pages.data_ptr(), 19000, dst.data_ptr(), False, st.cuda_stream)
torch.cuda.synchronize()
