import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, numpy as np
import cxl_speckv_amd as pkg
kv = pkg.CxlSpeckvKVAllocator(pkg.library_path(), "hip:0")
lib = kv.lib
lib.set_compression_scheme(2)
T, L = 4096, 32
h = kv.allocate(T, L, 8, 128, 2)
n = T * L * 8 * 128 * 2 * 2 // 4096
g = torch.Generator(device="cuda"); g.manual_seed(2001)
src = torch.randn((n, 2048), generator=g, device="cuda").to(torch.float16)
lib.write(h, 0, src.data_ptr(), src.numel() * 2, True)
dst = torch.empty_like(src)
s = torch.cuda.Stream()
K = 200
evs = [torch.cuda.Event(enable_timing=True) for _ in range(K + 1)]
for _ in range(5): lib.fetch_range(h, 0, n, dst.data_ptr(), False, s.cuda_stream)
torch.cuda.synchronize()
evs[0].record(s)
for i in range(K):
    lib.fetch_range(h, 0, n, dst.data_ptr(), False, s.cuda_stream)
    evs[i + 1].record(s)
torch.cuda.synchronize()
t = np.array([evs[i].elapsed_time(evs[i + 1]) * 1e3 for i in range(K)])
for lo in range(0, K, 20):
    print(f"steps {lo:3d}-{lo+19:3d}: mean {t[lo:lo+20].mean():.1f} us  min {t[lo:lo+20].min():.1f}  max {t[lo:lo+20].max():.1f}")
