"""fetch+decompress at a given footprint: python footprint.py T LAYERS [passes] [scheme]"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import torch
import cxl_speckv_amd as pkg
T, Lyr = int(sys.argv[1]), int(sys.argv[2])
passes = int(sys.argv[3]) if len(sys.argv) > 3 else 200
scheme = int(sys.argv[4]) if len(sys.argv) > 4 else 2
PAGE = 4096
lib = pkg.SpeckvLib(pkg.library_path(), "hip:0")
lib.set_compression_scheme(scheme)
n = T * Lyr * 8 * 128 * 2 * 2 // PAGE
h = lib.alloc(n * PAGE)
g = torch.Generator(device="cuda"); g.manual_seed(2004)
for p0 in range(0, n, 65536):
    x = torch.randn((min(65536, n - p0), 2048), generator=g, device="cuda", dtype=torch.float32).to(torch.float16)
    lib.write(h, p0 * PAGE, x.data_ptr(), x.numel() * 2, True)
del x
dst = torch.empty((n, 2048), dtype=torch.float16, device="cuda")
s = torch.cuda.Stream()
rec = lib.stats().compressed_bytes
for _ in range(passes // 4 + 5):
    lib.fetch_range(h, 0, n, dst.data_ptr(), False, s.cuda_stream)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record(s)
for _ in range(passes):
    lib.fetch_range(h, 0, n, dst.data_ptr(), False, s.cuda_stream)
b.record(s); torch.cuda.synchronize()
ms = a.elapsed_time(b) / passes
alg = rec + n * (4 + PAGE)
print(f"footprint T={T} L={Lyr} blocks={n} wgs_per_cu={os.environ.get('SPECKV_WGS_PER_CU','default')} ms={ms:.4f} GB/s={alg/ms/1e6:.1f} frac={alg/ms/1e6/8000:.4f}", flush=True)
lib.finalize()
