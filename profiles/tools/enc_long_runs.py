"""Block codec (k_compress<2> / k_fetch_decompress<2>) on blocks made of LONG runs (equal values for 200 .. 900 elements: every run
is split at 255 by the reference's rule): python profiles/tools/enc_long_runs.py [lo hi]"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import torch
from tests._gpu import load_raw_lib
lo = int(sys.argv[1]) if len(sys.argv) > 1 else 200
hi = int(sys.argv[2]) if len(sys.argv) > 2 else 900
lib = load_raw_lib()
n = 131072
g = torch.Generator(device="cuda"); g.manual_seed(11)
tot = n * 2048
m = tot // lo + 1
lens = torch.randint(lo, hi, (m,), generator=g, device="cuda")
vals = torch.randn(m, generator=g, device="cuda")
x = torch.repeat_interleave(vals, lens)[:tot].to(torch.float16).reshape(n, 2048).contiguous()
recs = torch.empty((n, 4096), dtype=torch.uint8, device="cuda")
lens_o = torch.empty(n, dtype=torch.int32, device="cuda"); scales = torch.empty(n, dtype=torch.float32, device="cuda")
out = torch.empty((n, 2048), dtype=torch.float16, device="cuda")
s = torch.cuda.Stream()
def enc(): assert lib.speckv_ext_codec_compress(x.data_ptr(), n, recs.data_ptr(), 4096, lens_o.data_ptr(), scales.data_ptr(), 2, 0, s.cuda_stream) == 0
def dec(): assert lib.speckv_ext_codec_decompress(recs.data_ptr(), 4096, lens_o.data_ptr(), scales.data_ptr(), n, out.data_ptr(), 0, 2, 0, s.cuda_stream) == 0
enc(); torch.cuda.synchronize()
comp = int(lens_o.to(torch.int64).sum().item())
for name, fn in (("compress", enc), ("decompress", dec)):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(s)
    for _ in range(20): fn()
    b.record(s); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 20
    byt = n * 4096 + comp + n * 8
    print(f"long runs {lo}..{hi}: {name} {ms*1e3:.1f} us  {n/ms/1e3:.0f} M blocks/s  record {comp/n:.0f} B/block  frac {byt/ms/1e6/8000:.3f}", flush=True)
