import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
import cxl_speckv_amd as pkg
from bench import ramp, PAGE, BLOCK_ELEMS, HBM_PEAK_GBPS, EXTRAS_RAMP_MS
kv = pkg.CxlSpeckvKVAllocator(pkg.library_path(), "hip:0")
lib = kv.lib
scheme = int(sys.argv[1]) if len(sys.argv) > 1 else 3
n_seq, T = int(os.environ.get("NSEQ", "256")), int(sys.argv[2]) if len(sys.argv) > 2 else 8192
lib.set_compression_scheme(scheme)
g = torch.Generator(device="cuda"); g.manual_seed(2004)
n_pages = T * 8 * 128 * 2 * 2 // PAGE
x = torch.randn((n_pages, BLOCK_ELEMS), generator=g, device="cuda", dtype=torch.float32).to(torch.float16)
handles = []
for _ in range(n_seq):
    h = lib.alloc(n_pages * PAGE); lib.set_layout(h, T, 1, 8, 128, 2); lib.write(h, 0, x.data_ptr(), x.numel() * 2, True); handles.append(h)
q = torch.randn((n_seq, 8, 8, 128), generator=g, device="cuda", dtype=torch.float32).to(torch.float16)
o = torch.empty((n_seq, 8, 8, 128), dtype=torch.float32, device="cuda")
s = torch.cuda.Stream()
lens = [T] * n_seq
fn = lib.attend_int4_batch if scheme == 3 else lib.attend_fp8_batch
def step(): fn(handles, 0, q.data_ptr(), 8, lens, 0.0883883, o.data_ptr(), None, s.cuda_stream)
for tps in ([None] + [int(v) for v in sys.argv[3:]]):
    if tps: os.environ["SPECKV_ATTEND_TILES_PER_SPLIT"] = str(tps)
    step(); torch.cuda.synchronize()
    ramp(step, torch.cuda.synchronize, EXTRAS_RAMP_MS)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(s)
    for _ in range(10): step()
    b.record(s); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 10
    rb = n_seq * n_pages * (1152 if scheme == 3 else 2048)
    print("scheme", scheme, "T", T, "tps", tps, "ms", round(ms, 4), "frac", round(rb / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4), flush=True)
