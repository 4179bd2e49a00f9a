import sys, time
sys.path.insert(0, "/root/repo")
import torch, numpy as np
import cxl_speckv_amd as pkg
kv = pkg.CxlSpeckvKVAllocator(pkg.library_path(), "hip:0"); lib = kv.lib
g = torch.Generator(device="cuda"); g.manual_seed(9)
emb = (torch.rand((32000, 64), generator=g, device="cuda") - 0.5) * 0.1
wout = (torch.rand((32000, 128), generator=g, device="cuda") - 0.5) * 0.1
lib.predictor_load(emb.data_ptr(), wout.data_ptr(), 32000, True)
n = 1
hist = torch.randint(0, 32000, (n, 16), generator=g, device="cuda", dtype=torch.int32)
tok = torch.empty((n, 4), dtype=torch.int32, device="cuda"); conf = torch.empty((n, 4), dtype=torch.float32, device="cuda")
s = torch.cuda.Stream()
def call(): lib.predict_batch(n, hist.data_ptr(), 4, tok.data_ptr(), conf.data_ptr(), s.cuda_stream)
call(); torch.cuda.synchronize()
def timeit(fn, reps=50):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(s)
    for _ in range(reps): fn()
    b.record(s); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
print("eager back to back", round(timeit(call), 2), "us")
# one call at a time, wall clock to completion
ws = []
for _ in range(200):
    torch.cuda.synchronize(); t0 = time.perf_counter_ns(); call(); s.synchronize(); ws.append((time.perf_counter_ns() - t0) / 1e3)
print("eager one at a time: launch + wait wall", round(float(np.median(ws)), 2), "us")
try:
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr, stream=s):
        call()
    torch.cuda.synchronize()
    want = tok.clone()
    tok.zero_(); gr.replay(); torch.cuda.synchronize()
    assert torch.equal(tok, want)
    def rep(): gr.replay()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(50): gr.replay()
    b.record(); torch.cuda.synchronize()
    print("graph replay back to back", round(a.elapsed_time(b) / 50 * 1e3, 2), "us")
    ws = []
    for _ in range(200):
        torch.cuda.synchronize(); t0 = time.perf_counter_ns(); gr.replay(); torch.cuda.synchronize(); ws.append((time.perf_counter_ns() - t0) / 1e3)
    print("graph one at a time: replay + wait wall", round(float(np.median(ws)), 2), "us")
except Exception as e:
    print("graph capture failed:", repr(e)[:300])
kv.close()
