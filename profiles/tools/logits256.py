# the reference-cell predictor for n requests, 30 times (PMC / trace target): python profiles/tools/logits256.py [n] [vocab]
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import cxl_speckv_amd as pkg
kv = pkg.CxlSpeckvKVAllocator(pkg.library_path(), "hip:0")
lib = kv.lib
g = torch.Generator(device="cuda"); g.manual_seed(10)
rnd = lambda *s_: (torch.rand(s_, generator=g, device="cuda") - 0.5) * 0.2
V = int(sys.argv[2]) if len(sys.argv) > 2 else 32000
emb, wout = rnd(V, 64), rnd(V, 128)
lib.predictor_load(emb.data_ptr(), wout.data_ptr(), V, True)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
hist = torch.randint(0, V, (n, 16), generator=g, device="cuda", dtype=torch.int32)
tok = torch.empty((n, 4), dtype=torch.int32, device="cuda"); conf = torch.empty((n, 4), dtype=torch.float32, device="cuda")
s = torch.cuda.Stream()
for _ in range(30):
    lib.predict_batch(n, hist.data_ptr(), 4, tok.data_ptr(), conf.data_ptr(), s.cuda_stream)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record(s)
for _ in range(20):
    lib.predict_batch(n, hist.data_ptr(), 4, tok.data_ptr(), conf.data_ptr(), s.cuda_stream)
b.record(s); torch.cuda.synchronize()
print('predict', n, 'requests, vocab', V, ':', round(a.elapsed_time(b) / 20 * 1e3, 1), 'us')
