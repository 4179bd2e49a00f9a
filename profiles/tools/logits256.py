# the reference-cell predictor for n requests, 30 times (PMC / trace target): python profiles/tools/logits256.py [n]
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import cxl_speckv_amd as pkg
kv = pkg.CxlSpeckvKVAllocator(pkg.library_path(), "hip:0")
lib = kv.lib
g = torch.Generator(device="cuda"); g.manual_seed(10)
rnd = lambda *s_: (torch.rand(s_, generator=g, device="cuda") - 0.5) * 0.2
emb, wout = rnd(32000, 64), rnd(32000, 128)
lib.predictor_load(emb.data_ptr(), wout.data_ptr(), 32000, True)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
hist = torch.randint(0, 32000, (n, 16), generator=g, device="cuda", dtype=torch.int32)
tok = torch.empty((n, 4), dtype=torch.int32, device="cuda"); conf = torch.empty((n, 4), dtype=torch.float32, device="cuda")
s = torch.cuda.Stream()
for _ in range(30):
    lib.predict_batch(n, hist.data_ptr(), 4, tok.data_ptr(), conf.data_ptr(), s.cuda_stream)
torch.cuda.synchronize()
