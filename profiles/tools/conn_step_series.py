import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
import cxl_speckv_amd as pkg
kv = pkg.CxlSpeckvKVAllocator(pkg.library_path(), "hip:0")
if os.environ.get("PRE") == "1":
    bench.ragged_batch_extra(torch, kv, scheme=4, hi=32768, tail=True)
from cxl_speckv_amd.kv_connector import SpeckvKVConnector
n_seq, Lyr, ctx, T = 256, 8, 2048, 4096
conn = SpeckvKVConnector(kv.lib, num_layers=Lyr, max_tokens=T, scheme="fp8")
ids = list(range(n_seq))
g = torch.Generator(device="cuda"); g.manual_seed(2006)
kp = torch.randn((Lyr, ctx, 8, 128), generator=g, device="cuda").to(torch.float16); vp = kp.clone()
for r in ids:
    conn.add_request(r); conn.write_prefill(r, kp, vp)
q = torch.randn((n_seq, 8, 8, 128), generator=g, device="cuda").to(torch.float16)
k = torch.randn((n_seq, Lyr, 8, 128), generator=g, device="cuda").to(torch.float16); v = k.clone()
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    for step in range(12):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        conn.begin_step(ids, depth_k=0); t1 = time.perf_counter()
        for layer in range(Lyr):
            out = conn.attend(layer, ids, q, 0.0884, stream=s)
        torch.cuda.synchronize(); t2 = time.perf_counter()
        keep = conn.append(ids, k, v, stream=s)
        torch.cuda.synchronize(); t3 = time.perf_counter()
        print(step, "begin %.3f attend %.3f append %.3f" % ((t1-t0)*1e3, (t2-t1)*1e3, (t3-t2)*1e3), flush=True)
kv.close()
