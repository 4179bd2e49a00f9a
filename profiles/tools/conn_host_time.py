"""Host time of the pieces of a connector decode step's front end (begin_step, plan_step), GPU idle at the start of each.
python profiles/tools/conn_host_time.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import cxl_speckv_amd as pkg
from cxl_speckv_amd.kv_connector import SpeckvKVConnector
n_seq, Lyr, ctx, T = 256, 8, 2048, 4096
lib = pkg.SpeckvLib(pkg.library_path(), "hip:0")
conn = SpeckvKVConnector(lib, num_layers=Lyr, max_tokens=T, scheme="fp8")
ids = list(range(n_seq))
g = torch.Generator(device="cuda"); g.manual_seed(1)
kp = torch.randn((Lyr, ctx, 8, 128), generator=g, device="cuda").to(torch.float16)
for r in ids:
    conn.add_request(r); conn.write_prefill(r, kp, kp)
s = torch.cuda.Stream()
def t(fn, reps=200):
    fn(); torch.cuda.synchronize()
    tot = 0.0
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter(); fn(); tot += time.perf_counter() - t0
    return tot / reps * 1e6
L = Lyr
def prep():
    reqs = np.repeat(np.asarray(ids, dtype=np.uint32), L)
    layers = np.tile(np.arange(L, dtype=np.uint16), len(ids))
    pos = np.repeat(np.asarray([max(conn.requests[r].length - 1, 0) for r in ids], dtype=np.uint32), L)
    depth = np.full(len(ids) * L, 0, np.uint32)
    return reqs, layers, pos, depth
arrs = prep()
print("numpy columns        %.1f us" % t(prep))
print("prefetch_batch       %.1f us" % t(lambda: lib.prefetch_batch(*arrs)))
def pf(): lib.prefetch_batch(*arrs); lib.prefetch_flush(want_count=False)
print("batch + flush        %.1f us" % t(pf))
print("begin_step           %.1f us" % t(lambda: conn.begin_step(ids, depth_k=0)))
print("plan_step            %.1f us" % t(lambda: conn.plan_step(ids, s)))
