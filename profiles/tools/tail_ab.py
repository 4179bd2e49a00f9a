"""A/B of one planned MXFP4 layer with and without the tail position (256 x 2k): python profiles/tools/tail_ab.py [scheme]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, ctypes
import cxl_speckv_amd as pkg
from cxl_speckv_amd.kv_connector import SpeckvKVConnector
SCH = sys.argv[1] if len(sys.argv) > 1 else "mxfp4"
SID = {"fp8": 4, "int4": 3, "mxfp4": 5}[SCH]
n_seq, Lyr, ctx, T = 256, 8, 2048, int(os.environ.get("TMAX", "4096"))
lib = pkg.SpeckvLib(pkg.library_path(), "hip:0")
conn = SpeckvKVConnector(lib, num_layers=Lyr, max_tokens=T, scheme=SCH)
ids = list(range(n_seq))
g = torch.Generator(device="cuda"); g.manual_seed(1)
kp = torch.randn((Lyr, ctx, 8, 128), generator=g, device="cuda").to(torch.float16)
vp = torch.randn((Lyr, ctx, 8, 128), generator=g, device="cuda").to(torch.float16)
for r in ids:
    conn.add_request(r); conn.write_prefill(r, kp, vp)
q = torch.randn((n_seq, 8, 8, 128), generator=g, device="cuda").to(torch.float16)
kt = torch.randn((n_seq, Lyr, 8, 128), generator=g, device="cuda").to(torch.float16)
vt = torch.randn((n_seq, Lyr, 8, 128), generator=g, device="cuda").to(torch.float16)
s = torch.cuda.Stream()
out = torch.empty((n_seq, 8, 8, 128), dtype=torch.float32, device="cuda"); lse = torch.empty((n_seq, 8, 8), dtype=torch.float32, device="cuda")
bound = conn.plan_step(ids, s)
plain = lambda l: lib.attend_planned(SID, conn._plan.data_ptr(), n_seq, l, q.data_ptr(), 8, bound, 0.0884, out.data_ptr(), lse.data_ptr(), s.cuda_stream)
tail = lambda l: lib.attend_planned_tail(SID, conn._plan.data_ptr(), n_seq, l, q.data_ptr(), 8, bound, 0.0884, out.data_ptr(), lse.data_ptr(), n_seq, 0, 0,
                                         kt.data_ptr(), vt.data_ptr(), Lyr * 8 * 128, s.cuda_stream)
for name, fn in (("planned", plain), ("planned_tail", tail), ("planned", plain), ("planned_tail", tail)):
    with torch.cuda.stream(s):
        for _ in range(5):
            for l in range(Lyr): fn(l)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(s)
        for _ in range(20):
            for l in range(Lyr): fn(l)
        b.record(s); torch.cuda.synchronize()
    print(SCH, name, "us per layer", round(a.elapsed_time(b) / 20 / Lyr * 1e3, 2))
# all layers in one call (MXFP4: one launch over layers x sequences), with the two-halves workgroups and without
import bench
qall = q[None].expand(Lyr, -1, -1, -1, -1).contiguous()
outall = torch.empty((Lyr, n_seq, 8, 8, 128), dtype=torch.float32, device="cuda"); lseall = torch.empty((Lyr, n_seq, 8, 8), dtype=torch.float32, device="cuda")
alll = lambda: lib.attend_planned_layers(SID, conn._plan.data_ptr(), n_seq, 0, Lyr, qall.data_ptr(), 8, bound, 0.0884, outall.data_ptr(), lseall.data_ptr(), s.cuda_stream)
for half in (0, 1, 0, 1):
    bench.set_tuning("attend_mx4_one_half", half)
    with torch.cuda.stream(s):
        for _ in range(5): alll()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(s)
        for _ in range(20): alll()
        b.record(s); torch.cuda.synchronize()
    print(SCH, "layers in one call, one_half =", half, "us per layer", round(a.elapsed_time(b) / 20 / Lyr * 1e3, 2))
bench.set_tuning("attend_mx4_one_half", 0)
