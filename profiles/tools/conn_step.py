"""Where a connector decode step spends its time: host time of each phase (no sync inside), device time of the step, and
the same attention layers replayed from a HIP graph (plan_step outside, planned launches captured).
Run on the GPU box:  python profiles/tools/conn_step.py [fp8|int4|mxfp4]"""
import os, sys, time, gc
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import cxl_speckv_amd as pkg
from cxl_speckv_amd.kv_connector import SpeckvKVConnector

n_seq, Lyr, ctx, T = 256, 8, 2048, 4096
lib = pkg.SpeckvLib(pkg.library_path(), "hip:0")
SCH = sys.argv[1] if len(sys.argv) > 1 else "fp8"
SID = {"fp8": 4, "int4": 3, "mxfp4": 5}[SCH]
REC = {"fp8": 1024, "int4": 576, "mxfp4": 544}[SCH]          # record bytes per position and kind
conn = SpeckvKVConnector(lib, num_layers=Lyr, max_tokens=T, scheme=SCH)
print("pool format:", SCH)
ids = list(range(n_seq))
g = torch.Generator(device="cuda"); g.manual_seed(1)
kp = torch.randn((Lyr, ctx, 8, 128), generator=g, device="cuda").to(torch.float16)
vp = torch.randn((Lyr, ctx, 8, 128), generator=g, device="cuda").to(torch.float16)
for r in ids:
    conn.add_request(r); conn.write_prefill(r, kp, vp)
q = torch.randn((n_seq, 8, 8, 128), generator=g, device="cuda").to(torch.float16)
k = torch.randn((n_seq, Lyr, 8, 128), generator=g, device="cuda").to(torch.float16)
v = torch.randn((n_seq, Lyr, 8, 128), generator=g, device="cuda").to(torch.float16)
s = torch.cuda.Stream()
rows = []
with torch.cuda.stream(s):
    for step in range(10):
        torch.cuda.synchronize()
        e = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        t0 = time.perf_counter(); e[0].record(s)
        conn.begin_step(ids, depth_k=0)
        t1 = time.perf_counter(); e[1].record(s)
        for layer in range(Lyr):
            out = conn.attend(layer, ids, q, 0.0884, stream=s)
        t2 = time.perf_counter(); e[2].record(s)
        keep = conn.append(ids, k, v, stream=s)
        t3 = time.perf_counter(); e[3].record(s)
        torch.cuda.synchronize()
        t4 = time.perf_counter()
        rows.append((step, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t0) * 1e3,
                     e[0].elapsed_time(e[1]), e[1].elapsed_time(e[2]), e[2].elapsed_time(e[3])))
print("step | host: begin attend append | wall | device: begin attend append")
for r in rows:
    print("%2d | %.3f %.3f %.3f | %.3f | %.3f %.3f %.3f" % r)

# the attention layers of a step as one graph
out = torch.empty((Lyr, n_seq, 8, 8, 128), dtype=torch.float32, device="cuda")
lse = torch.empty((Lyr, n_seq, 8, 8), dtype=torch.float32, device="cuda")
bound = conn.plan_step(ids, s)
def layers():
    for layer in range(Lyr):
        lib.attend_planned(SID, conn._plan.data_ptr(), n_seq, layer, q.data_ptr(), 8, bound, 0.0884, out[layer].data_ptr(), lse[layer].data_ptr(), s.cuda_stream)
layers(); torch.cuda.synchronize()
gr = torch.cuda.CUDAGraph()
gc.collect(); gc.disable()
with torch.cuda.graph(gr, stream=s):
    layers()
gc.enable()
for name, fn in (("eager planned x%d" % Lyr, layers), ("graph replay", gr.replay)):
    with torch.cuda.stream(s):
        for _ in range(20): fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter(); a.record(s)
        for _ in range(50): fn()
        host = (time.perf_counter() - t0) / 50 * 1e3
        b.record(s); torch.cuda.synchronize()
        ms = a.elapsed_time(b) / 50
    gb = n_seq * Lyr * 2 * ctx * REC / 1e9
    print("%-18s host %.3f ms  device %.3f ms  %.0f GB/s (%.3f of 8 TB/s)" % (name, host, ms, gb / (ms * 1e-3), gb / (ms * 1e-3) / 8000))
t0 = time.perf_counter()
for _ in range(50): conn.plan_step(ids, s)
torch.cuda.synchronize()
print("plan_step %.3f ms" % ((time.perf_counter() - t0) / 50 * 1e3))
