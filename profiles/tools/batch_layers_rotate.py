"""Does the batch attention's short-context figure owe anything to the Infinity Cache?  The bench's `*_attention_batch_decode_step_256xT`
repeats ONE layer (its records: 256 x T x rec bytes); a real decode step walks L layers, each with its own records.  Here the planned
form over allocations of L layers: the same layer again and again against the layers in turn.
    python profiles/tools/batch_layers_rotate.py [scheme=5] [T=2048] [L=8] [n_seq=256]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
import cxl_speckv_amd as pkg
scheme = int(sys.argv[1]) if len(sys.argv) > 1 else 5
T = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
L = int(sys.argv[3]) if len(sys.argv) > 3 else 8
n_seq = int(sys.argv[4]) if len(sys.argv) > 4 else 256
TA = int(sys.argv[5]) if len(sys.argv) > 5 else T          # capacity of the allocation in positions (>= T): changes the stride between sequences
kv = pkg.CxlSpeckvKVAllocator(pkg.library_path(), "hip:0")
lib = kv.lib
rec = {4: 2048, 3: 1152, 5: 1088}[scheme]
lib.set_compression_scheme(scheme)
g = torch.Generator(device="cuda"); g.manual_seed(2004)
pages_layer = T * 8 * 128 * 2 * 2 // bench.PAGE
pages_alloc = TA * 8 * 128 * 2 * 2 // bench.PAGE
x = torch.randn((pages_alloc * L, bench.BLOCK_ELEMS), generator=g, device="cuda", dtype=torch.float32).to(torch.float16)
GAP = int(os.environ.get('GAP', '0'))                     # throw-away allocations of (i % 13 + 1) * GAP pages between the sequences: perturbs the stride
handles, gaps = [], []
for i in range(n_seq):
    if GAP: gaps.append(lib.alloc((i % 13 + 1) * GAP * bench.PAGE))
    h = lib.alloc(pages_alloc * L * bench.PAGE)
    lib.set_layout(h, TA, L, 8, 128, 2)
    lib.write(h, 0, x.data_ptr(), x.numel() * 2, True)
    handles.append(h)
q = torch.randn((n_seq, 8, 8, 128), generator=g, device="cuda", dtype=torch.float32).to(torch.float16)
o = torch.empty((n_seq, 8, 8, 128), dtype=torch.float32, device="cuda")
lse = torch.empty((n_seq, 8, 8), dtype=torch.float32, device="cuda")
s = torch.cuda.Stream()
plan_bytes = lib.attend_plan_bytes(n_seq)
d_plan = torch.empty(plan_bytes, dtype=torch.uint8, device="cuda")
lib.attend_batch_plan(handles, [T] * n_seq, T, d_plan.data_ptr(), plan_bytes, s.cuda_stream)
def run(layers):
    for l in layers:
        lib.attend_planned(scheme, d_plan.data_ptr(), n_seq, l, q.data_ptr(), 8, T, 0.08838834764831845, o.data_ptr(), lse.data_ptr(), s.cuda_stream)
rec_bytes = n_seq * pages_layer * rec
for name, layers in (("one layer repeated", [0] * L), ("layers in turn", list(range(L))), ("one layer repeated", [3] * L), ("layers in turn", list(range(L)))):
    run(layers); torch.cuda.synchronize()
    bench.ramp(lambda: run(layers), torch.cuda.synchronize, bench.EXTRAS_RAMP_MS)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(s)
    for _ in range(5): run(layers)
    b.record(s); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / (5 * L)
    print(f"scheme {scheme} {n_seq} x {T} (capacity {TA}) x {L} layers, gap {GAP}, {name}: {ms * 1e3:.1f} us per layer, {rec_bytes / (ms * 1e-3) / 1e9 / 8000:.4f} of 8 TB/s ({rec_bytes / 1e6:.0f} MB per layer)", flush=True)
kv.close()
