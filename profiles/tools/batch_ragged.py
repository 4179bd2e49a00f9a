"""Batch attention over sequences of DIFFERENT lengths (uniform in [lo, hi] positions, multiples of 32), by pieces:
python profiles/tools/batch_ragged.py <schemes> <n_seq> <lo> <hi> <tps list, 0 = the engine's rule>"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, bench
import cxl_speckv_amd as pkg
schemes = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "4,3,5").split(",")]
n_seq = int(sys.argv[2]) if len(sys.argv) > 2 else 256
lo, hi = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (1024, 16384)
tpss = [int(v) for v in (sys.argv[5] if len(sys.argv) > 5 else "0").split(",")]
kv = pkg.CxlSpeckvKVAllocator(pkg.library_path(), "hip:0")
lib = kv.lib
PAGE, BLOCK = 4096, 2048
rng = np.random.default_rng(7)
lens = [int(v) * 32 for v in rng.integers(lo // 32, hi // 32 + 1, n_seq)]
if os.environ.get('DIST') == 'tail':                                 # heavy tail: one member in 16 at `hi`, the others uniform in lo .. hi / 8
    lens = [hi if i % 16 == 5 else int(v) * 32 for i, v in enumerate(rng.integers(lo // 32, max(lo // 32 + 1, hi // 256 + 1), n_seq))]
if os.environ.get('SORT') == '1': lens.sort(reverse=True)          # longest first: what an ordering by length inside the engine would give
if os.environ.get('SORT') == '2': lens.sort()
if os.environ.get('SORT', '').startswith('3'):                     # serpentine by rounds of R sequences: position p of round k holds rank p (k even) or the round's mirror (k odd)
    R = int(os.environ['SORT'].split(':')[1])
    d = sorted(lens, reverse=True)
    lens = []
    for k in range(0, len(d), R):
        chunk = d[k:k + R]
        lens += chunk if (k // R) % 2 == 0 else chunk[::-1]
T = hi
for scheme in schemes:
    rec = {4: 2048, 3: 1152, 5: 1088}[scheme]
    lib.set_compression_scheme(scheme)
    g = torch.Generator(device="cuda"); g.manual_seed(2004)
    n_pages = T * 8 * 128 * 2 * 2 // PAGE
    x = torch.randn((n_pages, BLOCK), generator=g, device="cuda", dtype=torch.float32).to(torch.float16)
    handles = []
    for _ in range(n_seq):
        h = lib.alloc(n_pages * PAGE); lib.set_layout(h, T, 1, 8, 128, 2); lib.write(h, 0, x.data_ptr(), x.numel() * 2, True); handles.append(h)
    q = torch.randn((n_seq, 8, 8, 128), generator=g, device="cuda", dtype=torch.float32).to(torch.float16)
    o = torch.empty((n_seq, 8, 8, 128), dtype=torch.float32, device="cuda")
    lse = torch.empty((n_seq, 8, 8), dtype=torch.float32, device="cuda")
    s = torch.cuda.Stream()
    fn = {4: lib.attend_fp8_batch, 3: lib.attend_int4_batch, 5: lib.attend_mx4_batch}[scheme]
    plan_bytes = lib.attend_plan_bytes(n_seq)
    d_plan = torch.empty(plan_bytes, dtype=torch.uint8, device="cuda")
    for tps in tpss:
        bench.set_tuning("attend_tiles_per_split", tps)
        lib.attend_batch_plan(handles, lens, T, d_plan.data_ptr(), plan_bytes, s.cuda_stream)
        res = []
        for step in (lambda: fn(handles, 0, q.data_ptr(), 8, lens, 0.0884, o.data_ptr(), None, s.cuda_stream),
                     lambda: lib.attend_planned(scheme, d_plan.data_ptr(), n_seq, 0, q.data_ptr(), 8, T, 0.0884, o.data_ptr(), lse.data_ptr(), s.cuda_stream)):
            step(); torch.cuda.synchronize()
            bench.ramp(step, torch.cuda.synchronize, 30)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(s)
            for _ in range(10): step()
            b.record(s); torch.cuda.synchronize()
            ms = a.elapsed_time(b) / 10
            res.append((round(ms, 4), round(sum(lens) * rec / (ms * 1e-3) / 8e12, 4)))
        bench.set_tuning("attend_tiles_per_split", 0)
        print(scheme, n_seq, f"{lo}-{hi}", "tps", tps, "batch", res[0], "planned", res[1], flush=True)
    for h in handles: lib.free(h)
kv.close()
