#!/usr/bin/env python3
"""A batch decode loop whose attention runs as ONE HIP graph replay per step.

The KV of every sequence lives compressed (FP8 or INT4 records) in the HBM pool, one allocation per sequence.  Per
generated token and sequence the loop

  1. plans the step (speckv_ext_attend_batch_plan: handle look-ups and one descriptor per sequence, outside the graph),
  2. replays a graph captured once: for every layer the planned batch attention (speckv_ext_attend_*_planned, kernel
     launches only) and -- on steps where the newest position still waits for its partner, pages hold position PAIRS --
     the tail fold (speckv_ext_attend_fold_tail),
  3. appends the step's K / V rows (the odd position goes to a persistent fp16 tail, a completed pair is written as
     2 x layers pages per sequence in one speckv_ext_write_strided_batch launch).

Two graphs exist (with and without the tail fold); both stay valid while the sequences grow, because the launches are
sized by a length bound and read the actual lengths from the plan on the device.  The result of every step is checked
against the connector's eager path on a copy of the same state.

    python examples/batch_decode_graph_example.py [--seqs 16] [--layers 4] [--prompt 96] [--steps 12] [--scheme fp8|int4|mxfp4]
"""
import argparse
import ctypes
import gc
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run(seqs=16, layers=4, prompt=96, steps=12, scheme="fp8", max_tokens=512, verbose=True):
    import torch
    import cxl_speckv_amd as pkg
    from cxl_speckv_amd.kv_connector import SpeckvKVConnector

    H, D, G = 8, 128, 4
    code = {"fp8": 4, "int4": 3, "mxfp4": 5}[scheme]
    lib = pkg.SpeckvLib(pkg.library_path(), "hip:0")
    conn = None
    try:
        # the connector holds the reference state (eager path); the graph path below shares its allocations
        conn = SpeckvKVConnector(lib, num_layers=layers, max_tokens=max_tokens, scheme=scheme)
        gen = torch.Generator(device="cuda"); gen.manual_seed(11)
        ids = list(range(seqs))
        for r in ids:
            conn.add_request(r)
            kp = torch.randn((layers, prompt, H, D), generator=gen, device="cuda").to(torch.float16)
            vp = torch.randn((layers, prompt, H, D), generator=gen, device="cuda").to(torch.float16)
            conn.write_prefill(r, kp, vp)
        handles = (ctypes.c_uint64 * seqs)(*[conn.requests[r].handle for r in ids])
        s = torch.cuda.Stream()
        plan = torch.zeros(lib.attend_plan_bytes(seqs), dtype=torch.uint8, device="cuda")
        q = torch.zeros((layers, seqs, H, G, D), dtype=torch.float16, device="cuda")          # static graph inputs
        tail_k = torch.zeros((seqs, layers, H, D), dtype=torch.float16, device="cuda")
        tail_v = torch.zeros_like(tail_k)
        out = torch.zeros((layers, seqs, H, G, D), dtype=torch.float32, device="cuda")        # static graph outputs
        lse = torch.zeros((layers, seqs, H, G), dtype=torch.float32, device="cuda")
        sm = 1.0 / D ** 0.5
        bound = max_tokens                                                                     # the graphs are sized for this

        def launches(fold):
            for layer in range(layers):
                lib.attend_planned(code, plan.data_ptr(), seqs, layer, q[layer].data_ptr(), G, bound, sm,
                                   out[layer].data_ptr(), lse[layer].data_ptr(), s.cuda_stream)
                if fold:
                    lib.attend_fold_tail(seqs, 0, H, G, q[layer].data_ptr(), tail_k.data_ptr() + layer * H * D * 2,
                                         tail_v.data_ptr() + layer * H * D * 2, layers * H * D, sm,
                                         out[layer].data_ptr(), lse[layer].data_ptr(), s.cuda_stream)

        def plan_now():
            lens = (ctypes.c_uint32 * seqs)(*[conn.length(r) & ~1 for r in ids])
            lib.attend_batch_plan(handles, lens, bound, plan.data_ptr(), plan.numel(), s.cuda_stream)

        plan_now()
        launches(True); torch.cuda.synchronize()                  # warm-up: sizes the engine's scratch outside any capture
        graphs = {}
        gc.collect(); gc.disable()                                # no collection (= no stray HIP calls) inside a capture
        try:
            for fold in (False, True):
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=s):
                    launches(fold)
                graphs[fold] = g
        finally:
            gc.enable()

        worst, t_graph, t_eager = 0.0, 0.0, 0.0
        with torch.cuda.stream(s):                                # the connector's torch glue runs on the current stream
            for step in range(steps):
                qs = torch.randn((layers, seqs, H, G, D), generator=gen, device="cuda").to(torch.float16)
                k_new = torch.randn((seqs, layers, H, D), generator=gen, device="cuda").to(torch.float16)
                v_new = torch.randn((seqs, layers, H, D), generator=gen, device="cuda").to(torch.float16)
                odd = conn.length(ids[0]) & 1                     # lockstep batch: everybody has a tail, or nobody
                # ---- graph path: copy the inputs into the static buffers, plan, replay
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                q.copy_(qs)
                plan_now()
                graphs[bool(odd)].replay()
                s.synchronize()
                t_graph += time.perf_counter() - t0
                # ---- eager reference: the connector's own per-layer calls
                t0 = time.perf_counter()
                ref = [conn.attend(layer, ids, qs[layer], sm, stream=s) for layer in range(layers)]
                s.synchronize()
                t_eager += time.perf_counter() - t0
                for layer in range(layers):
                    scale = float(ref[layer].abs().max()) + 1e-6
                    worst = max(worst, float((out[layer] - ref[layer]).abs().max()) / scale)
                # ---- append (both paths share the allocations; the graph path keeps its own copy of the tail)
                if not odd:
                    tail_k.copy_(k_new); tail_v.copy_(v_new)
                keep = conn.append(ids, k_new, v_new, stream=s)
                s.synchronize()
                del keep
        res = {"sequences": seqs, "layers": layers, "steps": steps, "final_length": conn.length(ids[0]),
               "max_rel_diff_graph_vs_eager": worst, "ms_per_step_graph": t_graph / steps * 1e3,
               "ms_per_step_eager": t_eager / steps * 1e3}
        if verbose:
            print(res)
        assert worst <= 1e-3, worst                               # two split arrangements of the same kernels (f16 weights)
        return res
    finally:
        if conn is not None:
            for r in list(conn.requests):
                conn.free_request(r)
        lib.finalize()


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--seqs", type=int, default=16)
    ap.add_argument("--layers", type=int, default=4)
    ap.add_argument("--prompt", type=int, default=96)
    ap.add_argument("--steps", type=int, default=12)
    ap.add_argument("--scheme", default="fp8", choices=["fp8", "int4", "mxfp4"])
    a = ap.parse_args()
    run(a.seqs, a.layers, a.prompt, a.steps, a.scheme, max_tokens=(a.prompt + a.steps + 511) // 512 * 512)
