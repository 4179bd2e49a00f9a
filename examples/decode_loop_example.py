#!/usr/bin/env python3
"""The decode loop the reference only sketches (and whose file does not parse):
reference host/python/vllm_speckv_backend.py:104-129 `decode_step_example`.

Here it runs for real on an MI355X through the same class and methods:
per generated token, one `prefetch_step` per layer (speculative look-ahead of the
next positions), then the attention-side reads `get_kv_ptr` the rows it needs, and
the engine's own token prediction is verified against the token that was produced.

    python examples/decode_loop_example.py [--steps 32] [--layers 4] [--tokens 512]
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run(steps=32, layers=4, tokens=512, heads=8, head_dim=128, scheme=2, verbose=True):
    import cxl_speckv_amd as pkg

    kv = pkg.CxlSpeckvKVAllocator(pkg.library_path(), "/dev/speckv0")     # any non-/dev/null path = HIP engine
    lib = kv.lib
    try:
        lib.set_compression_scheme(scheme)
        handle = kv.allocate(tokens, layers, heads, head_dim, 2)
        # the KV of the prompt (synthetic), compressed into the HBM pool
        rng = np.random.default_rng(0)
        kv_data = rng.standard_normal(tokens * layers * heads * head_dim * 2).astype(np.float16)
        lib.write(handle, 0, kv_data.ctypes.data, kv_data.nbytes, False)
        # a predictor with the reference's shape (weights would come from training)
        emb = ((rng.random((32000, 64)) - 0.5) * 0.1).astype(np.float32)
        wout = ((rng.random((32000, 128)) - 0.5) * 0.1).astype(np.float32)
        lib.predictor_load(emb.ctypes.data, wout.ctypes.data, 32000, False)

        state_tokens = list(range(1, 17))
        cur_pos = tokens // 2
        hits = 0
        for step in range(steps):
            # "forward pass": the attention of every layer reads the K/V rows up to cur_pos;
            # here it touches the row of the current position of every layer
            for layer in range(layers):
                for kind in (0, 1):
                    ptr = kv.get_kv_ptr(0, layer, 0, cur_pos, kind, head_dim * 2)
                    assert ptr
            new_token = int(rng.integers(0, 32000))
            # look-ahead for the next tokens, every layer (reference decode_step_example)
            last = state_tokens[-16:]
            for layer in range(layers):
                kv.prefetch_step(req_id=0, layer=layer, cur_pos=cur_pos, recent_tokens=last, depth_k=4)
            lib.sync()
            hit, depth = lib.verify(0, new_token)                 # against the engine's own prediction
            hits += int(hit)
            state_tokens.append(new_token)
            cur_pos += 1
        st = lib.stats()
        out = {"steps": steps, "l1_hits": st.l1_hits, "l2_hits": st.l2_hits, "l3_accesses": st.l3_accesses,
               "prefetched_pages": st.total_prefetches, "mispredictions": st.mispredictions, "depth": st.prefetch_depth}
        if verbose:
            print(out)
        return out
    finally:
        kv.close()


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=32)
    ap.add_argument("--layers", type=int, default=4)
    ap.add_argument("--tokens", type=int, default=512)
    a = ap.parse_args()
    run(a.steps, a.layers, a.tokens)
