#!/usr/bin/env python3
"""Generate tests/golden/* from the REFERENCE (dev container only).

Needs /root/reference and oracle/_ref/libspeckv_ref.so (make -C oracle ref).
The outputs are DATA: inputs and the reference's outputs.  No reference source
text is stored.  Re-run:  python tests/golden/generate_golden.py

Fixtures (SURVEY.md Appendix A names):
  cabi_trace.json     F-cabi    reference C ABI on the fake device "/dev/null"
  shim_offsets.json   F-offset  reference python shim: _calc_offset / get_kv_ptr
  codec_vectors.npz   F-codec   FPGACacheEngine compress/decompress (REF_EXACT)
  mm_trace.json       F-mm      CXLMemoryManager addresses / tiers / stats
  prefetch.json       F-prefetch SpeculativePrefetcher address lists, depth trace
"""
import ctypes as C
import json
import os
import sys

import numpy as np

sys.dont_write_bytecode = True   # never write into /root/reference

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REFROOT = "/root/reference"

from oracle.bindings import REF_SO, Reference, _ptr, u32p, u64p, i32p, f32p  # noqa: E402


def gen_cabi(ref):
    """Drive the reference's C ABI through the reference's OWN ctypes wrapper
    (host/python/speckv_ctypes.py, imported here only)."""
    sys.path.insert(0, os.path.join(REFROOT, "host", "python"))
    import speckv_ctypes  # the reference's module; never shipped
    trace = []

    def rec(op, args, status, value=None):
        trace.append({"op": op, "args": args, "status": status, "value": value})

    L = ref.lib
    rec("init", ["/dev/speckv0"], L.speckv_init(b"/dev/speckv0"))
    rec("free", [1], L.speckv_free(1))
    lib = speckv_ctypes.SpeckvLib(REF_SO, "/dev/null")
    rec("init", ["/dev/null"], 0)
    rec("init", ["/dev/null"], lib.lib.speckv_init(b"/dev/null"))
    for size in (1 << 20, 4096, 1):
        rec("alloc", [size], 0, lib.alloc(size))
    for h, off in ((1, 0), (1, 1024), (1, 4095), (1, 4096), (1, 8197), (1, 1048575), (1, 1048576),
                   (2, 0), (2, 4095), (2, 4096), (3, 0), (3, 1), (3, 4096), (999, 0)):
        p = C.c_void_p()
        st = lib.lib.speckv_access(h, off, 64, C.byref(p))
        rec("access", [h, off, 64], st, (p.value or 0) if st == 0 else None)
    tok = (C.c_int32 * 16)(*range(1, 17))
    rec("prefetch", [1, 0, 100, 4, list(range(1, 17))], lib.lib.speckv_prefetch(1, 0, 100, 4, tok, 16))
    rec("prefetch", [1, 0, 100, 4, []], lib.lib.speckv_prefetch(1, 0, 100, 4, tok, 0))
    rec("prefetch", [1, 0, 100, 4, None], lib.lib.speckv_prefetch(1, 0, 100, 4, None, 16))
    rec("set_prefetch_depth", [8], lib.lib.speckv_set_prefetch_depth(8))
    rec("set_compression_scheme", [2], lib.lib.speckv_set_compression_scheme(2))
    for h in (1, 1, 12345):
        rec("free", [h], lib.lib.speckv_free(h))
    rec("alloc", [0], 0, lib.alloc(0))
    p = C.c_void_p()
    rec("access", [4, 0, 1], lib.lib.speckv_access(4, 0, 1, C.byref(p)))
    rec("access", [1, 0, 1], lib.lib.speckv_access(1, 0, 1, C.byref(p)))
    h = C.c_uint64()
    rec("alloc_null_out", [10], lib.lib.speckv_alloc(10, None, None))
    rec("access_null_out", [2, 0, 1], lib.lib.speckv_access(2, 0, 1, None))
    lib.lib.speckv_finalize()
    rec("finalize", [], None)
    lib.lib.speckv_finalize()
    rec("finalize", [], None)
    rec("free", [1], lib.lib.speckv_free(1))
    rec("alloc", [10], lib.lib.speckv_alloc(10, None, C.byref(h)))
    rec("set_prefetch_depth", [4], lib.lib.speckv_set_prefetch_depth(4))
    rec("init", ["/dev/null"], lib.lib.speckv_init(b"/dev/null"))
    st = lib.lib.speckv_alloc(8192, None, C.byref(h))
    rec("alloc", [8192], st, h.value)
    lib.lib.speckv_finalize()
    rec("finalize", [], None)
    json.dump({"source": "reference host/src/speckv_c_api.cpp via host/python/speckv_ctypes.py on /dev/null",
               "trace": trace}, open(os.path.join(HERE, "cabi_trace.json"), "w"), indent=1)


def gen_shim(ref):
    """Reference shim class (vllm_speckv_backend.py:1-100; the file itself has a
    SyntaxError at line 104, so only the class body above the '# Example usage'
    marker is executed, in memory, with the relative import rewritten)."""
    sys.path.insert(0, os.path.join(REFROOT, "host", "python"))
    src = open(os.path.join(REFROOT, "host", "python", "vllm_speckv_backend.py")).read()
    src = src.split("# Example usage")[0].replace("from .speckv_ctypes import", "from speckv_ctypes import")
    ns = {}
    exec(compile(src, "<reference shim>", "exec"), ns)
    Alloc = ns["CxlSpeckvKVAllocator"]
    a = Alloc(REF_SO, "/dev/null")
    out = {"source": "reference host/python/vllm_speckv_backend.py:8-100 on /dev/null", "configs": []}
    rng = np.random.default_rng(42)
    for name, (T, L, H, D, bpe) in (("cfg1", (128, 1, 8, 128, 2)), ("8B@4k", (4096, 32, 8, 128, 2)),
                                    ("70B@8k", (8192, 80, 8, 128, 2)), ("odd", (100, 3, 5, 96, 2))):
        handle = a.allocate(T, L, H, D, bpe)
        eb = D * bpe
        pts = [(0, 0, 0, 0, 0), (0, L - 1, H - 1, T - 1, 1), (0, 0, H // 2, T // 2, 0), (0, 0, 0, 1, 0),
               (0, 0, 1, 0, 1), (1, 0, 0, 0, 0)]
        for _ in range(40):
            pts.append((0, int(rng.integers(0, L)), int(rng.integers(0, H)), int(rng.integers(0, T)),
                        int(rng.integers(0, 2))))
        rows = []
        for (req, layer, head, pos, kind) in pts:
            off = a._calc_offset(req, layer, head, pos, kind, eb)
            try:
                ptr = a.get_kv_ptr(req, layer, head, pos, kind, eb)
                st = 0
            except RuntimeError as e:
                ptr, st = None, int(str(e).rsplit(":", 1)[1])
            rows.append({"req": req, "layer": layer, "head": head, "pos": pos, "kind": kind,
                         "offset": off, "status": st, "ptr": ptr})
        out["configs"].append({"name": name, "T": T, "L": L, "H": H, "D": D, "bpe": bpe,
                               "handle": handle, "total_bytes": T * L * H * D * bpe * 2, "entries": rows})
    tokens = list(range(1, 17))
    a.prefetch_step(0, 0, 100, tokens, 4)
    out["prefetch_step_ok"] = True
    a._speckv.lib.speckv_finalize()
    json.dump(out, open(os.path.join(HERE, "shim_offsets.json"), "w"), indent=1)


def gen_codec(ref):
    rng = np.random.default_rng(1234)
    blocks = {
        "gauss_a": rng.standard_normal(2048).astype(np.float16).astype(np.float32),
        "gauss_b": (rng.standard_normal(2048) * 3.7).astype(np.float16).astype(np.float32),
        "zeros": np.zeros(2048, np.float32),
        "piecewise32": np.repeat(rng.standard_normal(64).astype(np.float16).astype(np.float32), 32),
        "const_run": np.full(2048, np.float32(np.float16(0.37)), np.float32),
        "ramp": np.linspace(-3, 3, 2048).astype(np.float16).astype(np.float32),
        "sparse": np.where(rng.random(2048) < 0.02, rng.standard_normal(2048), 0).astype(np.float16).astype(np.float32),
        "small": (rng.standard_normal(2048) * 1e-3).astype(np.float16).astype(np.float32),
        "f16_extremes": np.array([65504, -65504, 6.1e-5, 5.96e-8, -5.96e-8, 0, -0.0, 1, -1] * 227 + [0.5] * 5,
                                 np.float16).astype(np.float32),
        "kat": np.array([0, 1, -1, 0.5, 0.5, 0.5, 0.25, 127, 0.007874, 0.003937, -0.0039], np.float32),
        "short_257": rng.standard_normal(257).astype(np.float32),
    }
    out = {}
    for name, x in blocks.items():
        s, rle = ref.compress_f32(x)
        y = ref.decompress_f32(rle, s)
        out[f"{name}.x"] = x
        out[f"{name}.scale"] = np.array([s], np.float32)
        out[f"{name}.rle"] = rle
        out[f"{name}.y"] = y
    # the survey's large-vector summary numbers (Appendix A), recomputed here
    g = np.random.default_rng(2001).standard_normal(131072).astype(np.float32)
    s, rle = ref.compress_f32(g)
    y = ref.decompress_f32(rle, s)
    out["big.seed"] = np.array([2001]); out["big.n"] = np.array([131072])
    out["big.scale"] = np.array([s], np.float32)
    out["big.compressed_size"] = np.array([rle.size])
    out["big.rle_crc"] = np.array([int(np.bitwise_xor.reduce(rle.astype(np.uint64) * (np.arange(rle.size, dtype=np.uint64) % 251 + 1)))], np.uint64)
    out["big.y_sum_bits"] = np.array([int(y.view(np.uint32).astype(np.uint64).sum())], np.uint64)
    # malformed streams for the decoder
    for i, stream in enumerate(([5, 3, 7], [5, 0, 9, 2], [255, 200, 1, 255, 128, 1], [1], [])):
        rle = np.array(stream, np.uint8)
        cap = int(rle[1::2].astype(np.int64).sum()) if rle.size >= 2 else 0
        out[f"malformed{i}.rle"] = rle
        out[f"malformed{i}.y"] = ref.decompress_f32(rle, 0.5, cap=cap)
    np.savez_compressed(os.path.join(HERE, "codec_vectors.npz"), **out)


def gen_mm(ref):
    R = ref.lib
    mm = R.ref_mm_new(12, 3, 128)
    ev = []
    a0 = R.ref_mm_allocate(mm, 524288, 0, 2); ev.append(["allocate", [524288, 0, 2], a0])
    a1 = R.ref_mm_allocate(mm, 4096, 3, 0); ev.append(["allocate", [4096, 3, 0], a1])
    a2 = R.ref_mm_allocate(mm, 5000, 4, 1); ev.append(["allocate", [5000, 4, 1], a2])
    for va in (a0, a0 + 4096 * 3 + 17, a1, a1 + 5, a2 + 4096, a2 + 8191, a2 + 8192, 0x42, a0 - 1):
        ev.append(["translate", [va], R.ref_mm_translate(mm, va)])
    for va, t in ((a0, 2), (a0, 0), (a1, 0), (a2, 1), (a2 + 4096, 1), (0x42, 2)):
        ev.append(["is_in_cache", [va, t], R.ref_mm_is_in_cache(mm, va, t)])
    for i in range(11):
        R.ref_mm_update_access_tracking(mm, a0 + 4096)
        ev.append(["update_access_tracking", [a0 + 4096], None])
        ev.append(["is_hot_page", [a0 + 4096], R.ref_mm_is_hot_page(mm, a0 + 4096)])
    ev.append(["promote_to_l1", [a0 + 4096], R.ref_mm_promote_to_l1(mm, a0 + 4096)])
    ev.append(["promote_to_l1", [a0 + 4096], R.ref_mm_promote_to_l1(mm, a0 + 4096)])
    ev.append(["is_in_cache", [a0 + 4096, 0], R.ref_mm_is_in_cache(mm, a0 + 4096, 0)])
    ev.append(["promote_to_l1", [a2], R.ref_mm_promote_to_l1(mm, a2)])
    ev.append(["demote_to_l3", [a0 + 4096], R.ref_mm_demote_to_l3(mm, a0 + 4096)])
    ev.append(["demote_to_l3", [a0 + 4096], R.ref_mm_demote_to_l3(mm, a0 + 4096)])
    ev.append(["get_page_state", [a0], R.ref_mm_get_page_state(mm, a0)])
    R.ref_mm_mark_modified(mm, a0); ev.append(["mark_modified", [a0], None])
    ev.append(["get_page_state", [a0], R.ref_mm_get_page_state(mm, a0)])
    R.ref_mm_invalidate_page(mm, a0); ev.append(["invalidate_page", [a0], None])
    ev.append(["get_page_state", [a0], R.ref_mm_get_page_state(mm, a0)])
    ev.append(["get_page_state", [0x42], R.ref_mm_get_page_state(mm, 0x42)])
    R.ref_mm_deallocate(mm, a0); ev.append(["deallocate", [a0], None])
    ev.append(["translate", [a0], R.ref_mm_translate(mm, a0)])
    ev.append(["translate", [a0 + 4096], R.ref_mm_translate(mm, a0 + 4096)])
    u = (C.c_uint64 * 7)(); d = (C.c_double * 2)()
    R.ref_mm_get_statistics(mm, u, d)
    json.dump({"source": "reference src/cxl_memory/cxl_memory_manager.cpp, CXLMemoryManager(12,3,128)",
               "events": ev, "stats_u": list(u), "stats_d": list(d)},
              open(os.path.join(HERE, "mm_trace.json"), "w"), indent=1)
    R.ref_mm_delete(mm)


def gen_prefetch(ref):
    R = ref.lib
    mm = R.ref_mm_new(12, 3, 128)
    pf = R.ref_pf_new(mm, 4, 16)
    calls = []
    for hist, layer, depth in (([*range(1, 17)], 5, 0), ([*range(101, 117)], 0, 8), ([7, 8, 9], 79, 2),
                               ([1], 65535, 3), ([*range(1, 17)], 31, 1)):
        h = np.array(hist, np.uint32)
        va = np.zeros(16, np.uint64); tok = np.zeros(16, np.uint32); conf = np.zeros(16, np.float32)
        n = R.ref_pf_prefetch(pf, _ptr(h, u32p), h.size, layer, depth, _ptr(va, u64p), _ptr(tok, u32p),
                              _ptr(conf, f32p), 16)
        # predicted tokens / confidences of the FIRST predictor constructed in this process
        # (weights from glibc rand() in its default state, lstm_predictor.cpp:27-35)
        calls.append({"history": hist, "layer": layer, "depth": depth, "addresses": [int(v) for v in va[:n]],
                      "tokens": [int(t) for t in tok[:n]], "conf_bits": [int(b) for b in conf[:n].view(np.uint32)]})
    rng = np.random.default_rng(3)
    outcomes = [1] * 9 + [1] * 6 + [0] * 12 + [int(v) for v in (rng.random(200) < 0.9)]
    depths = []
    for i, ok in enumerate(outcomes):
        R.ref_pf_update_accuracy(pf, i, ok)
        depths.append(int(R.ref_pf_adaptive_depth(pf)))
    pred = np.array([1, 2, 3], np.uint32)
    m0 = int(R.ref_pf_handle_misprediction(pf, 2, _ptr(pred, u32p), 3))
    m1 = int(R.ref_pf_handle_misprediction(pf, 5, _ptr(pred, u32p), 3))
    json.dump({"source": "reference src/prefetcher/speculative_prefetcher.cpp, SpeculativePrefetcher(&mm,4,16)",
               "calls": calls, "initial_depth": 4, "outcomes": outcomes, "depth_trace": depths,
               "mispredictions_after_hit": m0, "mispredictions_after_miss": m1},
              open(os.path.join(HERE, "prefetch.json"), "w"), indent=1)
    R.ref_pf_delete(pf); R.ref_mm_delete(mm)


def gen_coherence():
    """Traces of the reference's CoherenceManager (oracle/_ref/libspeckv_ref_coh.so): the scenarios of the
    reference's own tests/test_coherence.cpp:57-398, plus one seeded random walk over every operation, with the
    result of every call, and the per-line (state, tier) and the seven counters at the end."""
    import numpy as np
    from oracle.bindings import ReferenceCoherence
    def run(line, has_driver, ops):
        rc = ReferenceCoherence(line, has_driver)
        results = rc.run([tuple(o) for o in ops])
        addrs = sorted({a for o in ops if len(o) > 1 for a in (o[1] if isinstance(o[1], list) else [o[1]])})
        final = {hex(a): [rc.op("get_state", a), rc.op("get_tier", a)] for a in addrs}
        st = rc.stats()
        rc.close()
        return {"cache_line_size": line, "has_driver": has_driver, "ops": ops, "results": results, "final": final, "stats": st}
    scen = {
        "read_operations": [["read", 0x10000], ["get_state", 0x10000], ["get_tier", 0x10000], ["read", 0x10000]],
        "write_operations": [["read", 0x20000], ["get_state", 0x20000], ["write", 0x20000], ["get_state", 0x20000]],
        "invalidation": [["read", 0x30000], ["invalidate", 0x30000], ["get_state", 0x30000]],
        "writeback": [["write", 0x40000], ["writeback", 0x40000], ["get_state", 0x40000]],
        "tier_promotion": [["get_tier", 0x50000], ["promote_to_l1", 0x50000], ["get_tier", 0x50000], ["get_state", 0x50000]],
        "tier_demotion": [["promote_to_l1", 0x60000], ["demote_to_l3", 0x60000], ["get_tier", 0x60000]],
        "batch_operations": [["read", 0x70000], ["read", 0x70040], ["read", 0x70080], ["read", 0x700C0],
                             ["batch_invalidate", [0x70000, 0x70040, 0x70080, 0x700C0]]],
        "flush_all": [["write", 0x80000], ["write", 0x80040], ["write", 0x80080], ["flush_all"]],
        "statistics": [["read", 0x90000], ["write", 0x90040], ["invalidate", 0x90000]],
        "state_transitions": [["get_state", 0xA0000], ["read", 0xA0000], ["write", 0xA0000], ["writeback", 0xA0000], ["invalidate", 0xA0000]],
        "multiple_addresses": [["read", 0xB0000 + i * 0x1000] for i in range(10)],
        "unaligned_and_modified_paths": [["write", 0x1234], ["get_state", 0x1200], ["get_state", 0x1240], ["demote_to_l3", 0x1239],
                                         ["write", 0x5000], ["invalidate", 0x5010], ["update_tier", 0x6000, 1], ["get_tier", 0x6000],
                                         ["get_state", 0x6000], ["read", 0x6000], ["batch_invalidate", [0x6000, 0x7000, 0x1234]],
                                         ["writeback", 0x9999]],
    }
    out = {"source": "reference src/cxl_memory/coherence_manager.cpp (CoherenceManager, opaque non-null driver)",
           "scenarios": {k: run(64, 1, v) for k, v in scen.items()}}
    rng = np.random.default_rng(77)
    names = ["read", "write", "invalidate", "writeback", "promote_to_l1", "demote_to_l3", "get_state", "get_tier"]
    walk = []
    for _ in range(400):
        r = rng.random()
        a = int(rng.integers(0, 24)) * 0x40 + int(rng.integers(0, 64))
        if r < 0.03: walk.append(["flush_all"])
        elif r < 0.06: walk.append(["batch_invalidate", [int(rng.integers(0, 24)) * 0x40 for _ in range(int(rng.integers(0, 6)))]])
        elif r < 0.09: walk.append(["update_tier", a, int(rng.integers(0, 3))])
        elif r < 0.10: walk.append(["reset_statistics"])
        else: walk.append([names[int(rng.integers(0, len(names)))], a])
    out["random_walk"] = run(64, 1, walk)
    out["no_driver"] = run(64, 0, [["read", 0x100], ["write", 0x100], ["promote_to_l1", 0x200], ["get_tier", 0x200],
                                   ["invalidate", 0x100], ["update_tier", 0x300, 0], ["demote_to_l3", 0x300], ["flush_all"]])
    out["line_128"] = run(128, 1, [["write", 0x1040], ["get_state", 0x1000], ["get_state", 0x1080], ["read", 0x10FF]])
    json.dump(out, open(os.path.join(HERE, "coherence_trace.json"), "w"), indent=1)


if __name__ == "__main__":
    ref = Reference()
    gen_codec(ref)
    gen_mm(ref)
    gen_prefetch(ref)
    gen_shim(ref)
    gen_coherence()
    gen_cabi(ref)   # last: leaves the reference C ABI finalized
    print("golden fixtures written to", HERE)
