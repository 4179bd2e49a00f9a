#!/usr/bin/env python3
"""Drop-in proof (dev container only): the REFERENCE's own Python callers driving the PRODUCT library.

  1. the reference's ctypes wrapper (host/python/speckv_ctypes.py, imported from /root/reference) and the class body of its
     vLLM shim (host/python/vllm_speckv_backend.py above the '# Example usage' marker -- the file has a SyntaxError behind
     it, SURVEY sect. 2 row 13), pointed at cxl-speckv_amd/lib/libcxlspeckv.so on the fake device "/dev/null": the F-cabi
     trace and every F-offset pointer of SURVEY Appendix A must come out as the reference produces them;
  2. a seeded random walk over the 8 C-ABI functions, product library against the compiled reference
     (oracle/_ref/libspeckv_ref.so) call for call: status codes, handles and returned pointers identical.

Nothing of the reference is stored here or anywhere in the repo: it is imported / loaded where it lies.  This file needs
/root/reference and therefore never runs on the GPU box (listed in .gpurunignore).
    python tests/golden/dropin_check.py [ops_per_seed=4000] [seeds=3]
"""
import ctypes as C
import json
import os
import random
import sys

sys.dont_write_bytecode = True          # never write into /root/reference

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REFROOT = "/root/reference"
REF_SO = os.path.join(ROOT, "oracle", "_ref", "libspeckv_ref.so")
PRODUCT_SO = os.path.join(ROOT, "cxl-speckv_amd", "lib", "libcxlspeckv.so")


def available():
    return (os.path.isdir(os.path.join(REFROOT, "host", "python")) and os.path.exists(REF_SO) and os.path.exists(PRODUCT_SO))


def _reference_python():
    """(speckv_ctypes module, CxlSpeckvKVAllocator class) of the reference, executed from where they lie."""
    pdir = os.path.join(REFROOT, "host", "python")
    if pdir not in sys.path:
        sys.path.insert(0, pdir)
    import speckv_ctypes                                    # the reference's module
    src = open(os.path.join(pdir, "vllm_speckv_backend.py")).read()
    src = src.split("# Example usage")[0].replace("from .speckv_ctypes import", "from speckv_ctypes import")
    ns = {}
    exec(compile(src, "<reference shim>", "exec"), ns)
    return speckv_ctypes, ns["CxlSpeckvKVAllocator"]


def replay_cabi_trace(so_path):
    """The F-cabi scenario of tests/golden/generate_golden.py through the REFERENCE's SpeckvLib class on `so_path`."""
    speckv_ctypes, _ = _reference_python()
    out = []
    raw = C.CDLL(so_path)
    raw.speckv_init.argtypes = [C.c_char_p]; raw.speckv_init.restype = C.c_int
    raw.speckv_free.argtypes = [C.c_uint64]; raw.speckv_free.restype = C.c_int
    out.append(("init", raw.speckv_init(b"/dev/speckv0")))
    out.append(("free", raw.speckv_free(1)))
    lib = speckv_ctypes.SpeckvLib(so_path, "/dev/null")
    out.append(("init_again", lib.lib.speckv_init(b"/dev/null")))
    for size in (1 << 20, 4096, 1):
        out.append(("alloc", size, lib.alloc(size)))
    for h, off in ((1, 0), (1, 1024), (1, 4095), (1, 4096), (1, 8197), (1, 1048575), (1, 1048576), (2, 0), (2, 4095), (2, 4096),
                   (3, 0), (3, 1), (3, 4096), (999, 0)):
        p = C.c_void_p()
        st = lib.lib.speckv_access(h, off, 64, C.byref(p))
        out.append(("access", h, off, st, (p.value or 0) if st == 0 else None))
    tok = (C.c_int32 * 16)(*range(1, 17))
    out.append(("prefetch", lib.lib.speckv_prefetch(1, 0, 100, 4, tok, 16)))
    out.append(("prefetch_empty", lib.lib.speckv_prefetch(1, 0, 100, 4, tok, 0)))
    out.append(("prefetch_null", lib.lib.speckv_prefetch(1, 0, 100, 4, None, 16)))
    out.append(("depth", lib.lib.speckv_set_prefetch_depth(8)))
    out.append(("scheme", lib.lib.speckv_set_compression_scheme(2)))
    for h in (1, 1, 12345):
        out.append(("free", h, lib.lib.speckv_free(h)))
    out.append(("alloc0", lib.alloc(0)))
    p = C.c_void_p()
    out.append(("access_empty_alloc", lib.lib.speckv_access(4, 0, 1, C.byref(p))))
    out.append(("access_freed", lib.lib.speckv_access(1, 0, 1, C.byref(p))))
    out.append(("alloc_null_out", lib.lib.speckv_alloc(10, None, None)))
    out.append(("access_null_out", lib.lib.speckv_access(2, 0, 1, None)))
    lib.lib.speckv_finalize(); lib.lib.speckv_finalize()
    h = C.c_uint64()
    out.append(("free_after_finalize", lib.lib.speckv_free(1)))
    out.append(("alloc_after_finalize", lib.lib.speckv_alloc(10, None, C.byref(h))))
    out.append(("init", lib.lib.speckv_init(b"/dev/null")))
    out.append(("alloc", lib.lib.speckv_alloc(8192, None, C.byref(h)), h.value))
    lib.lib.speckv_finalize()
    return out


def shim_pointers(so_path):
    """SURVEY Appendix A F-offset: the reference's CxlSpeckvKVAllocator (its _calc_offset, its get_kv_ptr) on `so_path`."""
    _, Alloc = _reference_python()
    a = Alloc(so_path, "/dev/null")
    out = []
    try:
        for (T, L, H, D, bpe) in ((128, 1, 8, 128, 2), (4096, 32, 8, 128, 2), (8192, 80, 8, 128, 2), (100, 3, 5, 96, 2)):
            handle = a.allocate(T, L, H, D, bpe)
            eb = D * bpe
            rnd = random.Random(T * 31 + L)
            pts = [(0, 0, 0, 0, 0), (0, L - 1, H - 1, T - 1, 1), (0, 0, H // 2, T // 2, 0), (0, 0, 0, 1, 0), (0, 0, 1, 0, 1), (1, 0, 0, 0, 0)]
            pts += [(0, rnd.randrange(L), rnd.randrange(H), rnd.randrange(T), rnd.randrange(2)) for _ in range(60)]
            for (req, layer, head, pos, kind) in pts:
                off = a._calc_offset(req, layer, head, pos, kind, eb)
                try:
                    ptr, st = a.get_kv_ptr(req, layer, head, pos, kind, eb), 0
                except RuntimeError as e:
                    ptr, st = None, int(str(e).rsplit(":", 1)[1])
                out.append((handle, req, layer, head, pos, kind, off, st, ptr))
            a.prefetch_step(0, 0, min(100, T - 1), list(range(1, 17)), 4)
    finally:
        a._speckv.lib.speckv_finalize()
    return out


class _Raw:
    """The 8 functions of host/include/speckv.h on one library, plain ctypes."""

    def __init__(self, path):
        L = self.lib = C.CDLL(path)
        L.speckv_init.argtypes = [C.c_char_p]; L.speckv_init.restype = C.c_int
        L.speckv_finalize.argtypes = []; L.speckv_finalize.restype = None
        L.speckv_alloc.argtypes = [C.c_size_t, C.c_void_p, C.POINTER(C.c_uint64)]; L.speckv_alloc.restype = C.c_int
        L.speckv_free.argtypes = [C.c_uint64]; L.speckv_free.restype = C.c_int
        L.speckv_access.argtypes = [C.c_uint64, C.c_uint64, C.c_size_t, C.POINTER(C.c_void_p)]; L.speckv_access.restype = C.c_int
        L.speckv_prefetch.argtypes = [C.c_uint32, C.c_uint16, C.c_uint32, C.c_uint32, C.POINTER(C.c_int32), C.c_uint32]
        L.speckv_prefetch.restype = C.c_int
        L.speckv_set_prefetch_depth.argtypes = [C.c_uint32]; L.speckv_set_prefetch_depth.restype = C.c_int
        L.speckv_set_compression_scheme.argtypes = [C.c_int]; L.speckv_set_compression_scheme.restype = C.c_int

    def do(self, op):
        L = self.lib
        k = op[0]
        if k == "init":
            return (L.speckv_init(op[1]),)
        if k == "finalize":
            L.speckv_finalize(); return (None,)
        if k == "alloc":
            h = C.c_uint64(0xDEAD)
            hint = (C.c_uint32 * 2)(op[2], 0) if op[2] is not None else None
            st = L.speckv_alloc(op[1], hint, C.byref(h) if op[3] else None)
            return (st, h.value if (st == 0 and op[3]) else None)
        if k == "free":
            return (L.speckv_free(op[1]),)
        if k == "access":
            p = C.c_void_p(0)
            st = L.speckv_access(op[1], op[2], op[3], C.byref(p) if op[4] else None)
            return (st, (p.value or 0) if (st == 0 and op[4]) else None)
        if k == "prefetch":
            tok = (C.c_int32 * max(len(op[5]), 1))(*op[5]) if op[5] is not None else None
            return (L.speckv_prefetch(op[1], op[2], op[3], op[4], tok, op[6]),)
        if k == "depth":
            return (L.speckv_set_prefetch_depth(op[1]),)
        if k == "scheme":
            return (L.speckv_set_compression_scheme(op[1]),)
        raise ValueError(k)


def fuzz(product_so, ref_so, ops_per_seed=4000, seeds=3):
    """Random walk over the C ABI on "/dev/null": returns (ops run, list of differences)."""
    prod, ref = _Raw(product_so), _Raw(ref_so)
    diffs, total = [], 0
    for seed in range(seeds):
        rnd = random.Random(1000 + seed)
        live, sizes = [], {}                              # handles believed live: only used to aim the walk
        for lib in (prod, ref):                           # every seed starts from a clean, initialised library
            lib.do(("finalize",)); lib.do(("init", b"/dev/null"))
        for _ in range(ops_per_seed):
            r = rnd.random()
            h = rnd.choice(live) if live and rnd.random() < 0.85 else rnd.choice([0, 1, 7, 999, 1 << 40])
            if r < 0.22:
                size = rnd.choice([0, 1, 4095, 4096, 4097, 65536, 1 << 20, (1 << 20) + 1, rnd.randrange(1, 1 << 22)])
                op = ("alloc", size, rnd.choice([None, 0, 1, 5]), rnd.random() > 0.03)
            elif r < 0.34:
                op = ("free", h)
            elif r < 0.74:
                size = sizes.get(h, 8192)
                off = rnd.choice([0, 1, 4095, 4096, max(size - 1, 0), size, size + 4096, rnd.randrange(0, max(size, 1) + 8192)])
                op = ("access", h, off, rnd.choice([0, 1, 256, 4096, 10000]), rnd.random() > 0.03)
            elif r < 0.88:
                hist = rnd.choice([16, 16, 16, 1, 0, 3])
                toks = None if rnd.random() < 0.05 else [rnd.randrange(0, 32000) for _ in range(max(hist, 1))]
                op = ("prefetch", rnd.randrange(0, 4), rnd.randrange(0, 80), rnd.randrange(0, 8192), rnd.choice([0, 1, 4, 8, 16]), toks, hist)
            elif r < 0.92:
                op = ("depth", rnd.choice([0, 1, 4, 8, 16, 1000]))
            elif r < 0.96:
                op = ("scheme", rnd.choice([0, 1, 2]))
            elif r < 0.98:
                op = ("finalize",)
            else:
                op = ("init", rnd.choice([b"/dev/null", b"/dev/null", b"/nonexistent/speckv0"]))
            a, b = prod.do(op), ref.do(op)
            total += 1
            if a != b:
                diffs.append({"seed": seed, "op": repr(op), "product": repr(a), "reference": repr(b)})
            if op[0] == "alloc" and b[0] == 0 and b[1] is not None:
                live.append(b[1]); sizes[b[1]] = op[1]
            elif op[0] == "free" and op[1] in live:
                live.remove(op[1])
            elif op[0] == "finalize" or (op[0] == "init" and b[0] == 0):
                live.clear(); sizes.clear()
        for lib in (prod, ref):
            lib.do(("finalize",))
    return total, diffs


def run(ops_per_seed=4000, seeds=3):
    report = {"available": available()}
    if not report["available"]:
        return report
    for lib in (_Raw(PRODUCT_SO), _Raw(REF_SO)):
        lib.do(("finalize",))
    report["cabi_product"] = replay_cabi_trace(PRODUCT_SO)
    report["cabi_reference"] = replay_cabi_trace(REF_SO)
    report["shim_product"] = shim_pointers(PRODUCT_SO)
    report["shim_reference"] = shim_pointers(REF_SO)
    report["fuzz_ops"], report["fuzz_diffs"] = fuzz(PRODUCT_SO, REF_SO, ops_per_seed, seeds)
    return report


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
    s = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    rep = run(n, s)
    if not rep["available"]:
        print("drop-in check needs /root/reference, oracle/_ref/libspeckv_ref.so and the product library"); sys.exit(2)
    ok = rep["cabi_product"] == rep["cabi_reference"] and rep["shim_product"] == rep["shim_reference"] and not rep["fuzz_diffs"]
    print(json.dumps({"cabi_trace_equal": rep["cabi_product"] == rep["cabi_reference"], "cabi_steps": len(rep["cabi_product"]),
                      "shim_pointers_equal": rep["shim_product"] == rep["shim_reference"], "shim_points": len(rep["shim_product"]),
                      "fuzz_ops": rep["fuzz_ops"], "fuzz_differences": len(rep["fuzz_diffs"]), "first_differences": rep["fuzz_diffs"][:5]}, indent=1))
    sys.exit(0 if ok else 1)
