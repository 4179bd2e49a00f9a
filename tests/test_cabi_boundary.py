"""-m "not gpu": the drop-in boundary on CPU.

* libcxlspeckv.so loads (RTLD_NOW: no undefined symbols) and exports every
  function include/speckv.h and include/speckv_ext.h declare;
* on the reference's fake device "/dev/null" (page-table emulation, no data
  path) every status code and every logical address equals the reference's,
  replaying the traces captured from the reference (tests/golden/);
* ports of the reference's own tests (tests/test_allocator.cpp, test_c_api.c,
  test_python.py scenarios);
* a data-path call without a HIP device fails loudly instead of falling back.
No compute is executed here."""
import ctypes as C
import json
import os
import re

import pytest

import cxl_speckv_amd as pkg
from cxl_speckv_amd.speckv_ctypes import SpeckvError

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def libpath():
    return pkg.build_library()


@pytest.fixture()
def nulllib(libpath):
    lib = pkg.SpeckvLib(libpath, "/dev/null")
    yield lib
    lib.finalize()


def declared_functions(header):
    src = open(os.path.join(ROOT, "include", header)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(speckv_[a-z0-9_]+)\s*\(", src)))


def test_the_tree_builds():
    """`make` over cxl-speckv_amd/csrc succeeds and leaves the library newer than every source: a compile error in one
    translation unit must not hide behind a library built from an older tree (it did once: the object rule sends the
    compiler's remarks to a file, and a stale libcxlspeckv.so kept every test green)."""
    import subprocess
    src = os.path.join(ROOT, "cxl-speckv_amd", "csrc")
    r = subprocess.run(["make", "-s", "-j8", "-C", src], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout[-3000:]
    lib = os.path.join(ROOT, "cxl-speckv_amd", "lib", "libcxlspeckv.so")
    newest = max(os.path.getmtime(os.path.join(src, f)) for f in os.listdir(src) if f.endswith((".hip", ".cpp", ".hpp", ".map")))
    assert os.path.getmtime(lib) >= newest


def test_library_exports_every_declared_symbol(libpath):
    lib = C.CDLL(libpath, mode=os.RTLD_NOW)
    names = declared_functions("speckv.h") + declared_functions("speckv_ext.h")
    assert len(declared_functions("speckv.h")) == 8
    assert len(names) > 30
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/ but not exported"
    assert lib.speckv_ext_backend is not None
    lib.speckv_ext_backend.restype = C.c_char_p
    assert lib.speckv_ext_backend() == b"hip"


def test_abi_struct_sizes():
    from cxl_speckv_amd.speckv_ctypes import DmaDesc, PageInfo
    assert C.sizeof(DmaDesc) == 24            # driver/uapi/speckv_ioctl.h:10-15
    assert C.sizeof(PageInfo) == 64


def test_cabi_trace_replay(libpath, golden_dir):
    """F-cabi: the reference's C ABI on /dev/null, call by call."""
    trace = json.load(open(os.path.join(golden_dir, "cabi_trace.json")))["trace"]
    lib = pkg.load_library(libpath)
    lib.speckv_alloc.argtypes = [C.c_size_t, C.c_void_p, C.POINTER(C.c_uint64)]
    lib.speckv_free.argtypes = [C.c_uint64]
    lib.speckv_access.argtypes = [C.c_uint64, C.c_uint64, C.c_size_t, C.POINTER(C.c_void_p)]
    lib.speckv_prefetch.argtypes = [C.c_uint32, C.c_uint16, C.c_uint32, C.c_uint32, C.POINTER(C.c_int32), C.c_uint32]
    lib.speckv_init.argtypes = [C.c_char_p]
    lib.speckv_finalize.restype = None
    for e in trace:
        op, a, st, val = e["op"], e["args"], e["status"], e["value"]
        if op == "init":
            if a[0] == "/dev/speckv0":
                # no HIP device here: like the reference without its char device -> -1
                import torch
                if torch.cuda.is_available():
                    continue
            assert lib.speckv_init(a[0].encode()) == st, e
        elif op == "finalize":
            lib.speckv_finalize()
        elif op == "alloc":
            h = C.c_uint64()
            assert lib.speckv_alloc(a[0], None, C.byref(h)) == st, e
            if st == 0: assert h.value == val, e
        elif op == "alloc_null_out":
            assert lib.speckv_alloc(a[0], None, None) == st
        elif op == "free":
            assert lib.speckv_free(a[0]) == st, e
        elif op == "access":
            p = C.c_void_p()
            assert lib.speckv_access(a[0], a[1], a[2], C.byref(p)) == st, e
            if st == 0: assert (p.value or 0) == val, e
        elif op == "access_null_out":
            assert lib.speckv_access(a[0], a[1], a[2], None) == st
        elif op == "prefetch":
            toks = a[4]
            arr = (C.c_int32 * 16)(*(toks if toks else [0] * 16))
            ptr = None if toks is None else arr
            n = 16 if toks is None else len(toks)
            assert lib.speckv_prefetch(a[0], a[1], a[2], a[3], ptr, n) == st, e
        elif op == "set_prefetch_depth":
            assert lib.speckv_set_prefetch_depth(a[0]) == st, e
        elif op == "set_compression_scheme":
            assert lib.speckv_set_compression_scheme(a[0]) == st, e
        else:
            raise AssertionError(op)
    lib.speckv_finalize()


def test_shim_offsets_and_pointers(libpath, golden_dir):
    """F-offset: our CxlSpeckvKVAllocator against the reference shim's outputs."""
    g = json.load(open(os.path.join(golden_dir, "shim_offsets.json")))
    kv = pkg.CxlSpeckvKVAllocator(libpath, "/dev/null")
    try:
        for cfg in g["configs"]:
            T, L, H, D, bpe = (cfg[k] for k in ("T", "L", "H", "D", "bpe"))
            handle = kv.allocate(T, L, H, D, bpe)
            assert handle == cfg["handle"]
            for e in cfg["entries"]:
                off = kv._calc_offset(e["req"], e["layer"], e["head"], e["pos"], e["kind"], D * bpe)
                assert off == e["offset"]
                if e["status"] == 0:
                    assert kv.get_kv_ptr(e["req"], e["layer"], e["head"], e["pos"], e["kind"], D * bpe) == e["ptr"]
                else:
                    with pytest.raises(RuntimeError, match=f"speckv_access failed: {e['status']}"):
                        kv.get_kv_ptr(e["req"], e["layer"], e["head"], e["pos"], e["kind"], D * bpe)
        kv.prefetch_step(0, 0, 100, list(range(1, 17)), 4)
    finally:
        kv.close()


def test_translate_matches_oracle_ids(nulllib, oracle):
    O = oracle.lib
    h1 = nulllib.alloc(5 << 20)
    h2 = nulllib.alloc(12345)
    for h, size in ((h1, 5 << 20), (h2, 12345)):
        for off in (0, 1, 4095, 4096, 8197, size - 1):
            info = nulllib.translate(h, off)
            p = off // 4096
            assert info.virt_page_id == O.orc_virt_page_id(h, p)
            assert info.phys_page_id == O.orc_phys_page_id(h, p)
            assert info.page_size == 4096 and info.pool_device == -1
            d = nulllib.fetch_desc(h, off)
            assert (d.fpga_addr, d.gpu_addr, d.bytes, d.flags) == (info.phys_page_id, O.orc_desc_gpu_addr(info.virt_page_id), 4096, 0)
    # residency flag after access (speckv_allocator.cpp:135): bit1, first page only
    assert nulllib.translate(h1, 3 * 4096).flags == 0
    nulllib.access(h1, 3 * 4096 + 5, 9000)
    assert nulllib.translate(h1, 3 * 4096).flags == 2
    assert nulllib.translate(h1, 4 * 4096).flags == 0          # length ignored by the reference
    with pytest.raises(SpeckvError) as ei:
        nulllib.translate(h1, 5 << 20)
    assert ei.value.status == -1


def test_reference_test_allocator_scenarios(nulllib):
    """Port of reference tests/test_allocator.cpp:11-136 on the fake device."""
    h = nulllib.alloc(1024 * 1024)
    assert h != 0
    nulllib.free(h)
    hs = [nulllib.alloc(4096 * (i + 1)) for i in range(10)]
    assert all(hs) and len(set(hs)) == 10
    for x in hs:
        nulllib.free(x)
    h = nulllib.alloc(4096)
    p = [nulllib.access(h, o, 1024) for o in (0, 1024, 2048)]
    base = 0x4000000000 + (h << 20)
    assert p == [base, base + 0x400, base + 0x800]            # what the reference prints for handle 1: 0x4000100000...
    nulllib.free(h)
    nulllib.prefetch(1, 0, 100, 4, list(range(1, 17)))


def test_reference_test_python_scenarios(libpath):
    """Port of reference tests/test_python.py:17-99.  The reference test re-inits
    without finalize and fails there; ours finalizes."""
    for _ in range(3):
        lib = pkg.SpeckvLib(libpath, "/dev/null")
        with pytest.raises(SpeckvError) as ei:                 # double init -> -1, as the reference
            pkg.SpeckvLib(libpath, "/dev/null")
        assert ei.value.status == -1
        h = lib.alloc(1024 * 1024)
        assert lib.access(h, 0, 4096) == 0x4000000000 + (h << 20)
        lib.prefetch(req_id=1, layer=0, cur_pos=100, depth_k=4, tokens=list(range(1, 17)))
        lib.free(h)
        lib.finalize()


def test_no_cpu_fallback_for_the_data_path(nulllib):
    """The fake device has no data path: every data call fails with DRIVER (-2)."""
    h = nulllib.alloc(1 << 20)
    buf = (C.c_uint8 * 4096)()
    for call in (lambda: nulllib.write(h, 0, C.addressof(buf), 4096, False),
                 lambda: nulllib.read(h, 0, C.addressof(buf), 4096, False),
                 lambda: nulllib.fetch_range(h, 0, 1, C.addressof(buf)),
                 lambda: nulllib.write_strided(h, 0, 1, 1, C.addressof(buf), 1),
                 lambda: nulllib.write_strided_batch([h], [0], [C.addressof(buf)], 1, 1, 1),
                 lambda: nulllib.attend_fp8(h, 0, 1, C.addressof(buf), 8, 0, 32, 0.1, C.addressof(buf)),
                 lambda: nulllib.attend_int4(h, 0, 1, C.addressof(buf), 8, 0, 32, 0.1, C.addressof(buf)),
                 lambda: nulllib.attend_fp8_batch([h], 0, C.addressof(buf), 8, [32], 0.1, C.addressof(buf), None, 1),
                 lambda: nulllib.attend_batch_plan([h], [32], 32, C.addressof(buf), 64, 1),
                 lambda: nulllib.attend_planned(4, C.addressof(buf), 1, 0, C.addressof(buf), 8, 32, 0.1, C.addressof(buf), None, 1),
                 lambda: nulllib.attend_planned(3, C.addressof(buf), 1, 0, C.addressof(buf), 8, 32, 0.1, C.addressof(buf), None, 1),
                 lambda: nulllib.attend_fold_tail(1, 0, 8, 8, C.addressof(buf), C.addressof(buf), C.addressof(buf), 1024, 0.1,
                                                  C.addressof(buf), C.addressof(buf), 1),
                 lambda: nulllib.poll_complete()):
        with pytest.raises(SpeckvError) as ei:
            call()
        assert ei.value.status == -2
    with pytest.raises(SpeckvError) as ei:
        nulllib.set_compression_scheme(2)                      # reference on /dev/null: ioctl fails -> -2
    assert ei.value.status == -2


def test_init_without_gpu_fails_loudly(libpath, capfd):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a HIP device is present")
    with pytest.raises(SpeckvError) as ei:
        pkg.SpeckvLib(libpath, "/dev/speckv0")
    assert ei.value.status == -1                               # reference: open() fails -> exception -> -1
    assert "no usable HIP device" in capfd.readouterr().err


def test_verify_and_adaptive_depth_host_logic(nulllib, golden_dir):
    """speckv_ext_verify drives the same depth trace as the reference prefetcher."""
    g = json.load(open(os.path.join(golden_dir, "prefetch.json")))
    assert nulllib.prefetch_depth() == g["initial_depth"]
    for ok, want in zip(g["outcomes"], g["depth_trace"]):
        hit, depth = nulllib.verify(0, 5 if ok else 9, [1, 5, 7])
        assert hit == bool(ok) and depth == want
    st = nulllib.stats()
    assert st.mispredictions == g["outcomes"].count(0)
    addrs = (C.c_uint64 * 16)(); n = C.c_uint32()
    for c in g["calls"]:
        k = c["depth"] or 4
        assert nulllib.lib.speckv_ext_prefetch_legacy_addrs(c["layer"], k, addrs, C.byref(n)) == 0
        assert list(addrs[:n.value]) == c["addresses"]


def test_layer_ratio_table(libpath, oracle):
    lib = pkg.load_library(libpath)
    lib.speckv_ext_layer_compression_ratio.restype = C.c_double
    lib.speckv_ext_layer_compression_ratio.argtypes = [C.c_uint32]
    for layer in list(range(90)) + [1000]:
        assert lib.speckv_ext_layer_compression_ratio(layer) == oracle.lib.orc_layer_compression_ratio(layer)


def test_div127_identity():
    """The divide-free dequantiser in kernels.hip (div127): q*fl(1/127) plus one
    Newton step equals float(q)/127.0f for every int8 q."""
    import numpy as np
    f32 = np.float32
    q = np.arange(-128, 128).astype(f32)
    rcp = f32(1.0) / f32(127.0)
    assert rcp == np.float32(float.fromhex("0x1.020408p-7"))
    r0 = (q * rcp).astype(f32)
    e = (-127.0 * r0.astype(np.float64) + q.astype(np.float64)).astype(f32)       # exact fma, one rounding
    r1 = (e.astype(np.float64) * np.float64(rcp) + r0.astype(np.float64)).astype(f32)
    assert np.array_equal(r1, (q / f32(127.0)).astype(f32))


def test_int4_encoder_short_cuts():
    """The INT4_G32 encoder (kernels.hip, k_compress<3>) forms a group scale as fp16(max|x| / 7) and quantises by it.  Three
    facts it relies on, exhaustively over all positive finite fp16 values on the host (the device repeats the first and the
    reciprocal in test_fast_division_is_exact):  m/7 as fma(m, hi, m*lo) is the correctly rounded quotient;  with a NORMAL
    stored scale no quotient reaches 7.5, so the clamp to [-7, 7] is only needed under subnormal scales;  the nibbles of a
    dword can be summed as signed i << 4k, biased by 0x88888888 and flipped back."""
    import numpy as np
    f32 = np.float32
    m = np.arange(1, 0x7C00, dtype=np.uint16).view(np.float16).astype(f32)
    hi, lo = f32(float.fromhex("0x1.24924ap-3")), f32(float.fromhex("-0x1.b6db6ep-28"))
    assert hi == f32(1.0 / 7.0) and lo == f32(1.0 / 7.0 - float(hi))
    t = (m * lo).astype(f32)
    got = (m.astype(np.float64) * float(hi) + t.astype(np.float64)).astype(f32)               # the fma: one rounding
    assert np.array_equal(got.view(np.uint32), (m / f32(7.0)).astype(f32).view(np.uint32))
    s16 = (m / f32(7.0)).astype(np.float16)
    normal = (s16.view(np.uint16) & 0x7C00) != 0
    y = m[normal] / s16[normal].astype(f32)
    assert float(np.trunc(y + f32(0.5)).max()) == 7.0 and float(y.max()) < 7.01
    sub = ~normal & (s16.view(np.uint16) != 0)
    assert float((m[sub] / s16[sub].astype(f32)).max()) > 7.5                                  # ... and there it is needed
    rng = np.random.default_rng(1)
    i = rng.integers(-7, 8, (100000, 8)).astype(np.int64)
    n = np.zeros(100000, np.int64)
    for k in range(8):
        n = (n + (i[:, k] << (4 * k))) & 0xFFFFFFFF
    packed = ((n + 0x88888888) & 0xFFFFFFFF) ^ 0x88888888
    want = np.zeros(100000, np.int64)
    for k in range(8):
        want |= (i[:, k] & 0xF) << (4 * k)
    assert np.array_equal(packed, want)


def test_address_encodings(libpath, oracle, reference=None):
    """the reference's address encodings as pure functions,
    against the oracle (itself pinned to the reference's TLB in test_oracle_vs_ref)."""
    import numpy as np
    lib = pkg.load_library(libpath)
    for f, args in (("speckv_ext_encode_virt_page", [C.c_uint32, C.c_uint16, C.c_uint16, C.c_uint32, C.c_uint8]),
                    ("speckv_ext_rtl_prefetch_vaddr", [C.c_uint32, C.c_uint16, C.c_uint32]),
                    ("speckv_ext_atu_translate", [C.c_uint64])):
        getattr(lib, f).restype = C.c_uint64
        getattr(lib, f).argtypes = args
    lib.speckv_ext_codec_model_throughput_gbps.restype = C.c_double
    lib.speckv_ext_codec_model_throughput_gbps.argtypes = [C.c_uint32, C.c_double, C.c_uint32]
    O = oracle.lib
    rng = np.random.default_rng(1)
    for _ in range(500):
        req, layer, head, pos, kind = (int(rng.integers(0, 2**32)), int(rng.integers(0, 2**16)), int(rng.integers(0, 2**16)),
                                       int(rng.integers(0, 2**32)), int(rng.integers(0, 2)))
        assert lib.speckv_ext_encode_virt_page(req, layer, head, pos, kind) == O.orc_encode_virt_page(req, layer, head, pos, kind)
        assert lib.speckv_ext_rtl_prefetch_vaddr(req, layer, pos) == O.orc_rtl_prefetch_vaddr(req, layer, pos)
    # known answers: encode (1,2,3,4,1) and the survey's ATU example 0x123456789 -> 0x4123456789
    assert lib.speckv_ext_encode_virt_page(1, 2, 3, 4, 1) == (1 << 32) | (2 << 16) | (3 << 8) | (4 << 1) | 1
    assert lib.speckv_ext_rtl_prefetch_vaddr(0, 5, 101) == (101 << 1) | (5 << 41)
    assert lib.speckv_ext_atu_translate(0x123456789) == 0x4123456789
    tlb = O.orc_tlb_new(1024)
    for va in [int(v) for v in rng.integers(0, 2**63, 200, dtype=np.uint64)]:
        assert lib.speckv_ext_atu_translate(va) == O.orc_tlb_translate(tlb, va, None)     # every first touch is a miss
    O.orc_tlb_delete(tlb)
    assert lib.speckv_ext_codec_model_throughput_gbps(1, 800.0, 512) == O.orc_codec_throughput_gbps(1, 800.0, 512) == 51.2


def _build_c_demo():
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pkg.library_path()                                   # make sure the library exists
    subprocess.check_call(["make", "-s", "-C", os.path.join(root, "examples")])
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = "/opt/rocm/lib:" + env.get("LD_LIBRARY_PATH", "")
    return os.path.join(root, "examples", "_build", "cxlspeckv_demo"), env


def test_headers_are_c99_and_the_c_demo_runs_on_the_fake_device():
    """include/*.h compile as strict C99 (-Wall -Wextra -pedantic) and a plain C program drives the ABI:
    the page-table part of examples/cxlspeckv_demo.c on the reference's /dev/null device."""
    import subprocess
    exe, env = _build_c_demo()
    out = subprocess.run([exe, "/dev/null"], env=env, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    assert "0x4000102005" in out.stdout and "0x4123456789" in out.stdout      # F-cabi / ATU goldens


def test_calls_from_many_threads(libpath):
    """SURVEY 8b threading: the C ABI is callable from any thread (one process-global lock, speckv_c_api.cpp:10).
    Eight threads allocate, access, translate and free at once on the fake device; every result must be the value
    the single-threaded formulas give and every handle must be unique."""
    import threading
    lib = pkg.SpeckvLib(libpath, "/dev/null")
    errors, handles = [], []
    lock = threading.Lock()

    def worker(seed):
        try:
            mine = []
            for i in range(200):
                size = 4096 * (1 + (seed * 7 + i) % 13)
                h = lib.alloc(size)
                mine.append(h)
                off = ((seed + i) * 977) % size
                assert lib.access(h, off, 16) == 0x4000000000 + (h << 20) + off
                info = lib.translate(h, off)
                assert info.virt_page_id == ((h << 32) | ((off // 4096) << 12)) and info.flags == 2
                if i % 3 == 0:
                    lib.free(mine.pop(0))
                lib.prefetch(seed, i % 4, i, 4, list(range(1, 17)))
            with lock:
                handles.extend(mine)
            for h in mine:
                lib.free(h)
        except Exception as e:                      # noqa: BLE001 - collected and re-raised below
            with lock:
                errors.append(repr(e))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(8)]
    for t in threads: t.start()
    for t in threads: t.join()
    lib.finalize()
    assert not errors, errors[:3]
    assert len(set(handles)) == len(handles)


def test_reference_callers_drive_the_product_library_unchanged():
    """The drop-in proof, kept permanent (dev container only; skipped wherever /root/reference is absent, i.e. on the GPU
    box): the reference's own speckv_ctypes.SpeckvLib and the class body of its vLLM shim, imported from where they lie,
    run on the PRODUCT library on "/dev/null" and produce the F-cabi trace and every F-offset pointer the reference
    library produces; then a seeded random walk over the 8 C-ABI functions, product against the compiled reference
    (oracle/_ref/libspeckv_ref.so), call for call.  tests/golden/dropin_check.py holds the driver."""
    import importlib.util
    path = os.path.join(ROOT, "tests", "golden", "dropin_check.py")
    if not os.path.exists(path):
        pytest.skip("tests/golden/dropin_check.py does not travel to the GPU box")
    spec = importlib.util.spec_from_file_location("dropin_check", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    if not mod.available():
        pytest.skip("needs /root/reference, oracle/_ref/libspeckv_ref.so and the built product library")
    rep = mod.run(ops_per_seed=3000, seeds=3)
    assert rep["cabi_product"] == rep["cabi_reference"] and len(rep["cabi_product"]) >= 35
    assert rep["shim_product"] == rep["shim_reference"] and len(rep["shim_product"]) >= 250
    # SURVEY Appendix A pointers, spelled out (cfg1, 8B@4k, 70B@8k; req_id 1 overflows the single-request allocation)
    by_key = {(r[0],) + tuple(r[1:6]): r for r in rep["shim_product"]}
    assert by_key[(1, 0, 0, 7, 127, 1)][8] == 0x400017ff00
    assert by_key[(2, 0, 31, 7, 4095, 1)][8] == 0x40201fff00
    assert by_key[(3, 0, 79, 7, 8191, 1)][8] == 0x40a02fff00
    assert by_key[(1, 1, 0, 0, 0, 0)][7] == -1
    assert rep["fuzz_ops"] == 9000 and rep["fuzz_diffs"] == [], rep["fuzz_diffs"][:5]
