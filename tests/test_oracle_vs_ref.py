"""Pins oracle/speckv_oracle.c against the REFERENCE ITSELF
(oracle/_ref/libspeckv_ref.so, built from /root/reference by oracle/Makefile).
Runs only where that build exists (dev container, or a GPU box that received
the prebuilt .so); the same facts are frozen in tests/golden/ for everywhere else.
"""
import ctypes as C

import numpy as np
import pytest

from oracle.bindings import MMStats, f32p, u8p, u32p, u64p, f64p, i32p, _ptr


def _inputs():
    rng = np.random.default_rng(1234)
    yield "gauss2048", rng.standard_normal(2048).astype(np.float32)
    yield "gauss_f16", rng.standard_normal(2048).astype(np.float16).astype(np.float32)
    yield "zeros", np.zeros(2048, np.float32)
    yield "const", np.full(2048, 0.37, np.float32)
    yield "piecewise32", np.repeat(rng.standard_normal(64).astype(np.float32), 32)
    yield "ramp", np.linspace(-3, 3, 2048, dtype=np.float32)
    yield "one", np.array([1.5], np.float32)
    yield "long_zero_run", np.zeros(1000, np.float32)
    yield "kat", np.array([0, 1, -1, 0.5, 0.5, 0.5, 0.25, 127, 0.007874, 0.003937, -0.0039], np.float32)
    yield "tiny", (rng.standard_normal(777) * 1e-30).astype(np.float32)
    yield "huge", (rng.standard_normal(513) * 1e30).astype(np.float32)
    yield "denorm", (rng.standard_normal(300) * 1e-41).astype(np.float32)
    yield "big131072", rng.standard_normal(131072).astype(np.float32)
    x = rng.standard_normal(2048).astype(np.float32); x[5] = np.inf
    yield "with_inf", x
    x = rng.standard_normal(2048).astype(np.float32); x[7] = np.nan
    yield "with_nan", x


@pytest.mark.parametrize("name,x", list(_inputs()), ids=[n for n, _ in _inputs()])
def test_codec_bit_exact(oracle, reference, name, x):
    s_ref, rle_ref = reference.compress_f32(x)
    s_orc, rle_orc = oracle.compress_f32(x, mode=oracle.REF_EXACT)
    assert s_ref.tobytes() == s_orc.tobytes()
    assert rle_ref.tobytes() == rle_orc.tobytes()
    y_ref = reference.decompress_f32(rle_ref, s_ref)
    y_orc = oracle.decompress_f32(rle_orc, s_orc, mode=oracle.REF_EXACT)
    assert y_ref.shape == y_orc.shape == x.shape
    # NaN payloads are compared as a class, everything else bit for bit
    nan = np.isnan(y_ref)
    assert np.array_equal(nan, np.isnan(y_orc))
    assert y_ref[~nan].tobytes() == y_orc[~nan].tobytes()


def test_codec_empty(oracle, reference):
    x = np.zeros(0, np.float32)
    s_ref, rle_ref = reference.compress_f32(x)
    s_orc, rle_orc = oracle.compress_f32(x)
    assert s_ref == s_orc == 1.0 and rle_ref.size == rle_orc.size == 0


def test_decode_malformed_streams(oracle, reference):
    # odd trailing byte dropped; count 0 emits nothing; counts > 127
    for rle in ([5, 3, 7], [5, 0, 9, 2], [255, 200, 1, 255, 128, 1], [1], []):
        rle = np.array(rle, np.uint8)
        cap = int(rle[1::2].astype(np.int64).sum()) if rle.size >= 2 else 0
        y_ref = reference.decompress_f32(rle, 0.5, cap=cap)
        y_orc = oracle.decompress_f32(rle, 0.5, cap=cap)
        assert y_ref.tobytes() == y_orc.tobytes()


def test_layer_ratio_and_throughput(oracle, reference):
    for layer in list(range(0, 90)) + [99, 1000]:
        assert oracle.lib.orc_layer_compression_ratio(layer) == reference.lib.ref_engine_ratio(reference.engine, layer)
    assert oracle.lib.orc_codec_throughput_gbps(1, 800.0, 512) == reference.lib.ref_engine_throughput(reference.engine)


def test_tlb_translate(oracle, reference):
    eng = reference.lib.ref_engine_new()
    tlb = oracle.lib.orc_tlb_new(1024)
    rng = np.random.default_rng(7)
    vas = [0x123456789, 0x123456000, 0x123456fff, 0x1000123456789, 0xFFFF_FFFF_FFFF_FFFF, 0]
    vas += [int(v) for v in rng.integers(0, 2**63, 200, dtype=np.uint64)]
    vas += [0x123456789 + 1024 * 4096, 0x123456789]   # conflict eviction then re-miss
    for va in vas:
        assert oracle.lib.orc_tlb_translate(tlb, va, None) == reference.lib.ref_engine_translate(eng, va), hex(va)
    oracle.lib.orc_tlb_delete(tlb)
    reference.lib.ref_engine_delete(eng)


def _ref_stats(reference, mm):
    u = (C.c_uint64 * 7)(); d = (C.c_double * 2)()
    reference.lib.ref_mm_get_statistics(mm, u, d)
    return list(u), list(d)


def _orc_stats(oracle, mm):
    s = MMStats()
    oracle.lib.orc_mm_get_statistics(mm, C.byref(s))
    return [s.l1_hits, s.l1_misses, s.l2_hits, s.l2_misses, s.l3_accesses,
            s.migrations_l1_to_l3, s.migrations_l3_to_l1], [s.l1_hit_rate, s.l2_hit_rate]


def test_memory_manager_trace(oracle, reference):
    """Random op trace on both; every return value and the stats must agree.
    (No trace reaches the reference's eviction path: it self-deadlocks there.)"""
    R, O = reference.lib, oracle.lib
    rm, om = R.ref_mm_new(12, 3, 128), O.orc_mm_new(12, 3, 128, 4096)
    rng = np.random.default_rng(99)
    bases = []
    for i in range(12):
        size = int(rng.integers(1, 40000)); layer = int(rng.integers(0, 80)); tier = int(rng.integers(0, 3))
        a, b = R.ref_mm_allocate(rm, size, layer, tier), O.orc_mm_allocate(om, size, layer, tier)
        assert a == b
        bases.append((a, size))
    for step in range(3000):
        base, size = bases[int(rng.integers(0, len(bases)))]
        va = base + int(rng.integers(0, size + 5000))
        op = int(rng.integers(0, 10))
        if op == 0: assert R.ref_mm_translate(rm, va) == O.orc_mm_translate(om, va)
        elif op == 1:
            t = int(rng.integers(0, 3)); assert R.ref_mm_is_in_cache(rm, va, t) == O.orc_mm_is_in_cache(om, va, t)
        elif op == 2: assert R.ref_mm_promote_to_l1(rm, va) == O.orc_mm_promote_to_l1(om, va)
        elif op == 3: assert R.ref_mm_demote_to_l3(rm, va) == O.orc_mm_demote_to_l3(om, va)
        elif op == 4: R.ref_mm_update_access_tracking(rm, va); O.orc_mm_update_access_tracking(om, va)
        elif op == 5: assert R.ref_mm_is_hot_page(rm, va) == O.orc_mm_is_hot_page(om, va)
        elif op == 6: R.ref_mm_mark_modified(rm, va); O.orc_mm_mark_modified(om, va)
        elif op == 7: R.ref_mm_invalidate_page(rm, va); O.orc_mm_invalidate_page(om, va)
        elif op == 8: assert R.ref_mm_get_page_state(rm, va) == O.orc_mm_get_page_state(om, va)
        elif op == 9 and step % 50 == 0: R.ref_mm_deallocate(rm, va); O.orc_mm_deallocate(om, va)
    assert _ref_stats(reference, rm) == _orc_stats(oracle, om)
    assert R.ref_mm_translate(rm, 0x42) == O.orc_mm_translate(om, 0x42) == 0
    R.ref_mm_delete(rm); O.orc_mm_delete(om)


def test_cxl_access_policy(oracle, reference):
    """memory_allocator.cpp:105-143 : L1 hit / L2 hot-promote / L3 promote."""
    R, O = reference.lib, oracle.lib
    ca = R.ref_ca_new(12, 3, 128)
    om = O.orc_mm_new(12, 3, 128, 4096)
    h = R.ref_ca_malloc(ca, 16 * 4096, 5)
    b = O.orc_mm_allocate(om, 16 * 4096, 5, 2)
    assert h == b == 0x100000000
    rng = np.random.default_rng(5)
    for _ in range(500):
        off = int(rng.integers(0, 16 * 4096))
        assert R.ref_ca_access(ca, h, off, 64) == O.orc_mm_cxl_access(om, b, off)
    R.ref_ca_delete(ca); O.orc_mm_delete(om)


def test_prefetch_legacy_addresses_and_depth(oracle, reference):
    R, O = reference.lib, oracle.lib
    rm, om = R.ref_mm_new(12, 3, 128), O.orc_mm_new(12, 3, 128, 4096)
    pf = R.ref_pf_new(rm, 4, 16)
    for hist, layer, depth in (([*range(1, 17)], 5, 0), ([*range(101, 117)], 0, 8), ([7, 8, 9], 79, 2)):
        h = np.array(hist, np.uint32)
        out_r = np.zeros(16, np.uint64); out_o = np.zeros(16, np.uint64)
        eff = depth if depth else R.ref_pf_adaptive_depth(pf)
        n_r = R.ref_pf_prefetch(pf, _ptr(h, u32p), h.size, layer, depth, _ptr(out_r, u64p), None, None, 16)
        n_o = O.orc_prefetch_legacy(om, layer, eff, eff, _ptr(out_o, u64p))
        assert n_r == n_o and out_r.tolist() == out_o.tolist()
    # adaptive depth trace, same outcomes on both
    ad = O.orc_adapt_new(4)
    rng = np.random.default_rng(3)
    outcomes = [1] * 9 + [1] * 6 + [0] * 12 + [int(v) for v in (rng.random(400) < 0.9)]
    for i, ok in enumerate(outcomes):
        R.ref_pf_update_accuracy(pf, i, ok); O.orc_adapt_update(ad, ok)
        assert R.ref_pf_adaptive_depth(pf) == O.orc_adapt_depth(ad), i
    pred = np.array([1, 2, 3], np.uint32)
    base = R.ref_pf_handle_misprediction(pf, 2, _ptr(pred, u32p), 3)
    assert R.ref_pf_handle_misprediction(pf, 5, _ptr(pred, u32p), 3) == base + 1
    assert O.orc_is_misprediction(5, _ptr(pred, u32p), 3) == 1 and O.orc_is_misprediction(2, _ptr(pred, u32p), 3) == 0
    O.orc_adapt_delete(ad); R.ref_pf_delete(pf); R.ref_mm_delete(rm); O.orc_mm_delete(om)


def test_cabi_model_vs_reference_cabi(oracle, reference):
    """The reference's own speckv_* on the fake device vs the oracle's model."""
    R, O = reference.lib, oracle.lib
    m = O.orc_cabi_new()
    assert R.speckv_init(b"/dev/speckv0") == O.orc_cabi_init(m, b"/dev/speckv0") == -1
    assert R.speckv_free(1) == O.orc_cabi_free(m, 1) == -4
    assert R.speckv_init(b"/dev/null") == O.orc_cabi_init(m, b"/dev/null") == 0
    assert R.speckv_init(b"/dev/null") == O.orc_cabi_init(m, b"/dev/null") == -1
    rng = np.random.default_rng(11)
    handles = []
    for size in [1 << 20, 4096, 1, 0, 12345, 5 << 20]:
        hr, ho = C.c_uint64(), C.c_uint64()
        assert R.speckv_alloc(size, None, C.byref(hr)) == O.orc_cabi_alloc(m, size, C.byref(ho)) == 0
        assert hr.value == ho.value
        handles.append((hr.value, size))
    assert R.speckv_alloc(10, None, None) == O.orc_cabi_alloc(m, 10, None) == -4
    for _ in range(2000):
        h, size = handles[int(rng.integers(0, len(handles)))]
        if rng.random() < 0.05: h = 999
        off = int(rng.integers(0, size + 9000))
        pr, po = C.c_void_p(), C.c_uint64()
        sr = R.speckv_access(h, off, 64, C.byref(pr)); so = O.orc_cabi_access(m, h, off, 64, C.byref(po))
        assert sr == so
        if sr == 0: assert (pr.value or 0) == po.value
    tok = np.arange(1, 17, dtype=np.int32)
    assert R.speckv_prefetch(1, 0, 100, 4, _ptr(tok, i32p), 16) == O.orc_cabi_prefetch(m, 1, 0, 100, 4, _ptr(tok, i32p), 16) == 0
    assert R.speckv_prefetch(1, 0, 100, 4, _ptr(tok, i32p), 0) == O.orc_cabi_prefetch(m, 1, 0, 100, 4, _ptr(tok, i32p), 0) == -4
    assert R.speckv_prefetch(1, 0, 100, 4, None, 16) == O.orc_cabi_prefetch(m, 1, 0, 100, 4, None, 16) == -4
    assert R.speckv_set_prefetch_depth(8) == O.orc_cabi_set_prefetch_depth(m, 8) == -2
    assert R.speckv_set_compression_scheme(2) == O.orc_cabi_set_compression_scheme(m, 2) == -2
    for h in (1, 1, 12345):
        assert R.speckv_free(h) == O.orc_cabi_free(m, h) == 0
    pr, po = C.c_void_p(), C.c_uint64()
    assert R.speckv_access(1, 0, 1, C.byref(pr)) == O.orc_cabi_access(m, 1, 0, 1, C.byref(po)) == -1
    R.speckv_finalize(); O.orc_cabi_finalize(m)
    assert R.speckv_free(1) == O.orc_cabi_free(m, 1) == -4
    assert R.speckv_init(b"/dev/null") == O.orc_cabi_init(m, b"/dev/null") == 0
    hr, ho = C.c_uint64(), C.c_uint64()
    R.speckv_alloc(8192, None, C.byref(hr)); O.orc_cabi_alloc(m, 8192, C.byref(ho))
    assert hr.value == ho.value == 1
    R.speckv_finalize(); O.orc_cabi_finalize(m); O.orc_cabi_delete(m)
