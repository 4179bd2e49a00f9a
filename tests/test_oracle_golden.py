"""Oracle (oracle/speckv_oracle.c) against the committed golden vectors that
tests/golden/generate_golden.py captured from the reference build.  Runs
everywhere (no /root/reference, no GPU)."""
import ctypes as C
import json
import os

import numpy as np

from oracle.bindings import MMStats, _ptr, u32p, u64p, i32p


def test_cabi_trace(oracle, golden_dir):
    trace = json.load(open(os.path.join(golden_dir, "cabi_trace.json")))["trace"]
    O = oracle.lib
    m = O.orc_cabi_new()
    for e in trace:
        op, a, st, val = e["op"], e["args"], e["status"], e["value"]
        if op == "init":
            assert O.orc_cabi_init(m, a[0].encode()) == st
        elif op == "finalize":
            O.orc_cabi_finalize(m)
        elif op == "alloc":
            h = C.c_uint64()
            assert O.orc_cabi_alloc(m, a[0], C.byref(h)) == st
            if st == 0: assert h.value == val
        elif op == "alloc_null_out":
            assert O.orc_cabi_alloc(m, a[0], None) == st
        elif op == "free":
            assert O.orc_cabi_free(m, a[0]) == st
        elif op == "access":
            p = C.c_uint64()
            assert O.orc_cabi_access(m, a[0], a[1], a[2], C.byref(p)) == st, e
            if st == 0: assert p.value == val, e
        elif op == "access_null_out":
            assert O.orc_cabi_access(m, a[0], a[1], a[2], None) == st
        elif op == "prefetch":
            toks = a[4]
            arr = np.array(toks if toks else [0], np.int32)
            ptr = None if toks is None else _ptr(arr, i32p)
            n = 16 if toks is None else len(toks)
            assert O.orc_cabi_prefetch(m, a[0], a[1], a[2], a[3], ptr, n) == st
        elif op == "set_prefetch_depth":
            assert O.orc_cabi_set_prefetch_depth(m, a[0]) == st
        elif op == "set_compression_scheme":
            assert O.orc_cabi_set_compression_scheme(m, a[0]) == st
        else:
            raise AssertionError(op)
    O.orc_cabi_delete(m)


def test_shim_offsets(oracle, golden_dir):
    g = json.load(open(os.path.join(golden_dir, "shim_offsets.json")))
    O = oracle.lib
    m = O.orc_cabi_new()
    assert O.orc_cabi_init(m, b"/dev/null") == 0
    for cfg in g["configs"]:
        T, L, H, D, bpe = (cfg[k] for k in ("T", "L", "H", "D", "bpe"))
        total = O.orc_shim_total_bytes(T, L, H, D, bpe)
        assert total == cfg["total_bytes"]
        h = C.c_uint64()
        assert O.orc_cabi_alloc(m, total, C.byref(h)) == 0 and h.value == cfg["handle"]
        for e in cfg["entries"]:
            off = O.orc_calc_offset(e["req"], e["layer"], e["head"], e["pos"], e["kind"], D * bpe, L, T, H)
            assert off == e["offset"]
            p = C.c_uint64()
            st = O.orc_cabi_access(m, h.value, off, D * bpe, C.byref(p))
            assert st == e["status"]
            if st == 0: assert p.value == e["ptr"]
    O.orc_cabi_delete(m)


def test_codec_vectors(oracle, golden_dir):
    g = np.load(os.path.join(golden_dir, "codec_vectors.npz"))
    names = sorted({k.split(".")[0] for k in g.files if k.endswith(".x")})
    assert len(names) >= 10
    for name in names:
        x, s, rle, y = g[f"{name}.x"], g[f"{name}.scale"][0], g[f"{name}.rle"], g[f"{name}.y"]
        s_o, rle_o = oracle.compress_f32(x, mode=oracle.REF_EXACT)
        assert s_o.tobytes() == s.tobytes(), name
        assert rle_o.tobytes() == rle.tobytes(), name
        y_o = oracle.decompress_f32(rle, s, mode=oracle.REF_EXACT)
        assert y_o.tobytes() == y.tobytes(), name
    # survey Appendix A known answer
    assert g["kat.rle"].astype(np.int8).tolist() == [0, 1, 127, 1, 2, 1, -65, 1, 0, 2, -32, 1, -31, 1, 0, 1, -1, 1, 0, 1]
    for i in range(5):
        rle, y = g[f"malformed{i}.rle"], g[f"malformed{i}.y"]
        assert oracle.decompress_f32(rle, 0.5, cap=y.size).tobytes() == y.tobytes()
    # large-vector digest
    x = np.random.default_rng(int(g["big.seed"][0])).standard_normal(int(g["big.n"][0])).astype(np.float32)
    s, rle = oracle.compress_f32(x)
    assert s.tobytes() == g["big.scale"][0].tobytes() and rle.size == int(g["big.compressed_size"][0])
    crc = int(np.bitwise_xor.reduce(rle.astype(np.uint64) * (np.arange(rle.size, dtype=np.uint64) % 251 + 1)))
    assert crc == int(g["big.rle_crc"][0])
    y = oracle.decompress_f32(rle, s)
    assert int(y.view(np.uint32).astype(np.uint64).sum()) == int(g["big.y_sum_bits"][0])


def test_block_f16_forms_consistent(oracle, golden_dir):
    """The fp16 block wrappers are the f32 reference maths bracketed by exact
    widening and one RNE narrowing (checked against numpy's fp16)."""
    g = np.load(os.path.join(golden_dir, "codec_vectors.npz"))
    for name in ("gauss_a", "gauss_b", "piecewise32", "zeros", "f16_extremes", "small"):
        x32, s, rle, y = g[f"{name}.x"], g[f"{name}.scale"][0], g[f"{name}.rle"], g[f"{name}.y"]
        x16 = x32.astype(np.float16)
        assert np.array_equal(x16.astype(np.float32), x32)
        s_o, rec = oracle.compress_block_f16(x16, scheme=2, mode=0)
        assert s_o.tobytes() == s.tobytes() and rec.tobytes() == rle.tobytes()
        y16 = oracle.decompress_block_f16(rec, s_o, scheme=2, mode=0, cap=2048)
        with np.errstate(over="ignore"):
            assert y16.tobytes() == y.astype(np.float16).tobytes(), name
        y32 = oracle.decompress_block_f32(rec, s_o, scheme=2, mode=0, cap=2048)
        assert y32.tobytes() == y.tobytes()


def test_half_conversions_exhaustive(oracle):
    L = oracle.lib
    allh = np.arange(65536, dtype=np.uint16)
    ref32 = allh.view(np.float16).astype(np.float32)
    got = np.array([L.orc_half_to_float(int(h)) for h in allh], np.float32)
    nan = np.isnan(ref32)
    assert np.array_equal(np.isnan(got), nan)
    assert got[~nan].tobytes() == ref32[~nan].tobytes()
    rng = np.random.default_rng(0)
    f = np.concatenate([rng.standard_normal(20000).astype(np.float32) * np.float32(10.0) ** rng.integers(-9, 6, 20000).astype(np.float32),
                        ref32[~nan], np.array([65519.9, 65520.0, 65504.0, 2.0**-25, 2.0**-25 * 1.0001, 2.0**-24, 6.1e-5, np.inf, -np.inf], np.float32)])
    # halfway cases between adjacent halves
    h = allh[(allh & 0x7FFF) < 0x7BFF]
    mid = (h.view(np.float16).astype(np.float64) + (h + 1).astype(np.uint16).view(np.float16).astype(np.float64)) / 2
    f = np.concatenate([f, mid.astype(np.float32)])
    with np.errstate(over="ignore"):
        want = f.astype(np.float16).view(np.uint16)
    got = np.array([L.orc_float_to_half(float(v)) for v in f], np.uint16)
    assert np.array_equal(got, want)


def test_mm_trace(oracle, golden_dir):
    g = json.load(open(os.path.join(golden_dir, "mm_trace.json")))
    O = oracle.lib
    mm = O.orc_mm_new(12, 3, 128, 4096)
    for op, a, want in g["events"]:
        fn = getattr(O, "orc_mm_" + op)
        got = fn(mm, *a)
        if want is not None:
            assert got == want, (op, a, got, want)
    s = MMStats(); O.orc_mm_get_statistics(mm, C.byref(s))
    assert [s.l1_hits, s.l1_misses, s.l2_hits, s.l2_misses, s.l3_accesses, s.migrations_l1_to_l3,
            s.migrations_l3_to_l1] == g["stats_u"]
    assert [s.l1_hit_rate, s.l2_hit_rate] == g["stats_d"]
    O.orc_mm_delete(mm)


def test_prefetch_golden(oracle, golden_dir):
    g = json.load(open(os.path.join(golden_dir, "prefetch.json")))
    O = oracle.lib
    depth = g["initial_depth"]
    for c in g["calls"]:
        eff = c["depth"] or depth
        out = np.zeros(16, np.uint64)
        n = O.orc_prefetch_legacy(None, c["layer"], eff, eff, _ptr(out, u64p))
        assert out[:n].tolist() == c["addresses"]
    assert g["calls"][0]["addresses"] == [0x50001, 0x50002, 0x50003, 0x50004]
    # token predictor: the oracle's restatement with the reference's
    # constructor weights reproduces the reference's predictions bit for bit
    emb, wout = oracle.lstm_reference_weights(1)
    for c in g["calls"]:
        eff = c["depth"] or depth
        tok, prob = oracle.lstm_predict(emb, wout, c["history"], eff)
        assert tok.tolist() == c["tokens"]
        assert prob.view(np.uint32).tolist() == c["conf_bits"]
    assert g["calls"][0]["tokens"] == [11465, 24880, 10938, 28629]
    ad = O.orc_adapt_new(depth)
    for ok, want in zip(g["outcomes"], g["depth_trace"]):
        O.orc_adapt_update(ad, ok)
        assert O.orc_adapt_depth(ad) == want
    O.orc_adapt_delete(ad)
    assert g["mispredictions_after_miss"] == g["mispredictions_after_hit"] + 1


def test_engine_prefetch_pages_properties(oracle):
    """Own-semantics lookup (RTL intent through the shim layout): pages returned
    cover exactly the K and V rows of positions cur_pos+1..cur_pos+k."""
    T, L, H, D, bpe = 4096, 32, 8, 128, 2
    pages_total = T * L * H * D * bpe * 2 // 4096
    rng = np.random.default_rng(8)
    for _ in range(200):
        layer = int(rng.integers(0, L)); pos = int(rng.integers(0, T)); k = int(rng.integers(1, 9))
        got = oracle.prefetch_pages(0, layer, pos, k, L, T, H, D, bpe, pages_total).tolist()
        want = []
        for kind in (0, 1):
            seen = []
            for p in range(pos + 1, min(pos + k, T - 1) + 1):
                off = ((((0 * L + layer) * 2 + kind) * T + p) * H) * D * bpe
                for pg in range(off // 4096, (off + H * D * bpe - 1) // 4096 + 1):
                    if pg not in seen: seen.append(pg)
            want += seen
        assert got == want
    flags = np.zeros(pages_total, np.uint32)
    base = oracle.prefetch_pages(0, 3, 100, 4, L, T, H, D, bpe, pages_total).tolist()
    flags[base[0]] = 2; flags[base[-1]] = 1
    assert oracle.prefetch_pages(0, 3, 100, 4, L, T, H, D, bpe, pages_total, flags).tolist() == base[1:-1]
    # req_id beyond the single-request allocation -> nothing
    assert oracle.prefetch_pages(1, 0, 0, 4, L, T, H, D, bpe, pages_total).size == 0
