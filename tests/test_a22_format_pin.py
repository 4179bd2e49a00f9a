"""-m "not gpu": an INDEPENDENT pin for the format half of SURVEY 8a row A22.

The reference has no int4/fp8 code, so the oracle's e4m3 conversions are checked against a third-party
implementation present in the image: PyTorch's CPU `torch.float8_e4m3fn` casts (OCP e4m3fn: bias 7, no
infinities, S.1111.111 = NaN, max finite 448).

Conventions pinned here (and stated in DESIGN.md):
  * decode: all 256 byte values, bit-identical fp32 (NaN for 0x7F / 0xFF on both sides);
  * encode: round-to-nearest-even of every FINITE fp16 value with |x| <= 448 -- the only inputs the codec feeds the
    converter (it clamps x/scale to +-448 first) -- byte-identical;
  * saturation: beyond 448 the oracle SATURATES to +-448 (0x7E / 0xFE), which is what the gfx950 conversion
    instruction does with its clamp and what the block codec relies on; torch's cast turns those into NaN
    instead, so that range is pinned to the stated convention, not to torch.

The INT4_G32 block format and both attention checkers (orc_attend_f16, orc_attend_fp8) are pinned further down against a
numpy float64 restatement written from the format description alone (it never calls the oracle).
"""
import numpy as np
import pytest

torch = pytest.importorskip("torch")


def test_e4m3_decode_of_all_256_bytes_equals_torch(oracle):
    L = oracle.lib
    b = np.arange(256, dtype=np.uint8)
    want = torch.from_numpy(b).view(torch.float8_e4m3fn).to(torch.float32).numpy()
    got = np.array([L.orc_e4m3_to_f32(int(v)) for v in b], np.float32)
    assert np.array_equal(np.isnan(got), np.isnan(want))
    assert np.isnan(want).sum() == 2 and np.isnan(got[0x7F]) and np.isnan(got[0xFF])
    ok = ~np.isnan(want)
    assert np.array_equal(got[ok].view(np.uint32), want[ok].view(np.uint32))
    assert got[0x7E] == 448.0 and got[0xFE] == -448.0 and got[0x01] == 2.0 ** -9


def test_e4m3_rne_encode_of_every_finite_fp16_equals_torch(oracle):
    L = oracle.lib
    h = np.arange(65536, dtype=np.uint32).astype(np.uint16).view(np.float16)
    f = h.astype(np.float32)
    finite = np.isfinite(f)
    inrange = finite & (np.abs(f) <= 448.0)
    want = torch.from_numpy(f).to(torch.float8_e4m3fn).view(torch.uint8).numpy()
    got = np.array([L.orc_f32_to_e4m3(float(v)) if fin else 0 for v, fin in zip(f, finite)], np.uint8)
    # -0.0 and +0.0 keep their sign on both sides; everything in range is byte-identical
    assert np.array_equal(got[inrange], want[inrange])
    assert int(inrange.sum()) == 48642          # every fp16 with |x| <= 448, both signs, zeros and subnormals included
    # halfway cases really are ties-to-even (spot check: 17 is halfway between 16 and 18 in e4m3 -> 16)
    assert L.orc_f32_to_e4m3(17.0) == L.orc_f32_to_e4m3(16.0)
    assert L.orc_f32_to_e4m3(19.0) == L.orc_f32_to_e4m3(20.0)
    # stated convention beyond the range: saturate, never NaN / wrap
    over = finite & (np.abs(f) > 448.0)
    assert set(got[over & (f > 0)].tolist()) == {0x7E}
    assert set(got[over & (f < 0)].tolist()) == {0xFE}


def test_e4m3_round_trip_is_identity_on_representable_values(oracle):
    L = oracle.lib
    for b in range(256):
        if b in (0x7F, 0xFF):
            continue
        assert L.orc_f32_to_e4m3(L.orc_e4m3_to_f32(b)) == b


def test_fp8_block_format_against_torch(oracle):
    """The whole FP8_E4M3 block format (scale = max|x|/448, bytes = e4m3(clamp(x/scale))) restated with torch's cast."""
    rng = np.random.default_rng(44)
    x = rng.standard_normal((16, 2048)).astype(np.float16)
    x[3] = 0
    x[5, ::7] *= 30
    scales, lens, recs = oracle.compress_blocks_f16(x, 4, 0)
    for i in range(x.shape[0]):
        xf = x[i].astype(np.float32)
        mx = np.abs(xf).max()
        s = np.float32(mx) / np.float32(448.0) if mx > 0 else np.float32(1.0)
        assert np.float32(scales[i]).tobytes() == np.float32(s).tobytes()
        v = np.clip((xf / s).astype(np.float32), -448.0, 448.0)
        want = torch.from_numpy(v).to(torch.float8_e4m3fn).view(torch.uint8).numpy()
        assert lens[i] == 2048 and np.array_equal(recs[i, :2048], want)
    y = oracle.decompress_blocks_f16(recs, lens, scales, 4, 0)
    for i in range(x.shape[0]):
        dec = torch.from_numpy(recs[i, :2048].copy()).view(torch.float8_e4m3fn).to(torch.float32).numpy() * np.float32(scales[i])
        assert np.array_equal(y[i].view(np.uint16), dec.astype(np.float16).view(np.uint16))


# ---- INT4_G32 and the attention arithmetic: restated here from the format description alone -------------------------
# include/speckv_ext.h ("SPECKV_COMP_INT4_G32: record 1152 B = 64 fp16 group scales + 2048 nibbles") and
# oracle/speckv_oracle.h (per group of 32: s = fp16(max|x| / 7); q = clamp(round-half-away(x / s), -7, 7), 0 when s == 0
# or x is NaN; nibble of element 2i in the low half of byte i, two's complement; y = fp16(q * s)).  Written in numpy
# float64 WITHOUT calling the oracle; float64 is exact enough to reproduce the fp32 arithmetic of the definition: max|x|/7
# and x/s of 11-bit operands never come within 2^-24 of a rounding boundary they do not sit on exactly.
def int4_g32_encode_numpy(x16):
    x = np.asarray(x16, np.float16).astype(np.float64).reshape(-1, 32)
    mx = np.max(np.where(np.isnan(x), 0.0, np.abs(x)), axis=1)
    with np.errstate(over="ignore"):
        s16 = (mx / 7.0).astype(np.float16)
    s = s16.astype(np.float64)
    with np.errstate(divide="ignore", invalid="ignore"):
        v = x / s[:, None]
    r = np.sign(v) * np.floor(np.abs(v) + 0.5)
    r = np.where(np.isnan(r), 0.0, r)
    r = np.clip(r, -7.0, 7.0)
    r = np.where((s[:, None] == 0.0) | np.isnan(s[:, None]), 0.0, r)
    q = r.astype(np.int64).reshape(-1) & 0xF
    nib = (q[0::2] | (q[1::2] << 4)).astype(np.uint8)
    return np.concatenate([s16.view(np.uint8), nib])


def int4_g32_decode_numpy(rec):
    rec = np.asarray(rec, np.uint8)
    groups = (rec.size * 2) // (32 + 4)                      # 2 B of scale + 16 B of nibbles per group
    s = rec[:2 * groups].view(np.float16).astype(np.float64)
    nib = rec[2 * groups:]
    q = np.empty(2 * nib.size, np.int64)
    q[0::2] = nib & 0xF
    q[1::2] = nib >> 4
    q = np.where(q >= 8, q - 16, q)
    with np.errstate(invalid="ignore", over="ignore"):
        return (q.astype(np.float64) * np.repeat(s, 32)).astype(np.float16)


def attention_numpy(q, k, v, sm_scale):
    """softmax(q.k^T * sm_scale).v in float64: q [g][d], k / v [n][d] -> out [g][d], lse [g], mag [g][d]."""
    q = np.asarray(q, np.float64); k = np.asarray(k, np.float64); v = np.asarray(v, np.float64)
    s = (q @ k.T) * float(sm_scale)
    m = s.max(axis=1, keepdims=True)
    p = np.exp(s - m)
    l = p.sum(axis=1, keepdims=True)
    return (p @ v) / l, (m + np.log(l))[:, 0], (p @ np.abs(v)) / l


def _int4_test_blocks():
    rng = np.random.default_rng(45)
    x = (rng.standard_normal((24, 2048)) * rng.uniform(0.01, 40.0, (24, 1))).astype(np.float16)
    x[3] = 0
    x[4, :32] = 0                                             # one zero group inside a block
    x[5, ::5] *= 50
    x[6] = np.float16(65504.0)                                # group scale = fp16(65504 / 7)
    x[7] = (rng.standard_normal(2048) * 6e-8).astype(np.float16)    # fp16 subnormals: scales underflow to 0 or denormals
    x[8] = np.repeat(np.arange(-7, 8, dtype=np.float16), 137)[:2048]   # values that sit exactly on the grid and on ties
    x[9] = (np.arange(2048) % 15 - 7) * np.float16(0.5) + np.float16(0.25)
    return x


def test_int4_g32_block_format_against_the_numpy_restatement(oracle):
    x = _int4_test_blocks()
    scales, lens, recs = oracle.compress_blocks_f16(x, 3, 0)
    y = oracle.decompress_blocks_f16(recs, lens, scales, 3, 0)
    for i in range(x.shape[0]):
        want = int4_g32_encode_numpy(x[i])
        assert lens[i] == 1152 and want.size == 1152
        assert np.array_equal(recs[i, :1152], want), i
        assert np.array_equal(y[i].view(np.uint16), int4_g32_decode_numpy(want).view(np.uint16)), i
    # every nibble value x a spread of scales decodes as the description says (incl. -8, which the encoder never emits)
    rec = np.zeros(1152, np.uint8)
    rec[:128] = np.array([0.0, 1.0, 0.333, 65504.0 / 7, 6e-8, 9360.0, 1e-3, 2.5] * 8, np.float16).view(np.uint8)
    rec[128:] = (np.arange(1024) * 37 + 11).astype(np.uint8)
    got = oracle.decompress_blocks_f16(rec[None, :].repeat(2, 0), np.array([1152, 1151], np.uint32), np.ones(2, np.float32), 3, 0)
    assert np.array_equal(got[0].view(np.uint16), int4_g32_decode_numpy(rec).view(np.uint16))
    assert not got[1].any()                                    # a short record decodes to zeros


def test_attention_oracles_against_the_numpy_restatement(oracle):
    """orc_attend_f16 (checker of the INT4 fused attention) and orc_attend_fp8 against the float64 numpy attention over
    the values the format descriptions give (INT4 pages through int4_g32_decode_numpy; e4m3 bytes through torch)."""
    from oracle.bindings import _ptr, u8p, u16p, f32p
    L = oracle.lib
    rng = np.random.default_rng(46)
    G, D, NPOS = 8, 128, 300
    qh = (rng.standard_normal((G, D)) * 2).astype(np.float16)
    sm = 1.0 / np.sqrt(D)
    # fp16 K / V rows as the INT4 format decodes them
    pages = (rng.standard_normal((2 * NPOS * D // 2048 + 1, 2048)) * 3).astype(np.float16)
    dec = np.stack([int4_g32_decode_numpy(int4_g32_encode_numpy(p)) for p in pages]).reshape(-1, D)
    k16 = np.ascontiguousarray(dec[:NPOS]); v16 = np.ascontiguousarray(dec[NPOS:2 * NPOS])
    o = np.zeros((G, D), np.float32); l = np.zeros(G, np.float32); m = np.zeros((G, D), np.float32)
    L.orc_attend_f16(_ptr(qh.view(np.uint16).reshape(-1), u16p), G, _ptr(k16.view(np.uint16).reshape(-1), u16p),
                     _ptr(v16.view(np.uint16).reshape(-1), u16p), NPOS, D, float(sm), _ptr(o, f32p), _ptr(l, f32p), _ptr(m, f32p))
    wo, wl, wm = attention_numpy(qh, k16, v16, np.float32(sm))
    assert np.allclose(o, wo, rtol=2e-6, atol=1e-7) and np.allclose(l, wl, rtol=2e-6, atol=2e-6) and np.allclose(m, wm, rtol=2e-6, atol=1e-7)
    # FP8: e4m3 bytes with per-row scales, query rows quantised per row (scale max|q|/448)
    kb = rng.integers(0, 256, (NPOS, D)).astype(np.uint8); vb = rng.integers(0, 256, (NPOS, D)).astype(np.uint8)
    kb[(kb & 0x7F) == 0x7F] = 0x3C; vb[(vb & 0x7F) == 0x7F] = 0x3C           # no NaN bytes
    ks = rng.uniform(0.001, 0.05, NPOS).astype(np.float32); vs = rng.uniform(0.001, 0.05, NPOS).astype(np.float32)
    q8 = np.zeros((G, D), np.uint8); qs = np.zeros(G, np.float32)
    L.orc_quantize_rows_e4m3(_ptr(qh.view(np.uint16).reshape(-1), u16p), G, D, _ptr(q8, u8p), _ptr(qs, f32p))
    f8 = lambda b: torch.from_numpy(np.ascontiguousarray(b)).view(torch.float8_e4m3fn).to(torch.float32).numpy().astype(np.float64)
    # the query quantisation itself, restated with torch's cast
    for r in range(G):
        qf = qh[r].astype(np.float32); s = np.float32(np.abs(qf).max()) / np.float32(448.0)
        assert np.float32(qs[r]).tobytes() == np.float32(s).tobytes()
        assert np.array_equal(q8[r], torch.from_numpy(np.clip(qf / s, -448, 448).astype(np.float32)).to(torch.float8_e4m3fn).view(torch.uint8).numpy())
    L.orc_attend_fp8(_ptr(q8, u8p), _ptr(qs, f32p), G, _ptr(kb, u8p), _ptr(ks, f32p), _ptr(vb, u8p), _ptr(vs, f32p), NPOS, D,
                     float(sm), _ptr(o, f32p), _ptr(l, f32p), _ptr(m, f32p))
    wo, wl, wm = attention_numpy(f8(q8) * qs[:, None].astype(np.float64), f8(kb) * ks[:, None].astype(np.float64),
                                 f8(vb) * vs[:, None].astype(np.float64), np.float32(sm))
    assert np.allclose(o, wo, rtol=2e-6, atol=1e-7) and np.allclose(l, wl, rtol=2e-6, atol=2e-6) and np.allclose(m, wm, rtol=2e-6, atol=1e-7)


# ---- MXFP4 (scheme 5): restated from the OCP Microscaling Formats (MX) v1.0 text alone --------------------------------
# Block of 32 elements, one E8M0 scale X = 2^(code - 127); elements FP4 E2M1 = {0, 0.5, 1, 1.5, 2, 3, 4, 6} with a sign bit.
# Conversion (spec section 6.3): shared exponent = floor(log2(max|x|)) - emax_elem (2 for E2M1, 8 for E4M3); elements are
# x / X rounded to nearest (ties to even), clamped to the largest magnitude.  Stated conventions beyond the spec's text
# (oracle/speckv_oracle.h): NaN inputs are skipped in the maximum and stored as +0, inf counts as 65504, an all-zero block
# gets code 0.  Which elements form a block is the format's own choice: the two 1024-element halves of a page are interleaved
# element by element (stream = e0, e1024, e1, e1025, ...) and the MX blocks are 32 consecutive elements of THAT stream; record =
# 1024 nibble bytes (stream element 2i in the low half of byte i, i.e. byte i = element i low, element 1024 + i high) then 64 codes.
E2M1_GRID = np.array([0.0, 0.5, 1.0, 1.5, 2.0, 3.0, 4.0, 6.0])


def e2m1_rne_numpy(v):
    """codes of float64 values, nearest grid point, ties to the even code, saturating; written as a search over the grid"""
    v = np.asarray(v, np.float64)
    a = np.minimum(np.abs(v), 6.0)
    d = np.abs(a[..., None] - E2M1_GRID)                       # distance to each of the 8 magnitudes
    best = d.min(axis=-1, keepdims=True)
    cand = d == best                                           # one candidate, or two on a tie (neighbouring codes)
    even = cand & (np.arange(8) % 2 == 0)
    code = np.where(cand.sum(-1) == 2, np.argmax(even, -1), np.argmax(cand, -1))
    return (code | np.where(np.signbit(v), 8, 0)).astype(np.uint8)


def mx_code_numpy(amax, emax_elem):
    amax = np.asarray(amax, np.float64)
    with np.errstate(divide="ignore"):
        e = np.floor(np.log2(np.where(amax > 0, amax, 1.0))).astype(np.int64)       # exact for these magnitudes (checked below)
    return np.where(amax > 0, np.clip(e - emax_elem + 127, 0, 254), 0).astype(np.uint8)


def mx_interleave(x):
    x = np.asarray(x).reshape(-1)
    return np.stack([x[:x.size // 2], x[x.size // 2:]], axis=1).reshape(-1)


def mx_deinterleave(s):
    s = np.asarray(s).reshape(-1, 2)
    return np.concatenate([s[:, 0], s[:, 1]])


def mxfp4_encode_numpy(x16):
    x = mx_interleave(np.asarray(x16, np.float16)).astype(np.float64).reshape(-1, 32)
    fin = np.where(np.isnan(x), 0.0, np.clip(x, -65504.0, 65504.0))
    code = mx_code_numpy(np.abs(fin).max(axis=1), 2)
    q = e2m1_rne_numpy(fin / np.exp2(code.astype(np.float64) - 127.0)[:, None])
    q = np.where(np.isnan(x), 0, q).reshape(-1).astype(np.uint8)
    return np.concatenate([(q[0::2] | (q[1::2] << 4)).astype(np.uint8), code])


def mxfp4_decode_numpy(rec):
    """float64 values of a record in ELEMENT order (not yet rounded to fp16)"""
    return mx_deinterleave(mxfp4_decode_stream_numpy(rec))


def mxfp4_decode_stream_numpy(rec):
    """float64 values of a record in stream (interleaved) order"""
    rec = np.asarray(rec, np.uint8)
    n = rec.size * 32 // 17                                     # 16 nibble bytes + 1 code per 32 elements
    nib = rec[:n // 2]
    q = np.empty(n, np.int64); q[0::2] = nib & 0xF; q[1::2] = nib >> 4
    val = E2M1_GRID[q & 7] * np.where(q & 8, -1.0, 1.0)
    codes = rec[n // 2:].astype(np.float64)
    scale = np.where(codes == 255, np.nan, np.exp2(codes - 127.0))
    return val * np.repeat(scale, 32)


def _mx_test_blocks():
    rng = np.random.default_rng(51)
    x = (rng.standard_normal((32, 2048)) * rng.uniform(0.01, 40.0, (32, 1))).astype(np.float16)
    x[3] = 0
    x[4, :32] = 0
    x[5, ::5] *= 50
    x[6] = np.float16(65504.0)
    x[7] = (rng.standard_normal(2048) * 6e-8).astype(np.float16)                    # fp16 subnormals
    x[8] = np.tile(np.array([0.25, 0.75, 1.25, 1.75, 2.5, 3.5, 5.0, 6.0, 7.0, 7.99, -0.25, -0.75, -1.25, -1.75, -2.5, -3.5,
                             -5.0, -7.0, 4.0, 0.5, 0.1, 0.24, 0.26, 0.74, 0.76, 1.24, 1.26, 2.49, 2.51, 4.99, 5.01, 0.0], np.float16), 64)   # every tie, with 7.99 fixing the scale at 1
    x[9] = x[8] * np.float16(2.0 ** -9)
    x[10, 7] = np.float16(np.inf); x[10, 40] = np.float16(-np.inf)
    x[11, 3] = np.float16(np.nan); x[11, 64:96] = np.float16(np.nan)               # a NaN among values, a group of NaNs only
    x[12] = np.float16(2.0 ** -24)                                                   # the smallest fp16 subnormal everywhere
    x[13] = (np.exp2(rng.integers(-20, 15, 2048)) * rng.choice([-1, 1], 2048) * rng.choice([1.0, 1.5, 1.25, 1.75], 2048)).astype(np.float16)
    return x


def test_mx_scalar_conversions_against_the_spec_text(oracle):
    L = oracle.lib
    # E2M1: every code decodes to the grid; every fp16 value rounds as the grid search says (ties to even, saturating)
    for c in range(16):
        assert L.orc_e2m1_to_f32(c) == E2M1_GRID[c & 7] * (-1 if c & 8 else 1)
    h = np.arange(65536, dtype=np.uint32).astype(np.uint16).view(np.float16)
    f = h.astype(np.float32)
    ok = ~np.isnan(f)
    got = np.array([L.orc_f32_to_e2m1(float(v)) for v in f[ok]], np.uint8)
    assert np.array_equal(got, e2m1_rne_numpy(f[ok].astype(np.float64)))
    assert L.orc_f32_to_e2m1(float("nan")) == 0
    assert [L.orc_f32_to_e2m1(v) for v in (0.25, 0.75, 1.25, 1.75, 2.5, 3.5, 5.0)] == [0, 2, 2, 4, 4, 6, 6]      # the seven ties
    # E8M0: 2^(code-127) bit for bit as torch's float8_e8m0fnu decodes it; 255 is NaN on both sides
    codes = np.arange(256, dtype=np.uint8)
    want = torch.from_numpy(codes).view(torch.float8_e8m0fnu).to(torch.float32).numpy()
    got = np.array([L.orc_e8m0_to_f32(int(c)) for c in codes], np.float32)
    assert np.isnan(got[255]) and np.isnan(want[255]) and np.array_equal(got[:255].view(np.uint32), want[:255].view(np.uint32))
    # shared exponent: floor(log2(amax)) - emax + 127 for every positive fp16 magnitude, both element types
    pos = f[(f > 0) & np.isfinite(f)]
    man, ex = np.frexp(pos.astype(np.float64))                 # independent of log2: amax = man * 2^ex, man in [0.5, 1)
    for emax in (2, 8):
        got = np.array([L.orc_mx_scale_code(float(v), emax) for v in pos], np.int64)
        assert np.array_equal(got, ex - 1 - emax + 127)
        assert np.array_equal(got, mx_code_numpy(pos, emax))
    assert L.orc_mx_scale_code(0.0, 2) == 0 and L.orc_mx_scale_code(float("inf"), 2) == 15 - 2 + 127


def test_mxfp4_block_format_against_the_numpy_restatement(oracle):
    x = _mx_test_blocks()
    scales, lens, recs = oracle.compress_blocks_f16(x, 5, 0)
    y = oracle.decompress_blocks_f16(recs, lens, scales, 5, 0)
    for i in range(x.shape[0]):
        want = mxfp4_encode_numpy(x[i])
        assert lens[i] == 1088 and want.size == 1088 and scales[i] == 1.0
        assert np.array_equal(recs[i, :1088], want), i
        with np.errstate(over="ignore", invalid="ignore"):
            assert np.array_equal(y[i].view(np.uint16), mxfp4_decode_numpy(want).astype(np.float16).view(np.uint16)), i
    # the largest element of every non-zero group lands on 4 or 6 (floor-type shared exponent: amax / X in [4, 8))
    dec = mxfp4_decode_stream_numpy(recs[0, :1088]).reshape(-1, 32)
    top = np.abs(dec).max(axis=1) / np.exp2(recs[0, 1024:1088].astype(np.float64) - 127)
    assert set(np.unique(top)) <= {4.0, 6.0}
    # a block really is 16 elements of the first half with the 16 matching elements of the second
    assert np.array_equal(mx_interleave(np.arange(2048)).reshape(-1, 32)[3], np.stack([np.arange(48, 64), np.arange(1072, 1088)], 1).reshape(-1))
    # quantisation error bound of the format: |x - y| <= X/2 * (grid step at |x|/X) and <= amax/4 when clamped (|x|/X in (6, 8))
    x0 = mx_interleave(x[0]).astype(np.float64).reshape(-1, 32); X = np.exp2(recs[0, 1024:1088].astype(np.float64) - 127)[:, None]
    assert np.all(np.abs(x0 - dec) <= np.where(np.abs(x0) / X > 6, 2.0, 1.0) * X + 1e-12)
    # every nibble value x a spread of codes decodes as described; a short record decodes to zeros, code 255 to NaN
    rec = np.zeros(1088, np.uint8)
    rec[:1024] = (np.arange(1024) * 37 + 11).astype(np.uint8)
    rec[1024:] = np.array([0, 1, 100, 101, 126, 127, 128, 140, 141, 150, 254, 255, 103, 110, 120, 130] * 4, np.uint8)
    got32 = np.stack([oracle.decompress_block_f32(rec, 1.0, 5, 0), oracle.decompress_block_f32(rec[:1087], 1.0, 5, 0)])
    want = mxfp4_decode_numpy(rec)
    nan = np.isnan(want)
    assert nan.sum() == 4 * 32 and np.array_equal(np.isnan(got32[0]), nan)
    with np.errstate(over="ignore"):
        w32 = want[~nan].astype(np.float32)                     # (6 x 2^127 overflows to inf on both sides)
    assert np.array_equal(got32[0][~nan].view(np.uint32), w32.view(np.uint32))      # fp32 output is exact
    assert not got32[1].any()
    got16 = oracle.decompress_block_f16(rec, 1.0, 5, 0)
    with np.errstate(over="ignore", invalid="ignore"):
        w16 = want.astype(np.float16)
    assert np.array_equal(got16[~nan].view(np.uint16), w16[~nan].view(np.uint16)) and np.isnan(got16[nan]).all()


def test_mxfp8_query_rows_and_the_mx4_attention_oracle_against_numpy(oracle):
    from oracle.bindings import _ptr, u8p, u16p, f32p
    L = oracle.lib
    rng = np.random.default_rng(52)
    G, D, NPOS = 8, 128, 300
    qh = (rng.standard_normal((G, D)) * rng.uniform(0.05, 30.0, (G, 1))).astype(np.float16)
    qh[1, :32] = 0; qh[2, 5] = np.float16(480.0); qh[3, 64:96] *= np.float16(2.0 ** -12)
    f8 = lambda b: torch.from_numpy(np.ascontiguousarray(b)).view(torch.float8_e4m3fn).to(torch.float32).numpy().astype(np.float64)
    for QB in (16, 32):
        q8 = np.zeros((G, D), np.uint8); qc = np.zeros((G, D // QB), np.uint8)
        L.orc_quantize_rows_mxfp8(_ptr(qh.view(np.uint16).reshape(-1), u16p), G, D, QB, _ptr(q8, u8p), _ptr(qc, u8p))
        # restated with torch's e4m3 cast: code = floor(log2 amax) - 8 + 127, bytes = e4m3(clip(x / 2^(code-127), +-448))
        xb = qh.astype(np.float64).reshape(G, D // QB, QB)
        code = mx_code_numpy(np.abs(xb).max(axis=2), 8)
        assert np.array_equal(qc, code)
        v = np.clip(xb / np.exp2(code.astype(np.float64) - 127)[..., None], -448.0, 448.0).astype(np.float32).reshape(G, D)
        assert np.array_equal(q8, torch.from_numpy(v).to(torch.float8_e4m3fn).view(torch.uint8).numpy())
        assert (np.abs(xb / np.exp2(code.astype(np.float64) - 127)[..., None]).max() > 448)        # the clamp was exercised (values in (448, 512))
    QB = 16
    qd = f8(q8) * 0                                              # (recomputed for the block size the attention uses)
    q8 = np.zeros((G, D), np.uint8); qc = np.zeros((G, D // QB), np.uint8)
    L.orc_quantize_rows_mxfp8(_ptr(qh.view(np.uint16).reshape(-1), u16p), G, D, QB, _ptr(q8, u8p), _ptr(qc, u8p))
    qd = f8(q8) * np.repeat(np.exp2(qc.astype(np.float64) - 127), QB, axis=1)
    # K / V page rows of ONE kv head out of 8: a page = [2 positions][8 heads][128]; the record's bytes 128 h .. 128 h + 127 and
    # codes 8 h .. 8 h + 7 belong to head h
    n_pages = NPOS // 2
    kp = (rng.standard_normal((n_pages, 2048)) * rng.uniform(0.2, 4.0, (n_pages, 1))).astype(np.float16)
    vp = (rng.standard_normal((n_pages, 2048)) * rng.uniform(0.2, 4.0, (n_pages, 1))).astype(np.float16)
    head = 5
    def rows_of(pages):
        recs = np.stack([mxfp4_encode_numpy(p) for p in pages])
        dec = np.stack([mxfp4_decode_numpy(r) for r in recs]).reshape(n_pages, 2, 8, D)[:, :, head, :].reshape(NPOS, D)
        return np.ascontiguousarray(recs[:, 128 * head:128 * head + 128]), np.ascontiguousarray(recs[:, 1024 + 8 * head:1024 + 8 * head + 8]), dec
    kr, kc, kdec = rows_of(kp)
    vr, vc, vdec = rows_of(vp)
    o = np.zeros((G, D), np.float32); l = np.zeros(G, np.float32); m = np.zeros((G, D), np.float32)
    sm = 1.0 / np.sqrt(D)
    for npos in (NPOS, NPOS - 1, 2, 1):
        L.orc_attend_mx4(_ptr(q8, u8p), _ptr(qc, u8p), QB, G, _ptr(kr, u8p), _ptr(kc, u8p), _ptr(vr, u8p), _ptr(vc, u8p), npos, D, float(sm),
                         _ptr(o, f32p), _ptr(l, f32p), _ptr(m, f32p))
        wo, wl, wm = attention_numpy(qd, kdec[:npos], vdec[:npos], np.float32(sm))
        assert np.allclose(o, wo, rtol=2e-6, atol=1e-7) and np.allclose(l, wl, rtol=2e-6, atol=2e-6) and np.allclose(m, wm, rtol=2e-6, atol=1e-7)
