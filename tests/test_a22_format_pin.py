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
