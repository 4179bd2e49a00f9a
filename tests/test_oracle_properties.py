"""Property-based pinning of the oracle (hypothesis): arbitrary inputs, not just the
hand-picked ones.  Against the reference build where it exists (dev container), and
codec laws that must hold anywhere."""
import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings
from hypothesis import strategies as st
from hypothesis.extra import numpy as hnp

COMMON = dict(deadline=None, suppress_health_check=[HealthCheck.function_scoped_fixture, HealthCheck.too_slow])

finite_f32 = hnp.arrays(np.float32, st.integers(1, 600),
                        elements=st.floats(-1e6, 1e6, width=32, allow_nan=False, allow_infinity=False))
any_f32 = hnp.arrays(np.float32, st.integers(1, 300), elements=st.floats(width=32, allow_nan=True, allow_infinity=True))
byte_streams = hnp.arrays(np.uint8, st.integers(0, 400))


@settings(max_examples=150, **COMMON)
@given(x=finite_f32)
def test_compress_equals_reference(oracle, reference, x):
    s_r, rle_r = reference.compress_f32(x)
    s_o, rle_o = oracle.compress_f32(x, mode=0)
    assert s_r.tobytes() == s_o.tobytes() and rle_r.tobytes() == rle_o.tobytes()


@settings(max_examples=100, **COMMON)
@given(x=any_f32)
def test_compress_equals_reference_nonfinite(oracle, reference, x):
    s_r, rle_r = reference.compress_f32(x)
    s_o, rle_o = oracle.compress_f32(x, mode=0)
    assert rle_r.tobytes() == rle_o.tobytes()
    assert (np.isnan(s_r) and np.isnan(s_o)) or s_r.tobytes() == s_o.tobytes()


@settings(max_examples=200, **COMMON)
@given(rle=byte_streams, scale=st.floats(2.0 ** -20, 1024.0, width=32))
def test_decode_arbitrary_streams_equals_reference(oracle, reference, rle, scale):
    cap = int(rle[1::2].astype(np.int64).sum()) if rle.size >= 2 else 0
    y_r = reference.decompress_f32(rle, scale, cap=cap)
    y_o = oracle.decompress_f32(rle, scale, mode=0, cap=cap)
    assert y_r.tobytes() == y_o.tobytes()


@settings(max_examples=150, **COMMON)
@given(x=finite_f32, mode=st.sampled_from([0, 1]))
def test_rle_laws(oracle, x, mode):
    """Format laws (anywhere): counts in 1..255 sum to n, no two neighbouring pairs carry
    the same value unless the first is a full 255-run, decode(encode) has n elements."""
    s, rle = oracle.compress_f32(x, mode=mode)
    pairs = rle.reshape(-1, 2)
    assert int(pairs[:, 1].astype(np.int64).sum()) == x.size
    assert (pairs[:, 1] >= 1).all()
    same = pairs[1:, 0] == pairs[:-1, 0]
    assert (pairs[:-1, 1][same] == 255).all()
    y = oracle.decompress_f32(rle, s, mode=mode)
    assert y.size == x.size
    if mode == 1 and np.abs(x).max() > 0:                      # INTENT: a real quantiser
        assert np.abs(y - x).max() <= float(s) * 0.5 * 1.0001 + 1e-30


@settings(max_examples=100, **COMMON)
@given(h=st.integers(1, 2**20), i=st.integers(0, 2**19), off=st.integers(0, 4095))
def test_page_id_arithmetic(oracle, h, i, off):
    O = oracle.lib
    assert O.orc_virt_page_id(h, i) == ((h << 32) | (i << 12)) & (2**64 - 1)
    assert O.orc_phys_page_id(h, i) == 0x4000000000 + (h << 20) + (i << 12)
    assert O.orc_desc_gpu_addr(O.orc_virt_page_id(h, i)) == 0x8000000000 + (O.orc_virt_page_id(h, i) & 0xFFFFFFFFFFFF)
