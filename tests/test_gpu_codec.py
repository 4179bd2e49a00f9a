"""-m gpu: the HIP codec kernels against the oracle, through the C ABI
(speckv_ext_codec_compress / speckv_ext_codec_decompress).

Bar: compressed bytes, record lengths and scale bits identical; decoded fp32 and
fp16 bit-identical (NaN payloads excepted) in both quantiser modes."""
import os

import numpy as np
import pytest

from tests._gpu import N, assert_same_float_bits, gpu_compress, gpu_decompress, load_debug_lib, load_raw_lib, stream_ptr, torch_mod, set_tuning

pytestmark = pytest.mark.gpu

SCHEMES = [0, 1, 2]
MODES = [0, 1]


def make_blocks(seed=1234):
    rng = np.random.default_rng(seed)
    blocks = [
        rng.standard_normal(N),                                   # gaussian
        rng.standard_normal(N) * 3.7,
        np.zeros(N),                                              # all zero -> runs of 255
        np.repeat(rng.standard_normal(N // 32), 32),              # piecewise constant
        np.full(N, 0.37),
        np.linspace(-3, 3, N),
        np.where(rng.random(N) < 0.02, rng.standard_normal(N), 0),  # sparse
        rng.standard_normal(N) * 1e-3,
        rng.standard_normal(N) * 6e-6,                            # fp16 subnormals
        np.resize(np.array([65504, -65504, 6.1e-5, 5.96e-8, -5.96e-8, 0, -0.0, 1, -1]), N),
        np.concatenate([np.zeros(255), [1.0], np.zeros(510), [2.0], np.zeros(N - 767)]),   # run splits at 255
        np.concatenate([np.full(256, 1.0), np.full(N - 256, -1.0)]),
        np.resize(np.array([1.0, 1.0, 2.0]), N),
        np.arange(N) % 7 - 3.0,
    ]
    x = np.stack(blocks)
    with np.errstate(over="ignore"):
        x16 = x.astype(np.float16)
    # inf / nan edge cases (cache_engine.cpp:176-180,190-191 through cvttss2si)
    e1 = x16[0].copy(); e1[5] = np.inf
    e2 = x16[0].copy(); e2[7] = np.nan
    e3 = x16[0].copy(); e3[0] = -np.inf; e3[100] = np.nan
    return np.concatenate([x16, np.stack([e1, e2, e3])])


@pytest.fixture(scope="module")
def lib():
    return load_raw_lib()


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("scheme", SCHEMES)
def test_compress_matches_oracle(lib, oracle, scheme, mode):
    x16 = make_blocks()
    scales, lens, recs = gpu_compress(lib, x16, scheme, mode)
    o_scales, o_lens, o_recs = oracle.compress_blocks_f16(x16, scheme, mode)
    assert np.array_equal(lens, o_lens), (lens, o_lens)
    if scheme != 0:
        nan = np.isnan(o_scales)
        assert np.array_equal(np.isnan(scales), nan)
        assert scales[~nan].tobytes() == o_scales[~nan].tobytes()
    for b in range(x16.shape[0]):
        assert recs[b, :lens[b]].tobytes() == o_recs[b, :lens[b]].tobytes(), f"block {b}"


@pytest.mark.parametrize("out_f32", [False, True])
@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("scheme", SCHEMES)
def test_decompress_matches_oracle(lib, oracle, scheme, mode, out_f32):
    x16 = make_blocks()
    o_scales, o_lens, o_recs = oracle.compress_blocks_f16(x16, scheme, mode)
    y = gpu_decompress(lib, o_recs, o_lens, o_scales, scheme, mode, out_f32)
    for b in range(x16.shape[0]):
        if out_f32:
            want = np.zeros(N, np.float32)
            got = oracle.decompress_block_f32(o_recs[b, :o_lens[b]], o_scales[b], scheme, mode, N)
        else:
            want = np.zeros(N, np.float16)
            got = oracle.decompress_block_f16(o_recs[b, :o_lens[b]], o_scales[b], scheme, mode, N)
        want[:got.size] = got
        assert_same_float_bits(y[b], want, f"scheme {scheme} mode {mode} block {b}")


@pytest.mark.parametrize("scheme", [3, 4])
def test_extension_formats_match_oracle(lib, oracle, scheme):
    """INT4_G32 / FP8_E4M3 (BASELINE config 5; no reference counterpart, parity is
    against oracle/ only): record bytes and decoded fp16/fp32 bit-identical."""
    x16 = make_blocks()
    rng = np.random.default_rng(77)
    extra = np.stack([rng.standard_normal(N) * 10.0 ** rng.integers(-4, 3), rng.standard_normal(N) * 300,
                      np.where(rng.random(N) < 0.5, 0, rng.standard_normal(N)),
                      np.repeat(rng.standard_normal(N // 32) * np.array([1e-6, 1, 100, 6e4] * (N // 128)), 32)])
    with np.errstate(over="ignore"):
        x16 = np.concatenate([x16, extra.astype(np.float16)])
    finite = np.isfinite(x16.astype(np.float32)).all(axis=1)     # inf/nan blocks: covered separately below
    scales, lens, recs = gpu_compress(lib, x16, scheme, 0)
    o_scales, o_lens, o_recs = oracle.compress_blocks_f16(x16, scheme, 0)
    assert np.array_equal(lens, o_lens) and (lens == (1152 if scheme == 3 else 2048)).all()
    for b in np.flatnonzero(finite):
        assert recs[b, :lens[b]].tobytes() == o_recs[b, :lens[b]].tobytes(), f"block {b}"
        if scheme == 4:
            assert scales[b].tobytes() == o_scales[b].tobytes()
    for out_f32 in (False, True):
        y = gpu_decompress(lib, o_recs, o_lens, o_scales, scheme, 0, out_f32)
        for b in np.flatnonzero(finite):
            if out_f32:
                want = oracle.decompress_block_f32(o_recs[b, :o_lens[b]], o_scales[b], scheme, 0, N)
            else:
                want = oracle.decompress_block_f16(o_recs[b, :o_lens[b]], o_scales[b], scheme, 0, N)
            assert_same_float_bits(y[b], want, f"scheme {scheme} f32={out_f32} block {b}")
    # every e4m3 byte / every nibble decodes like the oracle
    if scheme == 4:
        rec = np.resize(np.arange(256, dtype=np.uint8), (1, 4096)); ln = np.array([2048], np.uint32)
        y = gpu_decompress(lib, rec, ln, np.array([0.5], np.float32), 4, 0, True)[0]
        want = oracle.decompress_block_f32(rec[0, :2048], 0.5, 4, 0, N)
        assert_same_float_bits(y, want, "all e4m3 bytes")
    # round-trip error bounds (size-independent property)
    y = gpu_decompress(lib, recs, lens, scales, scheme, 0, True)
    xf = x16.astype(np.float32)
    for b in np.flatnonzero(finite):
        if scheme == 3:
            g = np.abs(xf[b]).reshape(-1, 32).max(axis=1)
            bound = np.repeat(g / 7 * 0.5 * 1.01 + g * 2 ** -10, 32) + 1e-12
        else:
            bound = np.abs(xf[b]) * 2 ** -4 + np.abs(xf[b]).max() / 448 * 2 ** -10 + 1e-12
        assert (np.abs(y[b] - xf[b]) <= bound).all(), f"block {b}"
    # short records decode to zeros (INT4) / a zero tail (FP8)
    lens2 = lens.copy(); lens2[0] = 100
    y = gpu_decompress(lib, recs, lens2, scales, scheme, 0, True)
    want = np.zeros(N, np.float32)
    got = oracle.decompress_block_f32(recs[0, :100], scales[0], scheme, 0, N)
    want[:got.size] = got
    assert_same_float_bits(y[0], want, "short record")


def test_mxfp4_block_format_matches_oracle(lib, oracle):
    """Scheme 5, MXFP4 (OCP MX v1.0: E2M1 elements, one E8M0 scale per 32; BASELINE configs[4] "4:1 ratio"; no reference
    counterpart: the oracle is pinned to a numpy restatement of the spec text by tests/test_a22_format_pin.py): record bytes
    and decoded fp16 / fp32 bit-identical, INCLUDING blocks with inf / NaN (NaN -> +0, inf -> 65504 in the maximum)."""
    x16 = make_blocks()
    rng = np.random.default_rng(78)
    ties = np.tile(np.array([0.25, 0.75, 1.25, 1.75, 2.5, 3.5, 5.0, 6.0, 7.0, 7.99, -0.25, -0.75, -1.25, -1.75, -2.5, -3.5,
                             -5.0, -7.0, 4.0, 0.5, 0.1, 0.24, 0.26, 0.74, 0.76, 1.24, 1.26, 2.49, 2.51, 4.99, 5.01, -0.0]), N // 32)
    extra = np.stack([rng.standard_normal(N) * 10.0 ** rng.integers(-4, 3), rng.standard_normal(N) * 300,
                      np.where(rng.random(N) < 0.5, 0, rng.standard_normal(N)),
                      np.repeat(rng.standard_normal(N // 32) * np.array([1e-6, 1, 100, 6e4] * (N // 128)), 32),
                      ties, ties * 2.0 ** -9, ties * 2.0 ** 12, ties * 2.0 ** -22,
                      np.exp2(rng.integers(-24, 16, N)) * rng.choice([-1.0, 1.0], N) * rng.choice([1.0, 1.25, 1.5, 1.75], N),
                      np.full(N, 2.0 ** -24), np.full(N, -65504.0)])
    with np.errstate(over="ignore"):
        x16 = np.concatenate([x16, extra.astype(np.float16)])
    e = x16[0].copy(); e[64:96] = np.nan; e[200] = np.inf; e[201] = -np.inf; e[300:332] = np.inf       # a group of NaNs only, a group of infs only
    x16 = np.concatenate([x16, e[None]])
    scales, lens, recs = gpu_compress(lib, x16, 5, 0)
    o_scales, o_lens, o_recs = oracle.compress_blocks_f16(x16, 5, 0)
    assert np.array_equal(lens, o_lens) and (lens == 1088).all() and (scales == 1.0).all()
    for b in range(x16.shape[0]):
        assert recs[b, :1088].tobytes() == o_recs[b, :1088].tobytes(), f"block {b}: first byte {np.flatnonzero(recs[b, :1088] != o_recs[b, :1088])[:4]}"
    for out_f32 in (False, True):
        y = gpu_decompress(lib, o_recs, o_lens, o_scales, 5, 0, out_f32)
        for b in range(x16.shape[0]):
            want = (oracle.decompress_block_f32 if out_f32 else oracle.decompress_block_f16)(o_recs[b, :1088], 1.0, 5, 0, N)
            assert_same_float_bits(y[b], want, f"mxfp4 f32={out_f32} block {b}")
    # every nibble under a spread of codes (0, 1: subnormal floats; 254: overflow of 6 x 2^127 to inf; 255: NaN)
    rec = np.zeros((4, 4096), np.uint8)
    rec[:, :1024] = (np.arange(1024) * 37 + 11).astype(np.uint8)
    rec[0, 1024:1088] = np.array([0, 1, 100, 101, 126, 127, 128, 140, 141, 150, 254, 255, 103, 110, 120, 130] * 4, np.uint8)
    rec[1, 1024:1088] = np.arange(64) + 96
    rec[2, 1024:1088] = np.arange(64) * 4
    rec[3, 1024:1088] = 255 - np.arange(64)
    ln = np.full(4, 1088, np.uint32)
    for out_f32 in (False, True):
        y = gpu_decompress(lib, rec, ln, np.ones(4, np.float32), 5, 0, out_f32)
        for b in range(4):
            want = (oracle.decompress_block_f32 if out_f32 else oracle.decompress_block_f16)(rec[b, :1088], 1.0, 5, 0, N)
            assert_same_float_bits(y[b], want, f"all nibbles f32={out_f32} row {b}")
    # round trip (size-independent): |x - y| <= half a grid step of the group, a whole one where the format clamps (|x| / X in (6, 8))
    y = gpu_decompress(lib, recs, lens, scales, 5, 0, True)
    xf = x16.astype(np.float32)
    ilv = lambda v: np.stack([v[:N // 2], v[N // 2:]], axis=1).reshape(-1)      # the format's stream order: halves interleaved, blocks of 32
    for b in np.flatnonzero(np.isfinite(xf).all(axis=1)):
        X = np.repeat(np.exp2(recs[b, 1024:1088].astype(np.float64) - 127), 32)
        xs, ys = ilv(xf[b]), ilv(y[b])
        bound = X * np.where(np.abs(xs) / X > 6, 2.0, 1.0)
        assert (np.abs(ys - xs) <= bound + 1e-30).all(), f"block {b}"
    # a short record decodes to zeros
    lens2 = lens.copy(); lens2[0] = 1087
    y = gpu_decompress(lib, recs, lens2, scales, 5, 0, True)
    assert not y[0].any() and y[1].any()


def test_golden_reference_vectors(lib, golden_dir):
    """Reference-generated vectors (tests/golden/codec_vectors.npz): the HIP path
    reproduces the reference's bytes and fp32 outputs without the oracle in between."""
    g = np.load(os.path.join(golden_dir, "codec_vectors.npz"))
    names = [k[:-2] for k in g.files if k.endswith(".x") and g[k].size == N]
    assert len(names) >= 8
    x16 = np.stack([g[f"{n}.x"] for n in names]).astype(np.float16)
    scales, lens, recs = gpu_compress(lib, x16, 2, 0)
    for i, n in enumerate(names):
        assert scales[i].tobytes() == g[f"{n}.scale"][0].tobytes(), n
        assert recs[i, :lens[i]].tobytes() == g[f"{n}.rle"].tobytes(), n
    y = gpu_decompress(lib, recs, lens, scales, 2, 0, out_f32=True)
    for i, n in enumerate(names):
        assert_same_float_bits(y[i], g[f"{n}.y"], n)


def test_decode_malformed_and_ragged(lib, oracle):
    """Decoder edge cases of cache_engine.cpp:241-258: odd trailing byte, zero
    counts, counts > 127, short / empty / overlong streams."""
    rng = np.random.default_rng(5)
    streams = [
        [5, 3, 7],                       # odd tail dropped
        [5, 0, 9, 2],                    # count 0
        [255, 200, 1, 255, 128, 1],
        [1],
        [],
        [3, 255] * 8 + [4, 8],           # exactly 2048
        [3, 255] * 9,                    # overlong: clipped at the block
        list(rng.integers(0, 256, 4096)),  # random pairs, random counts (sum >> 2048)
        [b for _ in range(2048) for b in (int(rng.integers(0, 256)), 1)],   # 2048 runs of 1
        [7, 0] * 100 + [9, 5],           # many zero-count pairs before data
    ]
    recs = np.zeros((len(streams), 4096), np.uint8)
    lens = np.zeros(len(streams), np.uint32)
    for i, s in enumerate(streams):
        recs[i, :len(s)] = s
        lens[i] = len(s)
    scales = np.full(len(streams), 0.5, np.float32)
    for mode in MODES:
        y = gpu_decompress(lib, recs, lens, scales, 2, mode, out_f32=True)
        for i in range(len(streams)):
            want = np.zeros(N, np.float32)
            got = oracle.decompress_block_f32(recs[i, :lens[i]], 0.5, 2, mode, N)
            want[:got.size] = got
            assert_same_float_bits(y[i], want, f"stream {i} mode {mode}")
    # streams whose value bytes are all zero take a shortcut (plain stores of +0) when the scale is finite and >= 0:
    # short, exact, overlong, empty, with zero counts -- and the scales for which 0/127 * scale is NOT +0 (negative: -0,
    # infinite or NaN: NaN) must still come out as the reference computes them
    zstreams = [[0, 255] * 8 + [0, 8], [0, 100], [], [0, 255] * 12, [0, 0, 0, 7, 0, 0], [0, 1] * 2048, [0, 255] * 8 + [0, 8, 3]]
    zscales = [0.5, 0.0, -0.5, float("inf"), float("nan"), -0.0, 3.0e38]
    recs = np.zeros((len(zstreams) * len(zscales), 4096), np.uint8)
    lens = np.zeros(recs.shape[0], np.uint32)
    scales = np.zeros(recs.shape[0], np.float32)
    for i, st in enumerate(zstreams):
        for j, sc in enumerate(zscales):
            k = i * len(zscales) + j
            recs[k, :len(st)] = st; lens[k] = len(st); scales[k] = sc
    for mode in MODES:
        for f32 in (True, False):
            y = gpu_decompress(lib, recs, lens, scales, 2, mode, out_f32=f32)
            for k in range(recs.shape[0]):
                dec = oracle.decompress_block_f32 if f32 else oracle.decompress_block_f16
                want = np.zeros(N, np.float32 if f32 else np.float16)
                got = dec(recs[k, :lens[k]], float(scales[k]), 2, mode, N)
                want[:got.size] = got
                assert_same_float_bits(y[k], want, f"zero-value stream {k // len(zscales)} scale {scales[k]!r} mode {mode} f32 {f32}")
    # ragged INT8 / FP16 records
    for scheme, ls in ((1, [0, 1, 7, 8, 9, 2047, 2048]), (0, [0, 2, 3, 14, 16, 18, 4094, 4096])):
        recs = rng.integers(0, 256, (len(ls), 4096)).astype(np.uint8)
        if scheme == 0:
            recs = (rng.standard_normal((len(ls), N)).astype(np.float16)).view(np.uint8).reshape(len(ls), 4096)
        lens = np.array(ls, np.uint32)
        scales = np.full(len(ls), 0.25, np.float32)
        y = gpu_decompress(lib, recs, lens, scales, scheme, 0, out_f32=False)
        for i in range(len(ls)):
            want = np.zeros(N, np.float16)
            got = oracle.decompress_block_f16(recs[i, :lens[i]], 0.25, scheme, 0, N)
            want[:got.size] = got
            assert_same_float_bits(y[i], want, f"scheme {scheme} len {ls[i]}")


def test_many_random_blocks(lib, oracle):
    """2048 seeded blocks with mixed statistics (1/4 smooth, 1/4 sparse)."""
    rng = np.random.default_rng(2001)
    B = 2048
    x = rng.standard_normal((B, N))
    x[: B // 4] = np.repeat(rng.standard_normal((B // 4, N // 16)), 16, axis=1)
    x[B // 4: B // 2] *= rng.random((B // 4, N)) < 0.05
    x16 = x.astype(np.float16)
    for mode in MODES:
        scales, lens, recs = gpu_compress(lib, x16, 2, mode)
        o_scales, o_lens, o_recs = oracle.compress_blocks_f16(x16, 2, mode)
        assert np.array_equal(lens, o_lens)
        assert scales.tobytes() == o_scales.tobytes()
        mask = np.arange(4096)[None, :] < lens[:, None]
        assert np.array_equal(recs[mask], o_recs[mask])
        y = gpu_decompress(lib, recs, lens, scales, 2, mode)
        want = oracle.decompress_blocks_f16(o_recs, o_lens, o_scales, 2, mode)
        assert_same_float_bits(y, want, f"mode {mode}")


def test_run_lengths_around_the_split_limit(lib, oracle):
    """The RLE encoder keeps its fast path while every run stays under the 255 limit (bounded by a per-lane estimate
    that is up to 7 elements pessimistic) and hands longer stretches to the run splitter (cache_engine.cpp:224: a run
    closes at count 255).  Piecewise-constant blocks whose zero-delta stretches are 230..270 long, at every alignment
    against the 8-element lanes and the 512-element chunks, plus stretches that begin in one chunk and end in the
    next, pin that boundary byte for byte; all-zero blocks with a single spike pin the direct zero-record path."""
    rng = np.random.default_rng(424242)
    blocks = []
    for seg in range(230, 271):
        for shift in (0, 1, 3, 7, 8, 129, 255, 500):
            levels = rng.standard_normal((N + 600) // seg + 3)
            x = np.repeat(levels, seg)[shift:shift + N]
            blocks.append(x)
    for start in (0, 1, 300, 511, 512, 1000, 1790, 1800):            # one long stretch inside noise, crossing chunk borders
        for length in (240, 247, 248, 249, 254, 255, 256, 257, 300, 520):
            x = rng.standard_normal(N)
            x[start:start + length] = x[start]
            blocks.append(x)
    for pos in (0, 1, 254, 255, 256, 1023, 2047):                    # zeros with one spike; and exact zeros
        x = np.zeros(N); x[pos] = 1.0
        blocks.append(x)
    blocks.append(np.zeros(N)); blocks.append(-np.zeros(N))
    x16 = np.stack(blocks).astype(np.float16)
    for mode in MODES:
        scales, lens, recs = gpu_compress(lib, x16, 2, mode)
        o_scales, o_lens, o_recs = oracle.compress_blocks_f16(x16, 2, mode)
        assert np.array_equal(lens, o_lens), np.nonzero(lens != o_lens)[0][:8]
        assert scales.tobytes() == o_scales.tobytes()
        mask = np.arange(4096)[None, :] < lens[:, None]
        bad = np.nonzero((recs != o_recs) & mask)
        assert bad[0].size == 0, (mode, bad[0][:5], bad[1][:5])
        y = gpu_decompress(lib, recs, lens, scales, 2, mode)
        assert_same_float_bits(y, oracle.decompress_blocks_f16(o_recs, o_lens, o_scales, 2, mode), f"mode {mode}")


def test_long_stretches_take_the_split_form_of_the_fast_encoder(lib, oracle):
    """Blocks made of LONG stretches of equal deltas (the data the scheme compresses 100 : 1): the fast encoder's SPLIT form
    starts a run at every 255th element of a stretch, across lanes and 512-element chunks.  Random piecewise-constant blocks
    with stretch lengths around every multiple of 255 and up to the whole block, at random phases; stretches of a constant
    NON-ZERO delta (ramps whose quantised steps are equal); whole-block constants; long stretches between bursts of noise --
    3 000 blocks, both quantiser modes, records byte for byte against the oracle, then decoded."""
    rng = np.random.default_rng(20250404)
    blocks = []
    special = [254, 255, 256, 509, 510, 511, 764, 765, 766, 1019, 1020, 1021, 1275, 1530, 1785, 2040, 2047, 2048]
    for i in range(2000):
        x = np.empty(N + 4096)
        pos = 0
        while pos < x.size:
            kind = rng.integers(0, 4)
            if kind == 0: m = int(rng.choice(special)) + int(rng.integers(-1, 2))
            elif kind == 1: m = int(rng.integers(256, 2300))
            elif kind == 2: m = int(rng.integers(1, 40))
            else: m = int(rng.integers(40, 256))
            x[pos:pos + m] = rng.standard_normal()
            pos += m
        ph = int(rng.integers(0, 4096))
        blocks.append(x[ph:ph + N])
    for i in range(500):                                             # ramps: equal quantised steps over hundreds of elements
        step = rng.choice([1.0, 2.0, 3.0, -1.0, -2.0]) / 127.0
        x = np.zeros(N)
        a = int(rng.integers(0, 600)); b = int(rng.integers(a + 60, a + 127))
        x[a:b] = np.arange(b - a) * step                              # |x| < 1
        x[0] = 1.0                                                    # pins the scale: max|x| = 1 -> q = round(127 x)
        x[b:] = x[b - 1] if i % 2 else 0.0
        blocks.append(x)
    for i in range(250):                                             # constants, and long stretches between noise
        x = np.full(N, rng.standard_normal())
        if i % 3 == 0:
            k = int(rng.integers(1, 6))
            for _ in range(k):
                a = int(rng.integers(0, N - 8)); x[a:a + int(rng.integers(1, 8))] = rng.standard_normal(1)
        blocks.append(x)
    for i in range(250):
        x = rng.standard_normal(N)
        a = int(rng.integers(0, N - 300)); m = int(rng.integers(250, N - a))
        x[a:a + m] = x[a]
        blocks.append(x)
    x16 = np.stack(blocks).astype(np.float16)
    for mode in MODES:
        scales, lens, recs = gpu_compress(lib, x16, 2, mode)
        o_scales, o_lens, o_recs = oracle.compress_blocks_f16(x16, 2, mode)
        assert np.array_equal(lens, o_lens), np.nonzero(lens != o_lens)[0][:8]
        assert scales.tobytes() == o_scales.tobytes()
        mask = np.arange(4096)[None, :] < lens[:, None]
        bad = np.nonzero((recs != o_recs) & mask)
        assert bad[0].size == 0, (mode, bad[0][:5], bad[1][:5])
        y = gpu_decompress(lib, recs, lens, scales, 2, mode)
        assert_same_float_bits(y, oracle.decompress_blocks_f16(o_recs, o_lens, o_scales, 2, mode), f"mode {mode}")


@pytest.mark.parametrize("mode", [0, 1])
def test_structured_hint_decoder_is_bit_identical(lib, oracle, mode):
    """SPECKV_CODEC_HINT_STRUCTURED selects a separate instantiation of the fetch kernel (decode_rle_fast<.., FLAT>: constant
    runs on 8-element boundaries skip the per-element recurrences).  Its output must be the oracle's, bit for bit, on the data
    it is meant for (runs of 32, of 8, of 64; zeros) AND on everything else a caller may wrongly hint: runs off the 8-element
    grid, runs of 7, noise, short malformed records -- fp16 and fp32 outputs."""
    rng = np.random.default_rng(404)
    blocks = []
    for run in (32, 8, 16, 64, 128, 256, 2048):
        blocks.append(np.repeat(rng.standard_normal(N // run), run))
    for run in (16, 32):                                                       # large jumps: the int8 prefix wraps many times inside a chunk
        blocks.append(np.repeat(rng.choice([-3.0, 3.0, -2.5, 2.9], N // run) + 0.01 * rng.standard_normal(N // run), run))
    for chunk in range(4):                                                     # flat everywhere but in one 512-element chunk
        b = np.repeat(rng.standard_normal(N // 32), 32); b[512 * chunk + 100:512 * chunk + 103] = (0.7, -0.4, 0.2)
        blocks.append(b)
    blocks.append(np.zeros(N))
    off = np.repeat(rng.standard_normal(N // 32 + 1), 32)[5:5 + N]            # runs of 32 that start 5 elements late
    blocks.append(off)
    blocks.append(np.repeat(rng.standard_normal(N // 7 + 1), 7)[:N])           # runs of 7: never on the grid
    mixed = np.repeat(rng.standard_normal(N // 32), 32); mixed[700:760] = rng.standard_normal(60)   # mostly flat, one noisy stretch
    blocks.append(mixed)
    blocks.append(rng.standard_normal(N))                                      # does not compress: the hint is simply wrong
    x = np.stack(blocks).astype(np.float16)
    scales, lens, recs = oracle.compress_blocks_f16(x, 2, mode)
    want16 = oracle.decompress_blocks_f16(recs, lens, scales, 2, mode)
    HINT = 0x100
    for out_f32 in (False, True):
        plain = gpu_decompress(lib, recs, lens, scales, 2, mode, out_f32)
        hinted = gpu_decompress(lib, recs, lens, scales, 2, mode | HINT, out_f32)
        assert_same_float_bits(hinted, plain, f"hint vs plain, f32={out_f32}")
        if not out_f32:
            assert_same_float_bits(hinted, want16, "hint vs oracle")
    # the hint on other schemes is ignored, not an error
    s1, l1, r1 = oracle.compress_blocks_f16(x, 1, mode)
    assert_same_float_bits(gpu_decompress(lib, r1[:, :2048], l1, s1, 1, mode | HINT), gpu_decompress(lib, r1[:, :2048], l1, s1, 1, mode), "int8")


def test_full_size_roundtrip_properties(lib):
    """BASELINE config 2 size (131072 blocks = 512 MiB fp16): size-independent
    properties instead of the oracle -- decode(encode(x)) is a fixed point of a
    second encode/decode pass (idempotence), INTENT-mode error is bounded by
    scale/2, FP16 scheme is the identity."""
    import torch
    B = 131072
    g = torch.Generator(device="cuda"); g.manual_seed(2001)
    x = torch.randn((B, N), generator=g, device="cuda", dtype=torch.float32).to(torch.float16)
    recs = torch.empty((B, 4096), dtype=torch.uint8, device="cuda")
    lens = torch.empty(B, dtype=torch.int32, device="cuda")
    scales = torch.empty(B, dtype=torch.float32, device="cuda")
    y = torch.empty_like(x); y2 = torch.empty_like(x)
    s = torch.cuda.current_stream().cuda_stream
    for scheme, mode in ((0, 0), (1, 1), (2, 1), (2, 0)):
        assert lib.speckv_ext_codec_compress(x.data_ptr(), B, recs.data_ptr(), 4096, lens.data_ptr(), scales.data_ptr(), scheme, mode, s) == 0
        assert lib.speckv_ext_codec_decompress(recs.data_ptr(), 4096, lens.data_ptr(), scales.data_ptr(), B, y.data_ptr(), 0, scheme, mode, s) == 0
        torch.cuda.synchronize()
        if scheme == 0:
            assert torch.equal(y.view(torch.int16), x.view(torch.int16))
            continue
        if mode == 1:
            err = (y.float() - x.float()).abs()
            bound = scales[:, None] * 0.5 + (y.float().abs() * 2 ** -11) + 1e-7
            assert bool((err <= bound).all())
            # idempotence: the INTENT quantiser is a projection (scale may shrink by < 1 ulp of fp16 max)
            lens2 = torch.empty_like(lens); scales2 = torch.empty_like(scales)
            assert lib.speckv_ext_codec_compress(y.data_ptr(), B, recs.data_ptr(), 4096, lens2.data_ptr(), scales2.data_ptr(), scheme, mode, s) == 0
            assert lib.speckv_ext_codec_decompress(recs.data_ptr(), 4096, lens2.data_ptr(), scales2.data_ptr(), B, y2.data_ptr(), 0, scheme, mode, s) == 0
            torch.cuda.synchronize()
            err2 = (y2.float() - y.float()).abs()
            assert bool((err2 <= bound).all())
        if scheme == 2:
            # record length law: every pair is 2 bytes, counts sum to 2048
            l = lens.cpu().numpy().astype(np.int64)
            assert (l % 2 == 0).all() and (l >= 2 * 9).all() and (l <= 4096).all()
            sample = recs[:64].cpu().numpy()
            for b in range(64):
                assert int(sample[b, 1:l[b]:2].astype(np.int64).sum()) == N


def test_fast_division_is_exact():
    """The compressor divides by the block scale through one reciprocal per block
    (kernels.hip: div_by_scale).  Exhaustive device check: every finite fp16
    dividend against every divisor the codec can form (m/127, m/448, and fp16 group
    scales) gives the same quotient bits as the IEEE divide (and the same stored byte).  Third counter: the cheap
    rounding truncate(y + copysign(0.5, y)) equals roundf(y) for every product the codec rounds (kernels.hip:
    round_to_int)."""
    import ctypes as C
    import torch
    lib = load_debug_lib()
    lib.speckv_debug_divcheck.argtypes = [C.c_float, C.c_void_p, C.c_void_p]
    for den in (127.0, 448.0, 0.0):
        cnt = torch.zeros(4, dtype=torch.int64, device="cuda")
        assert lib.speckv_debug_divcheck(den, cnt.data_ptr(), None) == 0
        torch.cuda.synchronize()
        assert cnt.tolist() == [0, 0, 0, 0], (den, cnt.tolist())


def test_fast_fp32_division_is_exact():
    """The tensor codec divides fp32 sources by the tensor's scale through one reciprocal and two refinements
    (codec_device.hpp: div_f32_by_scale) while the scale lies in 2^-60 .. 2^60.  Device check over EVERY fp32 bit pattern of the
    dividend a tensor with that scale can hold, per divisor: same quotient bits as the IEEE divide, same stored byte in both
    modes.  Divisors: significands all ones / 1.0 / random, at both ends and the middle of the admitted exponent range, and
    mx / 127 for round and random mx."""
    import ctypes as C
    import torch
    lib = load_debug_lib()
    lib.speckv_debug_divcheck_f32.argtypes = [C.c_float, C.c_void_p, C.c_void_p]
    rng = np.random.default_rng(20261004)
    scales = []
    for e in (-60, -59, -17, -1, 0, 1, 9, 59, 60):
        for mant in (0, 0x7FFFFF, 0x400000, 1, int(rng.integers(1, 0x7FFFFF)), int(rng.integers(1, 0x7FFFFF))):
            scales.append(np.uint32(((e + 127) << 23) | mant).view(np.float32))
    for mx in (1.0, 3.0, 127.0, 4.75, 65504.0, float(np.float32(rng.uniform(0.1, 20.0))), float(np.float32(rng.lognormal(0.0, 4.0)))):
        scales.append(np.float32(mx) / np.float32(127.0))
    checked = 0
    for s in scales:
        cnt = torch.zeros(4, dtype=torch.int64, device="cuda")
        assert lib.speckv_debug_divcheck_f32(float(s), cnt.data_ptr(), None) == 0
        torch.cuda.synchronize()
        c = cnt.tolist()
        assert c[:3] == [0, 0, 0] and c[3] > 2 ** 30, (float(s), hex(int(np.float32(s).view(np.uint32))), c)
        checked += c[3]
    assert checked > len(scales) * 2 ** 30


def test_wave_primitives():
    """DPP scans / shifts on the hardware (the folded wave_shr form once miscompiled)."""
    import ctypes as C
    import torch
    lib = load_debug_lib()
    lib.speckv_debug_wave_primitives.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    rng = np.random.default_rng(0)
    for trial in range(4):
        v = rng.integers(0, 1000, 64).astype(np.uint32) if trial else np.arange(64, dtype=np.uint32) + 100
        d_in = torch.from_numpy(v.view(np.int32)).cuda()
        d_out = torch.zeros(5 * 64, dtype=torch.int32, device="cuda")
        assert lib.speckv_debug_wave_primitives(C.c_void_p(d_in.data_ptr()), C.c_void_p(d_out.data_ptr()), None) == 0
        torch.cuda.synchronize()
        o = d_out.cpu().numpy().view(np.uint32).reshape(5, 64)
        assert np.array_equal(o[0], np.concatenate([[0xABCD], v[:-1]]))
        assert np.array_equal(o[1], np.cumsum(v).astype(np.uint32))
        assert np.array_equal(o[2], np.maximum.accumulate(v))
        assert (o[3] == v[63]).all()
        w = ((v.astype(np.uint64) * 2654435761) & 0xFFFFFFFF) >> 24
        prev = np.concatenate([[7], w[:-1]]).astype(np.int64)
        assert np.array_equal(o[4], ((w.astype(np.int64) - prev) & 0xFF).astype(np.uint32))


# ---- the reference's own call shape: a tensor of any length (speckv_ext_codec_compress_tensor / _decompress_tensor) ----
def gpu_compress_tensor(lib, x, mode=0, misalign=0):
    """x: 1-d float32 or float16 numpy.  Returns (scale f32, rle u8[compressed_size]).  misalign: elements the source is
    shifted off its 256-byte aligned allocation by."""
    torch = torch_mod()
    x = np.ascontiguousarray(x)
    f32 = x.dtype == np.float32
    assert f32 or x.dtype == np.float16
    n = x.size
    if misalign:
        pad = np.zeros(misalign, x.dtype)
        d_full = torch.from_numpy(np.concatenate([pad, x]) if f32 else np.concatenate([pad, x]).view(np.int16)).cuda()
        d_x = d_full[misalign:]
    else:
        d_x = torch.from_numpy(x if f32 else x.view(np.int16)).cuda() if n else torch.zeros(1, device="cuda")
    ws_bytes = lib.speckv_ext_codec_tensor_workspace_bytes(n)
    d_ws = torch.empty(ws_bytes + 256, dtype=torch.uint8, device="cuda")
    ws_ptr = (d_ws.data_ptr() + 255) & ~255
    d_rle = torch.full(((2 * n + 15) // 16 * 16 + 16,), 0xA5, dtype=torch.uint8, device="cuda")
    d_meta = torch.zeros(4, dtype=torch.int64, device="cuda")             # [0] compressed_size, [1] scale bits
    rc = lib.speckv_ext_codec_compress_tensor(d_x.data_ptr(), n, int(f32), d_rle.data_ptr(), d_meta.data_ptr(), d_meta.data_ptr() + 8,
                                              ws_ptr, ws_bytes, mode, stream_ptr())
    assert rc == 0, rc
    torch.cuda.synchronize()
    meta = d_meta.cpu().numpy()
    size = int(meta[0])
    scale = np.array([meta[1]], np.int64).view(np.float32)[0]
    rle = d_rle.cpu().numpy()
    assert size <= 2 * n
    return np.float32(scale), rle[:size].copy()


def gpu_decompress_tensor(lib, rle, scale, cap, mode=0, out_f32=True):
    torch = torch_mod()
    rle = np.ascontiguousarray(rle, dtype=np.uint8)
    d_rle = torch.from_numpy(np.concatenate([rle, np.zeros(16, np.uint8)])).cuda()
    ws_bytes = lib.speckv_ext_codec_tensor_decode_workspace_bytes(rle.size)
    d_ws = torch.empty(ws_bytes + 256, dtype=torch.uint8, device="cuda")
    ws_ptr = (d_ws.data_ptr() + 255) & ~255
    d_y = torch.full((cap + 16,), float("nan"), dtype=torch.float32 if out_f32 else torch.float16, device="cuda")
    d_n = torch.zeros(1, dtype=torch.int64, device="cuda")
    rc = lib.speckv_ext_codec_decompress_tensor(d_rle.data_ptr(), rle.size, float(scale), d_y.data_ptr(), cap, int(out_f32), d_n.data_ptr(),
                                                ws_ptr, ws_bytes, mode, stream_ptr())
    assert rc == 0, rc
    torch.cuda.synchronize()
    n = int(d_n.cpu().numpy()[0])
    y = d_y.cpu().numpy()
    assert np.isnan(y[max(n, cap):].astype(np.float32)).all()             # nothing written behind the buffer
    return y[:n].copy()


def test_tensor_codec_golden_reference_vectors(lib, golden_dir):
    """Every reference-generated vector of tests/golden/codec_vectors.npz through the tensor form -- including the ones the
    2048-element block form cannot take: `kat` (n = 11, the survey's known-answer test), `short_257`, and the 131 072-element
    `big` vector, whose stream is pinned by its length, a position-weighted checksum and the bit sum
    of its decoded output.  Bytes of the stream, bits of the scale, bits of every decoded fp32."""
    g = np.load(os.path.join(golden_dir, "codec_vectors.npz"))
    names = [k[:-2] for k in g.files if k.endswith(".x")]
    assert "kat" in names and "short_257" in names and len(names) >= 11
    for name in names:
        x = g[f"{name}.x"].astype(np.float32)
        scale, rle = gpu_compress_tensor(lib, x, 0)
        assert scale.tobytes() == g[f"{name}.scale"][0].tobytes(), name
        assert rle.tobytes() == g[f"{name}.rle"].tobytes(), name
        y = gpu_decompress_tensor(lib, g[f"{name}.rle"], g[f"{name}.scale"][0], x.size, 0, True)
        assert_same_float_bits(y, g[f"{name}.y"], name)
    # the known-answer stream itself (SURVEY Appendix A, F-codec-KAT)
    scale, rle = gpu_compress_tensor(lib, g["kat.x"].astype(np.float32), 0)
    assert scale == np.float32(1.0)
    assert rle.view(np.int8).tolist() == [0, 1, 127, 1, 2, 1, -65, 1, 0, 2, -32, 1, -31, 1, 0, 1, -1, 1, 0, 1]
    # big: regenerate the input exactly as the generator did
    n = int(g["big.n"][0])
    x = np.random.default_rng(int(g["big.seed"][0])).standard_normal(n).astype(np.float32)
    scale, rle = gpu_compress_tensor(lib, x, 0)
    assert scale.tobytes() == g["big.scale"][0].tobytes()
    assert rle.size == int(g["big.compressed_size"][0])
    crc = int(np.bitwise_xor.reduce(rle.astype(np.uint64) * (np.arange(rle.size, dtype=np.uint64) % 251 + 1)))
    assert crc == int(g["big.rle_crc"][0])
    y = gpu_decompress_tensor(lib, rle, scale, n, 0, True)
    assert y.size == n and int(y.view(np.uint32).astype(np.uint64).sum()) == int(g["big.y_sum_bits"][0])
    # malformed streams (odd tail, zero counts, counts > 127, one byte, empty)
    for i in range(5):
        rle = g[f"malformed{i}.rle"]
        want = g[f"malformed{i}.y"]
        y = gpu_decompress_tensor(lib, rle, 0.5, max(want.size, 1) + 40, 0, True)
        assert_same_float_bits(y, want, f"malformed{i}")


@pytest.mark.parametrize("form", ["single_pass", "no_split_tiles", "one_pass_decoder", "grids", "per_element_expand", "wg", "serial", "no_pre"])
def test_tensor_codec_scan_forms(lib, oracle, form):
    """Compress exists as ONE pass with look-back across workgroups (k_tc_fused, the default since round 4) and as the
    multi-launch form (speckv_ext_set_tuning: tc_multipass) whose scans across tiles come in three shapes -- grids of one wave per step, one
    workgroup, one wave (tc_scan; fp16 sources: with and without the summary pass emitting,
    tc_no_pre) -- which must all produce the oracle's stream and output: noise with long flat stretches (runs,
    255-splits and the delta chain cross tiles, workgroups and steps of 64 tiles), 70 to 900 000 elements."""
    rng = np.random.default_rng(5)
    if form == "no_pre":
        set_tuning("tc_no_pre", "1")                         # fp16 sources: summary and emit as two plain passes
    elif form == "grids":
        set_tuning("tc_multipass", "1")
    elif form == "per_element_expand":                               # the multi-launch decoder with the expand loop of rounds 2-3
        set_tuning("tc_multipass", "1"); set_tuning("td_expand_per_element", "1")
    elif form == "no_split_tiles":                                   # the one-pass compressor with the element-wise loop for long stretches
        set_tuning("tc_no_split_tiles", "1")
    elif form == "one_pass_decoder":                                 # the one-pass decoder also for streams of few pairs
        set_tuning("td_one_pass", "1")
    elif form != "single_pass":
        set_tuning("tc_scan", form)
    try:
        for n in (70, 2048 * 63 + 5, 2048 * 64, 2048 * 65 + 1, 900000):
            x = rng.standard_normal(n).astype(np.float32)
            for _ in range(6):                                       # flat stretches at random places, some longer than a step
                a = int(rng.integers(0, n)); b = min(n, a + int(rng.integers(1, 200000)))
                x[a:b] = np.float32(rng.standard_normal())
            for mode in MODES:
                o_scale, o_rle = oracle.compress_f32(x, mode)
                scale, rle = gpu_compress_tensor(lib, x, mode)
                assert np.float32(scale).tobytes() == np.float32(o_scale).tobytes() and rle.tobytes() == o_rle.tobytes(), (form, n, mode)
                y = gpu_decompress_tensor(lib, o_rle, o_scale, n + 3, mode, True)
                assert_same_float_bits(y, oracle.decompress_f32(o_rle, o_scale, mode), f"{form} {n} {mode}")
                if n >= 2048:                                        # the same as an fp16 source (tiles emitted by the summary pass unless no_pre)
                    x16 = x.astype(np.float16)
                    o_scale, o_rle = oracle.compress_f32(x16.astype(np.float32), mode)
                    scale, rle = gpu_compress_tensor(lib, x16, mode)
                    assert np.float32(scale).tobytes() == np.float32(o_scale).tobytes() and rle.tobytes() == o_rle.tobytes(), (form, n, mode, "fp16")
    finally:
        set_tuning("tc_scan", 0)
        set_tuning("tc_no_pre", 0)
        set_tuning("tc_multipass", 0)
        set_tuning("td_expand_per_element", 0)
        set_tuning("td_one_pass", 0)
        set_tuning("tc_no_split_tiles", 0)


def test_tensor_codec_look_back_over_many_workgroups(lib, oracle):
    """Both single-pass kernels hand their prefixes from workgroup to workgroup through status words that are looked back over
    64 at a time: a tensor of 6 Mi elements is 3072 tiles / chunks = 192 workgroups, three look-back windows deep, with flat
    stretches longer than a workgroup's 16 tiles (chain 1 of the compressor walks past workgroups that have no stretch start)
    and runs that cross many chunks of the decoder (several 4096-element windows per chunk, output clipped inside one)."""
    rng = np.random.default_rng(4242)
    n = 6 * 1024 * 1024 + 77
    x = rng.standard_normal(n).astype(np.float32)
    for a, m in ((100000, 70000), (1500000, 400000), (3000001, 33000), (5200000, 2049), (6000000, 255 * 40)):
        x[a:a + m] = np.float32(rng.standard_normal())
    x[4000000:4300000] = np.linspace(-2, 2, 300000, dtype=np.float32)       # constant deltas between quantisation steps
    for mode in MODES:
        o_scale, o_rle = oracle.compress_f32(x, mode)
        scale, rle = gpu_compress_tensor(lib, x, mode)
        assert np.float32(scale).tobytes() == np.float32(o_scale).tobytes(), mode
        assert rle.size == o_rle.size and rle.tobytes() == o_rle.tobytes(), (mode, rle.size, o_rle.size)
        want = oracle.decompress_f32(o_rle, o_scale, mode)
        y = gpu_decompress_tensor(lib, o_rle, o_scale, n + 9, mode, True)
        assert_same_float_bits(y, want, f"6Mi mode {mode}")
        y = gpu_decompress_tensor(lib, o_rle, o_scale, 1500000 + 123457, mode, False)       # clipped inside the long flat stretch, fp16 out
        assert_same_float_bits(y, want[:1500000 + 123457].astype(np.float16), f"6Mi clipped mode {mode}")
    # a hand-made stream: long counts everywhere (every chunk spans many windows) and a zero count in some chunks
    pairs = 300000
    stream = rng.integers(0, 256, 2 * pairs).astype(np.uint8)
    stream[1::2] = rng.integers(200, 256, pairs).astype(np.uint8)
    stream[1 + 2 * 5000] = 0; stream[1 + 2 * 123456] = 0
    want = oracle.decompress_f32(stream, 0.125, 0)
    y = gpu_decompress_tensor(lib, stream, 0.125, want.size + 5, 0, True)
    assert_same_float_bits(y, want, "long counts")


def test_tensor_codec_matches_oracle_over_sizes_and_structures(lib, oracle):
    """Lengths around the tile size and far beyond it, data whose runs, 255-splits and delta chain cross tile boundaries
    (constant tensors, long piecewise-constant stretches, ramps), fp32 and fp16 sources, both quantiser modes, and the
    decoder clipped by a short output buffer."""
    rng = np.random.default_rng(77)
    cases = []
    for n in (0, 1, 2, 11, 63, 64, 65, 254, 255, 256, 257, 2047, 2048, 2049, 4095, 4097, 6000, 131072, 300001):
        cases.append((f"gauss{n}", rng.standard_normal(n).astype(np.float32)))
    cases.append(("const600k", np.full(600001, np.float32(0.37))))
    cases.append(("zeros70k", np.zeros(70000, np.float32)))
    cases.append(("ramp", np.linspace(-3, 3, 50000).astype(np.float32)))                  # constant deltas: long stretches
    pw = np.repeat(rng.standard_normal(400).astype(np.float32), rng.integers(1, 3000, 400))
    cases.append(("piecewise", pw))
    sp = np.where(rng.random(200000) < 0.001, rng.standard_normal(200000), 0).astype(np.float32)
    cases.append(("sparse", sp))
    wide = rng.standard_normal(5000).astype(np.float32); wide[17] = np.inf; wide[99] = np.nan; wide[4000] = -3e38
    cases.append(("nonfinite", wide))
    # fp32 quotients on and beside the quantiser's rounding boundaries (n + 0.5, its two neighbours; the same over 127 for
    # REF_EXACT), for scales inside the range of the reciprocal short cut (div_f32_by_scale: 2^-60 .. 2^60), at its ends, outside
    # it (the IEEE divide) and with a full significand
    for e, mant in ((0, 0), (-60, 0), (60, 0x7FFFFF), (-61, 0x7FFFFF), (61, 0), (-100, 0x123456), (90, 0x7FFFFF), (3, 0x7FFFFF), (-7, 0x2AAAAA)):
        sc = np.uint32(((e + 127) << 23) | mant).view(np.float32)
        halves = np.arange(0, 127, dtype=np.float32) + np.float32(0.5)
        qs = np.concatenate([halves, np.nextafter(halves, np.float32(0)), np.nextafter(halves, np.float32(200)),
                             halves / np.float32(127), np.nextafter(halves / np.float32(127), np.float32(0)), np.nextafter(halves / np.float32(127), np.float32(2))]).astype(np.float32)
        qs = np.concatenate([qs, -qs, rng.uniform(-127, 127, 8192 - 2 * qs.size - 1).astype(np.float32)])
        with np.errstate(over="ignore", under="ignore"):
            xb = np.concatenate([[np.float32(127) * sc], rng.permutation(qs) * sc]).astype(np.float32)
        if np.isfinite(xb).all():
            cases.append((f"boundaries 2^{e} {mant:#x}", xb))
    for name, x in cases:
        for mode in MODES:
            o_scale, o_rle = oracle.compress_f32(x, mode)
            scale, rle = gpu_compress_tensor(lib, x, mode)
            assert np.float32(scale).tobytes() == np.float32(o_scale).tobytes(), (name, mode)
            assert rle.tobytes() == o_rle.tobytes(), (name, mode, rle.size, o_rle.size)
            want = oracle.decompress_f32(o_rle, o_scale, mode)
            y = gpu_decompress_tensor(lib, o_rle, o_scale, x.size + 5, mode, True)
            assert_same_float_bits(y, want, f"{name} mode {mode}")
            if x.size > 100:                                                              # a buffer shorter than the stream's total
                y = gpu_decompress_tensor(lib, o_rle, o_scale, x.size - 37, mode, True)
                assert_same_float_bits(y, want[:x.size - 37], f"{name} clipped")
    # the same structures as fp16 sources: whole tiles take the block encoder's 8-elements-per-lane path, which must hand the
    # tiles it cannot take (inf / NaN, stretches that may reach 255 elements, run starts of a stretch entering the tile) to the
    # element-wise loop, tile by tile; also sources that are not 16-byte aligned (element-wise throughout)
    noisy_then_flat = np.concatenate([rng.standard_normal(5000), np.full(9000, 0.5), rng.standard_normal(3000), np.zeros(300),
                                      rng.standard_normal(7), np.full(254, -1.25), rng.standard_normal(2000), np.full(255, 2.0),
                                      rng.standard_normal(1793), np.full(256, 0.75), rng.standard_normal(4000)]).astype(np.float32)
    for name, x in cases + [("noisy_then_flat", noisy_then_flat)]:
        if x.size < 2048:
            continue
        with np.errstate(over="ignore"):
            x16 = np.clip(x, -60000.0, 60000.0).astype(np.float16) if name != "nonfinite" else x.astype(np.float16)
        for mode in MODES:
            o_scale, o_rle = oracle.compress_f32(x16.astype(np.float32), mode)
            for misalign in (0, 3):
                scale, rle = gpu_compress_tensor(lib, x16, mode, misalign)
                assert np.float32(scale).tobytes() == np.float32(o_scale).tobytes(), (name, mode, misalign, "fp16 source")
                assert rle.tobytes() == o_rle.tobytes(), (name, mode, misalign, "fp16 source", rle.size, o_rle.size)
    # fp16 source and fp16 output: the same maths on exactly converted inputs, one RNE rounding on the way out
    for n in (11, 2049, 100000):
        x16 = (rng.standard_normal(n) * 3).astype(np.float16)
        o_scale, o_rle = oracle.compress_f32(x16.astype(np.float32), 0)
        scale, rle = gpu_compress_tensor(lib, x16, 0)
        assert np.float32(scale).tobytes() == np.float32(o_scale).tobytes() and rle.tobytes() == o_rle.tobytes()
        y16 = gpu_decompress_tensor(lib, o_rle, o_scale, n, 0, False)
        assert_same_float_bits(y16, oracle.decompress_f32(o_rle, o_scale, 0).astype(np.float16), f"fp16 {n}")
    # a tensor of one 2048-element block is the block codec's record, byte for byte
    xb = rng.standard_normal(2048).astype(np.float16)
    scales, lens, recs = gpu_compress(lib, xb[None, :], 2, 0)
    scale, rle = gpu_compress_tensor(lib, xb, 0)
    assert scale.tobytes() == scales[0].tobytes() and rle.tobytes() == recs[0, :lens[0]].tobytes()
    # random byte streams (zero counts, long counts) through the tensor decoder
    for n_pairs in (1, 100, 2048, 2049, 10000):
        stream = rng.integers(0, 256, 2 * n_pairs + 1).astype(np.uint8)
        counts = stream[1::2]
        counts[rng.random(counts.size) < 0.2] = 0
        want = oracle.decompress_f32(stream, 0.25, 0)
        y = gpu_decompress_tensor(lib, stream, 0.25, want.size + 3, 0, True)
        assert_same_float_bits(y, want, f"random stream {n_pairs}")


def test_engine_picks_the_flat_run_decoder_by_itself(oracle):
    """an allocation that holds data the reference's scheme compresses is read with the flat-run decoder without
    anybody passing a hint -- the compress kernel leaves the record length of every 1024th page in host-visible memory and
    speckv_ext_fetch_range looks at the mean (never-sealed allocations; sealed ones by their packed size, as before) -- through
    the fused kernel AND through the copy engines (staged records: k_fetch_decompress_flat_staged).  The bytes are the
    oracle's either way; speckv_ext_stats_t.flat_decoder_fetches shows which decoder ran."""
    import os
    import cxl_speckv_amd as pkg
    from cxl_speckv_amd.speckv_ctypes import SpeckvLib
    torch = __import__("torch")
    os.environ["SPECKV_POOL_DEVICES"] = "0,0"
    try:
        lib = SpeckvLib(pkg.library_path(), "hip:0")
    finally:
        os.environ.pop("SPECKV_POOL_DEVICES", None)
    try:
        lib.set_compression_scheme(2)
        n_pages = 4096
        rng = np.random.default_rng(31)
        flat = np.repeat(rng.standard_normal((n_pages, N // 32)), 32, axis=1).astype(np.float16)       # runs of 32: ~130-byte records
        flat[7] = rng.standard_normal(N).astype(np.float16)                                            # one page that does not compress
        noise = rng.standard_normal((n_pages, N)).astype(np.float16)
        seen = {}
        for name, x in (("flat", flat), ("noise", noise)):
            h = lib.alloc(n_pages * 4096)
            lib.write(h, 0, x.ctypes.data, x.nbytes, False)
            scales, lens, recs = oracle.compress_blocks_f16(x, 2, 0)
            want = oracle.decompress_blocks_f16(recs, lens, scales, 2, 0)
            before = lib.stats().flat_decoder_fetches
            for engine in (1, 2):
                out = torch.zeros((n_pages, N), dtype=torch.float16, device="cuda")
                torch.cuda.synchronize()
                lib.fetch_range(h, 0, n_pages, out.data_ptr(), False, torch.cuda.current_stream().cuda_stream, engine=engine)
                torch.cuda.synchronize()
                assert_same_float_bits(out.cpu().numpy(), want, f"{name} engine {engine}")
            seen[name] = lib.stats().flat_decoder_fetches - before
            lib.free(h)
        assert seen == {"flat": 2, "noise": 0}, seen
    finally:
        lib.finalize()


# ---- many tensors per launch, one workgroup per tensor (speckv_ext_codec_compress_tensors / _decompress_tensors, round 6) ----
def gpu_codec_tensors(lib, xs, mode=0, out_f32=True, misalign=0, room_extra=3):
    """xs: list of 1-d numpy arrays of ONE dtype (float32 or float16), any lengths.  One compress launch and one decompress launch over
    all of them.  Returns [(scale f32, rle bytes, decoded)] per tensor."""
    import ctypes as C
    torch = torch_mod()
    f32 = xs[0].dtype == np.float32
    esz = 4 if f32 else 2
    lib.speckv_ext_codec_tensors_workspace_bytes.argtypes = [C.c_uint32, C.c_uint64]; lib.speckv_ext_codec_tensors_workspace_bytes.restype = C.c_size_t
    lib.speckv_ext_codec_compress_tensors.argtypes = [C.c_uint32, C.c_void_p, C.c_uint64, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
    lib.speckv_ext_codec_decompress_tensors.argtypes = [C.c_uint32, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
    nt = len(xs)
    max_elems = max(x.size for x in xs) + room_extra
    ws_bytes = lib.speckv_ext_codec_tensors_workspace_bytes(nt, max_elems)
    d_ws = torch.full((ws_bytes + 256,), 0x5A, dtype=torch.uint8, device="cuda")               # (dirty: the call clears what it uses)
    ws_ptr = (d_ws.data_ptr() + 255) & ~255
    # sources back to back (each start 16-byte aligned + `misalign` elements), streams and outputs 16-byte aligned with guard bytes between
    src_off, rle_off, out_off = [], [], []
    so = ro = oo = 0
    osz = 4 if out_f32 else 2
    for x in xs:
        so = (so + 15) // 16 * 16 + misalign * esz; src_off.append(so); so += x.size * esz
        ro = (ro + 15) // 16 * 16; rle_off.append(ro); ro += (2 * x.size + 15) // 16 * 16 + 16
        oo = (oo + 15) // 16 * 16; out_off.append(oo); oo += (x.size + room_extra) * osz + 32
    h_src = np.zeros(so + 16, np.uint8)
    for x, o in zip(xs, src_off):
        h_src[o:o + x.size * esz] = np.ascontiguousarray(x).view(np.uint8)
    d_src = torch.from_numpy(h_src).cuda()
    d_rle = torch.full((ro + 16,), 0xA5, dtype=torch.uint8, device="cuda")
    d_out = torch.full((oo + 16,), 0xFF, dtype=torch.uint8, device="cuda")                  # (0xFFFF / 0xFFFFFFFF: NaN patterns)
    desc_c = np.zeros((nt, 4), np.uint64); desc_d = np.zeros((nt, 4), np.uint64)
    for i, x in enumerate(xs):
        desc_c[i] = (d_src.data_ptr() + src_off[i], x.size, d_rle.data_ptr() + rle_off[i], (2 * x.size + 15) // 16 * 16)
        desc_d[i] = (d_out.data_ptr() + out_off[i], x.size + room_extra, d_rle.data_ptr() + rle_off[i], 0)
    d_desc_c = torch.from_numpy(desc_c.view(np.int64)).cuda(); d_desc_d = torch.from_numpy(desc_d.view(np.int64)).cuda()
    d_bytes = torch.full((nt,), -1, dtype=torch.int64, device="cuda")
    d_scales = torch.full((nt,), float("nan"), dtype=torch.float32, device="cuda")
    d_nout = torch.full((nt,), -1, dtype=torch.int64, device="cuda")
    assert lib.speckv_ext_codec_compress_tensors(nt, d_desc_c.data_ptr(), max_elems, int(f32), d_bytes.data_ptr(), d_scales.data_ptr(), ws_ptr, ws_bytes, mode, stream_ptr()) == 0
    assert lib.speckv_ext_codec_decompress_tensors(nt, d_desc_d.data_ptr(), max_elems, d_bytes.data_ptr(), d_scales.data_ptr(), int(out_f32), d_nout.data_ptr(), ws_ptr, ws_bytes, mode,
                                                   stream_ptr()) == 0
    torch.cuda.synchronize()
    sizes, scales, nout = d_bytes.cpu().numpy(), d_scales.cpu().numpy(), d_nout.cpu().numpy()
    rle, out = d_rle.cpu().numpy(), d_out.cpu().numpy()
    res = []
    for i, x in enumerate(xs):
        assert 0 <= sizes[i] <= 2 * x.size, (i, sizes[i])
        r = rle[rle_off[i]:rle_off[i] + sizes[i]].copy()
        guard = rle[rle_off[i] + (2 * x.size + 15) // 16 * 16:rle_off[i] + (2 * x.size + 15) // 16 * 16 + 16]
        assert (guard == 0xA5).all(), f"tensor {i}: its stream ran over its room"
        assert nout[i] == x.size, (i, nout[i], x.size)
        y = out[out_off[i]:out_off[i] + x.size * osz].view(np.float32 if out_f32 else np.float16).copy()
        behind = out[out_off[i] + x.size * osz:out_off[i] + (x.size + room_extra) * osz + 32]
        assert (behind == 0xFF).all(), f"tensor {i}: something was written behind its last element"
        res.append((np.float32(scales[i]), r, y))
    return res


def test_tensors_batch_golden_reference_vectors(lib, golden_dir):
    """Every reference-generated codec vector as ONE batched launch each way (kat n = 11 beside short_257 beside the 2048-element blocks
    beside the 131 072-element `big` -- the reference's own call size): per tensor the stream's bytes, the scale's bits and every
    decoded fp32 bit are the reference's."""
    g = np.load(os.path.join(golden_dir, "codec_vectors.npz"))
    names = [k[:-2] for k in g.files if k.endswith(".x")]
    xs = [g[f"{nm}.x"].astype(np.float32) for nm in names]
    n_big = int(g["big.n"][0])
    xs.append(np.random.default_rng(int(g["big.seed"][0])).standard_normal(n_big).astype(np.float32))
    res = gpu_codec_tensors(lib, xs, 0, True)
    for nm, (scale, rle, y) in zip(names, res):
        assert scale.tobytes() == g[f"{nm}.scale"][0].tobytes(), nm
        assert rle.tobytes() == g[f"{nm}.rle"].tobytes(), nm
        assert_same_float_bits(y, g[f"{nm}.y"], nm)
    scale, rle, y = res[-1]
    assert scale.tobytes() == g["big.scale"][0].tobytes() and rle.size == int(g["big.compressed_size"][0])
    assert int(np.bitwise_xor.reduce(rle.astype(np.uint64) * (np.arange(rle.size, dtype=np.uint64) % 251 + 1))) == int(g["big.rle_crc"][0])
    assert int(y.view(np.uint32).astype(np.uint64).sum()) == int(g["big.y_sum_bits"][0])


@pytest.mark.parametrize("one_wg", [0, 1])
@pytest.mark.parametrize("dtype", [np.float32, np.float16])
def test_tensors_batch_matches_oracle_over_sizes_and_structures(lib, oracle, dtype, one_wg):
    """The batched form against the oracle, per tensor: lengths 0, 1, below / at / just over a tile, a round (16 tiles) and several
    rounds with ragged ends; noise, long flat stretches that cross tiles and rounds (255-splits, the delta chain and both carried
    chains), all zeros, inf / NaN; both quantiser modes; sources off their 16-byte alignment; fp16 and fp32 outputs."""
    rng = np.random.default_rng(61 if dtype == np.float32 else 62)
    lens = [0, 1, 7, 2047, 2048, 2049, 5000, 8 * 2048 - 1, 8 * 2048, 8 * 2048 + 1, 16 * 2048 - 1, 16 * 2048, 16 * 2048 + 1, 131072, 131072 + 77, 3 * 32768 + 2048 * 5 + 3, 300001]      # (workgroups take 16 tiles to compress, 8 chunks to decompress)
    xs = []
    for n in lens:
        x = rng.standard_normal(n).astype(np.float32)
        for _ in range(3 if n > 2048 else 0):
            a = int(rng.integers(0, n)); b = min(n, a + int(rng.integers(1, max(2, n))))
            x[a:b] = np.float32(rng.standard_normal())
        xs.append(x.astype(dtype))
    xs.append(np.zeros(40000, dtype))                                              # all zeros: scale 1, one run per 255
    z = rng.standard_normal(70000).astype(dtype); z[5] = np.inf; z[40000] = np.nan; z[69999] = -np.inf
    xs.append(z)
    xs.append(np.repeat(rng.standard_normal(300).astype(dtype), 700))              # runs of 700: every one split at 255, across tiles and rounds
    # both forms: several workgroups per tensor (look-back over the tensor's own words, max|x| by rendezvous) and one workgroup per tensor
    # (chains carried in LDS: what launches of tensors of at most 16 tiles take by themselves)
    set_tuning("tc_batch_one_wg", one_wg)
    try:
        _tensors_batch_against_oracle(lib, oracle, xs)
    finally:
        set_tuning("tc_batch_one_wg", 0)


def _tensors_batch_against_oracle(lib, oracle, xs):
    for mode in MODES:
        for misalign, out_f32 in ((0, True), (3, False)):
            res = gpu_codec_tensors(lib, xs, mode, out_f32, misalign)
            for i, (x, (scale, rle, y)) in enumerate(zip(xs, res)):
                o_scale, o_rle = oracle.compress_f32(x.astype(np.float32), mode)
                assert np.float32(scale).tobytes() == np.float32(o_scale).tobytes(), (i, x.size, mode, misalign)
                assert rle.tobytes() == o_rle.tobytes(), (i, x.size, mode, misalign)
                want = oracle.decompress_f32(o_rle, o_scale, mode)
                if out_f32:
                    assert_same_float_bits(y, want, f"tensor {i} n={x.size} mode={mode}")
                else:
                    assert_same_float_bits(y, want.astype(np.float16), f"tensor {i} n={x.size} mode={mode} fp16 out")


def test_tensors_batch_equals_the_single_tensor_entry_point(lib):
    """4 x 64 tensors of 131 072 elements (the reference's call size) in one launch: every stream equals the one the single-tensor
    entry point produces for the same tensor (which the tests above pin to the oracle and the reference)."""
    rng = np.random.default_rng(63)
    base = [rng.standard_normal(131072).astype(np.float32) * np.float32(rng.uniform(0.1, 30)) for _ in range(4)]
    xs = [base[i % 4] if i % 8 < 4 else np.roll(base[i % 4], i) for i in range(256)]
    res = gpu_codec_tensors(lib, xs, 0, True)
    single = {}
    for i in (0, 1, 2, 3, 5, 130, 255):
        scale, rle = gpu_compress_tensor(lib, xs[i], 0)
        assert res[i][0].tobytes() == scale.tobytes() and res[i][1].tobytes() == rle.tobytes(), i
        y = gpu_decompress_tensor(lib, rle, scale, xs[i].size, 0, True)
        assert_same_float_bits(res[i][2], y, f"tensor {i}")
    for i in range(256):                                                            # the same tensor always gives the same stream
        if i % 8 < 4:
            assert res[i][1].tobytes() == res[i % 4][1].tobytes(), i
