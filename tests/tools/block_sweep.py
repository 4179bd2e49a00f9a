"""A randomised sweep of the BLOCK codec's RLE scheme against the CPU oracle (test infrastructure): blocks made of random pieces --
noise, constants, ramps with equal quantised steps, zeros, sparse noise, runs of every length from 1 to 2048 at random phases -- so
that stretches of equal deltas end, begin and split (every 255 elements) at random places against the 8-element lanes and the
512-element chunks of the encoder; both quantiser modes; records byte for byte, lengths, scales, decoded bits (fp16; the flat-run kernel; fp32 outputs, sampled against the oracle).
    python tests/tools/block_sweep.py [rounds=20] [blocks_per_round=4096]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from tests._gpu import load_raw_lib, assert_same_float_bits
from tests.test_gpu_codec import gpu_compress, gpu_decompress, N
from oracle.bindings import Oracle, build_oracle

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
lib = load_raw_lib()
build_oracle()
oracle = Oracle()
t0 = time.time()
for r in range(rounds):
    rng = np.random.default_rng(9100 + r)
    x = np.empty(B * N + 8192)
    pos = 0
    while pos < x.size:
        kind = int(rng.integers(0, 7))
        if kind == 0: m = int(rng.integers(1, 3000)); p = rng.standard_normal(m) * 10.0 ** rng.uniform(-2, 2)
        elif kind == 1: m = int(rng.integers(1, 2300)); p = np.full(m, rng.standard_normal())
        elif kind == 2: m = int(rng.integers(200, 1100)); p = np.full(m, rng.standard_normal())
        elif kind == 3: m = int(rng.choice([254, 255, 256, 509, 510, 511, 765, 1020, 1275, 2040])) + int(rng.integers(-2, 3)); p = np.full(m, rng.standard_normal())
        elif kind == 4: m = int(rng.integers(1, 600)); p = np.zeros(m)
        elif kind == 5: m = int(rng.integers(30, 500)); p = rng.standard_normal() + np.arange(m) * rng.choice([1.0, 2.0, -1.0, 3.0]) * rng.uniform(0.001, 0.02)
        else: m = int(rng.integers(1, 400)); p = np.where(rng.random(m) < 0.02, rng.standard_normal(m), 0.0)
        x[pos:pos + m] = p[:x.size - pos]
        pos += m
    ph = int(rng.integers(0, 8192))
    x16 = x[ph:ph + B * N].astype(np.float16).reshape(B, N)
    for mode in (0, 1):
        scales, lens, recs = gpu_compress(lib, x16, 2, mode)
        o_scales, o_lens, o_recs = oracle.compress_blocks_f16(x16, 2, mode)
        assert np.array_equal(lens, o_lens), (r, mode, np.nonzero(lens != o_lens)[0][:8])
        assert scales.tobytes() == o_scales.tobytes(), (r, mode)
        mask = np.arange(4096)[None, :] < lens[:, None]
        bad = np.nonzero((recs != o_recs) & mask)
        assert bad[0].size == 0, (r, mode, bad[0][:5], bad[1][:5])
        y = gpu_decompress(lib, recs, lens, scales, 2, mode)
        assert_same_float_bits(y, oracle.decompress_blocks_f16(o_recs, o_lens, o_scales, 2, mode), f"round {r} mode {mode}")
        # the flat-run kernel (SPECKV_CODEC_HINT_STRUCTURED) and the fp32 outputs: the same values
        assert_same_float_bits(gpu_decompress(lib, recs, lens, scales, 2, mode | 0x100), y, f"round {r} mode {mode} hinted")
        y32 = gpu_decompress(lib, recs, lens, scales, 2, mode, True)
        assert_same_float_bits(gpu_decompress(lib, recs, lens, scales, 2, mode | 0x100, True), y32, f"round {r} mode {mode} hinted fp32")
        for b in rng.integers(0, B, 24):
            want = np.zeros(N, np.float32)
            got = oracle.decompress_block_f32(o_recs[b, :o_lens[b]], o_scales[b], 2, mode, N)
            want[:got.size] = got
            assert_same_float_bits(y32[b], want, f"round {r} mode {mode} block {b} fp32")
    if r % 5 == 4: print(f"round {r} ok, {(r + 1) * B} blocks, {time.time() - t0:.1f} s", flush=True)
print(f"block sweep clean: {rounds} rounds, {rounds * B} blocks")
