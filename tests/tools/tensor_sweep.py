"""A randomised sweep of the whole-tensor codec (speckv_ext_codec_compress_tensor / _decompress_tensor) against the CPU oracle
(test infrastructure): random lengths, data made of random pieces (noise at random magnitudes, constants, ramps, zeros, sparse
noise -- runs, 255-splits and delta chains crossing tiles at random places), both quantiser modes, fp32 and fp16 sources;
stream bytes, scale bits and decoded bits compared exactly; random byte streams (zero counts, long counts) through the decoder,
output buffers of random capacity.
    python tests/tools/tensor_sweep.py [rounds=200] [max_len=400000]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from tests._gpu import load_raw_lib, assert_same_float_bits
from tests.test_gpu_codec import gpu_compress_tensor, gpu_decompress_tensor
from oracle.bindings import Oracle, build_oracle

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 200
max_len = int(sys.argv[2]) if len(sys.argv) > 2 else 400000
lib = load_raw_lib()
build_oracle()
oracle = Oracle()
t0 = time.time()
elements = 0
for r in range(rounds):
    rng = np.random.default_rng(4200 + r)
    n = int(rng.integers(1, max_len)) if r % 5 else int(rng.integers(1, 5000))
    parts, left = [], n
    while left > 0:
        m = int(min(left, rng.integers(1, max(2, n // 3))))
        kind = int(rng.integers(0, 6))
        if kind == 0: p = rng.standard_normal(m) * 10.0 ** rng.uniform(-3, 3)
        elif kind == 1: p = np.full(m, rng.standard_normal())
        elif kind == 2: p = np.linspace(rng.standard_normal(), rng.standard_normal() * 4, m)
        elif kind == 3: p = np.zeros(m)
        elif kind == 4: p = np.where(rng.random(m) < 0.01, rng.standard_normal(m), 0.0)
        else: p = np.repeat(rng.standard_normal((m + 31) // 32), 32)[:m]
        parts.append(p); left -= m
    x = np.concatenate(parts).astype(np.float32)
    for mode in (0, 1):
        o_scale, o_rle = oracle.compress_f32(x, mode)
        scale, rle = gpu_compress_tensor(lib, x, mode)
        assert np.float32(scale).tobytes() == np.float32(o_scale).tobytes(), (r, mode, "scale")
        assert rle.tobytes() == o_rle.tobytes(), (r, mode, "stream", rle.size, o_rle.size)
        want = oracle.decompress_f32(o_rle, o_scale, mode)
        cap = n + 5 if r % 3 else max(1, int(rng.integers(1, n + 1)))
        y = gpu_decompress_tensor(lib, o_rle, o_scale, cap, mode, True)
        assert_same_float_bits(y, want[:cap], f"round {r} mode {mode}")
    if n >= 2048 and r % 2 == 0:                                     # fp16 source (whole tiles through the block encoder's path)
        x16 = np.clip(x, -60000.0, 60000.0).astype(np.float16)
        o_scale, o_rle = oracle.compress_f32(x16.astype(np.float32), 0)
        scale, rle = gpu_compress_tensor(lib, x16, 0, int(rng.integers(0, 2)) * 3)
        assert np.float32(scale).tobytes() == np.float32(o_scale).tobytes() and rle.tobytes() == o_rle.tobytes(), (r, "fp16 source")
    if r % 4 == 0:                                                   # random byte stream through the decoder
        n_pairs = int(rng.integers(1, 60000))
        stream = rng.integers(0, 256, 2 * n_pairs + int(rng.integers(0, 2))).astype(np.uint8)
        counts = stream[1::2]
        counts[rng.random(counts.size) < rng.uniform(0, 0.5)] = 0
        if r % 8 == 0: counts[:] = np.minimum(counts, 2)             # dense: short runs
        want = oracle.decompress_f32(stream, 0.25, 0)
        cap = want.size + 3 if r % 3 else max(1, want.size // 2)
        y = gpu_decompress_tensor(lib, stream, 0.25, cap, 0, True)
        assert_same_float_bits(y, want[:cap], f"random stream round {r}")
    elements += n
    if r % 20 == 19: print(f"round {r} ok, {elements} elements, {time.time() - t0:.1f} s", flush=True)
print(f"tensor sweep clean: {rounds} rounds, {elements} elements")
