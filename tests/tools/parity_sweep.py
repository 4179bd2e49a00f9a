"""A long randomised parity sweep of the codec kernels against the CPU oracle (test infrastructure): many seeds and block
statistics (Gaussian at several magnitudes, smooth, sparse, heavy-tailed, values at the fp16 limits, denormals), every
scheme and quantiser mode, compress bytes / lengths / scale bits and decoded bits compared exactly.
    python tests/tools/parity_sweep.py [rounds=8] [blocks_per_round=16384]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))   # tests/tools -> repo root
sys.path.insert(0, ROOT)
import numpy as np
from tests._gpu import N, load_raw_lib, gpu_compress, gpu_decompress
from oracle.bindings import Oracle, build_oracle

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 8
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
lib = load_raw_lib()
build_oracle()
oracle = Oracle()
t0 = time.time()
checked = 0
for r in range(rounds):
    rng = np.random.default_rng(9000 + r)
    x = rng.standard_normal((B, N)) * (10.0 ** rng.uniform(-6, 4, (B, 1)))
    k = B // 8
    x[0 * k:1 * k] = np.repeat(rng.standard_normal((k, N // 16)), 16, axis=1)                 # smooth (runs of 16)
    x[1 * k:2 * k] *= rng.random((k, N)) < 0.03                                                # sparse
    x[2 * k:3 * k] = rng.standard_cauchy((k, N)) * 3.0                                         # heavy tails
    x[3 * k:3 * k + 64] = 0.0                                                                  # exact zeros
    x[3 * k + 64:3 * k + 128, ::97] = 65504.0                                                  # fp16 maximum
    x[3 * k + 128:3 * k + 192] = rng.standard_normal((64, N)) * 6e-8                           # fp16 denormals
    x = np.clip(x, -65504.0, 65504.0)
    x16 = x.astype(np.float16)
    for scheme, modes in ((1, (0, 1)), (2, (0, 1)), (3, (0,)), (4, (0,))):
        for mode in modes:
            scales, lens, recs = gpu_compress(lib, x16, scheme, mode)
            o_scales, o_lens, o_recs = oracle.compress_blocks_f16(x16, scheme, mode)
            assert np.array_equal(lens, o_lens), (r, scheme, mode, "lens")
            assert scales.tobytes() == o_scales.tobytes(), (r, scheme, mode, "scales")
            mask = np.arange(recs.shape[1])[None, :] < lens[:, None]
            assert np.array_equal(recs[mask], o_recs[:, :recs.shape[1]][mask]), (r, scheme, mode, "records")
            y = gpu_decompress(lib, recs, lens, scales, scheme, mode)
            want = oracle.decompress_blocks_f16(o_recs, o_lens, o_scales, scheme, mode)
            assert y.view(np.uint16).tobytes() == np.asarray(want).view(np.uint16).tobytes() or \
                np.array_equal(np.isnan(y.astype(np.float32)), np.isnan(np.asarray(want).astype(np.float32))) and \
                np.array_equal(y.view(np.uint16)[~np.isnan(y.astype(np.float32))], np.asarray(want).view(np.uint16)[~np.isnan(np.asarray(want).astype(np.float32))]), (r, scheme, mode, "decoded")
            checked += B
    print("round", r, "ok,", checked, "block-checks,", round(time.time() - t0, 1), "s", flush=True)
print("parity sweep clean:", checked, "block-checks")
