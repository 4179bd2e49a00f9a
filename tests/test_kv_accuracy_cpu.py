"""CPU: the accuracy bounds of tests/test_gpu_accuracy.py have teeth, and the per-channel pre-scale of K is exact (numpy only)."""
import numpy as np

from cxl_speckv_amd.kv_accuracy import synth_kv, attention_f64, pow2_channel_scales, D
from tests.test_gpu_accuracy import BOUNDS


def test_a_dropped_tile_or_a_wrong_scale_fails_these_bounds():
    """The bounds have teeth (the old 'rel <= 0.6' would pass a broken kernel).  What checks the KERNEL is `kernel_rel_l2` -- the fused
    kernel against float64 attention over the same dequantised values: on this data, attention that skips ONE 32-position tile, or
    that reads K with every block scale one binade off, is 2 x and more outside every format's kernel bound."""
    K, V, qs, _ = synth_kv(2048, 8, 7001)
    sm = 1.0 / np.sqrt(D)
    rel = lambda a, b: float(np.sqrt(((a - b) ** 2).sum() / (b ** 2).sum()))
    for regime in ("peaky", "decode"):
        ref, top, _ = attention_f64(K, V, qs[regime], sm)
        t0 = int(np.bincount(top.ravel() // 32).argmax()) * 32           # the tile most rows look at
        keep = np.r_[0:t0, t0 + 32:2048]
        dropped = rel(attention_f64(K[keep], V[keep], qs[regime], sm)[0], ref)
        wrong = rel(attention_f64((K.astype(np.float32) * 2).astype(np.float16), V, qs[regime], sm)[0], ref)
        kern = max(b[regime][2] for b in BOUNDS.values())
        assert dropped > 2 * kern and wrong > 2 * kern, (regime, dropped, wrong, kern)


def test_power_of_two_channel_scales_are_exact():
    K, _, qs, _ = synth_kv(256, 8, 7003)
    s = pow2_channel_scales(K)
    assert np.all(np.log2(s) == np.round(np.log2(s)))
    Ks = (K.astype(np.float32) / s[None]).astype(np.float16)
    assert np.array_equal((Ks.astype(np.float32) * s[None]).astype(np.float16), K)          # (no value of this data leaves fp16's normal range)
