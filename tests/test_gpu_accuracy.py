"""-m gpu: what the pool formats cost in attention accuracy on KV-like data (VERDICT r5 missing #2, weak #2).

The reference claims "99.5 % preservation" / "minimal accuracy loss" (docs/ARCHITECTURE.md:246, README.md:18) and measures nothing; its
codec is INT8.  SURVEY row A22 (the 4:1 formats) is "parity unpinned by the reference", and the tolerance tests of the fused attention
compare the kernels with an oracle that quantises the query the way the kernel does -- so the price of quantising q, K and V was
visible nowhere.  Here it is: cxl-speckv_amd/kv_accuracy.py builds seeded K / V with the structure real KV has (outlier channels
10-50 x, RoPE pairs, log-normal channel scales, Student-t V; peaky / decode-like / flat softmax), the reference result is float64
attention over the ORIGINAL fp16 values with the UNQUANTISED query, and every bound below is the measured figure (MI355X, seeds 7001 and
7002: profiles/r06_kv_format_accuracy.txt) with a margin of a quarter -- a wrong scale, a dropped tile, a swapped nibble or a query
row quantised against the wrong block moves these numbers by factors, not by a quarter."""
import numpy as np
import pytest

import cxl_speckv_amd as pkg
from cxl_speckv_amd.kv_accuracy import kv_format_accuracy, synth_kv, attention_f64, pow2_channel_scales, D

pytestmark = pytest.mark.gpu

# format -> regime -> (rel_l2 upper bound, cosine_mean lower bound, kernel_rel_l2 upper bound)
BOUNDS = {
    "fp8_e4m3":        {"peaky": (0.065, 0.985, 0.015), "decode": (0.24, 0.975, 0.055), "flat": (0.040, 0.999, 0.010)},
    "int4_g32":        {"peaky": (0.30, 0.93, 0.001), "decode": (0.90, 0.76, 0.001), "flat": (0.21, 0.98, 0.001)},
    "int4_g32+kscale": {"peaky": (0.21, 0.94, 0.001), "decode": (0.53, 0.90, 0.001), "flat": (0.18, 0.985, 0.001)},
    "mxfp4":           {"peaky": (0.29, 0.92, 0.022), "decode": (0.94, 0.75, 0.072), "flat": (0.20, 0.98, 0.011)},
    "mxfp4+kscale":    {"peaky": (0.26, 0.91, 0.022), "decode": (0.78, 0.78, 0.078), "flat": (0.19, 0.98, 0.012)},
}


@pytest.fixture(scope="module")
def acc():
    kv = pkg.CxlSpeckvKVAllocator(pkg.library_path(), "hip:0")
    try:
        return {seed: kv_format_accuracy(kv, seed=seed) for seed in (7001, 7002)}
    finally:
        kv.close()


@pytest.mark.parametrize("fmt", list(BOUNDS))
def test_format_accuracy_on_kv_like_data(acc, fmt):
    for seed, table in acc.items():
        for regime, (rel_max, cos_min, kern_max) in BOUNDS[fmt].items():
            m = table[fmt][regime]
            assert m["rel_l2"] <= rel_max, (fmt, regime, seed, m)
            assert m["cosine_mean"] >= cos_min, (fmt, regime, seed, m)
            assert m["kernel_rel_l2"] <= kern_max, (fmt, regime, seed, m)
            # the split is consistent: total error within the sum of its parts
            assert m["rel_l2"] <= m["format_rel_l2"] + m["kernel_rel_l2"] + 1e-4, (fmt, regime, seed, m)


def test_formats_rank_as_their_bits_say(acc):
    """FP8 beats both 4-bit formats everywhere; on flat and peaky softmax the two 4-bit formats are within a few per cent of each
    other (MXFP4's coarser elements against INT4_G32's coarser scales); where the query weighs K's outlier channels, INT4_G32 with the
    per-channel pre-scale is the better 4-bit pool -- what the connector's documentation says about choosing a scheme."""
    for seed, t in acc.items():
        for regime in ("peaky", "decode", "flat"):
            assert t["fp8_e4m3"][regime]["rel_l2"] < 0.5 * min(t["int4_g32"][regime]["rel_l2"], t["mxfp4"][regime]["rel_l2"]), (seed, regime)
        for regime in ("peaky", "flat"):
            assert abs(t["int4_g32"][regime]["rel_l2"] - t["mxfp4"][regime]["rel_l2"]) <= 0.03, (seed, regime)
        assert t["int4_g32+kscale"]["decode"]["rel_l2"] < 0.75 * t["mxfp4"]["decode"]["rel_l2"], seed
        assert t["int4_g32+kscale"]["decode"]["rel_l2"] < 0.8 * t["int4_g32"]["decode"]["rel_l2"], seed


def test_connector_k_channel_scale_on_an_int4_pool():
    """SpeckvKVConnector.set_k_channel_scale: K / s into the pool, q * s to meet it (powers of two, calibrated from the prompt).  A decode
    loop over an INT4_G32 pool on KV-like data: prefill, appended positions (pairs and the odd tail), attention -- with the pre-scale the
    output is closer to float64 attention over the original K / V where the query weighs K's outlier channels, and kv_rows() hands
    back K in the caller's scaling."""
    import torch
    from cxl_speckv_amd.kv_connector import SpeckvKVConnector
    from cxl_speckv_amd.speckv_ctypes import SpeckvLib
    T, g, n_prompt, n_steps = 1024, 8, 601, 5
    K, V, qs, _ = synth_kv(T, g, 7005)
    sm = 1.0 / np.sqrt(D)
    lib = SpeckvLib(pkg.library_path(), "hip:0")
    try:
        err = {}
        for scaled in (False, True):
            conn = SpeckvKVConnector(lib, num_layers=1, num_kv_heads=8, head_dim=D, max_tokens=T, scheme="int4")
            if scaled:
                want = pow2_channel_scales(K[:n_prompt])[None]                                          # calibrated on the prompt only
                got = conn.calibrate_k_channel_scale(torch.from_numpy(K[None, :n_prompt]))            # ... by the connector itself, on the device
                assert np.array_equal(got.cpu().numpy(), want)
            conn.add_request(1)
            kd, vd = torch.from_numpy(K).cuda(), torch.from_numpy(V).cuda()
            keep = conn.write_prefill(1, kd[None, :n_prompt], vd[None, :n_prompt])
            for t in range(n_prompt, n_prompt + n_steps):
                keep += conn.append([1], kd[t][None, None], vd[t][None, None])
            n = n_prompt + n_steps
            assert conn.length(1) == n
            out = conn.attend(0, [1], torch.from_numpy(qs["decode"]).cuda()[None], sm)
            torch.cuda.synchronize()
            ref = attention_f64(K[:n], V[:n], qs["decode"], sm)[0]
            o = out[0].cpu().numpy().astype(np.float64)
            err[scaled] = float(np.sqrt(((o - ref) ** 2).sum() / (ref ** 2).sum()))
            rows = conn.kv_rows(1, 0, 0).cpu().numpy()
            assert rows.shape == (n, 8, D)
            # what comes back is INT4-quantised K in the caller's scaling: close to K channel by channel (a forgotten scale-back would be off by 2x .. 16x)
            assert np.abs(rows.astype(np.float32) - K[:n].astype(np.float32)).max() <= 0.16 * np.abs(K[:n].astype(np.float32)).max()
            conn.free_request(1)
        assert err[True] < 0.8 * err[False] and err[True] <= 0.55, err
        with pytest.raises(ValueError):
            c2 = SpeckvKVConnector(lib, num_layers=1, max_tokens=T, scheme="int4")
            c2.set_k_channel_scale(torch.full((1, 8, D), 3.0))                                              # not a power of two
    finally:
        lib.finalize()
