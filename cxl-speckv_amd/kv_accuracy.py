"""What the pool formats cost in attention accuracy, on data shaped like real KV (VERDICT r5 missing #2; the reference's claims:
docs/ARCHITECTURE.md:246 "Accuracy: 99.5 % preservation", README.md:18 "3-4x compression ratio with minimal accuracy loss" -- it
publishes no measurement and its codec is INT8, so these are figures for OUR pool formats).

N(0,1) blocks flatter every block format.  Real K / V have structure that block scaling reacts to:
  * K: a few channels per head carry values 10-50 x the rest (a bias that survives RoPE in the slow-rotating pairs), RoPE pairs
    (i, i + D/2) rotate with the position, per-channel magnitudes spread log-normally;
  * V: heavy tails (Student t, 4 degrees of freedom) on log-normal channel scales;
  * q: the same outlier channels as K (the model learns them together); softmax either PEAKY (one key takes most of the mass: a
    retrieval head) or FLAT (hundreds of keys share it: an averaging head), or in between (a decode-like query).
The reference result is float64 attention over the ORIGINAL fp16 K / V with the UNQUANTISED fp16 query -- nothing of the pool
format or of the kernel's own query quantisation is in it.  Measured per format: relative L2 error and cosine of the fused
attention's output rows, top-1 agreement of the attention weights (the position with the largest score, from the format's
dequantised K and the exact query), and the error split in two: `format_rel_l2` = float64 attention over the DEQUANTISED K / V
with the exact query against the reference (what 8 or 4 bits of K and V cost, whatever the kernel), `kernel_rel_l2` = the fused
kernel against that same float64 attention over the dequantised values (what the kernel adds: its query quantisation --
e4m3 per row for FP8, MXFP8 blocks for MXFP4, none for INT4_G32 -- f16 weights, fp32 accumulation).

`kv_format_accuracy(kv)` runs it on a CxlSpeckvKVAllocator over the HIP engine: bench.py prints the table (`kv_format_accuracy`),
tests/test_gpu_accuracy.py holds thresholds on it.  numpy float64 is the checker here; no oracle, no reference involved.
"""
import numpy as np

H, D, PAGE, N = 8, 128, 4096, 2048
SCHEMES = {"fp8_e4m3": 4, "int4_g32": 3, "mxfp4": 5}
REGIMES = ("peaky", "decode", "flat")


def synth_kv(T=2048, g=8, seed=7001, outlier_channels=2, outlier_lo=10.0, outlier_hi=50.0):
    """K, V fp16 [T][H][D] and fp16 queries {regime: [H][g][D]} with the structure described in the module text (seeded)."""
    rng = np.random.default_rng(seed)
    half = D // 2
    theta = 10000.0 ** (-np.arange(half) / half)                               # RoPE frequencies of the pairs (i, i + D/2)
    pos = np.arange(T)[:, None]
    ch_scale = np.exp(rng.normal(0.0, 0.5, (H, D)))                            # log-normal channel magnitudes
    bias = np.zeros((H, D))
    out_ch = np.zeros((H, outlier_channels), np.int64)
    for h in range(H):
        # outliers sit in slow pairs (high pair index): the rotation over T positions is a small angle, the bias survives
        pairs = rng.choice(np.arange(half * 3 // 4, half), outlier_channels, replace=False)
        out_ch[h] = pairs + half * rng.integers(0, 2, outlier_channels)
        bias[h, out_ch[h]] = rng.uniform(outlier_lo, outlier_hi, outlier_channels) * rng.choice([-1.0, 1.0], outlier_channels)
    k0 = rng.standard_normal((T, H, D)) * ch_scale + bias                      # content before the rotation
    ang = pos * theta                                                          # [T][half]
    c, s = np.cos(ang)[:, None, :], np.sin(ang)[:, None, :]
    K = np.concatenate([k0[..., :half] * c - k0[..., half:] * s, k0[..., :half] * s + k0[..., half:] * c], axis=-1)
    v_scale = np.exp(rng.normal(0.0, 0.3, (H, D)))
    V = rng.standard_t(4, (T, H, D)) * v_scale
    q = {}
    # decode-like: content with the same outlier channels (at 1-3 x the bulk: with K's 10-50 x that is a common score term of up to
    # 13 nats that moves slowly with the position), rotated to position T
    q0 = rng.standard_normal((H, g, D)) * ch_scale[:, None, :]
    for h in range(H):
        q0[h][:, out_ch[h]] += np.sign(bias[h, out_ch[h]]) * rng.uniform(1.0, 3.0, (g, outlier_channels))
    aT = T * theta
    q["decode"] = np.concatenate([q0[..., :half] * np.cos(aT) - q0[..., half:] * np.sin(aT), q0[..., :half] * np.sin(aT) + q0[..., half:] * np.cos(aT)], axis=-1)
    # peaky: each query row points at one key of its head (minus the head's mean key: the common bias does not pick a position)
    tgt = rng.integers(0, T, (H, g))
    kmean = K.mean(axis=0)
    qp = np.stack([np.stack([K[tgt[h, r], h] - kmean[h] for r in range(g)]) for h in range(H)])
    qp *= (14.0 * np.sqrt(D) / np.maximum((qp * qp).sum(-1, keepdims=True), 1e-9))      # margin of ~14 nats over an average key
    q["peaky"] = qp + rng.standard_normal((H, g, D)) * 0.05
    q["flat"] = rng.standard_normal((H, g, D)) * 0.15
    f16 = lambda a: np.ascontiguousarray(a.astype(np.float16))
    return f16(K), f16(V), {k: f16(v) for k, v in q.items()}, {"targets": tgt, "outlier_channels": out_ch}


def attention_f64(K, V, q, sm_scale):
    """float64 softmax(q.K^T * sm_scale).V per kv head: out [H][g][D], weights' argmax [H][g]"""
    Kd, Vd, qd = K.astype(np.float64), V.astype(np.float64), q.astype(np.float64)
    s = np.einsum("hgd,thd->hgt", qd, Kd) * sm_scale
    top1 = s.argmax(-1)
    p = np.exp(s - s.max(-1, keepdims=True))
    p /= p.sum(-1, keepdims=True)
    return np.einsum("hgt,thd->hgd", p, Vd), top1, p.max(-1)


def pow2_channel_scales(K):
    """Per-(head, channel) power-of-two pre-scale for K: the channel's max|k| relative to the head's median channel, rounded to a
    power of two (exact in fp16 both ways).  K / s is what goes into the pool, q * s is what meets it: q.k is unchanged, and an
    outlier channel no longer sets the block scale for the 15 channels that share its MX block."""
    amax = np.abs(K.astype(np.float32)).max(axis=0)                            # [H][D]
    med = np.median(amax, axis=-1, keepdims=True)
    return np.exp2(np.round(np.log2(np.maximum(amax / np.maximum(med, 1e-9), 1e-9)))).astype(np.float32)


def _rel(a, b):
    return float(np.sqrt(((a.astype(np.float64) - b.astype(np.float64)) ** 2).sum() / (b.astype(np.float64) ** 2).sum()))


def _metrics(got, ref, top1_fmt, top1_ref):
    g64, r64 = got.astype(np.float64), ref.astype(np.float64)
    num = np.sqrt(((g64 - r64) ** 2).sum(-1))
    den = np.sqrt((r64 ** 2).sum(-1))
    cos = (g64 * r64).sum(-1) / np.maximum(np.sqrt((g64 ** 2).sum(-1)) * den, 1e-300)
    return {"rel_l2": float(np.sqrt(((g64 - r64) ** 2).sum() / (r64 ** 2).sum())), "rel_l2_worst_row": float((num / np.maximum(den, 1e-300)).max()),
            "cosine_mean": float(cos.mean()), "cosine_min": float(cos.min()), "top1_agreement": float((top1_fmt == top1_ref).mean())}


def kv_format_accuracy(kv, T=2048, g=8, seed=7001, schemes=None, k_prescale=("mxfp4", "int4_g32")):
    """{format: {regime: metrics}} on the HIP engine behind `kv` (a CxlSpeckvKVAllocator); formats in `k_prescale` are also
    measured with the power-of-two per-channel pre-scale of K folded into the query ("<format>+kscale")."""
    import torch
    lib = kv.lib
    K, V, qs, info = synth_kv(T, g, seed)
    sm = 1.0 / np.sqrt(D)
    ref = {r: attention_f64(K, V, qs[r], sm) for r in REGIMES}
    n_pages = 2 * T * H * D * 2 // PAGE
    out = {"_data": {"T": T, "g": g, "seed": seed, "outlier_channels_per_head": int(info["outlier_channels"].shape[1]),
                     "reference": "float64 attention over the original fp16 K / V, unquantised fp16 query",
                     "softmax_top_weight_mean": {r: round(float(ref[r][2].mean()), 4) for r in REGIMES}}}
    scale = pow2_channel_scales(K)
    variants = []
    for name, scheme in (schemes or SCHEMES).items():
        variants.append((name, scheme, None))
        if name in k_prescale: variants.append((name + "+kscale", scheme, scale))
    st = torch.cuda.current_stream().cuda_stream
    for name, scheme, ks in variants:
        lib.set_compression_scheme(scheme)
        h = kv.allocate(T, 1, H, D, 2)
        Kw = K if ks is None else (K.astype(np.float32) / ks[None]).astype(np.float16)       # (a power of two: exact)
        buf = np.concatenate([Kw.reshape(-1), V.reshape(-1)])
        lib.write(h, 0, buf.ctypes.data, buf.nbytes, False)
        deq = torch.empty((n_pages, N), dtype=torch.float16, device="cuda")
        lib.fetch_range(h, 0, n_pages, deq.data_ptr(), False, st)
        torch.cuda.synchronize()
        flat = deq.cpu().numpy().reshape(-1)
        Kq, Vq = flat[:T * H * D].reshape(T, H, D), flat[T * H * D:].reshape(T, H, D)
        res = {}
        for r in REGIMES:
            qv = qs[r] if ks is None else (qs[r].astype(np.float32) * ks[:, None, :]).astype(np.float16)
            dq = torch.from_numpy(qv.view(np.int16)).cuda()
            o = torch.full((H, g, D), float("nan"), dtype=torch.float32, device="cuda")
            lse = torch.empty((H, g), dtype=torch.float32, device="cuda")
            {4: lib.attend_fp8, 3: lib.attend_int4, 5: lib.attend_mx4}[scheme](h, 0, 1, dq.data_ptr(), g, 0, T, sm, o.data_ptr(), lse.data_ptr(), st)
            torch.cuda.synchronize()
            of, top1_fmt, _ = attention_f64(Kq, Vq, qv, sm)              # the format alone: exact arithmetic over the dequantised values
            m = _metrics(o.cpu().numpy(), ref[r][0], top1_fmt, ref[r][1])
            m["format_rel_l2"] = _rel(of, ref[r][0])
            m["kernel_rel_l2"] = _rel(o.cpu().numpy(), of)
            res[r] = {k: round(v, 5) for k, v in m.items()}
        out[name] = res
        lib.free(h)
    return out


def format_table(acc):
    """markdown rows of kv_format_accuracy's result"""
    rows = ["| format | regime | rel. L2 error (format alone + kernel) | worst row | cosine (mean / min) | top-1 agreement |", "|---|---|---|---|---|---|"]
    for name, res in acc.items():
        if name.startswith("_"): continue
        for r in REGIMES:
            m = res[r]
            rows.append(f"| {name} | {r} | {m['rel_l2']:.4f} ({m['format_rel_l2']:.4f} + {m['kernel_rel_l2']:.4f}) | {m['rel_l2_worst_row']:.4f} | {m['cosine_mean']:.5f} / {m['cosine_min']:.5f} | {m['top1_agreement']:.3f} |")
    return "\n".join(rows)
