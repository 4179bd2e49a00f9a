"""-m gpu: BASELINE.json configs at their OWN size against the oracle.

  configs[4]  int4 / fp8 KV, 70B-shaped (80 layers, 8 kv heads x 128) @ 32k context: the fused attention of all 80
              layers (arithmetic-address form and the page-table form, default split counts) against the oracle's
              float64 attention on sampled (layer, head) rows; fetch + decompress of sampled pages bit-exact.
  configs[3]  shape of a decode step: 256 sequences @ 8k context, batch and planned (graph-capturable) forms of the
              fused attention against the oracle on sampled sequences.
  configs[1]  all 131 072 blocks of the 8B-shaped round trip against the (threaded) C oracle, bit for bit.

The oracle's checkers are themselves pinned: the codec half to the reference (tests/test_oracle_vs_ref.py), INT4_G32
and both attention checkers to a numpy float64 restatement of the format description (tests/test_a22_format_pin.py).
Stated tolerances are those of tests/test_gpu_engine.py (test_int4_fused_attention, test_fp8_fused_attention)."""
import os

import numpy as np
import pytest

import cxl_speckv_amd as pkg
from tests._gpu import N, assert_same_float_bits, torch_mod, set_tuning

pytestmark = pytest.mark.gpu
PAGE = 4096
H, D, G = 8, 128, 8                        # 70B-shaped: 8 kv heads x 128, 8 query heads per kv head


@pytest.fixture()
def eng():
    kv = pkg.CxlSpeckvKVAllocator(pkg.library_path(), "hip:0")
    yield kv
    kv.close()


def synth_pages(torch, seed, n_pages):
    """N(0,1) values with a per-page magnitude in [0.2, 3): generated on the GPU, identical for a given seed."""
    g = torch.Generator(device="cuda")
    g.manual_seed(seed)
    x = torch.randn((n_pages, N), generator=g, device="cuda", dtype=torch.float32)
    mag = torch.rand((n_pages, 1), generator=g, device="cuda", dtype=torch.float32) * 2.8 + 0.2
    return (x * mag).to(torch.float16)


class HeadChecker:
    """The oracle's attention of one kv head over the K / V regions of one layer, from the HOST copy of that layer's
    pages (region = T/2 pages of K followed by T/2 pages of V), through the oracle's own compress -> records."""

    def __init__(self, oracle, scheme, region_pages16, T):
        self.oracle, self.scheme, self.T = oracle, scheme, T
        self.scales, self.lens, self.recs = oracle.compress_blocks_f16(region_pages16, scheme, 0)
        if scheme == 3:
            self.dec = oracle.decompress_blocks_f16(self.recs, self.lens, self.scales, 3, 0).reshape(-1, 2, H, D)
        elif scheme == 5:
            self.lut = np.array([oracle.lib.orc_e4m3_to_f32(b) for b in range(256)], np.float64)
            self.lut[np.isnan(self.lut)] = 0.0
        else:
            self.lut = np.array([oracle.lib.orc_e4m3_to_f32(b) for b in range(256)], np.float32)
            self.lut[np.isnan(self.lut)] = 0.0

    def want(self, q_head, head, npos, sm):
        """q_head [G][D] fp16 -> out [G][D], lse [G], mag [G][D], delta (FP8: score error bound of the fp8 MFMA)."""
        from oracle.bindings import _ptr, u8p, u16p, f32p
        L = self.oracle.lib
        hp = self.T // 2                                               # pages per K / V region
        o = np.zeros((G, D), np.float32); l = np.zeros(G, np.float32); m = np.zeros((G, D), np.float32)
        if npos == 0:
            return o, np.full(G, -np.inf, np.float32), m, 0.0
        if self.scheme == 3:
            k16 = np.ascontiguousarray(self.dec[:hp, :, head, :].reshape(-1, D)[:npos]).view(np.uint16)
            v16 = np.ascontiguousarray(self.dec[hp:2 * hp, :, head, :].reshape(-1, D)[:npos]).view(np.uint16)
            L.orc_attend_f16(_ptr(np.ascontiguousarray(q_head).view(np.uint16).reshape(-1), u16p), G, _ptr(k16.reshape(-1), u16p),
                             _ptr(v16.reshape(-1), u16p), npos, D, float(sm), _ptr(o, f32p), _ptr(l, f32p), _ptr(m, f32p))
            return o, l, m, 0.0
        if self.scheme == 5:                                         # MXFP4: page rows of one head + their codes (tests/test_gpu_mx4.py)
            from tests.test_gpu_mx4 import head_rows, dequant_rows
            kr, kc = head_rows(self.recs, 0, npos, head)
            vr, vc = head_rows(self.recs, hp, npos, head)
            q8 = np.zeros((G, D), np.uint8); qc = np.zeros((G, D // 16), np.uint8)
            L.orc_quantize_rows_mxfp8(_ptr(np.ascontiguousarray(q_head).view(np.uint16).reshape(-1), u16p), G, D, 16, _ptr(q8, u8p), _ptr(qc, u8p))
            qd = self.lut[q8] * np.repeat(np.exp2(qc.astype(np.float64) - 127.0), 16, axis=1)
            delta = 3e-5 * float((np.abs(qd) @ np.abs(dequant_rows(kr, kc, npos)).T).max()) * sm
            L.orc_attend_mx4(_ptr(q8, u8p), _ptr(qc, u8p), 16, G, _ptr(kr, u8p), _ptr(kc, u8p), _ptr(vr, u8p), _ptr(vc, u8p), npos, D,
                             float(sm), _ptr(o, f32p), _ptr(l, f32p), _ptr(m, f32p))
            return o, l, m, delta
        r4 = self.recs[:, :N].reshape(-1, 2, H, D)
        krows = np.ascontiguousarray(r4[:hp, :, head, :].reshape(-1, D)[:npos])
        vrows = np.ascontiguousarray(r4[hp:2 * hp, :, head, :].reshape(-1, D)[:npos])
        ksc = np.ascontiguousarray(np.repeat(self.scales[:hp], 2)[:npos]); vsc = np.ascontiguousarray(np.repeat(self.scales[hp:2 * hp], 2)[:npos])
        q8 = np.zeros((G, D), np.uint8); qs = np.zeros(G, np.float32)
        L.orc_quantize_rows_e4m3(_ptr(np.ascontiguousarray(q_head).view(np.uint16).reshape(-1), u16p), G, D, _ptr(q8, u8p), _ptr(qs, f32p))
        smag = (np.abs(self.lut[q8]) @ np.abs(self.lut[krows]).T) * ksc[None, :] * qs[:, None] * sm
        delta = 3e-5 * float(smag.max())
        L.orc_attend_fp8(_ptr(q8, u8p), _ptr(qs, f32p), G, _ptr(krows, u8p), _ptr(ksc, f32p), _ptr(vrows, u8p), _ptr(vsc, f32p),
                         npos, D, float(sm), _ptr(o, f32p), _ptr(l, f32p), _ptr(m, f32p))
        return o, l, m, delta

    def check(self, got, got_lse, q_head, head, npos, sm, what):
        want, wlse, mag, delta = self.want(q_head, head, npos, sm)
        err = np.abs(np.asarray(got, np.float32) - want)
        tol = (2e-3 + 2 * delta) * mag + 1e-6
        assert np.all(err <= tol), (what, float((err / (mag + 1e-9)).max()), delta)
        if got_lse is not None and npos:
            assert np.all(np.abs(np.asarray(got_lse, np.float32) - wlse) <= 2e-3 + delta), (what, float(np.abs(got_lse - wlse).max()))


def sample_seed():
    """Which layers / heads / sequences the full-size tests check against the oracle rotates from run to run (the edge cases stay):
    SPECKV_SAMPLE_SEED pins it; the seed in use is part of every assertion message of these tests and is printed in the
    run's terminal summary whether the test passes or not (tests/conftest.py)."""
    import time
    env = os.environ.get("SPECKV_SAMPLE_SEED")
    seed = int(env) if env else int(time.time_ns() // 1_000_000) % (2 ** 31)
    from conftest import SAMPLE_SEEDS
    SAMPLE_SEEDS.append((os.environ.get("PYTEST_CURRENT_TEST", "?").split("::")[-1].split(" ")[0], seed))
    return seed


@pytest.mark.parametrize("scheme", [3, 4, 5])
def test_config5_70b_shaped_32k_context_at_full_size(eng, oracle, scheme):
    """BASELINE configs[4]: 80 layers x 32 768 positions x 8 kv heads x 128 in INT4_G32 (3.0 GB of records),
    FP8_E4M3 (5.4 GB) or MXFP4 (2.9 GB): one launch over all layers in the default geometry (the code that only runs at this size: 256-tile
    splits, split counts rounded to a multiple of 8, the merge over many splits), the page-table form of the same, a
    per-layer call, and sampled pages through fetch + decompress."""
    torch = torch_mod()
    lib = eng.lib
    T, L = 32768, 80
    lib.set_compression_scheme(scheme)
    h = eng.allocate(T, L, H, D, 2)
    layer_pages = T                                               # K + V pages of one layer
    seed = sample_seed()
    srng = np.random.default_rng(seed)
    mid = int(srng.integers(1, L - 1))                           # a layer in the middle: another one every run
    sampled = {0: None, mid: None, 79: None}
    for layer in range(L):
        x = synth_pages(torch, 2005_000 + layer, layer_pages)
        lib.write(h, layer * layer_pages * PAGE, x.data_ptr(), x.numel() * 2, True)
        if layer in sampled:
            sampled[layer] = x.cpu().numpy()
        del x
    gq = torch.Generator(device="cuda"); gq.manual_seed(2005)
    q = (torch.randn((L, H, G, D), generator=gq, device="cuda") * 1.5).to(torch.float16)
    qh = q.cpu().numpy()
    sm = 1.0 / np.sqrt(D)
    attend = {3: lib.attend_int4, 4: lib.attend_fp8, 5: lib.attend_mx4}[scheme]
    checkers = {layer: HeadChecker(oracle, scheme, pages, T) for layer, pages in sampled.items()}
    mh = int(srng.integers(0, H))
    heads = {0: tuple(int(v) for v in srng.choice(H, 2, replace=False)), mid: (mh,), 79: (7, int(srng.integers(0, 7)))}

    def run(general, layer0=0, n_layers=L, pos_end=T):
        if general: set_tuning("attend_general", str(int(general)))      # 1, 2: the table form of the fast kernel (record addresses from the page table)
        try:
            out = torch.full((n_layers, H, G, D), float("nan"), dtype=torch.float32, device="cuda")
            lse = torch.full((n_layers, H, G), float("nan"), dtype=torch.float32, device="cuda")
            attend(h, layer0, n_layers, q[layer0:layer0 + n_layers].data_ptr(), G, 0, pos_end, sm, out.data_ptr(), lse.data_ptr())
            torch.cuda.synchronize()
        finally:
            set_tuning("attend_general", 0)
        return out.cpu().numpy(), lse.cpu().numpy()

    for general in ((0, 2) if scheme == 5 else (0, 1, 2)):          # MXFP4 has one page-table form
        out, lse = run(general)
        assert np.isfinite(out).all() and np.isfinite(lse).all()
        for layer, hs in heads.items():
            for head in hs:
                checkers[layer].check(out[layer, head], lse[layer, head], qh[layer, head], head, T, sm,
                                      ("all layers", ("linear", "page table", "table form")[general], layer, head, "sample seed", seed))
    # one layer by itself (a per-layer decode call: other split geometry), and a context that ends inside a tile
    out, lse = run(False, mid, 1)
    checkers[mid].check(out[0, mh], lse[0, mh], qh[mid, mh], mh, T, sm, ("one layer alone", mid, mh, "sample seed", seed))
    out, lse = run(False, 79, 1, 32768 - 30)
    checkers[79].check(out[0, 7], lse[0, 7], qh[79, 7], 7, T - 30, sm, "layer 79, 32738 positions")
    # the table form on a ragged range, an odd number of tiles per split and a single layer
    out, lse = run(2, 79, 1, 32768 - 30)
    checkers[79].check(out[0, 7], lse[0, 7], qh[79, 7], 7, T - 30, sm, "layer 79, 32738 positions, table form")
    out, lse = run(2, mid, 1, 32768 - 96)
    checkers[mid].check(out[0, mh], lse[0, mh], qh[mid, mh], mh, T - 96, sm, ("32672 positions (1021 tiles), table form", mid, mh, "sample seed", seed))
    # fetch + decompress of sampled pages of the same allocation, bit for bit
    rng = np.random.default_rng(seed)
    for layer, pages16 in sampled.items():
        idx = np.sort(rng.choice(layer_pages, 1536, replace=False)).astype(np.uint32)
        d_idx = torch.from_numpy((idx + layer * layer_pages).astype(np.int64)).to(torch.int32).cuda()
        got = torch.empty((idx.size, N), dtype=torch.float16, device="cuda")
        lib.fetch_list(h, d_idx.data_ptr(), idx.size, got.data_ptr(), False)
        torch.cuda.synchronize()
        c = checkers[layer]
        want = oracle.decompress_blocks_f16(c.recs[idx], c.lens[idx], c.scales[idx], scheme, 0)
        assert_same_float_bits(got.cpu().numpy(), want, f"layer {layer} pages (sample seed {seed})")
        info = lib.translate(h, (layer * layer_pages + int(idx[7])) * PAGE)
        assert info.rec_bytes == c.lens[idx[7]] and np.float32(info.scale).tobytes() == c.scales[idx[7]].tobytes()
    lib.free(h)


@pytest.mark.parametrize("scheme", [5, 3, 4])
def test_config4_layout_70b_shaped_32k_context_striped_over_7_pools(oracle, scheme):
    """BASELINE configs[3]'s pool layout (1 compute + 7 pool GPUs; here seven runs on this GPU) at configs[4]'s size: 80 layers x 32 768
    positions of MXFP4 (or INT4_G32, or FP8_E4M3) KV striped page by page over the 7 runs -- the fused attention of all layers in one launch (the form that takes
    the range's pages by residue class: classes of 2341 and 2340 pages, ragged last tiles), a per-layer call on a range that does
    not start at 0, and sampled pages through fetch + decompress, against the oracle on sampled (layer, head) rows."""
    torch = torch_mod()
    os.environ["SPECKV_POOL_DEVICES"] = "0,0,0,0,0,0,0"
    try:
        kv = pkg.CxlSpeckvKVAllocator(pkg.library_path(), "hip:0")
    finally:
        os.environ.pop("SPECKV_POOL_DEVICES", None)
    try:
        lib = kv.lib
        T, L = 32768, 80
        attend = {3: lib.attend_int4, 4: lib.attend_fp8, 5: lib.attend_mx4}[scheme]
        lib.set_compression_scheme(scheme)
        h = kv.allocate(T, L, H, D, 2)
        layer_pages = T
        seed = sample_seed()
        srng = np.random.default_rng(seed)
        mid = int(srng.integers(1, L - 1))
        sampled = {0: None, mid: None, 79: None}
        for layer in range(L):
            x = synth_pages(torch, 2006_000 + layer, layer_pages)
            lib.write(h, layer * layer_pages * PAGE, x.data_ptr(), x.numel() * 2, True)
            if layer in sampled:
                sampled[layer] = x.cpu().numpy()
            del x
        assert {lib.translate(h, p * PAGE).pool_device for p in range(8)} == {0}        # (every "peer" is this GPU)
        gq = torch.Generator(device="cuda"); gq.manual_seed(2006)
        q = (torch.randn((L, H, G, D), generator=gq, device="cuda") * 1.5).to(torch.float16)
        qh = q.cpu().numpy()
        sm = 1.0 / np.sqrt(D)
        checkers = {layer: HeadChecker(oracle, scheme, pages, T) for layer, pages in sampled.items()}
        mh = int(srng.integers(0, H))
        heads = {0: tuple(int(v) for v in srng.choice(H, 2, replace=False)), mid: (mh,), 79: (7, int(srng.integers(0, 7)))}
        out = torch.full((L, H, G, D), float("nan"), dtype=torch.float32, device="cuda")
        lse = torch.full((L, H, G), float("nan"), dtype=torch.float32, device="cuda")
        attend(h, 0, L, q.data_ptr(), G, 0, T, sm, out.data_ptr(), lse.data_ptr())
        torch.cuda.synchronize()
        o, l_ = out.cpu().numpy(), lse.cpu().numpy()
        assert np.isfinite(o).all() and np.isfinite(l_).all()
        for layer, hs in heads.items():
            for head in hs:
                checkers[layer].check(o[layer, head], l_[layer, head], qh[layer, head], head, T, sm, ("striped x7, all layers", layer, head, "sample seed", seed))
        # one layer by itself on a ragged range (32738 positions: classes of unequal length)
        attend(h, 79, 1, q[79].data_ptr(), G, 0, T - 30, sm, out.data_ptr(), lse.data_ptr())
        torch.cuda.synchronize()
        checkers[79].check(out[0, 7].cpu().numpy(), lse[0, 7].cpu().numpy(), qh[79, 7], 7, T - 30, sm, ("striped x7, layer 79, 32738 positions", "sample seed", seed))
        rng = np.random.default_rng(seed)
        for layer, pages16 in sampled.items():
            idx = np.sort(rng.choice(layer_pages, 512, replace=False)).astype(np.uint32)
            d_idx = torch.from_numpy((idx + layer * layer_pages).astype(np.int64)).to(torch.int32).cuda()
            got = torch.empty((idx.size, N), dtype=torch.float16, device="cuda")
            lib.fetch_list(h, d_idx.data_ptr(), idx.size, got.data_ptr(), False)
            torch.cuda.synchronize()
            c = checkers[layer]
            want = oracle.decompress_blocks_f16(c.recs[idx], c.lens[idx], c.scales[idx], scheme, 0)
            assert_same_float_bits(got.cpu().numpy(), want, f"striped x7, layer {layer} pages (sample seed {seed})")
        lib.free(h)
    finally:
        kv.close()


@pytest.mark.parametrize("scheme", [3, 4, 5])
def test_config4_decode_step_256_sequences_8k_context(eng, oracle, scheme):
    """BASELINE configs[3]'s decode step on the config-5 formats: 256 sequences (one allocation each, 8 192 positions
    of two layers), lengths from empty to full, the batch form and the planned form of the fused attention against the
    oracle on sampled sequences."""
    torch = torch_mod()
    lib = eng.lib
    T, L, NSEQ = 8192, 2, 256
    lib.set_compression_scheme(scheme)
    rng = np.random.default_rng(2004)
    lens = [8192] * NSEQ
    for i, n in ((3, 0), (9, 2), (17, 8190), (40, 4098), (77, 6144), (130, 32), (200, 1000), (255, 8192 - 34)):
        lens[i] = n
    for i in range(100, 120):
        lens[i] = int(rng.integers(1, 4096)) * 2
    seed = sample_seed()
    srng = np.random.default_rng(seed)
    # the edge cases (a two-position sequence's neighbour, ragged ends, a random length) and three sequences that rotate
    sampled = {i: None for i in [17, 40, 111] + [int(v) for v in srng.choice([j for j in range(NSEQ) if j not in (3, 9, 17, 40, 111)], 3, replace=False)]}
    check_heads = tuple(int(v) for v in srng.choice(H, 2, replace=False))
    handles = []
    n_pages = T * L * H * D * 2 * 2 // PAGE
    for i in range(NSEQ):
        hnd = lib.alloc(n_pages * PAGE)
        lib.set_layout(hnd, T, L, H, D, 2)
        x = synth_pages(torch, 2004_000 + i, n_pages)
        lib.write(hnd, 0, x.data_ptr(), x.numel() * 2, True)
        if i in sampled:
            sampled[i] = x[T:2 * T].cpu().numpy()               # layer 1: its K region then its V region
        handles.append(hnd)
        del x
    gq = torch.Generator(device="cuda"); gq.manual_seed(2004)
    q = (torch.randn((NSEQ, H, G, D), generator=gq, device="cuda") * 1.5).to(torch.float16)
    qh = q.cpu().numpy()
    sm = 1.0 / np.sqrt(D)
    checkers = {i: HeadChecker(oracle, scheme, pages, T) for i, pages in sampled.items()}
    layer = 1

    def check(out, lse, what):
        assert float(np.abs(out[3]).max()) == 0.0                # the empty sequence
        for i, c in checkers.items():
            for head in check_heads:
                c.check(out[i, head], lse[i, head], qh[i, head], head, lens[i], sm, (what, i, head, "sample seed", seed))

    batch = {3: lib.attend_int4_batch, 4: lib.attend_fp8_batch, 5: lib.attend_mx4_batch}[scheme]
    out = torch.full((NSEQ, H, G, D), float("nan"), dtype=torch.float32, device="cuda")
    lse = torch.full((NSEQ, H, G), float("nan"), dtype=torch.float32, device="cuda")
    batch(handles, layer, q.data_ptr(), G, lens, sm, out.data_ptr(), lse.data_ptr())
    torch.cuda.synchronize()
    o1, l1 = out.cpu().numpy(), lse.cpu().numpy()
    check(o1, l1, "batch")
    # planned form: descriptors on the device, kernel launches only
    st = torch.cuda.Stream()
    plan_bytes = lib.attend_plan_bytes(NSEQ)
    d_plan = torch.empty(plan_bytes, dtype=torch.uint8, device="cuda")
    out.fill_(float("nan")); lse.fill_(float("nan"))
    torch.cuda.synchronize()
    lib.attend_batch_plan(handles, lens, T, d_plan.data_ptr(), plan_bytes, st.cuda_stream)
    lib.attend_planned(scheme, d_plan.data_ptr(), NSEQ, layer, q.data_ptr(), G, T, sm, out.data_ptr(), lse.data_ptr(), st.cuda_stream)
    torch.cuda.synchronize()
    check(out.cpu().numpy(), lse.cpu().numpy(), "planned")
    for hnd in handles:
        lib.free(hnd)


@pytest.mark.parametrize("T", [1024, 2048])
@pytest.mark.parametrize("scheme", [3, 4, 5])
def test_short_context_decode_step_256_sequences(eng, oracle, scheme, T):
    """The same decode step at SHORT contexts (256 sequences x 1k / 2k positions, one layer each): the shapes where a launch's
    fixed part weighs most and the batch forms differ most from the long-context ones (INT4: the two-halves workgroups with 16- and
    32-tile halves; FP8: whole sequences per workgroup, the first tile requested in front of the query).  Lengths from empty to
    full with ragged ends; batch and planned forms against the oracle on sampled sequences (rotating sample)."""
    torch = torch_mod()
    lib = eng.lib
    L, NSEQ = 1, 256
    lib.set_compression_scheme(scheme)
    seed = sample_seed()
    srng = np.random.default_rng(seed)
    rng = np.random.default_rng(3000 + T)
    lens = [T] * NSEQ
    for i, n in ((3, 0), (9, 2), (17, T - 2), (40, T // 2 + 2), (77, 34), (130, 32), (200, 30), (255, T - 34)):
        lens[i] = n
    for i in range(100, 120):
        lens[i] = int(rng.integers(1, T // 2)) * 2
    sampled = {i: None for i in [17, 40, 77, 111, 255] + [int(v) for v in srng.choice([j for j in range(NSEQ) if j not in (3, 9, 17, 40, 77, 111, 255)], 3, replace=False)]}
    check_heads = tuple(int(v) for v in srng.choice(H, 2, replace=False))
    handles = []
    n_pages = T * L * H * D * 2 * 2 // PAGE
    for i in range(NSEQ):
        hnd = lib.alloc(n_pages * PAGE)
        lib.set_layout(hnd, T, L, H, D, 2)
        x = synth_pages(torch, 3000_000 + 7 * T + i, n_pages)
        lib.write(hnd, 0, x.data_ptr(), x.numel() * 2, True)
        if i in sampled:
            sampled[i] = x.cpu().numpy()                         # layer 0: its K region then its V region
        handles.append(hnd)
        del x
    gq = torch.Generator(device="cuda"); gq.manual_seed(3000 + T)
    q = (torch.randn((NSEQ, H, G, D), generator=gq, device="cuda") * 1.5).to(torch.float16)
    qh = q.cpu().numpy()
    sm = 1.0 / np.sqrt(D)
    checkers = {i: HeadChecker(oracle, scheme, pages, T) for i, pages in sampled.items()}

    def check(out, lse, what):
        assert float(np.abs(out[3]).max()) == 0.0                # the empty sequence
        for i, c in checkers.items():
            for head in check_heads:
                c.check(out[i, head], lse[i, head], qh[i, head], head, lens[i], sm, (what, T, i, head, "sample seed", seed))

    batch = {3: lib.attend_int4_batch, 4: lib.attend_fp8_batch, 5: lib.attend_mx4_batch}[scheme]
    out = torch.full((NSEQ, H, G, D), float("nan"), dtype=torch.float32, device="cuda")
    lse = torch.full((NSEQ, H, G), float("nan"), dtype=torch.float32, device="cuda")
    batch(handles, 0, q.data_ptr(), G, lens, sm, out.data_ptr(), lse.data_ptr())
    torch.cuda.synchronize()
    check(out.cpu().numpy(), lse.cpu().numpy(), "batch")
    st = torch.cuda.Stream()
    plan_bytes = lib.attend_plan_bytes(NSEQ)
    d_plan = torch.empty(plan_bytes, dtype=torch.uint8, device="cuda")
    out.fill_(float("nan")); lse.fill_(float("nan"))
    torch.cuda.synchronize()
    lib.attend_batch_plan(handles, lens, T, d_plan.data_ptr(), plan_bytes, st.cuda_stream)
    lib.attend_planned(scheme, d_plan.data_ptr(), NSEQ, 0, q.data_ptr(), G, T, sm, out.data_ptr(), lse.data_ptr(), st.cuda_stream)
    torch.cuda.synchronize()
    check(out.cpu().numpy(), lse.cpu().numpy(), "planned")
    for hnd in handles:
        lib.free(hnd)


def test_config2_all_131072_blocks_against_the_oracle(eng, oracle):
    """BASELINE configs[1]: the 8B-shaped round trip, every one of its 131 072 blocks checked against the C oracle
    (record lengths, scale bits through the page table on a sample, decoded bits of ALL blocks) -- INT8_DELTA_RLE in
    both quantiser modes, with structured blocks mixed in."""
    torch = torch_mod()
    lib = eng.lib
    B = 131072
    x = synth_pages(torch, 2001, B)
    x[5::97] = 0
    x[7::101] = x[7::101, :64].repeat_interleave(32, dim=1)
    xh = x.cpu().numpy()
    out = torch.empty_like(x)
    for mode in (0, 1):
        lib.set_quant_mode(mode)
        lib.set_compression_scheme(2)
        h = eng.allocate(4096, 32, H, D, 2)
        lib.write(h, 0, x.data_ptr(), x.numel() * 2, True)
        lib.fetch_range(h, 0, B, out.data_ptr(), False)
        torch.cuda.synchronize()
        scales, lens, recs = oracle.compress_blocks_f16(xh, 2, mode)
        want = oracle.decompress_blocks_f16(recs, lens, scales, 2, mode)
        assert_same_float_bits(out.cpu().numpy(), want, f"mode {mode}")
        assert lib.stats().compressed_bytes == int(lens.astype(np.int64).sum())
        for p in (0, 5, 7, 4097, B - 1):
            info = lib.translate(h, p * PAGE)
            assert info.rec_bytes == lens[p] and np.float32(info.scale).tobytes() == scales[p].tobytes()
        lib.free(h)
    lib.set_quant_mode(0)
