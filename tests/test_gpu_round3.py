"""-m gpu: round-3 engine work through the C ABI -- ordering of the engine stream behind asynchronous writes, finalize
against a thread parked inside the engine, the prefetch queue under concurrent callers, the asynchronous write entry
points (speckv_ext_write_async / _write_runs)."""
import os
import threading
import time

import numpy as np
import pytest

import cxl_speckv_amd as pkg
from cxl_speckv_amd.speckv_ctypes import SpeckvError, SpeckvLib
from tests._gpu import N, assert_same_float_bits, dev_to_host, stored_record, torch_mod, set_tuning

pytestmark = pytest.mark.gpu
PAGE = 4096


def open_lib(**env):
    for k, v in env.items():
        os.environ[k] = str(v)
    try:
        return SpeckvLib(pkg.library_path(), "hip:0")
    finally:
        for k in env:
            os.environ.pop(k, None)


def synth(n_pages, seed):
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((n_pages, N))
    x[1::7] = 0.0
    return x.astype(np.float16)


@pytest.mark.parametrize("scheme", [2, 0])
def test_engine_reads_are_ordered_behind_asynchronous_writes(oracle, scheme):
    """a write on the CALLER's stream (write_strided / write_async / write_runs) followed, without any
    host synchronisation, by reads the engine does on its OWN stream -- speckv_access (synchronous miss), speckv_ext_read to
    a host buffer, a prefetch flush -- must see the new records, not the old ones and not half-written ones.  The caller's
    stream is kept busy in front of the write so that an unordered engine stream would win the race every time."""
    torch = torch_mod()
    lib = open_lib()
    try:
        lib.set_compression_scheme(scheme)
        T, L, H, D = 256, 2, 8, 128
        n_pages = T * L * H * D * 2 * 2 // PAGE
        h = lib.alloc(n_pages * PAGE)
        lib.set_layout(h, T, L, H, D, 2)
        old = synth(n_pages, 1)
        lib.write(h, 0, old.ctypes.data, old.nbytes, False)
        new = synth(n_pages, 2)
        d_new = torch.from_numpy(new.view(np.int16)).cuda()
        sc, ln, rc = oracle.compress_blocks_f16(new, scheme, 0)
        want = oracle.decompress_blocks_f16(rc, ln, sc, scheme, 0)
        user = torch.cuda.Stream()
        torch.cuda.synchronize()

        def busy():
            with torch.cuda.stream(user):
                torch.cuda._sleep(int(2.0e8))                      # ~0.1 s in front of the write

        # 1. write_async, then a synchronous miss through the drop-in entry point
        busy()
        lib.write_async(h, 16 * PAGE, d_new[16:48].data_ptr(), 32 * PAGE, user.cuda_stream)
        ptr = lib.access(h, 20 * PAGE, PAGE)
        assert dev_to_host(ptr, PAGE).tobytes() == want[20].tobytes()
        # 2. write_strided, then speckv_ext_read into a HOST buffer (no device-wide wait on that path)
        busy()
        lib.write_strided(h, 64, 3, 8, d_new[64:72].data_ptr(), user.cuda_stream)      # pages 64, 67, .. <- new[64..71]
        got = np.empty((32, N), np.float16)
        lib.read(h, 64 * PAGE, got.ctypes.data, got.nbytes, False)
        for j in range(8):
            assert got[3 * j].tobytes() == want[64 + j].tobytes(), j
        # 3. write_runs (two regions in one launch), then a look-ahead flush that fetches those very pages into L2
        busy()
        lib.write_runs(h, [128, 160], [d_new[128:136].data_ptr(), d_new[160:168].data_ptr()], 8, user.cuda_stream)
        lib.prefetch_batch([0], [0], [1], [12])                     # layer 0, positions 2..13: K pages 1..6, V pages 129..134
        assert lib.prefetch_flush() >= 12
        lib.sync()
        ptr = lib.access(h, 130 * PAGE, PAGE)
        assert dev_to_host(ptr, PAGE).tobytes() == want[130].tobytes()
        assert lib.stats().l2_hits >= 1
        lib.free(h)
    finally:
        lib.finalize()


def test_connector_append_then_block_table_without_a_sync(oracle):
    """The concrete case of the advice: SpeckvKVConnector.append() (asynchronous, side stream) directly followed by
    block_table() (engine stream), and write_prefill as ONE asynchronous launch (speckv_ext_write_runs)."""
    torch = torch_mod()
    from cxl_speckv_amd.kv_connector import SpeckvKVConnector
    lib = open_lib()
    try:
        L, H, D, T = 3, 8, 128, 128
        conn = SpeckvKVConnector(lib, num_layers=L, num_kv_heads=H, head_dim=D, max_tokens=T, scheme="int8_delta_rle")
        gen = torch.Generator(device="cuda"); gen.manual_seed(9)
        rnd = lambda *s: torch.randn(s, generator=gen, device="cuda", dtype=torch.float32).to(torch.float16)
        conn.add_request(1)
        k0, v0 = rnd(L, 21, H, D), rnd(L, 21, H, D)                # odd prompt: position 20 waits in the tail
        held = conn.write_prefill(1, k0, v0)
        rows = {0: [k0[:, p] for p in range(21)], 1: [v0[:, p] for p in range(21)]}
        for step in range(6):
            kn, vn = rnd(1, L, H, D), rnd(1, L, H, D)
            held += conn.append([1], kn, vn)
            rows[0].append(kn[0]); rows[1].append(vn[0])
            n = conn.length(1)
            if n % 2 == 0:                                          # the pair that was just written, read back at once
                for layer in range(L):
                    for kind in (0, 1):
                        addrs = conn.block_table(1, layer, kind, n - 2, n)
                        assert len(addrs) == 2 and all(addrs)
                        src = torch.stack((rows[kind][n - 2][layer], rows[kind][n - 1][layer])).cpu().numpy().reshape(1, N)
                        sc, ln, rc = oracle.compress_blocks_f16(src, 2, 0)
                        want = oracle.decompress_blocks_f16(rc, ln, sc, 2, 0).reshape(2, H * D)
                        for j, a in enumerate(addrs):
                            assert dev_to_host(a, H * D * 2).tobytes() == want[j].tobytes(), (step, layer, kind, j)
        # and the prompt itself, written by one launch
        torch.cuda.synchronize()
        for layer in range(L):
            for kind, t in ((0, k0), (1, v0)):
                got = conn.kv_rows(1, layer, kind)
                torch.cuda.synchronize()
                src = t[layer, :20].cpu().numpy().reshape(10, N)
                sc, ln, rc = oracle.compress_blocks_f16(src, 2, 0)
                want = oracle.decompress_blocks_f16(rc, ln, sc, 2, 0).reshape(20, H, D)
                assert_same_float_bits(got[:20].cpu().numpy(), want)
    finally:
        lib.finalize()


def test_write_entry_point_argument_errors():
    torch = torch_mod()
    lib = open_lib()
    try:
        lib.set_compression_scheme(2)
        h = lib.alloc(64 * PAGE)
        buf = torch.zeros((64, N), dtype=torch.float16, device="cuda")
        st = torch.cuda.Stream()
        with pytest.raises(SpeckvError):
            lib.write_async(h, 100, buf.data_ptr(), PAGE, st.cuda_stream)                 # offset not page aligned
        with pytest.raises(SpeckvError):
            lib.write_async(h, 0, buf.data_ptr(), 65 * PAGE, st.cuda_stream)              # past the end
        with pytest.raises(SpeckvError):
            lib.write_runs(h, [0, 4], [buf.data_ptr(), buf.data_ptr()], 8, st.cuda_stream)   # overlapping runs
        with pytest.raises(SpeckvError):
            lib.write_runs(h, [0, 60], [buf.data_ptr(), buf.data_ptr()], 8, st.cuda_stream)  # second run leaves the allocation
        with pytest.raises(SpeckvError):
            lib.write_runs(h, [0], [buf.data_ptr()], 8, None)                               # needs a stream
        lib.write_async(h, 0, buf.data_ptr(), 0, st.cuda_stream)                            # empty: fine
        lib.write_runs(h, [0, 8, 56], [buf.data_ptr()] * 3, 8, st.cuda_stream)
        torch.cuda.synchronize()
        assert lib.stats().total_compressions == 24
    finally:
        lib.finalize()


def test_finalize_waits_for_a_thread_parked_inside_the_engine():
    """a thread inside speckv_ext_sync has released the ABI mutex while it waits for the GPU; a
    concurrent speckv_finalize must not destroy the engine under it.  It waits until the engine is empty, entries that
    arrive meanwhile see the library as not initialised, and the parked call returns normally."""
    torch = torch_mod()
    lib = open_lib()
    lib.set_compression_scheme(2)
    n = 64
    h = lib.alloc(n * PAGE)
    x = synth(n, 3)
    d_x = torch.from_numpy(x.view(np.int16)).cuda()
    user = torch.cuda.Stream()
    results = {}

    def parked():
        try:
            # a synchronous miss: its fetch is ordered behind the asynchronous write below, which sits behind ~0.5 s of
            # work on the caller's stream -- the thread waits for the GPU inside the engine with the ABI lock released
            ptr = lib.access(h, 5 * PAGE, PAGE)
            results["parked"] = "ok" if ptr else "null pointer"      # (the pointer dies with the engine: not read here)
        except Exception as e:          # noqa: BLE001
            results["parked"] = repr(e)

    torch.cuda.synchronize()
    with torch.cuda.stream(user):
        torch.cuda._sleep(int(1.0e9))                              # ~0.5 s
    lib.write_async(h, 0, d_x.data_ptr(), n * PAGE, user.cuda_stream)
    t = threading.Thread(target=parked)
    t.start()
    time.sleep(0.15)                                               # the thread is inside the engine, waiting for the GPU
    t0 = time.perf_counter()
    lib.finalize()                                                 # must wait for it, not free the engine under it
    t.join(timeout=30)
    assert not t.is_alive()
    assert results.get("parked") == "ok", results
    assert time.perf_counter() - t0 < 20
    # the library is really finalized and can be opened again
    with pytest.raises(SpeckvError):
        lib.alloc(PAGE)
    lib2 = open_lib()
    try:
        assert lib2.alloc(PAGE) == 1
    finally:
        lib2.finalize()


def test_prefetch_from_two_threads_while_flushes_run(oracle):
    """speckv_prefetch from one thread while another thread's flush has let go of the ABI lock used to
    append to the very vectors that flush was copying from, and the flush then cleared them.  Two threads enqueue and
    flush concurrently for a while: no request may be lost (every page either fetched or already resident or counted as
    dropped), and nothing crashes."""
    torch = torch_mod()
    lib = open_lib(SPECKV_L2_MB=64)
    try:
        lib.set_compression_scheme(2)
        T, L, H, D = 2048, 4, 8, 128
        n_pages = T * L * H * D * 2 * 2 // PAGE
        handles = []
        for r in range(2):
            h = lib.alloc(n_pages * PAGE)
            lib.set_layout(h, T, L, H, D, 2)
            lib.bind_request(100 + r, h, 0)
            handles.append(h)
        errors = []

        def worker(r):
            try:
                rng = np.random.default_rng(r)
                for it in range(300):
                    pos = int(rng.integers(0, T - 40))
                    for layer in range(L):
                        lib.prefetch(100 + r, layer, pos, 4, list(range(1, 17)))
                    if it % 3 == 0:
                        lib.prefetch_flush(want_count=(it % 2 == 0))
            except Exception as e:      # noqa: BLE001
                errors.append(repr(e))

        ts = [threading.Thread(target=worker, args=(r,)) for r in range(2)]
        for t in ts: t.start()
        for t in ts: t.join(timeout=120)
        assert not any(t.is_alive() for t in ts) and not errors, errors
        lib.prefetch_flush()
        lib.sync()
        st = lib.stats()
        assert st.prefetch_dropped == 0
        assert st.total_prefetches > 0 and st.dma_completed >= st.total_prefetches
        # the look-ahead of a fresh position still lands and is served from L2
        lib.prefetch(100, 2, 1000, 4, list(range(1, 17)))
        lib.prefetch_flush()
        lib.sync()
        hits0 = lib.stats().l2_hits
        row = H * D * 2
        off = ((2 * 2 + 0) * T + 1002) * row
        assert lib.access(handles[0], off, 256)
        assert lib.stats().l2_hits == hits0 + 1
    finally:
        lib.finalize()


def test_decode_loop_does_not_accumulate_flights_or_events(oracle):
    """a decode loop that only appends, flushes and attends never called anything that retired finished
    flushes; completions are now harvested by the flush itself and counted when a flight finishes."""
    torch = torch_mod()
    lib = open_lib()
    try:
        lib.set_compression_scheme(2)
        T, L, H, D = 1024, 2, 8, 128
        h = lib.alloc(T * L * H * D * 2 * 2)
        lib.set_layout(h, T, L, H, D, 2)
        total = 0
        for it in range(200):
            lib.prefetch_batch([0, 0], [0, 1], [4 * it % (T - 16)] * 2, [4, 4])
            n = lib.prefetch_flush()
            total += n
        lib.sync()
        st = lib.stats()
        assert st.total_prefetches == total and total > 0
        assert lib.poll_complete() == total                        # descriptors completed since the last poll, then cleared
        assert lib.poll_complete() == 0
        assert st.dma_completed == total
    finally:
        lib.finalize()


@pytest.mark.parametrize("scheme", [3, 4, 5])
def test_residue_class_forms_on_short_and_ragged_ranges(scheme):
    """The residue-class forms of the fused attention over a pool striped x7 (k_attend_int4_wg8<.., CLS>, k_attend_fp8_dma<2>,
    k_attend_mx4<1>) at the small end: ranges with fewer pages than runs (empty classes), with one page per class, with classes of
    unequal length, a range that starts inside the region, several layers -- against a one-pool engine on the same data (other
    split boundaries: the tolerance of another summation order)."""
    torch = torch_mod()
    from tests.test_gpu_full_size import H, D, G
    T, L = 128, 3
    n_pages = T * L * H * D * 2 * 2 // PAGE
    rng = np.random.default_rng(310 + scheme)
    x = (rng.standard_normal((n_pages, N)) * rng.uniform(0.3, 2.0, (n_pages, 1))).astype(np.float16)
    q = (rng.standard_normal((L, H, G, D)) * 1.5).astype(np.float16)
    sm = 1.0 / np.sqrt(D)
    ranges = [(0, 2), (0, 8), (0, 14), (0, 16), (0, 30), (32, 64), (32, 46), (0, 128), (64, 126)]
    results = {}
    for name, env in (("one pool", {}), ("striped", {"SPECKV_POOL_DEVICES": "0,0,0,0,0,0,0"}), ("striped, layout set behind the write", {"SPECKV_POOL_DEVICES": "0,0,0,0,0,0,0"})):
        lib = open_lib(**env)
        try:
            lib.set_compression_scheme(scheme)
            h = lib.alloc(n_pages * PAGE)
            if "behind" in name:                     # (FP8: the scale tables -- page order and run order -- are then BUILT from the page table, not kept by the writes)
                lib.write(h, 0, x.ctypes.data, x.nbytes, False)
                lib.set_layout(h, T, L, H, D, 2)
            else:
                lib.set_layout(h, T, L, H, D, 2)
                lib.write(h, 0, x.ctypes.data, x.nbytes, False)
            d_q = torch.from_numpy(q.view(np.int16)).cuda()
            attend = {3: lib.attend_int4, 4: lib.attend_fp8, 5: lib.attend_mx4}[scheme]
            res = []
            for pb, pe in ranges:
                for layer0, nl in ((0, L), (1, 1)):
                    out = torch.full((nl, H, G, D), float("nan"), dtype=torch.float32, device="cuda")
                    lse = torch.full((nl, H, G), float("nan"), dtype=torch.float32, device="cuda")
                    attend(h, layer0, nl, d_q[layer0:layer0 + nl].data_ptr(), G, pb, pe, sm, out.data_ptr(), lse.data_ptr())
                    torch.cuda.synchronize()
                    res.append((out.cpu().numpy(), lse.cpu().numpy()))
            results[name] = res
        finally:
            lib.finalize()
    for which in ("striped", "striped, layout set behind the write"):
        for i, ((o_s, l_s), (o_o, l_o)) in enumerate(zip(results[which], results["one pool"])):
            assert np.isfinite(o_s).all() and np.isfinite(l_s).all(), (scheme, which, ranges[i // 2])
            scale = float(np.abs(o_o).max())
            assert float(np.abs(o_s - o_o).max()) <= 1e-3 * scale, (scheme, which, ranges[i // 2], i % 2)
            assert float(np.abs(l_s - l_o).max()) <= 2e-4, (scheme, which, ranges[i // 2], i % 2)


@pytest.mark.parametrize("pools", ["0,0,0", "0,0,0,0,0,0,0"])
@pytest.mark.parametrize("scheme", [3, 4])
def test_fused_attention_over_a_regularly_striped_pool_computes_its_addresses(oracle, scheme, pools):
    """an allocation striped `page % D` over several pools (every multi-GPU layout of configs[3] / [4])
    used to take the page-table form of the fused attention.  While the placement is regular the record of page p is
    base[p % D] + (p / D) * stride, and the fast kernels now compute exactly that: their result must be the oracle's (same
    tolerance as the linear form), and identical to what the page-table form (SPECKV_ATTEND_GENERAL=1) and a one-pool
    engine give up to split boundaries.  D = 3 and D = 7 (the 1 + 7 layout), ranges that start inside the region and end in
    a ragged tile, several layers per launch; after a migration the placement is no longer regular and the page table
    takes over again."""
    torch = torch_mod()
    from tests.test_gpu_full_size import HeadChecker, H, D, G
    T, L = 2048, 3
    n_pages = T * L * H * D * 2 * 2 // PAGE
    rng = np.random.default_rng(300 + scheme)
    x = (rng.standard_normal((n_pages, N)) * rng.uniform(0.2, 3.0, (n_pages, 1))).astype(np.float16)
    q = (rng.standard_normal((L, H, G, D)) * 1.5).astype(np.float16)
    sm = 1.0 / np.sqrt(D)
    results = {}
    for name, env in (("one pool", {}), ("striped", {"SPECKV_POOL_DEVICES": pools})):
        lib = open_lib(**env)
        try:
            lib.set_compression_scheme(scheme)
            h = lib.alloc(n_pages * PAGE)
            lib.set_layout(h, T, L, H, D, 2)
            lib.write(h, 0, x.ctypes.data, x.nbytes, False)
            d_q = torch.from_numpy(q.view(np.int16)).cuda()
            attend = lib.attend_int4 if scheme == 3 else lib.attend_fp8

            def run(pb, pe, general=False, layer0=0, nl=L):
                if general: set_tuning("attend_general", "1")
                try:
                    out = torch.full((nl, H, G, D), float("nan"), dtype=torch.float32, device="cuda")
                    lse = torch.full((nl, H, G), float("nan"), dtype=torch.float32, device="cuda")
                    attend(h, layer0, nl, d_q[layer0:layer0 + nl].data_ptr(), G, pb, pe, sm, out.data_ptr(), lse.data_ptr())
                    torch.cuda.synchronize()
                finally:
                    set_tuning("attend_general", 0)
                return out.cpu().numpy(), lse.cpu().numpy()

            results[name] = {"full": run(0, T), "part": run(64, T - 26, layer0=1, nl=2)}
            if name == "striped":
                results["page table"] = {"full": run(0, T, True), "part": run(64, T - 26, True, 1, 2)}
                info = [lib.translate(h, p * PAGE).pool_addr for p in range(2 * len(pools.split(",")))]
                assert len(set(info)) == len(info)
                # after a migration the placement is irregular: the page-table form must take over (same numbers)
                lib.migrate(h, 100, 300, 1)
                results["migrated"] = {"full": run(0, T), "part": run(64, T - 26, layer0=1, nl=2)}
        finally:
            lib.finalize()
    # the striped form against the oracle (layer 1, two heads, full range)
    checker = HeadChecker(oracle, scheme, x[T:2 * T], T)
    out, lse = results["striped"]["full"]
    for head in (0, 6):
        checker.check(out[1, head], lse[1, head], q[1, head], head, T, sm, ("striped vs oracle", head))
    # and against the other forms of the same computation
    for other in ("one pool", "page table", "migrated"):
        for case in ("full", "part"):
            o_s, l_s = results["striped"][case]
            o_o, l_o = results[other][case]
            scale = float(np.abs(o_o).max())
            assert float(np.abs(o_s - o_o).max()) <= 1e-3 * scale, (other, case)
            assert float(np.abs(l_s - l_o).max()) <= 2e-4, (other, case)


def test_real_lstm_cell_predictor_matches_torch_nn_lstm():
    """VERDICT r2 missing #3 / SURVEY 8f N1: the reference's cell ignores its weights, so the predictor's semantics as a MODEL
    are ours to define -- the standard LSTM in PyTorch's nn.LSTM conventions (speckv_ext_predictor_load_lstm).  Checked
    against an implementation that is not ours: torch.nn.LSTM on the CPU (2 layers, 64 -> 128, 16-token histories), then
    the output layer with bias, softmax and top-k.  Tolerance 2e-5 on the probabilities (device expf / tanhf against glibc,
    accumulated over 16 steps); the token ranking must agree wherever two probabilities are further apart than that.
    Loading the degenerate weights again switches back to the reference's cell."""
    torch = torch_mod()
    lib = open_lib()
    try:
        vocab, E, Hd, L, n, k = 4096, 64, 128, 2, 37, 8
        gen = torch.Generator().manual_seed(11)
        emb = torch.randn((vocab, E), generator=gen) * 0.7
        lstm = torch.nn.LSTM(E, Hd, num_layers=L, batch_first=True)
        with torch.no_grad():
            for p in lstm.parameters():
                p.copy_(torch.randn(p.shape, generator=gen) * 0.15)
        wout = torch.randn((vocab, Hd), generator=gen) * 0.5
        bout = torch.randn(vocab, generator=gen) * 0.3
        hist = torch.randint(0, vocab, (n, 16), generator=gen, dtype=torch.int32)
        hist[3, :5] = vocab + 7                                   # out-of-vocabulary ids embed as zeros (as the reference does)
        with torch.no_grad():
            x = torch.where((hist < vocab).unsqueeze(-1), emb[hist.clamp(max=vocab - 1).long()], torch.zeros(1))
            out, _ = lstm(x)
            probs = torch.softmax(out[:, -1] @ wout.T + bout, dim=-1)
        want_p, want_t = probs.topk(k, dim=-1)
        par = {name: p.detach().contiguous() for name, p in lstm.named_parameters()}
        ptr = lambda t: t.data_ptr()
        lib.predictor_load_lstm(ptr(emb), vocab, [ptr(par[f"weight_ih_l{l}"]) for l in range(L)], [ptr(par[f"weight_hh_l{l}"]) for l in range(L)],
                                [ptr(par[f"bias_ih_l{l}"]) for l in range(L)], [ptr(par[f"bias_hh_l{l}"]) for l in range(L)],
                                ptr(wout), ptr(bout), False)
        d_hist = hist.cuda()
        d_tok = torch.zeros((n, k), dtype=torch.int32, device="cuda")
        d_conf = torch.zeros((n, k), dtype=torch.float32, device="cuda")
        lib.predict_batch(n, d_hist.data_ptr(), k, d_tok.data_ptr(), d_conf.data_ptr())
        torch.cuda.synchronize()
        got_t, got_p = d_tok.cpu(), d_conf.cpu()
        assert float((got_p - want_p).abs().max()) <= 2e-5, float((got_p - want_p).abs().max())
        for r in range(n):
            for j in range(k):
                if int(got_t[r, j]) != int(want_t[r, j]):           # only a near-tie may swap places
                    assert abs(float(probs[r, got_t[r, j]]) - float(want_p[r, j])) <= 4e-5, (r, j)
        assert float(got_p.sum(-1).max()) <= 1.0 + 1e-5
        # through the engine's own path: histories arrive with speckv_prefetch, the flush predicts, verify() checks
        lib.set_compression_scheme(2)
        h = lib.alloc(64 * 2 * 2 * 8 * 128 * 2)
        lib.set_layout(h, 64, 2, 8, 128, 2)
        lib.prefetch(0, 0, 3, 4, [int(v) for v in hist[5]])
        lib.prefetch_flush()
        hit, _ = lib.verify(0, int(want_t[5, 0]))
        assert hit
        hit, _ = lib.verify(0, int(probs[5].argmin()))
        assert not hit
        # back to the reference's degenerate cell: the weights above are ignored again
        lib.predictor_load(ptr(emb), ptr(wout), vocab, False)
        lib.predict_batch(n, d_hist.data_ptr(), k, d_tok.data_ptr(), d_conf.data_ptr())
        torch.cuda.synchronize()
        assert not torch.equal(d_tok.cpu(), got_t)
    finally:
        lib.finalize()


@pytest.mark.parametrize("vocab", [1000, 32000])
def test_small_prediction_path_equals_the_batch_path(oracle, vocab):
    """One to four requests take two launches (k_predict_small: the logits of a workgroup's 128 rows reduced on the spot, no
    logits in memory; k_predict_small_merge) instead of the batch path's four.  Both paths on the same weights and histories
    (speckv_ext_set_tuning: predict_batch_path): the same tokens, confidences within 2e-5 (the dot products add in a
    different order), for the reference's degenerate cell -- also against the oracle -- and for the real LSTM cell with an
    output bias."""
    torch = torch_mod()
    lib = open_lib()
    try:
        rng = np.random.default_rng(vocab + 5)
        emb = rng.standard_normal((vocab, 64)).astype(np.float32) * 0.5
        wout = rng.standard_normal((vocab, 128)).astype(np.float32) * 0.5

        def run(n, k, H):
            d_h = torch.from_numpy(H).cuda()
            d_tok = torch.full((n, k), -7, dtype=torch.int32, device="cuda"); d_conf = torch.zeros((n, k), dtype=torch.float32, device="cuda")
            lib.predict_batch(n, d_h.data_ptr(), k, d_tok.data_ptr(), d_conf.data_ptr())
            torch.cuda.synchronize()
            return d_tok.cpu().numpy(), d_conf.cpu().numpy()

        def both(n, k, H):
            small = run(n, k, H)
            set_tuning("predict_batch_path", "1")
            try:
                batch = run(n, k, H)
            finally:
                set_tuning("predict_batch_path", 0)
            return small, batch

        lib.predictor_load(emb.ctypes.data, wout.ctypes.data, vocab, False)
        for n in (1, 2, 3, 4):
            for k in (1, 4, 8):
                H = rng.integers(0, vocab, (n, 16)).astype(np.int32)
                H[0, :2] = vocab + 3                                 # out-of-vocabulary ids embed as zeros
                (tok, conf), (btok, bconf) = both(n, k, H)
                assert tok.tolist() == btok.tolist(), (n, k)
                assert np.allclose(conf, bconf, rtol=2e-5, atol=1e-12), (n, k, np.abs(conf - bconf).max())
                for i in range(n):
                    o_tok, o_conf = oracle.lstm_predict(emb, wout, H[i].astype(np.uint32), k)
                    assert tok[i].tolist() == o_tok.astype(np.int32).tolist(), (n, k, i)
                    assert np.allclose(conf[i], o_conf, rtol=5e-4, atol=1e-12)
        # the real cell: k_lstm_cell's hidden vectors feed k_predict_small
        gen = torch.Generator().manual_seed(3)
        rnd = lambda *shape: (torch.rand(shape, generator=gen) - 0.5) * 0.4
        w_ih, w_hh = [rnd(512, 64), rnd(512, 128)], [rnd(512, 128), rnd(512, 128)]
        b_ih, b_hh = [rnd(512), rnd(512)], [rnd(512), rnd(512)]
        bout = rnd(vocab)
        t_emb, t_wout = torch.from_numpy(emb), torch.from_numpy(wout)
        lib.predictor_load_lstm(t_emb.data_ptr(), vocab, [t.data_ptr() for t in w_ih], [t.data_ptr() for t in w_hh],
                                [t.data_ptr() for t in b_ih], [t.data_ptr() for t in b_hh], t_wout.data_ptr(), bout.data_ptr(), False)
        for n in (1, 4):
            H = rng.integers(0, vocab, (n, 16)).astype(np.int32)
            (tok, conf), (btok, bconf) = both(n, 8, H)
            assert tok.tolist() == btok.tolist(), n
            assert np.allclose(conf, bconf, rtol=2e-5, atol=1e-12), (n, np.abs(conf - bconf).max())
            assert conf[0, 0] > conf[0, 7] > 0.0
    finally:
        lib.finalize()


@pytest.mark.parametrize("n_req", [5, 64, 200])
def test_one_workgroup_flush_equals_the_four_launch_pipeline(oracle, n_req):
    """Small flushes run the device-side pipeline (candidates, first-occurrence dedupe, ring run, ordered placement) as phases
    of one workgroup (k_flush_small) instead of four launches.  Same requests through both forms (speckv_ext_set_tuning: flush_no_small /
    flush_small_words), each in a fresh engine: the same pages become resident, in the SAME ring
    slots (entry order is part of the contract: the host derives residency from the slot's sequence number), with the same
    bytes; and the set of pages is the oracle's.  Requests repeat pages (dedupe) and some name pages already resident."""
    T, L, H, D, bpe = 512, 4, 8, 128, 2
    n_pages = T * L * H * D * bpe * 2 // PAGE
    rng = np.random.default_rng(n_req)
    layers = rng.integers(0, L, n_req).astype(np.uint16)
    pos = rng.integers(0, T - 8, n_req).astype(np.uint32)
    pos[n_req // 2:] = pos[: n_req - n_req // 2]                      # repeated positions: the same pages named again
    reqs = np.zeros(n_req, np.uint32)
    depth = np.full(n_req, 4, np.uint32)
    x = (rng.standard_normal((n_pages, N)) * 0.5).astype(np.float16)
    seen = {}
    for form, env in (("one workgroup", {"flush_small_words": 16384}), ("four launches", {"flush_no_small": 1})):
        for k_, v_ in env.items():
            set_tuning(k_, v_)
        try:
            lib = SpeckvLib(pkg.library_path(), "hip:0")
            try:
                lib.set_compression_scheme(2)
                h = lib.alloc(n_pages * PAGE)
                lib.set_layout(h, T, L, H, D, bpe)
                lib.write(h, 0, x.ctypes.data, x.nbytes, False)
                for p in (3, 40, 41):
                    lib.access(h, p * PAGE, 1)                        # resident before the flush: filtered out
                lib.prefetch_batch(reqs, layers, pos, depth)
                lib.prefetch_flush()
                lib.sync()
                info = [lib.translate(h, p * PAGE) for p in range(n_pages)]
                res = {p: (i.cache_addr, i.flags & 3) for p, i in enumerate(info) if i.flags & 3}
                data = {p: dev_to_host(res[p][0], PAGE).tobytes() for p in sorted(res)[:40]}
                seen[form] = (res, data, int(lib.stats().total_prefetches))
            finally:
                lib.finalize()
        finally:
            for k_ in env:
                set_tuning(k_, 0)
    a, b = seen["one workgroup"], seen["four launches"]
    base = lambda res: min(addr for addr, _ in res.values())
    rel = lambda res: {p: (addr - base(res), f) for p, (addr, f) in res.items()}
    assert rel(a[0]) == rel(b[0])                                    # same pages, same slots (relative to the cache base), same tier bits
    assert a[1] == b[1] and a[2] == b[2]
    expect = {3, 40, 41}
    for i in range(n_req):
        expect |= set(oracle.prefetch_pages(0, int(layers[i]), int(pos[i]), 4, L, T, H, D, bpe, n_pages).tolist())
    assert set(a[0]) == expect


def test_flush_time_predictions_under_concurrent_prefetch_and_verify(oracle):
    """A flush's predictions run on a stream of their own and are collected by the next speckv_ext_verify; collecting waits on
    an event with the ABI lock released, so another thread's flush may start the NEXT prediction meanwhile (generation
    counter in Engine::harvest_predictions).  Two threads for a second: one keeps sending speckv_prefetch calls with
    histories (a flush every 4 calls), one keeps verifying; then, quiet again, every request's prediction is the oracle's
    for its LAST history."""
    lib = open_lib()
    try:
        emb, wout = oracle.lstm_reference_weights(1)
        lib.predictor_load(emb.ctypes.data, wout.ctypes.data, 32000, False)
        lib.set_compression_scheme(2)
        T, L, H, D, bpe = 256, 4, 8, 128, 2
        h = lib.alloc(8 * T * L * H * D * bpe * 2)
        lib.set_layout(h, T, L, H, D, bpe)
        rng = np.random.default_rng(9)
        last, errors, stop = {}, [], threading.Event()

        def sender():
            try:
                step = 0
                while not stop.is_set():
                    r = step % 8
                    hist = [int(v) for v in rng.integers(0, 32000, 16)]
                    lib.prefetch(r, step % L, 8 + (step // 8) % 200, 4, hist)
                    last[r] = hist
                    step += 1
            except Exception as e:                                    # noqa: BLE001
                errors.append(repr(e))

        def checker():
            try:
                while not stop.is_set():
                    for r in range(8):
                        try:
                            lib.verify(r, 1)
                        except SpeckvError as e:                      # no prediction for this request yet
                            assert e.status == -4
            except Exception as e:                                    # noqa: BLE001
                errors.append(repr(e))

        ts = [threading.Thread(target=sender), threading.Thread(target=checker)]
        for t in ts:
            t.start()
        time.sleep(1.0)
        stop.set()
        for t in ts:
            t.join(timeout=30)
        assert not any(t.is_alive() for t in ts) and not errors, errors
        lib.prefetch_flush()
        for r, hist in last.items():
            o_tok, _ = oracle.lstm_predict(emb, wout, np.array(hist, np.uint32), 4)
            hit, _ = lib.verify(r, int(o_tok[0]))
            assert hit, r
        st = lib.stats()
        assert st.successful_prefetches >= len(last)
    finally:
        lib.finalize()


class _RawDevice:
    """A device buffer by address, for torch.as_tensor (CUDA array interface)."""
    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (int(ptr), False), "version": 2}


@pytest.mark.parametrize("spin", [True, False])
def test_access_miss_completion_word_orders_the_page_for_other_streams(spin):
    """speckv_access on a miss returns when the fetch kernel's last wave has stored a token to a pinned host word (the
    runtime's completion path costs 4 us more; SPECKV_ACCESS_NO_SPIN=1 keeps it).  The kernel is then still retiring, so
    what the caller does next must already see the page: 600 random misses (the ring of the default cache is recycled
    many times over), each read at once by a kernel on ANOTHER stream, and spans of 1..8 pages; against the source (FP16
    scheme: the stored page is the source page)."""
    torch = torch_mod()
    lib = open_lib(**({} if spin else {"SPECKV_ACCESS_NO_SPIN": 1}))
    try:
        lib.set_compression_scheme(0)
        n_pages = 4096
        h = lib.alloc(n_pages * PAGE)
        rng = np.random.default_rng(3)
        x = rng.integers(0, 2 ** 16, (n_pages, PAGE // 2), dtype=np.uint16)
        x &= 0x7BFF                                                  # finite fp16 bit patterns
        lib.write(h, 0, x.ctypes.data, x.nbytes, False)
        side = torch.cuda.Stream()
        held = []
        for i in range(600):
            p = int(rng.integers(0, n_pages - 8))
            span = 1 + (i % 8 if i % 5 == 0 else 0)
            ptr = lib.access(h, p * PAGE, span * PAGE)
            with torch.cuda.stream(side):
                held.append((p, span, torch.as_tensor(_RawDevice(ptr, span * PAGE), device="cuda").clone()))
            if len(held) == 50:                                      # (the copies are checked in batches: no host sync per miss)
                side.synchronize()
                for q, sp, t in held:
                    assert t.cpu().numpy().tobytes() == x[q:q + sp].tobytes(), (i, q, sp)
                held.clear()
        st = lib.stats()
        assert st.l3_accesses >= 500
    finally:
        lib.finalize()


@pytest.mark.parametrize("vocab,n,k", [(8, 1, 8), (33, 31, 8), (1007, 33, 5), (4097, 70, 8), (32768, 3, 1), (32769, 2, 8), (128256, 5, 8),
                                       (262144, 2, 4), (262145, 2, 8)])
def test_predictor_shapes_off_the_tile_sizes(oracle, vocab, n, k):
    """The predictor's kernels cut their work into fixed tiles -- 32 output rows and 32 requests per matrix tile, 4096 logits per
    top-k workgroup, up to 64 of those per request (128 256 tokens, Llama-3's vocabulary: 32; beyond 262 144 the one-workgroup kernel
    takes over) -- so: vocabularies
    and batch sizes that are not multiples of any of them, k = 8 = the whole vocabulary, against the oracle's
    restatement of the reference's predictor (tokens identical, confidences within 5e-4 relative, as in
    test_token_predictor_matches_oracle_and_reference)."""
    torch = torch_mod()
    lib = open_lib()
    try:
        rng = np.random.default_rng(vocab * 131 + n)
        emb = rng.standard_normal((vocab, 64)).astype(np.float32) * 0.5
        wout = rng.standard_normal((vocab, 128)).astype(np.float32) * 0.5
        lib.predictor_load(emb.ctypes.data, wout.ctypes.data, vocab, False)
        H = rng.integers(0, vocab, (n, 16)).astype(np.int32)
        H[0, :3] = vocab + 1                                        # out-of-vocabulary ids embed as zeros
        d_h = torch.from_numpy(H).cuda()
        d_tok = torch.full((n, k), -7, dtype=torch.int32, device="cuda"); d_conf = torch.zeros((n, k), dtype=torch.float32, device="cuda")
        lib.predict_batch(n, d_h.data_ptr(), k, d_tok.data_ptr(), d_conf.data_ptr())
        torch.cuda.synchronize()
        tok, conf = d_tok.cpu().numpy(), d_conf.cpu().numpy()
        for i in range(n):
            o_tok, o_conf = oracle.lstm_predict(emb, wout, H[i].astype(np.uint32), k)
            assert tok[i].tolist() == o_tok.astype(np.int32).tolist(), (i, tok[i], o_tok)
            assert np.allclose(conf[i], o_conf, rtol=5e-4, atol=1e-12), (i, conf[i], o_conf)
    finally:
        lib.finalize()


@pytest.mark.parametrize("pools", [None, "0,0,0"])
def test_compaction_packs_rle_records_and_frees_the_slots(oracle, pools):
    """INT8_DELTA_RLE records used to live in worst-case 4 KiB slots for good, so the
    reference's own scheme bought no pool capacity and the copy engine shipped slots.  speckv_ext_compact seals an
    allocation: records packed back to back (128-byte aligned), slots returned to the pool.  Checked: the pool really
    shrinks to the packed size (stats.pool_bytes_in_use, and a second allocation fits into what was freed), every read
    path still returns the oracle's bits (bulk fetch by both engines, random list, speckv_access through the ring), the
    copy engine now moves record bytes, a write unseals transparently, and Gaussian data (incompressible for this
    scheme) gains nothing."""
    torch = torch_mod()
    env = {"SPECKV_POOL_DEVICES": pools} if pools else {}
    lib = open_lib(SPECKV_STAGE_MB=2, **env)
    try:
        lib.set_compression_scheme(2)
        n = 6144
        rng = np.random.default_rng(17)
        x = np.zeros((n, N), np.float16)
        x[0::3] = np.repeat(rng.standard_normal((len(x[0::3]), N // 32)), 32, axis=1).astype(np.float16)   # runs of 32
        x[1::3] = 0                                                                                         # zeros
        x[2::3] = rng.standard_normal((len(x[2::3]), N)).astype(np.float16)                                 # incompressible
        x[5] = 0; x[4095] = 0
        sc, ln, rc = oracle.compress_blocks_f16(x, 2, 0)
        want = oracle.decompress_blocks_f16(rc, ln, sc, 2, 0)
        h = lib.alloc(n * PAGE)
        lib.write(h, 0, x.ctypes.data, x.nbytes, False)
        st0 = lib.stats()
        assert st0.pool_bytes_in_use == n * PAGE and st0.written_pages == n and st0.sealed_allocations == 0
        before, after = lib.compact(h)
        packed = int(((ln.astype(np.int64) + 127) // 128 * 128).sum())
        assert before == n * PAGE and after == packed and after < 0.45 * before
        st1 = lib.stats()
        assert st1.pool_bytes_in_use == packed and st1.sealed_allocations == 1 and st1.compactions == 1
        assert st1.compressed_bytes == int(ln.astype(np.int64).sum())
        assert lib.compact(h) == (packed, packed)                       # idempotent
        # reads: bulk (both engines), list, single pages through the ring
        out = torch.empty((n, N), dtype=torch.float16, device="cuda")
        s = torch.cuda.Stream()
        for engine in (1, 2):
            out.fill_(float("nan"))
            ce0 = lib.stats().copy_engine_bytes
            lib.fetch_range(h, 0, n, out.data_ptr(), False, s.cuda_stream, engine=engine)
            torch.cuda.synchronize()
            assert_same_float_bits(out.cpu().numpy(), want, f"engine {engine}")
            if engine == 2:
                assert lib.stats().copy_engine_bytes - ce0 == packed    # record bytes crossed, not 4 KiB slots
        out.fill_(float("nan"))
        lib.fetch_range(h, 1001, 777, out.data_ptr(), False, s.cuda_stream, engine=2)      # a range that starts inside the stripes
        torch.cuda.synchronize()
        assert_same_float_bits(out[:777].cpu().numpy(), want[1001:1778], "partial range, copy engine")
        perm = torch.from_numpy(rng.permutation(n)[:500].astype(np.int32)).cuda()
        sub = torch.empty((500, N), dtype=torch.float16, device="cuda")
        lib.fetch_list(h, perm.data_ptr(), 500, sub.data_ptr(), False)
        torch.cuda.synchronize()
        assert_same_float_bits(sub.cpu().numpy(), want[perm.cpu().numpy()], "list")
        for p in (0, 1, 2, 5, 4095, n - 1):
            assert dev_to_host(lib.access(h, p * PAGE, PAGE), PAGE).tobytes() == want[p].tobytes(), p
            info = lib.translate(h, p * PAGE)
            assert info.rec_bytes == ln[p] and stored_record(info, ln[p]).tobytes() == rc[p, :ln[p]].tobytes()
        # the freed slots are really back in the pool: a second allocation of the same size fits without growing it
        reserved = lib.stats().pool_bytes_reserved
        h2 = lib.alloc((n * PAGE - packed) // PAGE // 4 * PAGE)
        assert lib.stats().pool_bytes_reserved == reserved
        lib.free(h2)
        # a write unseals (back into slots), the rest of the data survives, and sealing again works
        y = rng.standard_normal((16, N)).astype(np.float16)
        lib.write(h, 300 * PAGE, y.ctypes.data, y.nbytes, False)
        st2 = lib.stats()
        assert st2.sealed_allocations == 0 and st2.pool_bytes_in_use == n * PAGE
        x2 = x.copy(); x2[300:316] = y
        sc2, ln2, rc2 = oracle.compress_blocks_f16(x2, 2, 0)
        want2 = oracle.decompress_blocks_f16(rc2, ln2, sc2, 2, 0)
        lib.fetch_range(h, 0, n, out.data_ptr(), False, s.cuda_stream)
        torch.cuda.synchronize()
        assert_same_float_bits(out.cpu().numpy(), want2, "after the write")
        d_y = torch.from_numpy(y.view(np.int16)).cuda()
        b2, a2 = lib.compact(h)
        assert b2 == n * PAGE and a2 == int(((ln2.astype(np.int64) + 127) // 128 * 128).sum())
        lib.write_async(h, 900 * PAGE, d_y.data_ptr(), y.nbytes, s.cuda_stream)             # asynchronous writes unseal too
        torch.cuda.synchronize()
        assert lib.stats().sealed_allocations == 0
        lib.compact(h)
        lib.migrate(h, 0, 64, 0)                                                             # and so does a migration
        assert lib.stats().sealed_allocations == 0
        lib.free(h)
        assert lib.stats().pool_bytes_in_use == 0
        # incompressible data: nothing to gain, nothing breaks; other schemes: a no-op
        g = rng.standard_normal((512, N)).astype(np.float16)
        hg = lib.alloc(512 * PAGE)
        lib.write(hg, 0, g.ctypes.data, g.nbytes, False)
        bg, ag = lib.compact(hg)
        assert 0.99 * bg <= ag <= bg
        lib.free(hg)
        lib.set_compression_scheme(4)
        hf = lib.alloc(64 * PAGE)
        lib.write(hf, 0, g.ctypes.data, 64 * PAGE, False)
        slack = 15 * 2048 * len(pools.split(",")) if pools else 0        # (FP8 runs of a striped allocation carry 15 records of slack: k_attend_fp8_dma<2>)
        assert lib.compact(hf) == (64 * 2048 + slack, 64 * 2048 + slack) and lib.stats().sealed_allocations == 0
    finally:
        lib.finalize()


@pytest.mark.parametrize("scheme", [3, 4, 5])
def test_batch_and_planned_attention_over_striped_pools(scheme):
    """A decode step of a batch whose sequences live in striped pools (the 1 + 7 layout of configs[3]): the batch and the
    planned forms used to refuse anything but single-run allocations.  Each sequence descriptor now carries its own run
    bases, so one launch serves striped sequences and single-run ones (placed on one pool with preferred_node) alike;
    results against the per-sequence entry point (same kernel family, other split boundaries: 1e-3 of the row scale)."""
    torch = torch_mod()
    lib = open_lib(SPECKV_POOL_DEVICES="0,0,0,0,0,0,0")
    try:
        lib.set_compression_scheme(scheme)
        T, L, H, D, G = 1024, 2, 8, 128, 8
        n_pages = T * L * H * D * 2 * 2 // PAGE
        rng = np.random.default_rng(600 + scheme)
        lens = [1024, 514, 0, 66, 1000, 32]
        handles = []
        for i, n in enumerate(lens):
            h = lib.alloc(n_pages * PAGE, preferred_node=(3 if i % 3 == 1 else 0))       # every third sequence: one run on pool 3
            lib.set_layout(h, T, L, H, D, 2)
            x = (rng.standard_normal((n_pages, N)) * rng.uniform(0.3, 2.0, (n_pages, 1))).astype(np.float16)
            lib.write(h, 0, x.ctypes.data, x.nbytes, False)
            handles.append(h)
        assert lib.translate(handles[0], 0).pool_addr != lib.translate(handles[0], PAGE).pool_addr - {4: 2048, 3: 1152, 5: 1024}[scheme]       # (striped: page 1 is not the next record; MXFP4 runs are tile-planar, nibble rows 1024 B apart)
        q = torch.from_numpy(rng.standard_normal((len(lens), H, G, D)).astype(np.float16)).cuda()
        sm = 1.0 / np.sqrt(D)
        single = {3: lib.attend_int4, 4: lib.attend_fp8, 5: lib.attend_mx4}[scheme]
        batch = {3: lib.attend_int4_batch, 4: lib.attend_fp8_batch, 5: lib.attend_mx4_batch}[scheme]
        st = torch.cuda.Stream()
        for layer in (0, 1):
            ref = torch.zeros((len(lens), H, G, D), dtype=torch.float32, device="cuda")
            ref_lse = torch.full((len(lens), H, G), float("-inf"), dtype=torch.float32, device="cuda")
            for i, (h, n) in enumerate(zip(handles, lens)):
                if n:
                    single(h, layer, 1, q[i].data_ptr(), G, 0, n, sm, ref[i].data_ptr(), ref_lse[i].data_ptr())
            torch.cuda.synchronize()
            out = torch.full((len(lens), H, G, D), float("nan"), dtype=torch.float32, device="cuda")
            lse = torch.full((len(lens), H, G), float("nan"), dtype=torch.float32, device="cuda")
            batch(handles, layer, q.data_ptr(), G, lens, sm, out.data_ptr(), lse.data_ptr())
            torch.cuda.synchronize()
            plan_bytes = lib.attend_plan_bytes(len(lens))
            d_plan = torch.empty(plan_bytes, dtype=torch.uint8, device="cuda")
            out2 = torch.full_like(out, float("nan")); lse2 = torch.full_like(lse, float("nan"))
            torch.cuda.synchronize()
            lib.attend_batch_plan(handles, lens, T, d_plan.data_ptr(), plan_bytes, st.cuda_stream)
            lib.attend_planned(scheme, d_plan.data_ptr(), len(lens), layer, q.data_ptr(), G, T, sm, out2.data_ptr(), lse2.data_ptr(), st.cuda_stream)
            torch.cuda.synchronize()
            for got, got_lse, what in ((out, lse, "batch"), (out2, lse2, "planned")):
                for i, n in enumerate(lens):
                    if n == 0:
                        assert float(got[i].abs().max()) == 0.0
                        continue
                    scale = float(ref[i].abs().max()) + 1e-6
                    assert float((got[i] - ref[i]).abs().max()) <= 1e-3 * scale, (what, layer, i)
                    assert float((got_lse[i] - ref_lse[i]).abs().max()) <= 2e-4, (what, layer, i)
        if scheme == 4:
            # FP8 batches take the page tables by default; the residue-class form of the register-staged kernel on request (layer 1: `ref`)
            set_tuning("attend_fp8_striped_table", -1)
            try:
                out6 = torch.full_like(out, float("nan")); lse6 = torch.full_like(lse, float("nan"))
                batch(handles, 1, q.data_ptr(), G, lens, sm, out6.data_ptr(), lse6.data_ptr())
                out7 = torch.full_like(out, float("nan")); lse7 = torch.full_like(lse, float("nan"))
                torch.cuda.synchronize()
                lib.attend_batch_plan(handles, lens, T, d_plan.data_ptr(), plan_bytes, st.cuda_stream)
                lib.attend_planned(scheme, d_plan.data_ptr(), len(lens), 1, q.data_ptr(), G, T, sm, out7.data_ptr(), lse7.data_ptr(), st.cuda_stream)
                torch.cuda.synchronize()
            finally:
                set_tuning("attend_fp8_striped_table", 0)
            for got, got_lse, what in ((out6, lse6, "batch, class form"), (out7, lse7, "planned, class form")):
                for i, n in enumerate(lens):
                    if n == 0:
                        assert float(got[i].abs().max()) == 0.0
                        continue
                    scale = float(ref[i].abs().max()) + 1e-6
                    assert float((got[i] - ref[i]).abs().max()) <= 1e-3 * scale, (what, i)
                    assert float((got_lse[i] - ref_lse[i]).abs().max()) <= 2e-4, (what, i)
        # a partly migrated sequence no longer has an arithmetic placement: the whole launch then reads its record addresses from
        # the page tables (the table forms of the kernels) -- batch and planned, same numbers as before the migration (layer 1: `ref`)
        lib.migrate(handles[0], 4, 8, 2)
        lib.migrate(handles[4], 300, 5, 6)
        out4 = torch.full_like(out, float("nan")); lse4 = torch.full_like(lse, float("nan"))
        batch(handles, 1, q.data_ptr(), G, lens, sm, out4.data_ptr(), lse4.data_ptr())
        out5 = torch.full_like(out, float("nan")); lse5 = torch.full_like(lse, float("nan"))
        torch.cuda.synchronize()
        lib.attend_batch_plan(handles, lens, T, d_plan.data_ptr(), plan_bytes, st.cuda_stream)
        lib.attend_planned(scheme, d_plan.data_ptr(), len(lens), 1, q.data_ptr(), G, T, sm, out5.data_ptr(), lse5.data_ptr(), st.cuda_stream)
        torch.cuda.synchronize()
        for got, got_lse, what in ((out4, lse4, "batch, table form"), (out5, lse5, "planned, table form")):
            for i, n in enumerate(lens):
                if n == 0:
                    assert float(got[i].abs().max()) == 0.0
                    continue
                scale = float(ref[i].abs().max()) + 1e-6
                assert float((got[i] - ref[i]).abs().max()) <= 1e-3 * scale, (what, i)
                assert float((got_lse[i] - ref_lse[i]).abs().max()) <= 2e-4, (what, i)
        # ... but a sequence migrated AS A WHOLE (a hot one pulled onto one pool GPU) is one run again and qualifies for every
        # arithmetic-address path: same results, records at base + page * stride
        lib.migrate(handles[0], 0, n_pages, 5)
        rec_off = (lambda p: (p >> 4) * 17408 + (p & 15) * 1024) if scheme == 5 else (lambda p: (2048 if scheme == 4 else 1152) * p)      # MXFP4: 16-record tiles of 136 lines
        base = lib.translate(handles[0], 0).pool_addr
        assert [lib.translate(handles[0], p * PAGE).pool_addr - base for p in (1, 2, 77, n_pages - 1)] == [rec_off(p) for p in (1, 2, 77, n_pages - 1)]
        if scheme == 5:
            assert [lib.translate(handles[0], p * PAGE).aux_offset for p in (0, 1, 15, 16, 77)] == [16384 - 960 * (p & 15) for p in (0, 1, 15, 16, 77)]
        out3 = torch.full_like(out, float("nan")); lse3 = torch.full_like(lse, float("nan"))
        batch(handles, 1, q.data_ptr(), G, lens, sm, out3.data_ptr(), lse3.data_ptr())
        torch.cuda.synchronize()
        for i, n in enumerate(lens):
            if n:
                scale = float(ref[i].abs().max()) + 1e-6
                assert float((out3[i] - ref[i]).abs().max()) <= 1e-3 * scale and float((lse3[i] - ref_lse[i]).abs().max()) <= 2e-4, i
    finally:
        lib.finalize()


@pytest.mark.parametrize("scheme", ["fp16", "int8_delta_rle", "fp8"])
def test_vllm_shaped_connector_round_trip(oracle, scheme):
    """VERDICT r2 missing #4: an adapter with the method surface of vLLM's v1 KV connector (vLLM is not in the image, so the
    scheduler / worker objects are stand-ins with vLLM's attribute names).  A prefill instance saves two prompts from its
    paged cache layer by layer; a second step presents the same request ids with a fresh, differently numbered set of blocks:
    the scheduler side reports the matched tokens (whole blocks, never the last token), the worker side loads them into the
    paged cache -- bit-exact for the fp16 pool, the oracle's decode of the oracle's records for INT8_DELTA_RLE, close for
    fp8."""
    torch = torch_mod()
    from types import SimpleNamespace as NS
    from cxl_speckv_amd.vllm_connector import SpeckvVllmConnector, slot_mapping_for
    lib = open_lib()
    try:
        L, H, D, BS, NB = 2, 8, 128, 16, 64
        c = SpeckvVllmConnector(lib, num_layers=L, num_kv_heads=H, head_dim=D, block_size=BS, max_tokens=256, scheme=scheme)
        gen = torch.Generator(device="cuda"); gen.manual_seed(21)
        names = [f"model.layers.{i}.self_attn.attn" for i in range(L)]
        caches = {n: torch.zeros((2, NB, BS, H, D), dtype=torch.float16, device="cuda") for n in names}
        c.register_kv_caches(caches)
        prompts = {"req-a": (40, [3, 9, 4]), "req-b": (64, [20, 21, 7, 30])}
        truth = {}
        # ---- step 1 (prefill): scheduler builds the metadata, the "model" fills the paged cache, the worker saves layer by layer
        new = [NS(req_id=rid, prompt_token_ids=list(range(n)), block_ids=[blk]) for rid, (n, blk) in prompts.items()]   # list per cache group
        for r in new:
            assert c.get_num_new_matched_tokens(NS(request_id=r.req_id, num_tokens=len(r.prompt_token_ids)), 0) == (0, False)
        meta = c.build_connector_meta(NS(scheduled_new_reqs=new))
        assert [m.is_store for m in meta.requests] == [True, True]
        c.bind_connector_metadata(meta)
        c.start_load_kv(None)                                          # nothing to load in this step
        for li, name in enumerate(names):
            flat = caches[name].reshape(2, NB * BS, H, D)
            for rid, (n, blk) in prompts.items():
                slots = torch.tensor(slot_mapping_for(blk, BS, n), device="cuda")
                rows = torch.randn((2, n, H, D), generator=gen, device="cuda").to(torch.float16)
                flat[:, slots] = rows
                truth[(rid, li)] = rows.clone()
            c.save_kv_layer(name, caches[name], None)
        c.wait_for_save()
        c.clear_connector_metadata()
        torch.cuda.synchronize()
        # ---- step 2 (another instance / a resumed request): fresh cache, other blocks
        for t in caches.values():
            t.zero_()
        resumed = {"req-a": [50, 51, 52], "req-b": [10, 11, 12, 13]}
        for rid, (n, _) in prompts.items():
            req = NS(request_id=rid, num_tokens=n)
            matched, is_async = c.get_num_new_matched_tokens(req, 0)
            assert matched == (n - 1) // BS * BS and is_async is False      # whole blocks, never the last token
            c.update_state_after_alloc(req, NS(get_block_ids=lambda b=resumed[rid]: [b]), matched)
        meta = c.build_connector_meta(NS(scheduled_new_reqs=[NS(req_id=rid, prompt_token_ids=list(range(n)), block_ids=[resumed[rid]])
                                                               for rid, (n, _) in prompts.items()]))
        assert [m.is_store for m in meta.requests] == [False, False]
        c.bind_connector_metadata(meta)
        c.start_load_kv(None)
        torch.cuda.synchronize()
        for li, name in enumerate(names):
            c.wait_for_layer_load(name)
            flat = caches[name].reshape(2, NB * BS, H, D)
            for rid, (n, _) in prompts.items():
                m = (n - 1) // BS * BS
                slots = torch.tensor(slot_mapping_for(resumed[rid], BS, m), device="cuda")
                got = flat[:, slots]                                     # [2][m][H][D]
                want = truth[(rid, li)][:, :m]
                if scheme == "fp16":
                    assert torch.equal(got, want), (rid, li)
                elif scheme == "int8_delta_rle":
                    for kind in (0, 1):
                        src = want[kind].cpu().numpy().reshape(-1, N)
                        sc, ln, rc = oracle.compress_blocks_f16(src, 2, 0)
                        dec = oracle.decompress_blocks_f16(rc, ln, sc, 2, 0).reshape(m, H, D)
                        assert_same_float_bits(got[kind].cpu().numpy(), dec, f"{rid} layer {li} kind {kind}")
                else:
                    err = (got.float() - want.float()).abs().max() / want.float().abs().max()
                    assert float(err) < 0.07, (rid, li, float(err))
            # blocks the loads did not name stay untouched
            assert float(flat[:, 40 * BS:50 * BS].abs().max()) == 0.0
        assert c.request_finished(NS(request_id="req-a"), [50, 51, 52]) == (False, None)
        c.free_request("req-a")
        assert c.get_num_new_matched_tokens(NS(request_id="req-a", num_tokens=40), 0) == (0, False)
    finally:
        lib.finalize()


def test_sealed_structured_allocation_takes_the_flat_run_decoder(oracle):
    """A sealed allocation whose packed records average under 512 bytes is fetched by the decoder instantiation with the
    flat-run fast path (Engine::fetch_range sets CodecArgs::structured_hint from the packed size): same bits as the oracle,
    fp16 and fp32, before and after the seal; a mostly-noise allocation (mean record ~4 KiB) keeps the plain decoder."""
    torch = torch_mod()
    lib = open_lib()
    try:
        lib.set_compression_scheme(2)
        n = 1024
        rng = np.random.default_rng(29)
        x = np.repeat(rng.standard_normal((n, N // 32)), 32, axis=1).astype(np.float16)     # runs of 32
        x[::5] = 0
        x[7, 100:140] = rng.standard_normal(40).astype(np.float16)                          # one block with a noisy stretch
        sc, ln, rc = oracle.compress_blocks_f16(x, 2, 0)
        want = oracle.decompress_blocks_f16(rc, ln, sc, 2, 0)
        h = lib.alloc(n * PAGE)
        lib.write(h, 0, x.ctypes.data, x.nbytes, False)
        s = torch.cuda.Stream()
        out = torch.empty((n, N), dtype=torch.float16, device="cuda")
        for sealed in (False, True):
            if sealed:
                before, after = lib.compact(h)
                assert after / n < 512 and after < before
            out.fill_(float("nan"))
            lib.fetch_range(h, 0, n, out.data_ptr(), False, s.cuda_stream)
            torch.cuda.synchronize()
            assert_same_float_bits(out.cpu().numpy(), want, f"sealed={sealed}")
        out32 = torch.full((n, N), float("nan"), dtype=torch.float32, device="cuda")
        lib.fetch_range(h, 3, n - 3, out32.data_ptr(), True, s.cuda_stream)
        torch.cuda.synchronize()
        sc2 = oracle.decompress_blocks_f16(rc, ln, sc, 2, 0)             # (fp16 of the fp32 result is the fp16 result)
        assert_same_float_bits(out32[:n - 3].cpu().numpy().astype(np.float16), sc2[3:], "fp32 output of the sealed allocation")
    finally:
        lib.finalize()


def test_vllm_connector_prefix_hit_chunked_prefill_and_separate_roles():
    """ADVICE r3 (both medium findings).  (1) A request whose first `num_computed_tokens` tokens already sit in vLLM's own
    prefix cache loads the tokens [computed, computed + matched) -- not [0, matched): the slots of the prefix stay untouched,
    the slots behind it receive the right rows.  (2) A prompt prefilled in chunks is stored once, in the step whose chunk
    completes it, from all of its slots.  (3) The scheduler-role and worker-role connectors are SEPARATE objects here (vLLM
    runs them in different processes); only the metadata object passes between them, and the scheduler-role one is built
    without a library."""
    torch = torch_mod()
    from types import SimpleNamespace as NS
    from cxl_speckv_amd.vllm_connector import SpeckvVllmConnector, slot_mapping_for
    lib = open_lib()
    try:
        L, H, D, BS, NB = 2, 8, 128, 16, 64
        sched = SpeckvVllmConnector(None, num_layers=L, num_kv_heads=H, head_dim=D, block_size=BS, max_tokens=256, scheme="fp16")
        work = SpeckvVllmConnector(lib, num_layers=L, num_kv_heads=H, head_dim=D, block_size=BS, max_tokens=256, scheme="fp16")
        with pytest.raises(RuntimeError, match="scheduler role"):
            sched.register_kv_caches({})
        gen = torch.Generator(device="cuda"); gen.manual_seed(77)
        names = [f"model.layers.{i}.self_attn.attn" for i in range(L)]
        caches = {n: torch.zeros((2, NB, BS, H, D), dtype=torch.float16, device="cuda") for n in names}
        work.register_kv_caches(caches)
        n, blk = 75, [5, 9, 2, 40, 41]                     # 75 tokens: an odd length (tail row), 5 blocks of 16
        truth = {li: torch.randn((2, n, H, D), generator=gen, device="cuda").to(torch.float16) for li in range(L)}

        def model_writes(lo, hi):                           # the forward pass of one chunk fills its slots
            slots = torch.tensor(slot_mapping_for(blk, BS, hi - lo, lo), device="cuda")
            for li, name in enumerate(names):
                caches[name].reshape(2, NB * BS, H, D)[:, slots] = truth[li][:, lo:hi]

        def worker_step(meta, lo, hi):
            work.bind_connector_metadata(meta)
            work.start_load_kv(None)
            model_writes(lo, hi)
            for name in names:
                work.save_kv_layer(name, caches[name], None)
            work.wait_for_save()
            work.clear_connector_metadata()

        # ---- chunked prefill: 32 + 32 + 11 tokens over three steps; only the last step stores
        req = NS(request_id="r1", num_tokens=n)
        assert sched.get_num_new_matched_tokens(req, 0) == (0, False)
        m1 = sched.build_connector_meta(NS(scheduled_new_reqs=[NS(req_id="r1", prompt_token_ids=list(range(n)), block_ids=[blk[:2]], num_computed_tokens=0)],
                                           num_scheduled_tokens={"r1": 32}))
        assert m1.requests == []
        worker_step(m1, 0, 32)
        assert work.conn.requests == {}                    # nothing reaches the pool before the prompt is complete
        m2 = sched.build_connector_meta(NS(scheduled_new_reqs=[], num_scheduled_tokens={"r1": 32},
                                           scheduled_cached_reqs=NS(req_ids=["r1"], new_block_ids=[[blk[2:4]]], num_computed_tokens=[32])))
        assert m2.requests == []
        worker_step(m2, 32, 64)
        m3 = sched.build_connector_meta(NS(scheduled_new_reqs=[], num_scheduled_tokens={"r1": 11},
                                           scheduled_cached_reqs=[NS(req_id="r1", new_block_ids=[blk[4:]], num_computed_tokens=64)]))
        assert [(r.is_store, r.first_token, r.num_tokens) for r in m3.requests] == [(True, 0, n)]
        assert m3.requests[0].slot_mapping == slot_mapping_for(blk, BS, n)
        worker_step(m3, 64, n)
        torch.cuda.synchronize()
        eid = m3.requests[0].engine_id
        for li in range(L):
            for kind in (0, 1):
                assert torch.equal(work.conn.kv_rows(eid, li, kind), truth[li][kind]), (li, kind)       # all 75 rows, tail included
                assert torch.equal(work.conn.kv_rows(eid, li, kind, 33, 70), truth[li][kind][33:70])    # a range that starts on an odd row

        # ---- the request comes back with a 32-token hit in vLLM's own prefix cache: external tokens are [32, 64)
        for t in caches.values():
            t.zero_()
        new_blk = [30, 31, 17, 18, 19]
        prefix = {li: caches[names[li]].reshape(2, NB * BS, H, D) for li in range(L)}
        pre_slots = torch.tensor(slot_mapping_for(new_blk, BS, 32), device="cuda")
        for li in range(L):
            prefix[li][:, pre_slots] = 7.0                  # what vLLM's own cache holds for the prefix: must not be rewritten
        req = NS(request_id="r1", num_tokens=n)
        matched, is_async = sched.get_num_new_matched_tokens(req, 32)
        assert (matched, is_async) == ((n - 1) // BS * BS - 32, False)      # 64 usable - 32 computed
        sched.update_state_after_alloc(req, NS(get_block_ids=lambda: [new_blk]), matched)
        m4 = sched.build_connector_meta(NS(scheduled_new_reqs=[NS(req_id="r1", prompt_token_ids=list(range(n)), block_ids=[new_blk], num_computed_tokens=64)],
                                           num_scheduled_tokens={"r1": n - 64}))
        assert [(r.is_store, r.first_token, r.num_tokens) for r in m4.requests] == [(False, 32, 32)]
        assert m4.requests[0].slot_mapping == slot_mapping_for(new_blk, BS, 32, 32)
        work.bind_connector_metadata(m4)
        work.start_load_kv(None)
        work.clear_connector_metadata()
        torch.cuda.synchronize()
        ext_slots = torch.tensor(slot_mapping_for(new_blk, BS, 32, 32), device="cuda")
        for li in range(L):
            assert torch.equal(prefix[li][:, ext_slots], truth[li][:, 32:64]), li
            assert bool((prefix[li][:, pre_slots] == 7.0).all())                                       # the prefix was left alone
            rest = torch.tensor(slot_mapping_for(new_blk, BS, n - 64, 64), device="cuda")
            assert float(prefix[li][:, rest].abs().max()) == 0.0                                       # nothing beyond the match

        # ---- the same request PREEMPTED and resumed -- vLLM lists it among the cached requests (parallel-list shape,
        # resumed_from_preemption) with a fresh block table, never in scheduled_new_reqs; its pool hit must still be loaded
        for t in caches.values():
            t.zero_()
        res_blk = [50, 51, 52, 53, 54]
        req = NS(request_id="r1", num_tokens=n)
        matched, _ = sched.get_num_new_matched_tokens(req, 0)
        assert matched == (n - 1) // BS * BS                                  # 64 of the 75 tokens
        sched.update_state_after_alloc(req, NS(get_block_ids=lambda: [res_blk]), matched)
        m_res = sched.build_connector_meta(NS(scheduled_new_reqs=[], num_scheduled_tokens={"r1": n - matched},
                                              scheduled_cached_reqs=NS(req_ids=["r1"], new_block_ids=[[res_blk]], num_computed_tokens=[matched],
                                                                       resumed_from_preemption=[True])))
        assert [(r.is_store, r.first_token, r.num_tokens) for r in m_res.requests] == [(False, 0, matched)]
        assert m_res.requests[0].slot_mapping == slot_mapping_for(res_blk, BS, matched, 0)
        work.bind_connector_metadata(m_res)
        work.start_load_kv(None)
        work.clear_connector_metadata()
        torch.cuda.synchronize()
        res_slots = torch.tensor(slot_mapping_for(res_blk, BS, matched, 0), device="cuda")
        for li in range(L):
            assert torch.equal(prefix[li][:, res_slots], truth[li][:, :matched]), li
        # ... and a hit that no entry of the step's output carries is an error, not a silent drop
        sched.update_state_after_alloc(req, NS(get_block_ids=lambda: [res_blk]), matched)
        sched._pending_free.append(4242)                                              # a free waiting for this step's metadata ...
        with pytest.raises(RuntimeError, match="without ever being filled"):
            sched.build_connector_meta(NS(scheduled_new_reqs=[], num_scheduled_tokens={}))
        after = sched.build_connector_meta(NS(scheduled_new_reqs=[]))
        assert after.requests == []                                                     # (the state was reset)
        assert after.free_engine_ids == [4242]                                          # ... survives the metadata that died with the error

        # ---- a load for a request this worker never saved fails with a clear error, not a KeyError
        bad = sched.build_connector_meta(NS(scheduled_new_reqs=[]))
        from cxl_speckv_amd.vllm_connector import ReqMeta
        bad.requests.append(ReqMeta("ghost", 999, [0], 0, 1, False))
        work.bind_connector_metadata(bad)
        with pytest.raises(RuntimeError, match="not in this worker's pool"):
            work.start_load_kv(None)
        work.clear_connector_metadata()

        # ---- free travels with the next step's metadata
        sched.free_request("r1")
        assert sched.get_num_new_matched_tokens(NS(request_id="r1", num_tokens=n), 0) == (0, False)
        m5 = sched.build_connector_meta(NS(scheduled_new_reqs=[]))
        assert m5.free_engine_ids == [eid]
        work.bind_connector_metadata(m5)
        assert eid not in work.conn.requests
    finally:
        lib.finalize()


@pytest.mark.parametrize("scheme", [3, 4, 5])
def test_attention_calls_on_different_streams_share_the_scratch_safely(scheme):
    """The split partials of every fused-attention call live in ONE scratch buffer of the engine.  Calls issued back to back on
    different caller streams (no host synchronisation in between) used to be the caller's problem ("use one stream"); the
    library now orders a call behind the previous user of the buffer when the stream changes.  Two sequences with different
    data, attended alternately on two streams that are each kept busy in front of their call, must give exactly what the
    same calls give one at a time."""
    torch = torch_mod()
    from tests.test_gpu_full_size import H, D, G
    T, L = 4096, 4
    n_pages = T * L * H * D * 2 * 2 // PAGE
    lib = open_lib()
    try:
        lib.set_compression_scheme(scheme)
        attend = {3: lib.attend_int4, 4: lib.attend_fp8, 5: lib.attend_mx4}[scheme]
        rng = np.random.default_rng(900 + scheme)
        hs, qs = [], []
        for i in range(2):
            x = (rng.standard_normal((n_pages, N)) * (1.0 + i)).astype(np.float16)
            h = lib.alloc(n_pages * PAGE)
            lib.set_layout(h, T, L, H, D, 2)
            lib.write(h, 0, x.ctypes.data, x.nbytes, False)
            hs.append(h)
            qs.append(torch.from_numpy((rng.standard_normal((L, H, G, D)) * 1.5).astype(np.float16).view(np.int16)).cuda())
        sm = 1.0 / np.sqrt(D)

        def call(i, out, lse, stream):
            attend(hs[i], 0, L, qs[i].data_ptr(), G, 0, T, sm, out.data_ptr(), lse.data_ptr(), stream)

        want = []
        for i in range(2):
            out = torch.empty((L, H, G, D), dtype=torch.float32, device="cuda")
            lse = torch.empty((L, H, G), dtype=torch.float32, device="cuda")
            call(i, out, lse, None)
            torch.cuda.synchronize()
            want.append((out.clone(), lse.clone()))
        streams = [torch.cuda.Stream(), torch.cuda.Stream()]
        ballast = torch.empty(64 << 20, dtype=torch.uint8, device="cuda")
        for rep in range(6):
            outs = [(k & 1, torch.full((L, H, G, D), float("nan"), dtype=torch.float32, device="cuda"),
                     torch.full((L, H, G), float("nan"), dtype=torch.float32, device="cuda")) for k in range(8)]
            torch.cuda.synchronize()
            for k, (i, out, lse) in enumerate(outs):          # back to back, no host synchronisation in between
                if (k + rep) % 3 == 0:
                    with torch.cuda.stream(streams[i]):
                        ballast.fill_(k)                      # this stream's call starts late: the other one overtakes it
                call(i, out, lse, streams[i].cuda_stream)
            torch.cuda.synchronize()
            for i, out, lse in outs:
                assert torch.equal(out, want[i][0]), (rep, i)
                assert torch.equal(lse, want[i][1]), (rep, i)
    finally:
        lib.finalize()
