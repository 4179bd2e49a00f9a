"""-m gpu: what round 2 added to the engine, against the oracle / against the round-1 paths.

  * device-side prefetch flush: first-occurrence order, ring slots in that order, many allocations per flush,
    ring wrap / eviction done by the fetch kernel;
  * copy-engine fetch (hipMemcpyPeerAsync runs + local decompress) == fused peer-load kernel, bit for bit;
  * migration frees exact record runs;
  * speckv_prefetch under the reference's unchanged shim (8 symbols only);
  * speckv_free does not stall on the device; batch attention refuses stream capture.
The "peers" are pools on the same GPU (SPECKV_POOL_DEVICES=0,0,0): placement, copies and bookkeeping are the code
that runs across xGMI; only the link is missing on a one-GPU box.
"""
import ctypes as C
import os
import time

import numpy as np
import pytest

import cxl_speckv_amd as pkg
from cxl_speckv_amd.speckv_ctypes import SpeckvError
from tests._gpu import N, assert_same_float_bits, dev_to_host, graph_capture, torch_mod, set_tuning
from tests.test_gpu_engine import synth

pytestmark = pytest.mark.gpu
PAGE = 4096


def open_lib(**env):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update({k: str(v) for k, v in env.items()})
    try:
        return pkg.SpeckvLib(pkg.library_path(), "hip:0")
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def expected_flush(oracle, reqs, geom, n_pages, resident=()):
    """Oracle: candidate pages of every request in request order, first occurrence kept, resident pages skipped."""
    T, L, H, D, bpe = geom
    seen, out = set(resident), []
    fl = np.zeros(n_pages, np.uint32)
    for p in resident:
        fl[p] = 2
    for (req, layer, pos, k) in reqs:
        for p in oracle.prefetch_pages(req, layer, pos, k, L, T, H, D, bpe, n_pages, fl).tolist():
            if p not in seen:
                seen.add(p); out.append(p)
    return out


@pytest.mark.parametrize("host_words", ["scatter", "fetch"])
def test_flush_keeps_first_occurrence_order_and_assigns_slots_in_it(oracle, host_words):
    """host_words: the pages' host-visible residency words stored by the scatter kernel (small flushes) or by the fetch
    launch itself (large ones) -- both forced here; translate() derives the L2 bit from exactly those words."""
    lib = open_lib(SPECKV_L2_MB=64, SPECKV_FLUSH_HOST_WORDS=host_words)
    try:
        lib.set_compression_scheme(2)
        geom = (T, L, H, D, bpe) = (1024, 6, 8, 128, 2)
        n_pages = 2 * T * L * H * D * bpe // PAGE
        h = lib.alloc(n_pages * PAGE)
        lib.set_layout(h, *geom)
        x = synth(n_pages, seed=5)
        lib.write(h, 0, x.ctypes.data, x.nbytes, False)
        rng = np.random.default_rng(77)
        n = 700
        reqs = [(0, int(rng.integers(0, L)), int(rng.integers(0, 64)) * 2, int(rng.integers(1, 9))) for _ in range(n)]
        reqs += reqs[:50]                                      # exact duplicates, overlapping look-aheads everywhere
        pre = [3, 4, 1025]                                     # already resident: must be filtered
        for p in pre:
            lib.access(h, p * PAGE, 8)
        want = expected_flush(oracle, reqs, geom, n_pages, resident=pre)
        lib.prefetch_batch([r[0] for r in reqs], [r[1] for r in reqs], [r[2] for r in reqs], [r[3] for r in reqs])
        issued = lib.prefetch_flush()
        assert issued == len(want)
        lib.sync()
        infos = [lib.translate(h, p * PAGE) for p in want]
        assert all(i.flags & 2 for i in infos)
        base = infos[0].cache_addr
        assert [i.cache_addr - base for i in infos] == [j * PAGE for j in range(len(want))]   # slot = first slot + rank
        resident = {p for p in range(n_pages) if lib.translate(h, p * PAGE).flags & 3}
        assert resident == set(want) | set(pre)
        scales, lens, recs = oracle.compress_blocks_f16(x, 2, 0)
        for j in (0, 1, len(want) // 2, len(want) - 1):
            p = want[j]
            dec = oracle.decompress_block_f16(recs[p, :lens[p]], scales[p], 2, 0, N)
            assert_same_float_bits(dev_to_host(infos[j].cache_addr, PAGE).view(np.float16), dec)
        st = lib.stats()
        assert st.total_prefetches == len(want) and st.prefetch_dropped == 0
        assert lib.poll_complete() >= len(want)
        # the same requests again: everything is resident, nothing is issued
        lib.prefetch_batch([r[0] for r in reqs], [r[1] for r in reqs], [r[2] for r in reqs], [r[3] for r in reqs])
        assert lib.prefetch_flush() == 0
    finally:
        lib.finalize()


@pytest.mark.parametrize("host_words", ["scatter", "fetch"])
def test_flush_over_many_allocations_with_request_bindings(oracle, host_words):
    """One allocation per sequence (the serving layout): request ids are bound to handles, one flush serves them all."""
    lib = open_lib(SPECKV_FLUSH_HOST_WORDS=host_words)
    try:
        lib.set_compression_scheme(1)
        geom = (T, L, H, D, bpe) = (256, 4, 8, 128, 2)
        n_pages = 2 * T * L * H * D * bpe // PAGE
        n_seq = 12
        hs, data = [], []
        for s in range(n_seq):
            h = lib.alloc(n_pages * PAGE)
            lib.set_layout(h, *geom)
            x = synth(n_pages, seed=100 + s)
            lib.write(h, 0, x.ctypes.data, x.nbytes, False)
            lib.bind_request(1000 + s, h, 0)
            hs.append(h); data.append(x)
        pos = [10 + 6 * s for s in range(n_seq)]
        reqs, layers, poss = [], [], []
        for s in range(n_seq):
            for layer in range(L):
                reqs.append(1000 + s); layers.append(layer); poss.append(pos[s])
        reqs += [5, 77777]; layers += [0, 1]; poss += [4, 4]        # unbound ids address the newest allocation as req 5 / 77777: out of range
        lib.prefetch_batch(reqs, layers, poss, [4] * len(reqs))
        issued = lib.prefetch_flush()
        lib.sync()
        total = 0
        for s in range(n_seq):
            want = expected_flush(oracle, [(0, layer, pos[s], 4) for layer in range(L)], geom, n_pages)
            got = {p for p in range(n_pages) if lib.translate(hs[s], p * PAGE).flags & 2}
            assert got == set(want), s
            total += len(want)
            p = want[-1]
            ptr = lib.access(hs[s], p * PAGE, 16)
            sc, ln, rc = oracle.compress_blocks_f16(data[s][p:p + 1], 1, 0)
            dec = oracle.decompress_block_f16(rc[0, :ln[0]], sc[0], 1, 0, N)
            assert_same_float_bits(dev_to_host(ptr, PAGE).view(np.float16), dec)
        assert issued == total
        st = lib.stats()
        assert st.prefetch_dropped == 2 and st.total_prefetches == total
        # a freed allocation takes its binding with it; its requests are dropped, the others still work
        lib.free(hs[0])
        lib.prefetch_batch([1000, 1001], [0, 0], [100, 100], [2, 2])
        assert lib.prefetch_flush() > 0
        assert lib.stats().prefetch_dropped == 3
    finally:
        lib.finalize()


@pytest.mark.parametrize("seq_limit,host_words", [(None, "scatter"), (700, "scatter"), (None, "fetch"), (700, "fetch")])
def test_ring_wrap_and_eviction_by_the_fetch_kernel(oracle, seq_limit, host_words):
    """A ring of 256 slots under flushes that wrap it several times: what is flagged resident really is there, evicted
    pages lose their bit, synchronous misses interleave with flushes, the host's ring hand stays in step.  The host derives
    "still in the ring" from the sequence number the fetch kernel stores for a page (its only host-visible store): with
    seq_limit the 32-bit sequence numbers are renumbered every few hundred slots instead of every 3 * 10^9, so the run
    crosses that code several times with live pages on both sides."""
    env = {"SPECKV_L2_MB": 1, "SPECKV_L1_MB": 1, "SPECKV_FLUSH_HOST_WORDS": host_words}
    if seq_limit:
        env["SPECKV_RING_SEQ_LIMIT"] = seq_limit
    lib = open_lib(**env)
    try:
        lib.set_compression_scheme(1)
        geom = (T, L, H, D, bpe) = (512, 2, 8, 128, 2)
        n_pages = 2 * T * L * H * D * bpe // PAGE            # 2048
        h = lib.alloc(n_pages * PAGE)
        lib.set_layout(h, *geom)
        x = synth(n_pages, seed=9)
        lib.write(h, 0, x.ctypes.data, x.nbytes, False)
        y = np.empty_like(x); lib.read(h, 0, y.ctypes.data, y.nbytes, False)
        rng = np.random.default_rng(3)
        for step in range(40):
            n = int(rng.integers(1, 30))
            lib.prefetch_batch([0] * n, [int(v) for v in rng.integers(0, L, n)], [int(v) * 2 for v in rng.integers(0, T // 2 - 10, n)],
                               [int(v) for v in rng.integers(1, 9, n)])
            issued = lib.prefetch_flush()
            assert issued <= 128                                  # never more than half the ring per flush
            if step % 3 == 0:                                     # a synchronous miss between flushes
                p = int(rng.integers(0, n_pages))
                ptr = lib.access(h, p * PAGE, 8)
                assert dev_to_host(ptr, PAGE).tobytes() == y[p].tobytes()
            if step % 5 == 4:
                lib.sync()
                infos = [(p, lib.translate(h, p * PAGE)) for p in range(n_pages)]
                res = [(p, i) for p, i in infos if i.flags & 2]
                assert len(res) <= 256
                assert len({i.cache_addr for _, i in res}) == len(res)              # one page per slot
                for p, i in res[::7]:
                    assert dev_to_host(i.cache_addr, PAGE).tobytes() == y[p].tobytes()
        assert lib.stats().total_prefetches > 256 * 3
    finally:
        lib.finalize()


@pytest.mark.parametrize("scheme", [2, 0, 4, 3, 5])
def test_copy_engine_fetch_equals_kernel_fetch(scheme):
    """speckv_ext_fetch_range_engine: engine 2 (coalesced hipMemcpyPeerAsync runs on per-peer streams into staging, local
    decompress, double-buffered) against engine 1 (fused peer-load kernel), pool striped over three peers."""
    torch = torch_mod()
    lib = open_lib(SPECKV_POOL_DEVICES="0,0,0", SPECKV_STAGE_MB=1)      # 1 MiB staging: many chunks
    try:
        lib.set_compression_scheme(scheme)
        n = 5000                                                      # not a multiple of 3, > several chunks
        h = lib.alloc(n * PAGE)
        x = synth(n, seed=33 + scheme)
        lib.write(h, 0, x.ctypes.data, x.nbytes, False)
        s = torch.cuda.Stream()
        for first, cnt, f32 in ((0, n, False), (1, n - 1, False), (7, 1000, True), (4999, 1, False), (2, 2, False), (100, 0, False)):
            a = torch.full((max(cnt, 1), N * (2 if f32 else 1)), -1, dtype=torch.int16, device="cuda")
            b = torch.full_like(a, -2)
            lib.fetch_range(h, first, cnt, a.data_ptr(), f32, s.cuda_stream, engine=1)
            lib.fetch_range(h, first, cnt, b.data_ptr(), f32, s.cuda_stream, engine=2)
            torch.cuda.synchronize()
            if cnt:
                assert torch.equal(a, b), (scheme, first, cnt, f32)
        assert lib.stats().copy_engine_runs > 0 and lib.stats().copy_engine_bytes > 0
        # on the engine's own stream too, and the default (auto) choice on a same-GPU pool is the kernel
        runs = lib.stats().copy_engine_runs
        a = torch.empty((n, N), dtype=torch.int16, device="cuda"); b = torch.empty_like(a)
        lib.fetch_range(h, 0, n, a.data_ptr(), False, None, engine=2)
        lib.fetch_range(h, 0, n, b.data_ptr(), False, None)
        lib.sync()
        assert torch.equal(a, b) and lib.stats().copy_engine_runs > runs
        runs = lib.stats().copy_engine_runs
        lib.fetch_range(h, 0, n, b.data_ptr(), False, s.cuda_stream)
        torch.cuda.synchronize()
        assert lib.stats().copy_engine_runs == runs
        # after a migration the records are no longer in striping order: engine 2 is refused, auto still works
        lib.migrate(h, 10, 31, 1)
        with pytest.raises(SpeckvError) as ei:
            lib.fetch_range(h, 0, n, b.data_ptr(), False, s.cuda_stream, engine=2)
        assert ei.value.status == -4
        lib.fetch_range(h, 0, n, b.data_ptr(), False, s.cuda_stream)
        torch.cuda.synchronize()
        assert torch.equal(a, b)
    finally:
        lib.finalize()


@pytest.mark.parametrize("scheme", [4, 3, 1, 5])
def test_migration_frees_exact_runs(scheme, oracle):
    """records are 2048 / 1152 B; migrating an odd number of pages out of a striped pool and then
    allocating and writing a second handle must leave every record of the first one intact."""
    lib = open_lib(SPECKV_POOL_DEVICES="0,0,0", SPECKV_SLAB_MB=8)
    try:
        lib.set_compression_scheme(scheme)
        n = 1001
        h = lib.alloc(n * PAGE)
        x = synth(n, seed=55)
        lib.write(h, 0, x.ctypes.data, x.nbytes, False)
        before = np.empty_like(x); lib.read(h, 0, before.ctypes.data, before.nbytes, False)
        reserved = lib.stats().pool_bytes_reserved
        lib.migrate(h, 3, 37, 1)                                   # 37 single-record runs leave pools 0 and 2 (and 1)
        lib.migrate(h, 500, 1, 0)
        h2 = lib.alloc(300 * PAGE)                                 # lands in the holes and around them
        z = synth(300, seed=56)
        lib.write(h2, 0, z.ctypes.data, z.nbytes, False)
        h3 = lib.alloc(64 * PAGE, preferred_node=1)
        lib.write(h3, 0, z.ctypes.data, 64 * PAGE, False)
        after = np.empty_like(x); lib.read(h, 0, after.ctypes.data, after.nbytes, False)
        assert after.tobytes() == before.tobytes()
        got2 = np.empty_like(z); lib.read(h2, 0, got2.ctypes.data, got2.nbytes, False)
        sc, ln, rc = oracle.compress_blocks_f16(z, scheme, 0)
        assert_same_float_bits(got2, oracle.decompress_blocks_f16(rc, ln, sc, scheme, 0))
        assert lib.stats().pool_bytes_reserved <= reserved + (16 << 20)
        lib.free(h); lib.free(h2); lib.free(h3)
        lib.sync()
        h4 = lib.alloc(n * PAGE)                                   # everything coalesced back: fits without new slabs
        assert lib.stats().pool_bytes_reserved <= reserved + (16 << 20)
        lib.free(h4)
    finally:
        lib.finalize()


def _raw_reference_surface():
    """The 8 reference symbols only, bound as the reference's speckv_ctypes.py binds them."""
    lib = pkg.load_library()
    lib.speckv_init.argtypes = [C.c_char_p]; lib.speckv_init.restype = C.c_int
    lib.speckv_finalize.restype = None
    lib.speckv_alloc.argtypes = [C.c_size_t, C.c_void_p, C.POINTER(C.c_uint64)]; lib.speckv_alloc.restype = C.c_int
    lib.speckv_free.argtypes = [C.c_uint64]
    lib.speckv_access.argtypes = [C.c_uint64, C.c_uint64, C.c_size_t, C.POINTER(C.c_void_p)]; lib.speckv_access.restype = C.c_int
    lib.speckv_prefetch.argtypes = [C.c_uint32, C.c_uint16, C.c_uint32, C.c_uint32, C.POINTER(C.c_int32), C.c_uint32]
    lib.speckv_prefetch.restype = C.c_int
    return lib


@pytest.mark.parametrize("mode", ["env", "inferred"])
def test_prefetch_under_the_reference_shim_unchanged(mode, oracle):
    """BASELINE north_star: 'drops into the existing vLLM integration shim unchanged'.  The reference's allocate() sends
    no geometry (vllm_speckv_backend.py:26-43).  Call order of its decode loop (:104-129) with ONLY the 8 reference
    functions: allocate -> per token, prefetch_step per layer -> get_kv_ptr.  Pages of the look-ahead must become
    L2-resident (observed through speckv_ext_translate, which the caller would not need)."""
    from cxl_speckv_amd.speckv_ctypes import PageInfo
    T, L, H, D, bpe = 256, 4, 8, 128, 2
    total = 2 * T * L * H * D * bpe
    n_pages = total // PAGE
    if mode == "env":
        os.environ["SPECKV_LAYOUT"] = f"{T},{L},{H},{D},{bpe}"
    lib = _raw_reference_surface()
    try:
        assert lib.speckv_init(b"hip:0") == 0
        h = C.c_uint64()
        assert lib.speckv_alloc(total, None, C.byref(h)) == 0
        toks = (C.c_int32 * 16)(*range(1, 17))
        out = C.c_void_p()
        entry = D * bpe

        def off(layer, head, pos, kind):
            return ((((0 * L + layer) * 2 + kind) * T + pos) * H + head) * entry
        if mode == "inferred":
            # the shim reads the current position first (that is how the entry size becomes known)
            assert lib.speckv_access(h, off(0, 0, 10, 0), entry, C.byref(out)) == 0
        for pos in (10, 11):                                   # two tokens: the layer index falling back ends a step
            for layer in range(L):
                assert lib.speckv_prefetch(0, layer, pos, 4, toks, 16) == 0
        assert lib.speckv_access(h, off(1, 3, 12, 1), entry, C.byref(out)) == 0 and out.value
        lib.speckv_ext_translate.argtypes = [C.c_uint64, C.c_uint64, C.POINTER(PageInfo)]
        want = set()
        for layer in range(L):
            want |= set(oracle.prefetch_pages(0, layer, 10, 4, L, T, H, D, bpe, n_pages).tolist())
        info = PageInfo()
        res = set()
        for p in range(n_pages):
            assert lib.speckv_ext_translate(h, p * PAGE, C.byref(info)) == 0
            if info.flags & 2:
                res.add(p)
        assert want <= res, sorted(want - res)[:8]
        assert lib.speckv_free(h) == 0
    finally:
        lib.speckv_finalize()
        os.environ.pop("SPECKV_LAYOUT", None)


def test_free_returns_without_waiting_for_the_device():
    """speckv_free used to hipDeviceSynchronize(); now an allocation a caller stream may still read is released later,
    and the call returns at once even while an unrelated long kernel occupies the GPU."""
    torch = torch_mod()
    lib = open_lib()
    try:
        lib.set_compression_scheme(2)
        n = 4096
        h = lib.alloc(n * PAGE)
        x = synth(n, seed=2)
        lib.write(h, 0, x.ctypes.data, x.nbytes, False)
        dst = torch.empty((n, N), dtype=torch.int16, device="cuda")
        busy, user = torch.cuda.Stream(), torch.cuda.Stream()
        torch.cuda.synchronize()
        with torch.cuda.stream(busy):
            torch.cuda._sleep(int(2.0e9))                          # ~1 s of GPU time on another stream
        with torch.cuda.stream(user):
            torch.cuda._sleep(int(4.0e8))                          # the caller's own stream is busy too (~0.2 s)
        lib.fetch_range(h, 0, n, dst.data_ptr(), False, user.cuda_stream)
        t0 = time.perf_counter()
        lib.free(h)
        dt = time.perf_counter() - t0
        assert dt < 0.05, dt
        h2 = lib.alloc(n * PAGE)                                   # still works while the old records wait for their stream
        lib.write(h2, 0, x.ctypes.data, x.nbytes, False)
        torch.cuda.synchronize()
        ref = np.empty_like(x); lib.read(h2, 0, ref.ctypes.data, ref.nbytes, False)
        assert dst.cpu().numpy().view(np.float16).tobytes() == ref.tobytes()       # the fetch read intact records
        lib.sync()
        lib.free(h2)
    finally:
        lib.finalize()


def test_batch_attention_refuses_stream_capture():
    """the batch forms stage their descriptors per call and must not be captured into a graph."""
    torch = torch_mod()
    lib = open_lib()
    try:
        lib.set_compression_scheme(4)
        T, H, D, G = 64, 8, 128, 8
        n_pages = 2 * T * H * D * 2 // PAGE
        h = lib.alloc(n_pages * PAGE)
        lib.set_layout(h, T, 1, H, D, 2)
        x = synth(n_pages, seed=4)
        lib.write(h, 0, x.ctypes.data, x.nbytes, False)
        q = torch.randn((1, H, G, D), device="cuda").to(torch.float16)
        out = torch.empty((1, H, G, D), dtype=torch.float32, device="cuda")
        s = torch.cuda.Stream()
        lib.attend_fp8_batch([h], 0, q.data_ptr(), G, [T], 0.1, out.data_ptr(), None, s.cuda_stream)    # eager: fine
        torch.cuda.synchronize()
        eager = out.clone()
        lib.attend_fp8(h, 0, 1, q.data_ptr(), G, 0, T, 0.1, out.data_ptr(), None, s.cuda_stream)         # warm-up sizes the scratch
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with graph_capture(g, s):
            with pytest.raises(SpeckvError) as ei:
                lib.attend_fp8_batch([h], 0, q.data_ptr(), G, [T], 0.1, out.data_ptr(), None, s.cuda_stream)
            assert ei.value.status == -4
            lib.attend_fp8(h, 0, 1, q.data_ptr(), G, 0, T, 0.1, out.data_ptr(), None, s.cuda_stream)     # per-sequence form: capturable
        out.zero_()
        g.replay(); torch.cuda.synchronize()
        assert torch.allclose(out, eager, rtol=1e-5, atol=1e-6)
    finally:
        lib.finalize()


@pytest.mark.parametrize("scheme", [4, 3, 5])
def test_planned_batch_attention_replays_under_a_graph(scheme):
    """The planned form of the batch attention: speckv_ext_attend_batch_plan once per step outside the graph, the
    per-layer speckv_ext_attend_*_planned calls captured ONCE and replayed while the sequences grow.  Every replay must
    equal the per-sequence entry point at the lengths of that step (the tolerance of two split arrangements, see
    test_fused_attention_batch_of_sequences); sequences that finish in one split are written by the attention kernel,
    the others by the merge, in the same launch."""
    torch = torch_mod()
    lib = open_lib()
    try:
        lib.set_compression_scheme(scheme)
        single_fn = {4: lib.attend_fp8, 3: lib.attend_int4, 5: lib.attend_mx4}[scheme]
        T, L, H, D, G = 2048, 3, 8, 128, 4
        rng = np.random.default_rng(97)
        steps = [[64, 1024, 2, 600, 0], [66, 1026, 4, 602, 0], [512, 2048, 34, 1600, 2], [2048, 2048, 2048, 2048, 2048]]
        n_seq = len(steps[0])
        handles = []
        for _ in range(n_seq):
            h = lib.alloc(T * L * H * D * 2 * 2)
            lib.set_layout(h, T, L, H, D, 2)
            n_pages = T * L * H * D * 2 * 2 // PAGE
            x = (rng.standard_normal((n_pages, N)) * rng.uniform(0.2, 2.0, (n_pages, 1))).astype(np.float16)
            lib.write(h, 0, x.ctypes.data, x.nbytes, False)
            handles.append(h)
        sm = 1.0 / np.sqrt(D)
        q = torch.from_numpy(rng.standard_normal((L, n_seq, H, G, D)).astype(np.float16)).cuda()
        out = torch.zeros((L, n_seq, H, G, D), dtype=torch.float32, device="cuda")
        lse = torch.zeros((L, n_seq, H, G), dtype=torch.float32, device="cuda")
        plan_bytes = lib.attend_plan_bytes(n_seq)
        assert plan_bytes == n_seq * (64 + 4)              # one descriptor and one dispatch-order index per sequence
        plan = torch.zeros(plan_bytes, dtype=torch.uint8, device="cuda")
        s = torch.cuda.Stream()
        for tps in (None, "8", "64"):                 # the rule's geometry, one with real splits, one without any (no merge launch)
            if tps is None: set_tuning("attend_tiles_per_split", 0)
            else: set_tuning("attend_tiles_per_split", tps)
            try:
                def run_layers():
                    for layer in range(L):
                        lib.attend_planned(scheme, plan.data_ptr(), n_seq, layer, q[layer].data_ptr(), G, T, sm,
                                           out[layer].data_ptr(), lse[layer].data_ptr(), s.cuda_stream)
                # no plan at that address yet / wrong shape -> refused, nothing launched
                other = torch.zeros(plan_bytes, dtype=torch.uint8, device="cuda")
                with pytest.raises(SpeckvError):
                    lib.attend_planned(scheme, other.data_ptr(), n_seq, 0, q[0].data_ptr(), G, T, sm, out[0].data_ptr(), None, s.cuda_stream)
                lib.attend_batch_plan(handles, steps[0], T, plan.data_ptr(), plan_bytes, s.cuda_stream)
                with pytest.raises(SpeckvError):
                    lib.attend_planned(scheme, plan.data_ptr(), n_seq, L, q[0].data_ptr(), G, T, sm, out[0].data_ptr(), None, s.cuda_stream)
                with pytest.raises(SpeckvError):
                    lib.attend_planned(scheme, plan.data_ptr(), n_seq, 0, q[0].data_ptr(), G, T - 2, sm, out[0].data_ptr(), None, s.cuda_stream)
                with pytest.raises(SpeckvError):
                    lib.attend_planned({4: 3, 3: 4, 5: 4}[scheme], plan.data_ptr(), n_seq, 0, q[0].data_ptr(), G, T, sm, out[0].data_ptr(), None, s.cuda_stream)     # another format than the plan's
                run_layers()                              # warm-up: sizes the scratch
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                with graph_capture(g, s):
                    with pytest.raises(SpeckvError):      # the plan itself stays outside
                        lib.attend_batch_plan(handles, steps[0], T, plan.data_ptr(), plan_bytes, s.cuda_stream)
                    run_layers()
                for lens in steps:
                    lib.attend_batch_plan(handles, lens, T, plan.data_ptr(), plan_bytes, s.cuda_stream)
                    s.synchronize()
                    out.fill_(float("nan")); lse.fill_(float("nan"))
                    torch.cuda.synchronize()
                    g.replay()
                    torch.cuda.synchronize()
                    for layer in range(L):
                        for i, (h, n) in enumerate(zip(handles, lens)):
                            if n == 0:
                                assert float(out[layer, i].abs().max()) == 0.0
                                continue
                            one = torch.empty((H, G, D), dtype=torch.float32, device="cuda")
                            one_lse = torch.empty((H, G), dtype=torch.float32, device="cuda")
                            single_fn(h, layer, 1, q[layer, i].data_ptr(), G, 0, n, sm, one.data_ptr(), one_lse.data_ptr())
                            torch.cuda.synchronize()
                            scale = float(one.abs().max()) + 1e-6
                            assert float((out[layer, i] - one).abs().max()) <= 1e-3 * scale, (tps, lens, layer, i)
                            assert float((lse[layer, i] - one_lse).abs().max()) <= 1e-4, (tps, lens, layer, i)
                # a length beyond the planned bound is refused by the plan
                with pytest.raises(SpeckvError):
                    lib.attend_batch_plan(handles, [T] * n_seq, T - 2, plan.data_ptr(), plan_bytes, s.cuda_stream)
                del g
            finally:
                set_tuning("attend_tiles_per_split", 0)
        for h in handles:
            lib.free(h)
    finally:
        lib.finalize()


@pytest.mark.parametrize("scheme", [4, 3, 5])
@pytest.mark.parametrize("first", ["tail", "equal"])
def test_a_plans_first_batch_fixes_the_room_for_pieces_under_a_graph(scheme, first):
    """Members of different lengths get pieces on account of the lengths (ring_rule.hpp ragged_tiles_per_piece) and the rows-first grid -- decided by
    the FIRST plan of a shape in a buffer, kept by every later plan of that shape there, because the launches may sit in a captured graph.  Capture
    once behind the first plan, then replay behind plans of other length sets (a heavy tail, nearly equal lengths, a tail elsewhere, empty members):
    every replay equals the per-sequence entry point.  `first` = what the first plan sees: a heavy tail (room for pieces, merge launch in the graph)
    or equal lengths (no room: later tails run as whole sequences)."""
    torch = torch_mod()
    lib = open_lib()
    try:
        lib.set_compression_scheme(scheme)
        single_fn = {4: lib.attend_fp8, 3: lib.attend_int4, 5: lib.attend_mx4}[scheme]
        T, L, H, D, G = 4096, 1, 8, 128, 8
        rng = np.random.default_rng(211)
        n_seq = 24
        tail = [int(v) * 2 for v in rng.integers(16, 200, n_seq)]
        tail[3], tail[17] = T, T - 64                          # 128 tiles against ~7: far over a CU's share
        equal = [2048 + 2 * i for i in range(n_seq)]
        tail2 = [int(v) * 2 for v in rng.integers(1, 100, n_seq)]
        tail2[0], tail2[23], tail2[7] = T, 0, 2
        order = ([tail, equal, tail2, tail] if first == "tail" else [equal, tail, tail2, equal])
        n_pages = T * L * H * D * 2 * 2 // PAGE
        x = (rng.standard_normal((n_pages, N)) * rng.uniform(0.2, 2.0, (n_pages, 1))).astype(np.float16)
        handles = []
        for _ in range(n_seq):
            h = lib.alloc(T * L * H * D * 2 * 2)
            lib.set_layout(h, T, L, H, D, 2)
            lib.write(h, 0, x.ctypes.data, x.nbytes, False)
            handles.append(h)
        sm = 1.0 / np.sqrt(D)
        q = torch.from_numpy(rng.standard_normal((n_seq, H, G, D)).astype(np.float16)).cuda()
        out = torch.zeros((n_seq, H, G, D), dtype=torch.float32, device="cuda")
        lse = torch.zeros((n_seq, H, G), dtype=torch.float32, device="cuda")
        plan_bytes = lib.attend_plan_bytes(n_seq)
        plan = torch.zeros(plan_bytes, dtype=torch.uint8, device="cuda")
        s = torch.cuda.Stream()
        def run():
            lib.attend_planned(scheme, plan.data_ptr(), n_seq, 0, q.data_ptr(), G, T, sm, out.data_ptr(), lse.data_ptr(), s.cuda_stream)
        lib.attend_batch_plan(handles, order[0], T, plan.data_ptr(), plan_bytes, s.cuda_stream)
        run(); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with graph_capture(g, s):
            run()
        one = torch.empty((H, G, D), dtype=torch.float32, device="cuda")
        one_lse = torch.empty((H, G), dtype=torch.float32, device="cuda")
        for lens in order:
            # (another shape through the same buffer in between -- half the bound: this shape's room must survive it)
            lib.attend_batch_plan(handles, [min(n, T // 2) for n in (tail if lens is equal else equal)], T // 2, plan.data_ptr(), plan_bytes, s.cuda_stream)
            lib.attend_batch_plan(handles, lens, T, plan.data_ptr(), plan_bytes, s.cuda_stream)
            s.synchronize()
            out.fill_(float("nan")); lse.fill_(float("nan"))
            torch.cuda.synchronize()
            g.replay()
            torch.cuda.synchronize()
            for i, (h, n) in enumerate(zip(handles, lens)):
                if n == 0:
                    assert float(out[i].abs().max()) == 0.0
                    continue
                single_fn(h, 0, 1, q[i].data_ptr(), G, 0, n, sm, one.data_ptr(), one_lse.data_ptr())
                torch.cuda.synchronize()
                scale = float(one.abs().max()) + 1e-6
                assert float((out[i] - one).abs().max()) <= 1e-3 * scale, (first, i, n)
                assert float((lse[i] - one_lse).abs().max()) <= 1e-4, (first, i, n)
        del g
        for h in handles:
            lib.free(h)
    finally:
        lib.finalize()


@pytest.mark.parametrize("subset", [False, True])
def test_fold_tail_adds_one_position(subset):
    """speckv_ext_attend_fold_tail against the same three lines in float64: rows chosen by an index list or all of them,
    a strided tail tensor ([rows][layers][heads][dim], one layer taken), rows with nothing stored (out 0, lse -inf) and
    scores far above / below the stored log-sum-exp.  Tolerance: fp32 dot product of 128 terms and v_exp / v_log, 2e-5
    relative on out, 1e-5 absolute on lse."""
    torch = torch_mod()
    lib = open_lib()
    try:
        rng = np.random.default_rng(131)
        n_seq, H, G, D, Lyr, layer = 9, 8, 5, 128, 3, 1
        rows = [7, 0, 3, 8] if subset else list(range(n_seq))
        q = rng.standard_normal((n_seq, H, G, D)).astype(np.float16)
        q[3] *= 40.0                                               # score far above the stored lse
        q[8] *= -40.0                                              # ... and far below (sign flips with k, both occur)
        out = rng.standard_normal((n_seq, H, G, D)).astype(np.float32)
        lse = rng.uniform(-3, 9, (n_seq, H, G)).astype(np.float32)
        out[0] = 0.0; lse[0] = -np.inf                              # a sequence without stored positions
        kt = rng.standard_normal((len(rows), Lyr, H, D)).astype(np.float16)
        vt = rng.standard_normal((len(rows), Lyr, H, D)).astype(np.float16)
        sm = 0.0884
        want_out, want_lse = out.astype(np.float64), lse.astype(np.float64)
        for i, b in enumerate(rows):
            sc = np.einsum("hgd,hd->hg", q[b].astype(np.float64), kt[i, layer].astype(np.float64)) * sm
            new = np.logaddexp(want_lse[b], sc)
            want_out[b] = want_out[b] * np.exp(want_lse[b] - new)[..., None] + vt[i, layer].astype(np.float64)[:, None, :] * np.exp(sc - new)[..., None]
            want_lse[b] = new
        d_q, d_out, d_lse = torch.from_numpy(q).cuda(), torch.from_numpy(out).cuda(), torch.from_numpy(lse).cuda()
        d_k, d_v = torch.from_numpy(kt).cuda(), torch.from_numpy(vt).cuda()
        d_rows = torch.tensor(rows, dtype=torch.int32).cuda() if subset else None
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        lib.attend_fold_tail(len(rows), d_rows.data_ptr() if subset else 0, H, G, d_q.data_ptr(), d_k.data_ptr() + layer * H * D * 2,
                             d_v.data_ptr() + layer * H * D * 2, Lyr * H * D, sm, d_out.data_ptr(), d_lse.data_ptr(), s.cuda_stream)
        torch.cuda.synchronize()
        got_out, got_lse = d_out.cpu().numpy(), d_lse.cpu().numpy()
        assert np.all(np.abs(got_out - want_out) <= 2e-5 * np.abs(want_out) + 2e-6)
        assert np.all(np.abs(got_lse - want_lse) <= 1e-5 * np.maximum(1.0, np.abs(want_lse)))
        untouched = [b for b in range(n_seq) if b not in rows]
        assert np.array_equal(got_out[untouched], out[untouched]) and np.array_equal(got_lse[untouched], lse[untouched])
        with pytest.raises(SpeckvError):                            # tails overlapping: stride below one row
            lib.attend_fold_tail(len(rows), 0, H, G, d_q.data_ptr(), d_k.data_ptr(), d_v.data_ptr(), H * D - 2, sm, d_out.data_ptr(),
                                 d_lse.data_ptr(), s.cuda_stream)
        with pytest.raises(SpeckvError):                            # the log-sum-exp is not optional here
            lib.attend_fold_tail(len(rows), 0, H, G, d_q.data_ptr(), d_k.data_ptr(), d_v.data_ptr(), H * D, sm, d_out.data_ptr(), 0, s.cuda_stream)
    finally:
        lib.finalize()


@pytest.mark.parametrize("scheme", ["fp8", "int4", "mxfp4"])
def test_kv_connector_toy_decode_loop(scheme):
    """a vLLM-shaped connector (one allocation per request, batched append / look-ahead / attention).
    A 2-layer toy decode loop over a batch of three requests: at every step the attention the connector computes
    straight from the compressed records (+ the fp16 tail position) equals torch attention over the pages fetched and
    decompressed from the same pool.  Tolerances as the per-sequence tests: INT4 2e-3 * sum p|v| (same arithmetic on
    both sides up to summation order), FP8 10 % of the largest output (its query is quantised to e4m3 in the kernel), MXFP4 the
    FP8 bound (K / V values are exactly the decompressed ones, the query is quantised to MXFP8 in the kernel)."""
    torch = torch_mod()
    from cxl_speckv_amd.kv_connector import SpeckvKVConnector
    lib = open_lib()
    try:
        L, H, D, G, T = 2, 8, 128, 4, 256
        conn = SpeckvKVConnector(lib, num_layers=L, num_kv_heads=H, head_dim=D, max_tokens=T, scheme=scheme)
        gen = torch.Generator(device="cuda"); gen.manual_seed(5)
        rnd = lambda *s: torch.randn(s, generator=gen, device="cuda", dtype=torch.float32).to(torch.float16)
        rids, prompt = [11, 12, 13], [37, 64, 1]
        for rid, n in zip(rids, prompt):
            conn.add_request(rid)
            conn.write_prefill(rid, rnd(L, n, H, D), rnd(L, n, H, D))
        sm = 1.0 / np.sqrt(D)
        keep = []
        for step in range(9):
            keep += conn.append(rids, rnd(len(rids), L, H, D), rnd(len(rids), L, H, D))
            for layer in range(L):
                q = rnd(len(rids), H, G, D)
                out = conn.attend(layer, rids, q, sm)
                torch.cuda.synchronize()
                for b, rid in enumerate(rids):
                    k = conn.kv_rows(rid, layer, 0).float()
                    v = conn.kv_rows(rid, layer, 1).float()
                    assert k.shape[0] == prompt[b] + step + 1 == conn.length(rid)
                    p = torch.softmax(torch.einsum("hgd,thd->hgt", q[b].float(), k) * sm, dim=-1)
                    ref = torch.einsum("hgt,thd->hgd", p, v)
                    err = (out[b] - ref).abs()
                    if scheme == "int4":
                        mag = torch.einsum("hgt,thd->hgd", p, v.abs())
                        assert bool((err <= 2e-3 * mag + 1e-5).all()), (step, layer, rid, float(err.max()))
                    else:
                        # e4m3 keeps 3 mantissa bits: every V element is off by <= 2^-4 of itself (0.0625 * sum p|v|), the
                        # e4m3 query / K perturb the weights of a short context by a few per cent of max|v| on top
                        mag = torch.einsum("hgt,thd->hgd", p, v.abs())
                        assert bool((err <= 0.07 * mag + 0.08 * v.abs().max()).all()), (step, layer, rid, float(err.max()))
        # a request leaves, another joins with the same id: the binding follows
        conn.free_request(12)
        conn.add_request(12)
        conn.write_prefill(12, rnd(L, 8, H, D), rnd(L, 8, H, D))
        out = conn.attend(0, [12], rnd(1, H, G, D), sm)
        torch.cuda.synchronize()
        assert bool(torch.isfinite(out).all())
        for rid in (11, 12, 13):
            conn.free_request(rid)
    finally:
        lib.finalize()


def test_kv_connector_block_tables_and_lookahead(oracle):
    """The same connector over a pool whose pages are consumed decompressed (INT8_DELTA_RLE, the reference's codec):
    one batched look-ahead per step, block tables of device addresses for a paged-attention consumer."""
    torch = torch_mod()
    from cxl_speckv_amd.kv_connector import SpeckvKVConnector
    lib = open_lib()
    try:
        L, H, D, T = 3, 8, 128, 128
        conn = SpeckvKVConnector(lib, num_layers=L, num_kv_heads=H, head_dim=D, max_tokens=T, scheme="int8_delta_rle")
        gen = torch.Generator(device="cuda"); gen.manual_seed(6)
        rnd = lambda *s: torch.randn(s, generator=gen, device="cuda", dtype=torch.float32).to(torch.float16)
        data = {}
        for rid, n in ((1, 40), (2, 64)):
            conn.add_request(rid)
            k, v = rnd(L, n, H, D), rnd(L, n, H, D)
            conn.write_prefill(rid, k, v)
            data[rid] = (k, v)
        # the look-ahead of position len-1 brings the rows after it; ask at an earlier position so that stored rows are hit
        for rid in (1, 2):
            conn.requests[rid].length -= 10
        conn.begin_step([1, 2], depth_k=4)
        lib.sync()
        for rid in (1, 2):
            conn.requests[rid].length += 10
        assert lib.stats().total_prefetches > 0 and lib.stats().prefetch_dropped == 0
        hits0 = lib.stats().l2_hits
        for rid in (1, 2):
            n = conn.length(rid)
            for layer in range(L):
                for kind in (0, 1):
                    addrs = conn.block_table(rid, layer, kind, n - 10, n - 6)
                    assert len(addrs) == 4 and all(addrs)
                    rows = conn.kv_rows(rid, layer, kind)
                    torch.cuda.synchronize()
                    for j, a in enumerate(addrs):
                        got = dev_to_host(a, H * D * 2)
                        assert got.tobytes() == rows[n - 10 + j].cpu().numpy().tobytes()
                    # and those rows are what the reference codec makes of the data (oracle, bit for bit)
                    src = data[rid][kind][layer, n - 10 - (n - 10) % 2: n - 10 - (n - 10) % 2 + 2].cpu().numpy().reshape(1, N)
                    sc, ln, rc = oracle.compress_blocks_f16(src, 2, 0)
                    want = oracle.decompress_blocks_f16(rc, ln, sc, 2, 0).reshape(2, H, D)
                    assert_same_float_bits(rows[n - 10 - (n - 10) % 2: n - 10 - (n - 10) % 2 + 2].cpu().numpy(), want)
        assert lib.stats().l2_hits > hits0                            # the block tables were served from prefetched pages
    finally:
        lib.finalize()


def test_int4_attention_with_group_scales_near_the_fp16_limit(oracle):
    """INT4_G32 groups whose largest |x| is beyond 57 000 get a scale above 8188: a dequantisation of the form
    fp16(u) * s + (-8 s) has no finite -8 s there (the first fast path of this kernel needed a checked second path for
    such groups).  The conversion now subtracts first for every group -- (1032 + q) - 1032, then one rounding in the
    product with the scale -- so these pages are ordinary; this test keeps them covered: the oracle's attention over the
    oracle's decompressed pages, tolerance as in test_int4_fused_attention, through the linear form and ((30, T)) through
    the page table, and an allocation that once held such groups answers exactly like one that never did."""
    torch = torch_mod()
    from oracle.bindings import _ptr, u16p, f32p
    lib = open_lib()
    try:
        lib.set_compression_scheme(3)
        T, L, H, D, bpe, G = 256, 1, 8, 128, 2, 8
        n_pages = T * L * H * D * bpe * 2 // PAGE
        rng = np.random.default_rng(97)
        sm = 1.0 / np.sqrt(D)
        q = (rng.standard_normal((H, G, D)) * 0.5).astype(np.float16)
        d_q = torch.from_numpy(q.view(np.int16)).cuda()

        def run_case(x, ranges):
            h = lib.alloc(n_pages * PAGE)
            lib.set_layout(h, T, L, H, D, bpe)
            lib.write(h, 0, x.ctypes.data, x.nbytes, False)
            scales, lens, recs = oracle.compress_blocks_f16(x, 3, 0)
            dec = oracle.decompress_blocks_f16(recs, lens, scales, 3, 0).reshape(n_pages, 2, H, D)
            for pb, pe in ranges:
                npos = pe - pb
                d_out = torch.full((H, G, D), float("nan"), dtype=torch.float32, device="cuda")
                lib.attend_int4(h, 0, 1, d_q.data_ptr(), G, pb, pe, sm, d_out.data_ptr())
                torch.cuda.synchronize()
                got = d_out.cpu().numpy()
                assert np.isfinite(got).all(), (pb, pe)
                kf, vf = pb // 2, T // 2 + pb // 2
                for head in range(H):
                    k16 = np.ascontiguousarray(dec[kf:kf + npos // 2, :, head, :].reshape(npos, D)).view(np.uint16)
                    v16 = np.ascontiguousarray(dec[vf:vf + npos // 2, :, head, :].reshape(npos, D)).view(np.uint16)
                    o = np.zeros((G, D), np.float32); l = np.zeros(G, np.float32); m = np.zeros((G, D), np.float32)
                    oracle.lib.orc_attend_f16(_ptr(np.ascontiguousarray(q[head]).view(np.uint16).reshape(-1), u16p), G,
                                              _ptr(k16.reshape(-1), u16p), _ptr(v16.reshape(-1), u16p), npos, D, float(sm),
                                              _ptr(o, f32p), _ptr(l, f32p), _ptr(m, f32p))
                    err = np.abs(got[head] - o)
                    assert np.all(err <= 2e-3 * m + 1e-6), (pb, pe, head, float((err / (m + 1e-9)).max()))
            return h

        plain = rng.standard_normal((n_pages, N)).astype(np.float16)
        big = plain.copy()
        # K pages (0 .. T/2) and V pages (T/2 .. T): a few groups of 32 elements with one value near the fp16 limit
        for page, elem, val in ((3, 40, 60000.0), (3, 1500, -64000.0), (70, 7, 58000.0), (T // 2 + 9, 300, -61000.0),
                                (T // 2 + 64, 2047, 65000.0), (T // 2 + 127, 0, 57400.0)):
            big[page, elem] = np.float16(val)
        assert np.abs(big.astype(np.float32)).max() / 7.0 > 8188.0
        ranges = [(0, T), (0, 64), (64, 192), (30, T)]            # linear form, and (30, T) through the page table
        h_big = run_case(big, ranges)
        h_plain = run_case(plain, ranges)
        # overwrite the large groups with ordinary data: nothing of them may linger
        lib.write(h_big, 0, plain.ctypes.data, plain.nbytes, False)
        d_out = torch.empty((H, G, D), dtype=torch.float32, device="cuda")
        d_ref = torch.empty((H, G, D), dtype=torch.float32, device="cuda")
        lib.attend_int4(h_big, 0, 1, d_q.data_ptr(), G, 0, T, sm, d_out.data_ptr())
        lib.attend_int4(h_plain, 0, 1, d_q.data_ptr(), G, 0, T, sm, d_ref.data_ptr())
        torch.cuda.synchronize()
        # same records, same split geometry: the same bits
        assert torch.equal(d_out, d_ref)
        lib.free(h_big); lib.free(h_plain)
    finally:
        lib.finalize()


@pytest.mark.parametrize("scheme", [2, 3, 4, 5])
def test_write_strided_batch_equals_per_allocation_writes(scheme):
    """speckv_ext_write_strided_batch (one launch for a batch of allocations: the append of a decode step) stores exactly
    what one speckv_ext_write_strided per allocation stores: record lengths, scales and decoded pages are identical; a page
    that was cached is invalidated; bad batches are refused."""
    torch = torch_mod()
    lib = open_lib()
    try:
        lib.set_compression_scheme(scheme)
        T, L, H, D, bpe = 64, 3, 8, 128, 2
        n_pages = 2 * T * L * H * D * bpe // PAGE
        region = T // 2
        n_alloc, n_each = 5, 2 * L
        st = torch.cuda.Stream()
        rng = np.random.default_rng(123 + scheme)
        batch, single = [], []
        for _ in range(n_alloc):
            for lst in (batch, single):
                h = lib.alloc(n_pages * PAGE); lib.set_layout(h, T, L, H, D, bpe); lst.append(h)
        firsts = [int(v) for v in rng.integers(0, region, n_alloc)]
        src = torch.from_numpy((rng.standard_normal((n_alloc, n_each, N)) * 3).astype(np.float16).view(np.int16)).cuda()
        # one of the target pages is resident before the write: the batch write has to invalidate it
        lib.access(batch[1], firsts[1] * PAGE, 8)
        assert lib.translate(batch[1], firsts[1] * PAGE).flags & 3
        torch.cuda.synchronize()
        lib.write_strided_batch(batch, firsts, [src[i].data_ptr() for i in range(n_alloc)], region, n_each, st.cuda_stream)
        for i in range(n_alloc):
            lib.write_strided(single[i], firsts[i], region, n_each, src[i].data_ptr(), st.cuda_stream)
        st.synchronize()
        assert not (lib.translate(batch[1], firsts[1] * PAGE).flags & 3)
        got = torch.empty((n_pages, N), dtype=torch.float16, device="cuda")
        want = torch.empty((n_pages, N), dtype=torch.float16, device="cuda")
        for i in range(n_alloc):
            lib.fetch_range(batch[i], 0, n_pages, got.data_ptr(), False, st.cuda_stream)
            lib.fetch_range(single[i], 0, n_pages, want.data_ptr(), False, st.cuda_stream)
            st.synchronize()
            assert torch.equal(got.view(torch.int16), want.view(torch.int16)), i
            for j in range(n_each):
                a, b = lib.translate(batch[i], (firsts[i] + j * region) * PAGE), lib.translate(single[i], (firsts[i] + j * region) * PAGE)
                assert (a.rec_bytes, np.float32(a.scale).tobytes()) == (b.rec_bytes, np.float32(b.scale).tobytes())
                assert a.rec_bytes > 0
        ptrs = [src[i].data_ptr() for i in range(n_alloc)]
        with pytest.raises(SpeckvError):
            lib.write_strided_batch(batch, firsts, ptrs, region, n_each, 0)                      # NULL stream
        with pytest.raises(SpeckvError):
            lib.write_strided_batch(batch[:2] + batch[:1], firsts[:3], ptrs[:3], region, n_each, st.cuda_stream)   # same allocation twice
        with pytest.raises(SpeckvError):
            lib.write_strided_batch(batch[:2], [region * 2 * L - 1, 0], ptrs[:2], region, n_each, st.cuda_stream)  # runs off the end
        lib.set_compression_scheme(1)
        other = lib.alloc(n_pages * PAGE)
        with pytest.raises(SpeckvError):
            lib.write_strided_batch(batch[:1] + [other], firsts[:2], ptrs[:2], region, n_each, st.cuda_stream)     # mixed schemes
        for h in batch + single + [other]:
            lib.free(h)
    finally:
        lib.finalize()


@pytest.mark.parametrize("scheme", ["fp8", "int4", "mxfp4"])
def test_batch_decode_graph_example_runs(scheme):
    """examples/batch_decode_graph_example.py: a batch decode loop whose per-step attention (all layers, tail fold on odd
    steps) is one HIP graph replay, planned outside the graph; every step equals the connector's eager path."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("batch_decode_graph_example",
                                                  os.path.join(os.path.dirname(__file__), "..", "examples", "batch_decode_graph_example.py"))
    mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
    res = mod.run(seqs=6, layers=2, prompt=70, steps=7, scheme=scheme, max_tokens=256, verbose=False)
    assert res["final_length"] == 77 and res["max_rel_diff_graph_vs_eager"] <= 1e-3


def test_flush_of_a_256_sequence_decode_step_at_full_size(oracle):
    """BASELINE configs[3] call count at full size: 256 sequences x 80 layers = 20 480 look-ahead requests (depth 4),
    one allocation per sequence, ONE device-side flush -> 122 880 pages.  At this size the pages' host-visible words are
    stored by the fetch launch itself.  Checked: the page count, that exactly the expected pages of sampled sequences
    are L2-resident by the host's view (translate) with slots in first-occurrence order, decoded contents of sampled
    pages bit for bit against the oracle, and that a second flush of the same requests issues nothing."""
    torch = torch_mod()
    lib = open_lib(SPECKV_L2_MB=2048)
    try:
        lib.set_compression_scheme(2)
        n_seq, Lyr, T, H, D = 256, 80, 128, 8, 128
        n_pages = 2 * T * Lyr * H * D * 2 // PAGE
        hs = []
        for s_ in range(n_seq):
            h = lib.alloc(n_pages * PAGE)
            lib.set_layout(h, T, Lyr, H, D, 2)
            lib.bind_request(s_, h, 0)
            hs.append(h)
        pos0 = 16
        # real data where the flush will look (positions pos0+1 .. pos0+4 of every layer and kind) for three sequences
        region = T // 2                                               # pages of one (layer, kind) region
        want_pages = sorted({(layer * 2 + kind) * region + p // 2 for layer in range(Lyr) for kind in (0, 1) for p in range(pos0 + 1, pos0 + 5)})
        assert len(want_pages) == 6 * Lyr                             # positions 17..20 touch pages 8, 9, 10 of each region
        sampled = (0, 101, 255)
        data = {}
        for sq in sampled:
            x = synth(len(want_pages), seed=300 + sq)
            for i, pg in enumerate(want_pages):
                lib.write(hs[sq], pg * PAGE, x[i].ctypes.data, PAGE, False)
            data[sq] = x
        n_req = n_seq * Lyr
        reqs = np.repeat(np.arange(n_seq, dtype=np.uint32), Lyr)
        layers = np.tile(np.arange(Lyr, dtype=np.uint16), n_seq)
        lib.prefetch_batch(reqs, layers, np.full(n_req, pos0, np.uint32), np.full(n_req, 4, np.uint32))
        issued = lib.prefetch_flush()
        assert issued == n_seq * len(want_pages) == 122880
        lib.sync()
        for sq in sampled:
            infos = {pg: lib.translate(hs[sq], pg * PAGE) for pg in want_pages}
            assert all(i.flags & 2 for i in infos.values()), sq
            # first-occurrence order inside a sequence: layer by layer, K pages then V pages, ascending positions
            order = []
            for layer in range(Lyr):
                for kind in (0, 1):
                    for p in range(pos0 + 1, pos0 + 5):
                        pg = (layer * 2 + kind) * region + p // 2
                        if pg not in order:
                            order.append(pg)
            addrs = [infos[pg].cache_addr for pg in order]
            assert all(b - a == PAGE for a, b in zip(addrs, addrs[1:])), sq          # one run of ring slots per sequence
            others = [pg for pg in range(0, n_pages, 97) if pg not in infos]
            assert not any(lib.translate(hs[sq], pg * PAGE).flags & 3 for pg in others)
            scales, lens, recs = oracle.compress_blocks_f16(data[sq], 2, 0)
            for i in (0, 1, len(want_pages) // 2, len(want_pages) - 1):
                dec = oracle.decompress_block_f16(recs[i, :lens[i]], scales[i], 2, 0, N)
                assert_same_float_bits(dev_to_host(infos[want_pages[i]].cache_addr, PAGE).view(np.float16), dec)
        st = lib.stats()
        assert st.prefetch_dropped == 0
        lib.prefetch_batch(reqs, layers, np.full(n_req, pos0, np.uint32), np.full(n_req, 4, np.uint32))
        assert lib.prefetch_flush() == 0
    finally:
        lib.finalize()


@pytest.mark.parametrize("scheme", [5, 4, 3])
def test_planned_layer_with_the_tail_position_in_one_call(scheme):
    """speckv_ext_attend_planned_tail = the planned layer followed by speckv_ext_attend_fold_tail, in one call: MXFP4 folds the position
    in the attention kernel's own epilogue (single-split sequences: into the final rows; with real splits: into split 0's partial, in
    front of the merge), the other formats by one fold launch inside the call.  Tails on some sequences only (index by sequence),
    on all of them (no index needed), on none; a batch with an empty member takes the launch form; as a HIP graph."""
    torch = torch_mod()
    lib = open_lib()
    try:
        lib.set_compression_scheme(scheme)
        T, L, H, D, G = 1024, 2, 8, 128, 8
        rng = np.random.default_rng(300 + scheme)
        lens_list = ([64, 1024, 34, 600], [64, 1024, 0, 600])
        n_seq = 4
        handles = []
        for _ in range(n_seq):
            h = lib.alloc(T * L * H * D * 2 * 2)
            lib.set_layout(h, T, L, H, D, 2)
            n_pages = T * L * H * D * 2 * 2 // PAGE
            x = (rng.standard_normal((n_pages, N)) * rng.uniform(0.2, 2.0, (n_pages, 1))).astype(np.float16)
            lib.write(h, 0, x.ctypes.data, x.nbytes, False)
            handles.append(h)
        sm = 1.0 / np.sqrt(D)
        q = torch.from_numpy(rng.standard_normal((L, n_seq, H, G, D)).astype(np.float16)).cuda()
        kt = torch.from_numpy((rng.standard_normal((n_seq, L, H, D)) * 1.5).astype(np.float16)).cuda()
        vt = torch.from_numpy(rng.standard_normal((n_seq, L, H, D)).astype(np.float16)).cuda()
        plan_bytes = lib.attend_plan_bytes(n_seq)
        plan = torch.zeros(plan_bytes, dtype=torch.uint8, device="cuda")
        s = torch.cuda.Stream()
        stride = L * H * D

        def both(layer, lens, rows, tps):
            """(separate calls, one call) for tails on the sequences `rows`"""
            set_tuning("attend_tiles_per_split", tps)
            try:
                lib.attend_batch_plan(handles, lens, T, plan.data_ptr(), plan_bytes, s.cuda_stream)
                res = []
                n_tail = len(rows)
                d_rows = torch.tensor(rows, dtype=torch.int32).cuda() if 0 < n_tail < n_seq else None
                inv = [-1] * n_seq
                for i, b in enumerate(rows): inv[b] = i
                d_idx = torch.tensor(inv, dtype=torch.int32).cuda() if 0 < n_tail < n_seq else None
                kk = kt[rows].contiguous() if n_tail else kt; vv = vt[rows].contiguous() if n_tail else vt
                for one_call in (False, True):
                    out = torch.full((n_seq, H, G, D), float("nan"), dtype=torch.float32, device="cuda")
                    lse = torch.full((n_seq, H, G), float("nan"), dtype=torch.float32, device="cuda")
                    if one_call:
                        lib.attend_planned_tail(scheme, plan.data_ptr(), n_seq, layer, q[layer].data_ptr(), G, T, sm, out.data_ptr(), lse.data_ptr(), n_tail,
                                                d_rows.data_ptr() if d_rows is not None else 0, d_idx.data_ptr() if d_idx is not None else 0,
                                                kk.data_ptr(), vv.data_ptr(), stride, s.cuda_stream)
                    else:
                        lib.attend_planned(scheme, plan.data_ptr(), n_seq, layer, q[layer].data_ptr(), G, T, sm, out.data_ptr(), lse.data_ptr(), s.cuda_stream)
                        if n_tail:
                            lib.attend_fold_tail(n_tail, d_rows.data_ptr() if d_rows is not None else 0, H, G, q[layer].data_ptr(),
                                                 kk.data_ptr() + layer * H * D * 2, vv.data_ptr() + layer * H * D * 2, stride, sm, out.data_ptr(), lse.data_ptr(),
                                                 s.cuda_stream)
                    s.synchronize()
                    res.append((out.cpu().numpy(), lse.cpu().numpy()))
                return res
            finally:
                set_tuning("attend_tiles_per_split", 0)

        for lens in lens_list:
            for rows in ([0, 1, 2, 3], [1, 3], [2], []):
                for tps in (0, 4):                                  # the rule's geometry (one split each here), and real splits with a merge behind
                    for layer in range(L):
                        (o0, l0), (o1, l1) = both(layer, lens, rows, tps)
                        assert np.isfinite(o1).all() and not np.isnan(l1).any()
                        scale = np.abs(o0).max(axis=-1, keepdims=True) + 1e-6
                        assert np.all(np.abs(o1 - o0) <= 2e-3 * scale + 1e-6), (scheme, lens, rows, tps, layer, float(np.abs(o1 - o0).max()))
                        fin = np.isfinite(l0)
                        assert np.array_equal(fin, np.isfinite(l1)) and np.all(np.abs(l1[fin] - l0[fin]) <= 2e-3), (scheme, lens, rows, tps, layer)
        # the launch form on request gives the same again (what the one call does for FP8 / INT4 anyway)
        set_tuning("attend_fold_launch", 1)
        try:
            (o0, l0), (o1, l1) = both(1, lens_list[0], [1, 3], 0)
            assert np.allclose(o0, o1, rtol=0, atol=1e-6) and np.allclose(l0, l1, rtol=0, atol=1e-6)
        finally:
            set_tuning("attend_fold_launch", 0)
        # captured: the one call replays with the tail rows read at replay time
        lib.attend_batch_plan(handles, lens_list[0], T, plan.data_ptr(), plan_bytes, s.cuda_stream)
        out = torch.zeros((n_seq, H, G, D), dtype=torch.float32, device="cuda"); lse = torch.zeros((n_seq, H, G), dtype=torch.float32, device="cuda")
        call = lambda: lib.attend_planned_tail(scheme, plan.data_ptr(), n_seq, 0, q[0].data_ptr(), G, T, sm, out.data_ptr(), lse.data_ptr(), n_seq, 0, 0,
                                               kt.data_ptr(), vt.data_ptr(), stride, s.cuda_stream)
        call(); s.synchronize()
        want = out.clone()
        g = torch.cuda.CUDAGraph()
        with graph_capture(g, s):
            call()
        out.zero_()
        g.replay(); torch.cuda.synchronize()
        assert torch.equal(out, want)
    finally:
        lib.finalize()


@pytest.mark.parametrize("scheme", ["mxfp4", "fp8", "int4"])
def test_connector_layers_in_one_call_equal_the_per_layer_calls(scheme):
    """SpeckvKVConnector.attend_layers (speckv_ext_attend_planned_layers): all layers of a step in one library call -- over an MXFP4 pool
    with one split per sequence ONE launch over layers x sequences, tails folded in by the kernel; per-layer launches from the one call
    otherwise -- equals attend() layer by layer: even and odd lengths (some sequences with a tail, all, none), bit for bit where the
    same kernels run (the loop form), within the tolerance of another summation order nowhere (the one-launch form runs the same
    workgroups on the same data)."""
    torch = torch_mod()
    from cxl_speckv_amd.kv_connector import SpeckvKVConnector
    lib = open_lib()
    try:
        L, H, D, G, T = 3, 8, 128, 8, 512
        conn = SpeckvKVConnector(lib, num_layers=L, num_kv_heads=H, head_dim=D, max_tokens=T, scheme=scheme)
        gen = torch.Generator(device="cuda"); gen.manual_seed(15)
        rnd = lambda *s: torch.randn(s, generator=gen, device="cuda", dtype=torch.float32).to(torch.float16)
        rids, prompt = [3, 4, 5, 6], [37, 64, 130, 2]
        for rid, n in zip(rids, prompt):
            conn.add_request(rid)
            conn.write_prefill(rid, rnd(L, n, H, D), rnd(L, n, H, D))
        sm = 1.0 / np.sqrt(D)
        keep = []
        for step in range(4):
            q = rnd(L, len(rids), H, G, D)
            for loop in (0, 1):
                set_tuning("attend_layers_loop", loop)
                try:
                    got = conn.attend_layers(0, L, rids, q, sm)
                finally:
                    set_tuning("attend_layers_loop", 0)
                torch.cuda.synchronize()
                for layer in range(L):
                    one = conn.attend(layer, rids, q[layer], sm)
                    torch.cuda.synchronize()
                    assert torch.equal(got[layer], one), (scheme, step, loop, layer, float((got[layer] - one).abs().max()))
            part = conn.attend_layers(1, 2, rids, q[1:], sm)                   # a sub-range of the layers
            torch.cuda.synchronize()
            assert torch.equal(part[1], conn.attend(2, rids, q[2], sm))
            keep += conn.append(rids, rnd(len(rids), L, H, D), rnd(len(rids), L, H, D))
        for rid in rids:
            conn.free_request(rid)
    finally:
        lib.finalize()
