"""-m "not gpu": host-side units of the product library, no GPU involved.

  * speckv_ext_mm_*  (SURVEY 8a rows A10-A14 in the PRODUCT): replay of the reference's memory-manager trace
    (tests/golden/mm_trace.json, recorded from the reference's CXLMemoryManager) and random walks against the oracle;
  * SlabPool bookkeeping (tests/csrc/slab_pool_test.cpp: the product's slab_pool.cpp over host-heap slabs):
    sub-range frees at record granularity never reach into live neighbours (the round-1 migrate bug);
  * speckv_ext_placement / speckv_ext_pool_shard_pages: the striping rule;
  * the drop-in .so exports the C ABI and nothing else.
"""
import ctypes as C
import json
import os
import subprocess

import numpy as np
import pytest

import cxl_speckv_amd as pkg

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class MMStats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("l1_hits", "l1_misses", "l2_hits", "l2_misses", "l3_accesses",
                                          "migrations_l1_to_l3", "migrations_l3_to_l1")] + \
               [("l1_hit_rate", C.c_double), ("l2_hit_rate", C.c_double)]


MM_SIG = {
    "new": (C.c_void_p, [C.c_uint64] * 4), "delete": (None, [C.c_void_p]),
    "allocate": (C.c_uint64, [C.c_void_p, C.c_uint64, C.c_uint32, C.c_int]),
    "deallocate": (None, [C.c_void_p, C.c_uint64]), "translate": (C.c_uint64, [C.c_void_p, C.c_uint64]),
    "is_in_cache": (C.c_int, [C.c_void_p, C.c_uint64, C.c_int]),
    "promote_to_l1": (C.c_int, [C.c_void_p, C.c_uint64]), "demote_to_l3": (C.c_int, [C.c_void_p, C.c_uint64]),
    "invalidate_page": (None, [C.c_void_p, C.c_uint64]), "mark_modified": (None, [C.c_void_p, C.c_uint64]),
    "get_page_state": (C.c_int, [C.c_void_p, C.c_uint64]),
    "update_access_tracking": (None, [C.c_void_p, C.c_uint64]), "is_hot_page": (C.c_int, [C.c_void_p, C.c_uint64]),
    "get_statistics": (None, [C.c_void_p, C.POINTER(MMStats)]),
    "cxl_access": (C.c_uint64, [C.c_void_p, C.c_uint64, C.c_uint64]),
}


@pytest.fixture(scope="module")
def product():
    lib = pkg.load_library()
    for name, (res, args) in MM_SIG.items():
        fn = getattr(lib, "speckv_ext_mm_" + name)
        fn.restype, fn.argtypes = res, args
    lib.speckv_ext_placement.argtypes = [C.c_uint64, C.c_uint32, C.c_uint64, C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)]
    lib.speckv_ext_placement.restype = None
    lib.speckv_ext_pool_shard_pages.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32]
    lib.speckv_ext_pool_shard_pages.restype = C.c_uint64
    return lib


def _stats(lib, mm, prefix):
    s = MMStats()
    getattr(lib, prefix + "get_statistics")(mm, C.byref(s))
    return [s.l1_hits, s.l1_misses, s.l2_hits, s.l2_misses, s.l3_accesses, s.migrations_l1_to_l3, s.migrations_l3_to_l1], \
           [s.l1_hit_rate, s.l2_hit_rate]


def test_product_memory_manager_replays_the_reference_trace(product, golden_dir):
    g = json.load(open(os.path.join(golden_dir, "mm_trace.json")))
    mm = product.speckv_ext_mm_new(12, 3, 128, 4096)
    for op, a, want in g["events"]:
        got = getattr(product, "speckv_ext_mm_" + op)(mm, *a)
        if want is not None:
            assert got == want, (op, a, got, want)
    u, d = _stats(product, mm, "speckv_ext_mm_")
    assert u == g["stats_u"] and d == g["stats_d"]
    product.speckv_ext_mm_delete(mm)
    # SURVEY appendix A, F-mm
    mm = product.speckv_ext_mm_new(12, 3, 128, 4096)
    a0 = product.speckv_ext_mm_allocate(mm, 524288, 0, 2)
    assert a0 == 0x100000000 and product.speckv_ext_mm_translate(mm, a0) == 0x20000000000
    assert product.speckv_ext_mm_allocate(mm, 4096, 3, 0) == 0x100080000
    assert product.speckv_ext_mm_translate(mm, 0x100080000) == 0x8000000000
    a2 = product.speckv_ext_mm_allocate(mm, 5000, 4, 1)
    assert a2 == 0x100081000 and product.speckv_ext_mm_translate(mm, a2) == 0x10000000000
    assert product.speckv_ext_mm_translate(mm, a2 + 8191) == 0x10000000000 + 8191      # 5000 B -> 2 pages
    assert product.speckv_ext_mm_translate(mm, a2 + 8192) == 0
    assert product.speckv_ext_mm_translate(mm, a0 + 4096 * 3 + 17) == 0x20000003011
    product.speckv_ext_mm_delete(mm)


@pytest.mark.parametrize("l1_gb", [12, 0])
def test_product_memory_manager_random_walk_equals_oracle(product, oracle, l1_gb):
    """Same call sequence into the product and into the oracle (itself pinned to the reference by
    tests/test_oracle_vs_ref.py): every return value and the statistics agree.  l1_gb = 0 makes every promotion
    evict the LRU entry first (cxl_memory_manager.cpp:139-142)."""
    O = oracle.lib
    rng = np.random.default_rng(100 + l1_gb)
    pm = product.speckv_ext_mm_new(l1_gb, 3, 128, 4096)
    om = O.orc_mm_new(l1_gb, 3, 128, 4096)
    bases = []
    for step in range(4000):
        r = rng.random()
        if r < 0.05 or not bases:
            size = int(rng.integers(1, 40000)); layer = int(rng.integers(0, 80)); tier = int(rng.integers(0, 3))
            a, b = product.speckv_ext_mm_allocate(pm, size, layer, tier), O.orc_mm_allocate(om, size, layer, tier)
            assert a == b
            bases.append((a, size))
            continue
        base, size = bases[int(rng.integers(0, len(bases)))]
        va = base + int(rng.integers(0, size + 6000))          # sometimes past the end / into the next allocation
        if rng.random() < 0.05:
            va = int(rng.integers(0, 1 << 34))
        op = rng.choice(["translate", "is_in_cache", "promote_to_l1", "demote_to_l3", "update_access_tracking",
                         "is_hot_page", "get_page_state", "mark_modified", "invalidate_page", "cxl_access", "deallocate"],
                        p=[.15, .1, .15, .1, .2, .08, .05, .03, .03, .1, .01])
        if op == "is_in_cache":
            t = int(rng.integers(0, 3))
            assert product.speckv_ext_mm_is_in_cache(pm, va, t) == O.orc_mm_is_in_cache(om, va, t), (step, hex(va))
        elif op == "cxl_access":
            off = va - base
            if off < 0:
                continue
            assert product.speckv_ext_mm_cxl_access(pm, base, off) == O.orc_mm_cxl_access(om, base, off)
        elif op == "deallocate":
            product.speckv_ext_mm_deallocate(pm, base); O.orc_mm_deallocate(om, base)
        else:
            a = getattr(product, "speckv_ext_mm_" + op)(pm, va)
            b = getattr(O, "orc_mm_" + op)(om, va)
            if op not in ("update_access_tracking", "mark_modified", "invalidate_page"):
                assert a == b, (step, op, hex(va), a, b)
    from tests.test_oracle_golden import MMStats as OStats
    s = OStats(); O.orc_mm_get_statistics(om, C.byref(s))
    u, d = _stats(product, pm, "speckv_ext_mm_")
    assert u == [s.l1_hits, s.l1_misses, s.l2_hits, s.l2_misses, s.l3_accesses, s.migrations_l1_to_l3, s.migrations_l3_to_l1]
    assert d == [s.l1_hit_rate, s.l2_hit_rate]
    product.speckv_ext_mm_delete(pm); O.orc_mm_delete(om)


# ----------------------------------------------------------------------------- slab pool
@pytest.fixture(scope="module")
def slab():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "csrc")])
    lib = C.CDLL(os.path.join(ROOT, "tests", "_build", "libslabpool_test.so"))
    lib.slabtest_new.restype = C.c_void_p; lib.slabtest_new.argtypes = [C.c_size_t, C.c_size_t]
    lib.slabtest_delete.argtypes = [C.c_void_p]
    lib.slabtest_alloc.restype = C.c_void_p; lib.slabtest_alloc.argtypes = [C.c_void_p, C.c_size_t]
    lib.slabtest_alloc_up_to.restype = C.c_void_p; lib.slabtest_alloc_up_to.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.POINTER(C.c_size_t)]
    lib.slabtest_free.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    for n in ("used", "reserved", "free_runs"):
        getattr(lib, "slabtest_" + n).restype = C.c_size_t; getattr(lib, "slabtest_" + n).argtypes = [C.c_void_p]
    return lib


@pytest.mark.parametrize("stride", [2048, 1152, 4096, 17408])            # (17 408: one tile of an MXFP4 run, 16 records)
def test_slab_subrange_free_never_reaches_live_records(slab, stride):
    """freeing record 1 of a run of 64 and allocating again must not hand out bytes of record 2."""
    p = slab.slabtest_new(2 << 20, 0)
    base = slab.slabtest_alloc(p, 64 * stride)
    assert base and slab.slabtest_used(p) == 64 * stride
    slab.slabtest_free(p, base + stride, stride)
    assert slab.slabtest_used(p) == 63 * stride
    again = slab.slabtest_alloc(p, min(stride, 4096))
    assert again == base + stride                                 # first fit: exactly the hole, nothing more
    if stride != 4096:
        nxt = slab.slabtest_alloc(p, max(4096, stride + 128))
        assert nxt >= base + 64 * stride                          # a bigger request cannot fit in the hole
    slab.slabtest_delete(p)


def test_slab_random_alloc_free_keeps_runs_disjoint(slab):
    rng = np.random.default_rng(5)
    p = slab.slabtest_new(256 << 10, 0)
    live = []            # (addr, bytes)
    for step in range(3000):
        if live and rng.random() < 0.45:
            i = int(rng.integers(0, len(live)))
            addr, n = live.pop(i)
            stride = 128 * int(rng.integers(1, 9))
            if n > 2 * stride and rng.random() < 0.5:             # free a middle piece first, then the two rests
                k = (n // stride) // 2 * stride
                slab.slabtest_free(p, addr + k, stride)
                slab.slabtest_free(p, addr, k)
                if n - k - stride:
                    slab.slabtest_free(p, addr + k + stride, n - k - stride)
            else:
                slab.slabtest_free(p, addr, n)
        else:
            n = 128 * int(rng.integers(1, 600))
            kind = rng.random()
            got = C.c_size_t()
            if kind < 0.2:
                got = C.c_size_t()
                want = max(n, 1152)
                a = slab.slabtest_alloc_up_to(p, want, 1152, C.byref(got))
                n = got.value
                assert a and n and n % 1152 == 0 and n <= want
            elif kind < 0.4:                                       # runs of MXFP4 tiles (16 records = 17 408 bytes = 136 lines)
                n = 17408 * int(rng.integers(1, 6))
                if rng.random() < 0.5:
                    got = C.c_size_t()
                    a = slab.slabtest_alloc_up_to(p, n, 17408, C.byref(got))
                    n = got.value
                    assert a and n and n % 17408 == 0
                else:
                    a = slab.slabtest_alloc(p, n)
                    assert a
            else:
                a = slab.slabtest_alloc(p, n)
                assert a
            assert a % 128 == 0 and n % 128 == 0                  # whole cache lines, starting on one
            for b, m in live:
                assert a + n <= b or b + m <= a, "overlapping live runs"
            live.append((a, n))
        assert slab.slabtest_used(p) == sum(m for _, m in live)
        assert slab.slabtest_reserved(p) >= slab.slabtest_used(p)
    for a, n in live:
        slab.slabtest_free(p, a, n)
    assert slab.slabtest_used(p) == 0
    assert slab.slabtest_free_runs(p) <= slab.slabtest_reserved(p) // (256 << 10) + 3   # everything coalesced again, slab by slab
    slab.slabtest_delete(p)


def test_slab_capacity_limit(slab):
    p = slab.slabtest_new(64 << 10, 128 << 10)
    a = slab.slabtest_alloc(p, 100 << 10)
    assert a and not slab.slabtest_alloc(p, 64 << 10)             # would pass the cap
    slab.slabtest_free(p, a, 100 << 10)
    assert slab.slabtest_alloc(p, 64 << 10)
    slab.slabtest_delete(p)


# ----------------------------------------------------------------------------- placement
def test_placement_rule(product):
    for n_pages, d in ((131072, 1), (131072, 7), (655360, 7), (10, 3), (5, 8), (0, 4), (1000, 0)):
        dd = d or 1
        shard = [product.speckv_ext_pool_shard_pages(n_pages, d, k) for k in range(dd)]
        assert sum(shard) == n_pages and max(shard) - min(shard) <= 1
        assert product.speckv_ext_pool_shard_pages(n_pages, d, dd) == 0
        seen = [[] for _ in range(dd)]
        for p in list(range(min(n_pages, 50))) + ([n_pages - 1] if n_pages else []):
            pool, rec = C.c_uint32(), C.c_uint64()
            product.speckv_ext_placement(n_pages, d, p, C.byref(pool), C.byref(rec))
            assert pool.value == p % dd and rec.value == p // dd and rec.value < shard[pool.value]
            seen[pool.value].append(rec.value)
        for s in seen:
            assert s == sorted(s)                                 # consecutive pages of a pool are consecutive records


# ----------------------------------------------------------------------------- symbol hygiene
def test_library_exports_only_the_c_abi():
    out = subprocess.check_output(["nm", "-D", "--defined-only", pkg.library_path()], text=True)
    names = [l.split()[-1] for l in out.splitlines() if l.strip()]
    assert len(names) >= 60
    stray = [n for n in names if not (n.startswith("speckv_") or n.startswith("coherence_manager_"))]
    assert stray == [], stray
    assert not [n for n in names if n.startswith("speckv_debug")], "self-check kernels belong to tests/_build/libspeckv_debug.so"


# ----------------------------------------------------------------------------- ring / split rules (ring_rule.hpp)
@pytest.fixture(scope="module")
def rules():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "csrc")])
    lib = C.CDLL(os.path.join(ROOT, "tests", "_build", "libhostrules_test.so"))
    lib.rules_ring_take.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32)]
    lib.rules_ring_live.restype = C.c_int; lib.rules_ring_live.argtypes = [C.c_uint32] * 3
    lib.rules_even_split.argtypes = [C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32)]
    lib.rules_int4_unequal.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32)]
    lib.rules_fp8_batch_tiles_per_split.restype = C.c_uint32
    lib.rules_fp8_batch_tiles_per_split.argtypes = [C.POINTER(C.c_uint32), C.c_uint32, C.c_uint32, C.c_uint32]
    lib.rules_balanced_tiles_per_piece.restype = C.c_uint32
    lib.rules_balanced_tiles_per_piece.argtypes = [C.POINTER(C.c_uint32), C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]
    return lib


@pytest.mark.parametrize("n,start", [(64, 0), (256, 0), (1000, 0), (256, 0xFFFFF000), (97, 12345)])
def test_ring_sequence_numbers_against_a_model_of_the_ring(rules, n, start):
    """The rule host and device share (ring_rule.hpp): runs never wrap, slot = seq % n, and `live` (derived from a stored
    sequence number and the hand alone) says exactly whether the slot still holds that page -- checked against a plain
    model of the ring that records which take last wrote every slot.  Includes a start near the 32-bit wrap: differences
    are taken modulo 2^32."""
    rng = np.random.default_rng(n + (start & 0xFFFF))
    hand = start - start % n                                   # a hand position that is a lap boundary, like a fresh ring
    owner = [None] * n                                         # slot -> sequence number stored there
    stored = []                                                # every (seq, slot) ever written
    out = (C.c_uint32 * 3)()
    for _ in range(600):
        m = int(rng.integers(1, n // 2 + 1))                   # the engine never takes more than half the ring
        rules.rules_ring_take(hand & 0xFFFFFFFF, m, n, out)
        seq, slot, nxt = out[0], out[1], out[2]
        assert slot == seq % n and slot + m <= n               # the run does not wrap
        assert (nxt - seq) & 0xFFFFFFFF == m
        skipped = (seq - hand) & 0xFFFFFFFF
        assert skipped == 0 or (slot == 0 and skipped < n)     # only the rest of a lap is ever skipped
        for i in range(m):
            owner[slot + i] = (seq + i) & 0xFFFFFFFF
            stored.append(((seq + i) & 0xFFFFFFFF, slot + i))
        hand = nxt
        # every sequence number ever stored: live <=> it is still the owner of its slot
        for q, s in stored[-3 * n:]:
            intact = owner[s] == q
            got = bool(rules.rules_ring_live(hand & 0xFFFFFFFF, q, n))
            if got:
                assert intact, (hand, q, s)                    # never a stale hit
            elif intact:
                # conservative only where the hand has passed q + n without writing the slot (skipped at the end of a lap)
                assert (hand - q) & 0xFFFFFFFF > n, (hand, q, s)
            if (hand - q) & 0xFFFFFFFF <= n:
                assert got and intact, (hand, q, s)            # within one ring length of the hand: always live


def test_ring_live_is_conservative_about_skipped_slots(rules):
    """A slot skipped at the end of a lap keeps its old page, but the hand has passed it: `live` reports it dead (a
    refetch, never a stale hit)."""
    out = (C.c_uint32 * 3)()
    n = 100
    rules.rules_ring_take(0, 90, n, out)                       # slots 0..89, hand 90
    rules.rules_ring_take(out[2], 30, n, out)                  # does not fit: skips 90..99, takes 0..29 of the next lap
    assert (out[0], out[1], out[2]) == (100, 0, 130)
    assert rules.rules_ring_live(130, 95, n) == 1              # 95 was never stored, but 89 ...
    assert rules.rules_ring_live(130, 89, n) == 1 and rules.rules_ring_live(130, 30, n) == 1    # ... 30..89 of lap 0 are intact
    assert rules.rules_ring_live(130, 29, n) == 0              # overwritten by sequence number 129
    rules.rules_ring_take(130, 50, n, out)                     # 30..79
    assert rules.rules_ring_live(out[2], 79, n) == 0 and rules.rules_ring_live(out[2], 80, n) == 1
    assert rules.rules_ring_live(out[2], out[2], n) == 0       # nothing is stored at the hand yet


def test_even_split_covers_the_tiles_evenly(rules):
    out = (C.c_uint32 * 2)()
    for n_tiles in list(range(0, 70)) + [255, 256, 257, 1023, 1024, 4097]:
        for want in (0, 1, 2, 3, 7, 8, 9, 64, 300, 5000):
            rules.rules_even_split(n_tiles, want, out)
            tps, ns = out[0], out[1]
            if n_tiles == 0:
                assert ns == 0 and tps >= 1
                continue
            assert 1 <= ns <= max(1, min(want, n_tiles)) or want == 0 and ns == 1
            assert tps * ns >= n_tiles > tps * (ns - 1)        # every split non-empty, all tiles covered
            assert tps - (n_tiles - tps * (ns - 1)) < ns or ns == 1 or tps <= (n_tiles + ns - 1) // ns   # no split shorter than needed by rounding


def test_fp8_batch_split_rule(rules):
    """The split rule of the FP8 batch attention (ring_rule.hpp::fp8_batch_tiles_per_split, measured in
    profiles/*_fp8_batch_split_rule_ab.txt): the shapes whose best split length was measured on the MI355X, and the
    properties every answer must have -- a split is never shorter than 8 tiles, batches that fill the chip on their own stay
    whole, a split launch stays within one residency (1024 workgroups) and is cheaper than whole sequences by the rule's
    own busiest-CU cost, and the rule
    for a uniform batch is the same whether the tiles come as a list or as a bound."""
    f = rules.rules_fp8_batch_tiles_per_split
    hq = 2                                                        # 8 kv heads: two workgroup columns per sequence

    def uniform(n_seq, tiles):
        a = f(None, n_seq, tiles, hq)
        b = f((C.c_uint32 * n_seq)(*([tiles] * n_seq)), n_seq, 0, hq)
        assert a == b
        return a

    def cost_whole(columns, tiles):
        return (-(-columns // 256) if columns <= 1024 else 4 * -(-columns // 1024)) * tiles

    # measured best split lengths (sequences, context / 32 tiles) -> tiles per split
    assert uniform(128, 64) == 64 and uniform(256, 64) == 64 and uniform(256, 256) == 256 and uniform(512, 32) == 32      # whole sequences
    assert uniform(64, 256) == 128                                # two splits: 256 workgroups
    assert uniform(32, 128) == 32 and uniform(16, 256) == 32      # 256 workgroups
    assert uniform(32, 1024) == 256 and uniform(4, 4096) == 128   # 256 workgroups
    assert uniform(48, 512) == 64                                 # 8 splits: 768 workgroups = three per CU (288 would be 0.50 of peak)
    assert uniform(300, 64) == 64                                 # 600 columns x 2k: pieces gave nothing (whole 0.66 of HBM peak, two pieces 0.62)
    # more than two rounds of whole sequences, 4k context and up: the pieces that balance the last round (balanced_tiles_per_piece,
    # profiles/r06_batch_over_cus.txt: 260 x 8k 0.59 -> 0.70 of HBM peak, 300 0.68 -> 0.74, 340 0.73 -> 0.78, 448 0.72 -> 0.76)
    assert uniform(260, 256) == 64 and uniform(300, 256) == 128 and uniform(340, 256) == 86 and uniform(448, 256) == 128
    assert uniform(384, 256) == 256 and uniform(480, 256) == 256 and uniform(512, 256) == 256      # whole rounds, or near enough: whole sequences
    # ... and between one round and two (130 x 8k: 0.62 -> 0.64, 200 x 8k: 0.69 -> 0.73)
    assert uniform(130, 256) == 32 and uniform(160, 256) == 64 and uniform(200, 256) == 86 and uniform(230, 256) == 256
    assert uniform(100, 128) == 26                                # 200 columns x 5 splits = 1000 workgroups (measured 0.61 of HBM peak, whole 0.54) since the merge is one wave per row
    rng = np.random.default_rng(5)
    for _ in range(400):
        n_seq = int(rng.integers(1, 700))
        tiles = int(rng.integers(1, 5000))
        tps = uniform(n_seq, tiles)
        splits = -(-tiles // tps)
        wgs = n_seq * hq * splits
        assert 1 <= tps <= tiles and (splits == 1 or tps >= 8) and splits <= 2048
        if n_seq * hq > 256 and tiles >= 128:                     # the balanced-pieces rule: at most 8 pieces of 32 tiles or more
            assert splits <= 8 and (splits == 1 or tps >= 32)
            continue
        if n_seq * hq >= 256:
            assert splits == 1 or wgs <= 1024                     # a batch that covers the chip is split only inside one residency
        if splits > 1:                                            # priced below whole sequences by the rule's own cost
            assert wgs <= 1024 and -(-wgs // 256) * (tps + 3) + (8 if splits <= 8 else 16) < cost_whole(n_seq * hq, tiles), (n_seq, tiles, tps, wgs)
    # ragged batches: priced on their real workgroup count, and every sequence's own split count stays within the merge's limit
    for _ in range(200):
        n_seq = int(rng.integers(1, 300))
        t = rng.integers(0, 3000, n_seq).astype(np.uint32)
        tps = f(t.ctypes.data_as(C.POINTER(C.c_uint32)), n_seq, 0, hq)
        assert tps >= 1 and (t.max() == 0 or tps <= max(int(t.max()), 8))
        assert int(np.max(-(-t.astype(np.int64) // tps))) <= 2048
    assert f(None, 0, 100, hq) == 8 and f(None, 10, 0, hq) == 8  # nothing to do: any legal length


def test_balanced_pieces_rule_for_batches_over_the_cu_count(rules):
    """ring_rule.hpp::balanced_tiles_per_piece (batch attention of more workgroup columns than whole rounds of the machine take, round 6,
    profiles/r06_batch_over_cus.txt): the measured shapes get pieces within 3 % of the best measured, batches that fill whole rounds stay
    whole, a piece is never shorter than the model's minimum, at most 8 pieces, the pieces cover the longest sequence, the answer for a
    uniform batch is the same from a list and from a bound, and it never prices pieces above whole sequences."""
    f = rules.rules_balanced_tiles_per_piece
    MX4, FP8, INT4 = 0, 1, 2

    def uniform(n_seq, tiles, cols, model, cus=256):
        a = f(None, n_seq, tiles, cols, cus, model)
        b = f((C.c_uint32 * n_seq)(*([tiles] * n_seq)), n_seq, 0, cols, cus, model)
        assert a == b
        return a

    # 8k context = 256 tiles; (sequences -> tiles per piece) as measured best or within 3 % of it
    assert [uniform(n, 256, 1, MX4) for n in (260, 300, 340, 384, 448, 512)] == [64, 64, 86, 256, 256, 256]
    assert [uniform(n, 256, 2, FP8) for n in (260, 300, 340, 384, 448, 480, 512)] == [64, 128, 86, 256, 128, 256, 256]
    assert [uniform(n, 256, 1, INT4) for n in (260, 300, 340, 384, 448, 512)] == [32, 64, 86, 128, 256, 256]
    assert uniform(260, 64, 1, MX4) == 32 and uniform(260, 64, 1, INT4) == 32          # 2k context: two pieces (0.48 -> 0.51, 0.36 -> 0.45 of HBM peak)
    HALVES = 3                                                    # INT4, at most CUs sequences: the 16-wave form
    assert [uniform(n, 256, 1, HALVES) for n in (130, 160, 200, 230, 256)] == [37, 86, 256, 256, 256]
    assert [uniform(n, 256, 1, MX4) for n in (130, 160, 200, 230, 256)] == [86, 86, 256, 256, 256]
    min_tiles = {MX4: 24, FP8: 32, INT4: 32, HALVES: 32}
    rng = np.random.default_rng(23)
    for _ in range(600):
        model = int(rng.integers(0, 4))
        cols = 2 if model == FP8 else 1
        n_seq = int(rng.integers(1, 1500))
        tiles = int(rng.integers(1, 3000))
        cus = int(rng.choice([256, 304, 64]))
        tps = uniform(n_seq, tiles, cols, model, cus)
        pieces = -(-tiles // tps)
        assert 1 <= tps <= tiles and pieces <= 8 and (pieces == 1 or tps >= min_tiles[model])
        if (n_seq * cols) % cus == 0:
            assert pieces == 1                                    # whole rounds of whole sequences: nothing to balance
    for _ in range(200):                                          # ragged batches: priced on their real workgroup count
        n_seq = int(rng.integers(1, 600))
        t = rng.integers(0, 2000, n_seq).astype(np.uint32)
        tps = f(t.ctypes.data_as(C.POINTER(C.c_uint32)), n_seq, 0, 1, 256, MX4)
        assert tps >= 1 and (t.max() == 0 or tps <= int(t.max()))
        assert int(np.max(-(-t.astype(np.int64) // tps))) <= 8
    assert f(None, 0, 100, 1, 256, MX4) == 100 and f(None, 10, 0, 1, 256, MX4) == 8     # nothing to do: any legal length


def test_dispatch_order_of_ragged_batches(rules):
    """ring_rule.hpp::dispatch_order_by_length (AttendArgs::order, round 6): a permutation; longest first; with rounds, every second round
    reversed, so that position p of round 2k and position p of round 2k + 1 hold a long and a short sequence (their sums even out);
    sequences of (nearly) one length keep the caller's order."""
    f = rules.rules_dispatch_order
    f.restype = C.c_int
    f.argtypes = [C.POINTER(C.c_uint32), C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32)]
    rng = np.random.default_rng(31)
    for n, rnd in ((256, 128), (512, 128), (512, 256), (300, 0), (7, 2), (2, 1), (1000, 128)):
        ln = rng.integers(32, 8192, n).astype(np.uint32)
        order = np.full(n, 0xFFFFFFFF, dtype=np.uint32)
        assert f(ln.ctypes.data_as(C.POINTER(C.c_uint32)), n, rnd, order.ctypes.data_as(C.POINTER(C.c_uint32))) == 1
        assert sorted(order.tolist()) == list(range(n))
        got = ln[order].astype(np.int64)
        if rnd == 0:
            assert np.all(np.diff(got) <= 0)                        # longest first
        else:
            for k in range(0, n, rnd):
                d = np.diff(got[k:k + rnd])
                assert np.all(d <= 0) if (k // rnd) % 2 == 0 else np.all(d >= 0)
            if n >= 2 * rnd:                                        # a CU's pair from rounds 0 and 1: sums within the spread of neighbouring ranks
                sums = got[:rnd] + got[rnd:2 * rnd]
                assert sums.max() - sums.min() <= 0.35 * sums.mean(), (n, rnd, sums.min(), sums.max())
    same = np.full(64, 4096, dtype=np.uint32); same[3] = 4000
    order = np.zeros(64, dtype=np.uint32)
    assert f(same.ctypes.data_as(C.POINTER(C.c_uint32)), 64, 16, order.ctypes.data_as(C.POINTER(C.c_uint32))) == 0      # one length: as given


def test_ragged_piece_length_rule(rules):
    """ring_rule.hpp::ragged_tiles_per_piece (members of different lengths): nothing while the longest member stays within 1.25 x a CU's fair share of the
    launch (tiles x workgroup columns / CUs); otherwise pieces of that share, 128 tiles at most, 32 at least, half the longest member at most."""
    f = rules.rules_ragged_tiles_per_piece
    f.restype = C.c_uint32
    f.argtypes = [C.POINTER(C.c_uint32), C.c_uint32, C.c_uint32, C.c_uint32]
    ptr = lambda a: a.ctypes.data_as(C.POINTER(C.c_uint32))
    rng = np.random.default_rng(7)
    t = rng.integers(32, 513, 256).astype(np.uint32)               # 256 x 1k .. 16k: share 272 (one column), 544 (FP8)
    assert f(ptr(t), 256, 256, 0) == 128 and f(ptr(t), 256, 256, 1) == 0      # MXFP4 / INT4: pieces; FP8: whole, in serpentine order
    t2 = rng.integers(32, 513, 512).astype(np.uint32)
    assert f(ptr(t2), 512, 256, 0) == 0                            # two machines' worth: whole sequences, longest first
    tail = rng.integers(32, 129, 256).astype(np.uint32); tail[5::16] = 1024
    assert 100 <= f(ptr(tail), 256, 256, 0) <= 128 and 55 <= f(ptr(tail), 256, 256, 1) <= 80       # FP8: a quarter of its share (four workgroups per CU)
    assert 32 <= f(ptr(tail[:64].copy()), 64, 256, 0) <= 40        # few members: short pieces, so that the long members spread over the machine
    same = np.full(256, 256, dtype=np.uint32)
    assert f(ptr(same), 256, 256, 0) == 0 and f(ptr(same[:1].copy()), 1, 256, 0) == 0
    assert f(ptr(same[:32].copy()), 32, 256, 1) == 0 and f(ptr(same[:8].copy()), 8, 256, 0) == 0      # few members of one length are over a CU's share too: not this rule's business
    for _ in range(300):
        n = int(rng.integers(1, 700)); model = int(rng.choice([0, 1, 3]))
        t = rng.integers(0, int(rng.integers(2, 3000)), n).astype(np.uint32)
        tps = f(ptr(t), n, 256, model)
        if tps:
            assert (16 if model == 1 else 32) <= tps <= max(32, int(t.max()) // 2) and -(-int(t.max()) // tps) <= 2048


def test_int4_batch_unequal_split_rule(rules):
    """The INT4 batch attention's two-pieces rule (ring_rule.hpp::int4_unequal_fraction / unequal_pieces, measured in
    profiles/r03_int4_batch_split_sweep.txt): only batches that fill between half and the whole machine with workgroup columns,
    only sequences of 6k positions or more; the first piece is the long one, both are non-empty and cover the sequence; the
    measured shapes get the measured fractions."""
    out = (C.c_uint32 * 3)()

    def rule(columns, tiles_max, n_tiles):
        rules.rules_int4_unequal(columns, tiles_max, n_tiles, out)
        return bool(out[0]), int(out[1]), int(out[2])

    assert rule(512, 256, 256) == (True, 205, 2)                  # 256 sequences x 8k: 0.8 of 256 tiles first
    assert rule(384, 256, 256) == (True, 141, 2)                  # 192 sequences: near-equal halves
    assert rule(512, 128, 128)[0] is False                        # 4k context: whole sequences
    assert rule(256, 256, 256)[0] is False and rule(768, 256, 256)[0] is False      # too few / enough columns
    rng = np.random.default_rng(17)
    for _ in range(2000):
        columns = int(rng.integers(1, 1200)); tiles_max = int(rng.integers(1, 5000)); n_tiles = int(rng.integers(0, tiles_max + 1))
        on, first, pieces = rule(columns, tiles_max, n_tiles)
        if not on:
            assert pieces == (1 if n_tiles else 0)
            continue
        assert 384 <= columns <= 672 and tiles_max >= 192
        if n_tiles < 192:
            assert pieces == (1 if n_tiles else 0) and (first == n_tiles or n_tiles == 0)      # short members of the batch stay whole
        else:
            assert pieces == 2 and n_tiles / 2 <= first < n_tiles
            second = n_tiles - first
            assert 1 <= second <= first and -(-n_tiles // first) == 2      # what the kernel computes from tiles_per_split = first



def test_mx4_class_enumeration_covers_every_page_once_and_stays_inside_the_planners_bound():
    """k_attend_mx4's striped form takes the pages of a range by residue class (kernels.hpp: mx4_class_tiles, mx4_striped_tiles):
    class c = pages j with j % n == c, in tiles of 16, every class with the tile count of the largest.  Restated here: every page
    of the range appears in exactly one (class, tile, row), rows past a class's end are the masked ones, and the tile count stays
    under the bound the planner sizes its splits with (engine_attend.cpp: plan_geometry, ceil(pages / 16) + runs + 1)."""
    def class_tiles(n_pages, n):
        return ((n_pages + n - 1) // n + 15) // 16

    for n in range(1, 9):
        for pages in list(range(1, 300)) + [511, 512, 513, 4095, 4096, 4097, 16384, 16385, 131071]:
            m = class_tiles(pages, n)
            total = n * m
            assert total <= (pages + 15) // 16 + n + 1, (n, pages, total)
            if pages > 2000:
                continue
            seen = set()
            for t in range(total):
                c, tm = divmod(t, m)
                for row in range(16):
                    j = c + n * (16 * tm + row)
                    if j < pages:                                   # (the kernel's mask: j >= n_pages)
                        assert j not in seen
                        seen.add(j)
            assert len(seen) == pages, (n, pages)
