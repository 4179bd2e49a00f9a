"""Coherence shadow directory: the reference's CoherenceManager behind its
coherence_manager_* C ABI.  Host bookkeeping, so everything here runs without a GPU:

  oracle  vs  the reference itself (oracle/_ref/libspeckv_ref_coh.so, dev container only)
  oracle  vs  golden traces recorded from the reference (tests/golden/coherence_trace.json)
  product (libcxlspeckv.so through the C ABI and the Python mirror)  vs  oracle and golden
"""
import ctypes as C
import json
import os
import re

import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings
from hypothesis import strategies as st

import cxl_speckv_amd as pkg
from cxl_speckv_amd.cxlspeckv_coherence import CoherenceManager, CoherenceState, MemoryTier, bind_coherence
from oracle.bindings import OracleCoherence, ReferenceCoherence, have_reference_coherence

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COMMON = dict(deadline=None, suppress_health_check=[HealthCheck.function_scoped_fixture, HealthCheck.too_slow])


class ProductCoherence:
    """The shipped library through the same op() interface as the oracle / reference drivers."""

    def __init__(self, line=64):
        self.m = CoherenceManager("/dev/null", line)
        self.data = bytes(64)

    def close(self):
        self.m.close()

    def op(self, name, a=0, t=0):
        m = self.m
        if name == "read": return int(m.request_read(a, 64) is not None)
        if name == "write": return int(m.request_write(a, self.data))
        if name == "writeback": return int(m.writeback(a, self.data))
        if name == "flush_all": return int(m.flush_all())
        if name == "update_tier": m.update_tier(a, MemoryTier(t)); return 0
        if name == "reset_statistics": m.reset_statistics(); return 0
        if name == "batch_invalidate": return int(m.batch_invalidate(list(a)))
        if name in ("get_state", "get_tier"): return int(getattr(m, name)(a))
        return int(getattr(m, name)(a))

    def run(self, ops):
        return [self.op(*o) for o in ops]

    def stats(self):
        s = self.m.get_statistics()
        return [s[k] for k in ("total_reads", "total_writes", "coherence_ops", "invalidations_sent", "writebacks_performed",
                                "directory_hits", "directory_misses")]


def traces():
    d = json.load(open(os.path.join(ROOT, "tests", "golden", "coherence_trace.json")))
    out = dict(d["scenarios"])
    for k in ("random_walk", "no_driver", "line_128"):
        out[k] = d[k]
    return out


def replay(impl, tr):
    ops = [tuple(o) for o in tr["ops"]]
    assert impl.run(ops) == tr["results"]
    for a, (state, tier) in tr["final"].items():
        assert (impl.op("get_state", int(a, 16)), impl.op("get_tier", int(a, 16))) == (state, tier), a
    assert impl.stats() == tr["stats"]


@pytest.mark.parametrize("name", sorted(traces()))
def test_oracle_replays_reference_traces(oracle, name):
    tr = traces()[name]
    oc = OracleCoherence(oracle, tr["cache_line_size"], tr["has_driver"])
    replay(oc, tr)
    oc.close()


@pytest.mark.parametrize("name", sorted(k for k, v in traces().items() if v["has_driver"]))
def test_product_replays_reference_traces(name):
    tr = traces()[name]
    pc = ProductCoherence(tr["cache_line_size"])
    replay(pc, tr)
    pc.close()


op_strategy = st.one_of(
    st.tuples(st.sampled_from(["read", "write", "invalidate", "writeback", "promote_to_l1", "demote_to_l3", "get_state", "get_tier"]),
              st.integers(0, 40).map(lambda i: i * 0x40 + (i * 7) % 64)),
    st.tuples(st.just("update_tier"), st.integers(0, 40).map(lambda i: i * 0x40), st.integers(0, 2)),
    st.tuples(st.just("batch_invalidate"), st.lists(st.integers(0, 40).map(lambda i: i * 0x40), max_size=6)),
    st.tuples(st.just("flush_all")), st.tuples(st.just("reset_statistics")))


@settings(max_examples=60, **COMMON)
@given(ops=st.lists(op_strategy, max_size=120), line=st.sampled_from([64, 128, 256, 4096]), drv=st.sampled_from([1, 1, 0]))
def test_oracle_equals_reference_on_random_walks(oracle, ops, line, drv):
    if not have_reference_coherence():
        pytest.skip("oracle/_ref/libspeckv_ref_coh.so not built (needs /root/reference)")
    oc, rc = OracleCoherence(oracle, line, drv), ReferenceCoherence(line, drv)
    try:
        assert oc.run(ops) == rc.run(ops)
        assert oc.stats() == rc.stats()
        for i in range(41):
            assert oc.op("get_state", i * 0x40) == rc.op("get_state", i * 0x40)
            assert oc.op("get_tier", i * 0x40) == rc.op("get_tier", i * 0x40)
    finally:
        oc.close(); rc.close()


@settings(max_examples=60, **COMMON)
@given(ops=st.lists(op_strategy, max_size=200), line=st.sampled_from([64, 128, 4096]))
def test_product_equals_oracle_on_random_walks(oracle, ops, line):
    oc, pc = OracleCoherence(oracle, line, 1), ProductCoherence(line)
    try:
        assert pc.run(ops) == oc.run(ops)
        assert pc.stats() == oc.stats()
        for i in range(41):
            assert pc.op("get_state", i * 0x40) == oc.op("get_state", i * 0x40)
            assert pc.op("get_tier", i * 0x40) == oc.op("get_tier", i * 0x40)
    finally:
        oc.close(); pc.close()


def test_directory_grows_past_its_first_table(oracle):
    """100 000 distinct lines (the flat table rehashes several times), then a flush and a batch."""
    oc, pc = OracleCoherence(oracle, 64, 1), ProductCoherence(64)
    addrs = (np.random.default_rng(3).permutation(100000).astype(np.uint64) * 64 + 0x4000000000)
    for a in addrs[:50000]:
        assert pc.op("write", int(a)) == 1
        oc.op("write", int(a))
    for a in addrs[50000:]:
        pc.op("read", int(a)); oc.op("read", int(a))
    assert pc.m.entry_count() == 100000
    assert pc.op("flush_all") == oc.op("flush_all") == 1
    assert pc.op("batch_invalidate", [int(a) for a in addrs[:2000]]) == oc.op("batch_invalidate", [int(a) for a in addrs[:2000]])
    assert pc.stats() == oc.stats()
    for a in addrs[::997]:
        assert pc.op("get_state", int(a)) == oc.op("get_state", int(a))
        assert pc.op("get_tier", int(a)) == oc.op("get_tier", int(a))
    oc.close(); pc.close()


def test_reference_test_scenarios_through_the_python_mirror():
    """tests/test_coherence.cpp of the reference, through the mirror of its Python binding.  Two of its own
    assertions do not hold for its own implementation (every stubbed device operation is booked as an access:
    coherence_manager.cpp:420): "total_reads == 2" after two reads of one line is 3, "total_writes == 1" is 2,
    and "total_reads == NUM_ADDRS" is 2*NUM_ADDRS.  The library reproduces the implementation, so those three
    are asserted with the value the reference code produces."""
    with CoherenceManager("/dev/null", 64) as mgr:                  # test_initialization
        s = mgr.get_statistics()
        assert (s["total_reads"], s["total_writes"], s["coherence_ops"]) == (0, 0, 0)
    with CoherenceManager("/dev/null") as mgr:                      # test_read_operations
        assert mgr.request_read(0x10000, 64) == bytes(64)
        assert mgr.get_state(0x10000) == CoherenceState.SHARED and mgr.get_tier(0x10000) == MemoryTier.L1_GPU
        assert mgr.request_read(0x10000, 64) is not None
        s = mgr.get_statistics()
        assert s["total_reads"] == 3 and s["directory_hits"] >= 1    # the reference's test says 2
    with CoherenceManager("/dev/null") as mgr:                      # test_write_operations
        mgr.request_read(0x20000, 64)
        assert mgr.request_write(0x20000, bytes([0xAB]) * 64)
        assert mgr.get_state(0x20000) == CoherenceState.MODIFIED
        s = mgr.get_statistics()
        assert s["total_writes"] == 2 and s["invalidations_sent"] >= 1   # the reference's test says 1
    with CoherenceManager("/dev/null") as mgr:                      # test_invalidation, test_writeback
        mgr.request_read(0x30000, 64)
        assert mgr.is_valid(0x30000) and mgr.invalidate(0x30000) and not mgr.is_valid(0x30000)
        mgr.request_write(0x40000, bytes(64))
        assert mgr.is_modified(0x40000) and mgr.writeback(0x40000, bytes(64))
        assert mgr.get_state(0x40000) == CoherenceState.SHARED and mgr.get_statistics()["writebacks_performed"] >= 1
    with CoherenceManager("/dev/null") as mgr:                      # tier promotion / demotion
        assert mgr.get_tier(0x50000) == MemoryTier.L3_CXL and mgr.promote_to_l1(0x50000)
        assert mgr.get_tier(0x50000) == MemoryTier.L1_GPU and mgr.demote_to_l3(0x50000)
        assert mgr.get_tier(0x50000) == MemoryTier.L3_CXL
    with CoherenceManager("/dev/null") as mgr:                      # batch, flush, statistics, multiple addresses
        addrs = [0x70000, 0x70040, 0x70080, 0x700C0]
        for a in addrs: mgr.request_read(a, 64)
        assert mgr.batch_invalidate(addrs) and all(mgr.get_state(a) == CoherenceState.INVALID for a in addrs)
        assert mgr.get_statistics()["invalidations_sent"] >= len(addrs)
        for a in (0x80000, 0x80040, 0x80080):
            mgr.request_write(a, bytes(64)); assert mgr.is_modified(a)
        assert mgr.flush_all() and not any(mgr.is_modified(a) for a in (0x80000, 0x80040, 0x80080))
        assert 0.0 <= mgr.get_statistics()["hit_rate"] <= 1.0
        mgr.reset_statistics()
        assert mgr.get_statistics()["total_reads"] == 0 and mgr.get_statistics()["total_writes"] == 0
        for i in range(10):
            mgr.request_read(0xB0000 + i * 0x1000, 64)
            assert mgr.get_state(0xB0000 + i * 0x1000) == CoherenceState.SHARED
        assert mgr.get_statistics()["total_reads"] == 20             # the reference's test says NUM_ADDRS = 10


def test_c_abi_conventions():
    """coherence_c_api.cpp:33-209: NULL-handle and NULL-buffer results, create failures."""
    lib = bind_coherence(pkg.load_library())
    header = open(os.path.join(ROOT, "include", "speckv_coherence.h")).read()
    declared = sorted(set(re.findall(r"\b(coherence_manager_[a-z0-9_]+)\s*\(", header)))
    assert len([n for n in declared if "_ext_" not in n]) == 14
    for n in declared:
        assert hasattr(lib, n), n
    assert lib.coherence_manager_create(None, 64) is None
    buf = C.create_string_buffer(64)
    assert lib.coherence_manager_request_read(None, 0, buf, 64) is False
    assert lib.coherence_manager_get_state(None, 0) == 0 and lib.coherence_manager_get_tier(None, 0) == 2
    assert lib.coherence_manager_flush_all(None) is False and lib.coherence_manager_invalidate(None, 0) is False
    lib.coherence_manager_destroy(None); lib.coherence_manager_reset_statistics(None)
    h = lib.coherence_manager_create(b"/dev/null", 64)
    assert h
    assert lib.coherence_manager_request_read(h, 0, None, 64) is False       # data_out == NULL
    assert lib.coherence_manager_request_write(h, 0, None, 64) is False
    assert lib.coherence_manager_writeback(h, 0, None, 64) is False
    assert lib.coherence_manager_batch_invalidate(h, None, 3) is False
    lib.coherence_manager_get_statistics(h, None)                            # ignored
    buf.raw = bytes([0x5A]) * 64
    assert lib.coherence_manager_request_read(h, 0x40, buf, 64) is True
    assert buf.raw == bytes([0x5A]) * 64                                     # no data path: buffer untouched
    lib.coherence_manager_destroy(h)
    # any other device path names the engine's GPU: without one the create fails like the reference's driver
    import torch
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError):
            CoherenceManager("/dev/speckv0")
