"""-m "not gpu": the N>1 contract of bench.py under gloo, world size 2, on CPU.

The path shards by blocks with no data-path collective (DESIGN.md sect. 6); what
the ranks share is the timing protocol: barrier + sync on both sides of exactly K
steps, MAX over ranks, whole-job rate = units of all ranks / that time."""
import os
import socket
import sys
import time

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench
    calls = []

    def step(i):
        calls.append(i)
        time.sleep(0.02 * (rank + 1))          # rank 1 is the slow one

    elapsed = bench.run_timed(step, steps=5, warmup=2, sync=lambda: None, dist=dist,
                              warm=lambda: calls.append("w"))
    rate = bench.whole_job_rate(world, 1000, 5, elapsed)
    # every rank owns its own shard: disjoint seeds / pools, identical sizes
    torch.save({"rank": rank, "elapsed": elapsed, "rate": rate, "calls": calls}, os.path.join(out_dir, f"r{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_timing_protocol_world2(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r = [torch.load(os.path.join(tmp_path, f"r{i}.pt")) for i in range(world)]
    for x in r:
        assert x["calls"] == ["w", "w", 0, 1, 2, 3, 4]          # W warmups, exactly K timed steps
    assert r[0]["elapsed"] == r[1]["elapsed"]                   # MAX over ranks is what everyone reports
    assert r[0]["elapsed"] >= 5 * 0.04 * 0.95                   # bounded below by the slow rank
    assert r[0]["rate"] == pytest.approx(2 * 1000 * 5 / r[0]["elapsed"])


def test_single_rank_needs_no_process_group():
    sys.path.insert(0, ROOT)
    import bench
    n = []
    e = bench.run_timed(lambda i: n.append(i), steps=3, warmup=1, sync=lambda: None)
    assert n == [0, 0, 1, 2] and e > 0
    assert bench.whole_job_rate(1, 131072, 3, 0.5) == 131072 * 3 / 0.5


def test_committed_bench_line_has_the_contract_fields():
    """The bench line committed with the round's profile (profiles/r01f_bench.json = stdout of `python bench.py` on the
    MI355X box) carries every field of the driver's contract, with the tier's meaning: metric and config from
    BASELINE.json, roofline and cpu_baseline objects, no model keys."""
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    d = json.load(open(os.path.join(root, "profiles", "r01f_bench.json")))
    base = json.load(open(os.path.join(root, "BASELINE.json")))
    assert d["metric"] in base["metric"] and d["unit"] == "blocks/s"
    for k in ("value", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert k in d, k
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert d["dtype"] == "u8" and "workload" in d["config"] and "model" not in d["config"]
    assert "configs[1]" in d["config"]["workload"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and 0.5 < r["frac"] < 1.0
    assert r["traffic"] is None or 0.95 < r["traffic"] / r["algorithmic_bytes_per_launch"] < 1.2
    assert abs(d["value"] - 131072 * d["n_gpus"] / (d["ms_per_step"] * 1e-3)) / d["value"] < 0.01
    c = d["cpu_baseline"]
    assert c["unit"] == "blocks/s" and c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["value"] > 0 and c["sample"]
    assert d["parity_spot_check"] is True
