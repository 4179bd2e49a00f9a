"""-m gpu: the N > 1 control flow of bench.py under the driver's eyes (BASELINE configs[2] and configs[3] in their REMOTE
form: 1 compute GPU + pool GPUs; the reference's path is speckv_allocator.cpp:115-138
`sync_fetch_page`, one DMA descriptor per page from the pool into the compute GPU).

The development pool has one GPU per box, so `SPECKV_BENCH_SINGLE_GPU_TEST=1` puts every rank and every "peer" on GPU 0
with gloo between the ranks: `python bench.py --gpus N` starts its own ranks (torch.distributed.run as a child), every rank
starts the remote-fetch child process, the striping (page % (N-1)), both fetch engines, the speculative-prefetch leg and the
accounting are the real code.  No link is crossed and the line says so itself; these tests check that the line is complete,
that the two fetch engines delivered the same bytes, and that nothing failed, hung or was skipped.
"""
import json
import os
import subprocess
import sys
import time

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ENGINES = ("fused_peer_load_kernel", "copy_engines_then_local_decompress")


def run_bench(n, steps=3, warmup=1, timeout=240):
    env = dict(os.environ, SPECKV_BENCH_SINGLE_GPU_TEST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "SPECKV_BENCH_HOST_DRY_RUN"):
        env.pop(k, None)
    t0 = time.time()
    # a FRESH child process: it starts its ranks itself, before anything in it touches the GPU
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", str(steps), "--warmup", str(warmup)],
                       env=env, capture_output=True, text=True, timeout=timeout, cwd=ROOT)
    took = time.time() - t0
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-3000:])
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-3000:]                          # rank 0 only, one line
    return json.loads(lines[0]), took


def check_line(d, n, steps, warmup):
    assert d["n_gpus"] == n and d["steps"] == steps and d["warmup"] == warmup
    assert "watchdog_fired" not in d and d["parity_spot_check"] is True
    assert d["scaling"] == "weak" and d["value"] > 0 and d["cpu_baseline"] is None          # measured by the N = 1 run only
    # whole job = every rank's blocks over the max-over-ranks time
    assert abs(d["value"] - n * d["config"]["blocks_per_gpu"] / (d["ms_per_step"] * 1e-3)) / d["value"] < 0.01
    x = d["xgmi"]
    assert "failed" not in x, x
    modes = {"cfg3": 1, "cfg4": n - 1, "symmetric": n - 1}
    for mode, links in modes.items():
        m = x[mode]
        if mode == "cfg4" and n == 2:
            assert m["same_as"] == "cfg3"
            continue
        assert "skipped" not in m and "same_as" not in m, (mode, m)
        assert m["links"] == links and m["pool_gpus_per_compute_gpu"] == links
        assert m["compute_ranks"] == (n if mode == "symmetric" else 1)
        for e in ENGINES:
            assert "skipped" not in m[e], (mode, e, m[e])
            assert m[e]["inbound_GBps_per_compute_gpu"] > 0 and m[e]["link_bytes_per_pass"] > 0 and m[e]["blocks_per_s_whole_job"] > 0
        # both engines fetched one whole allocation and the device compared the results bit for bit
        assert m["engines_bit_identical"] is True, (mode, m["engines_bit_identical"])
        ce = m["copy_engines_then_local_decompress"]
        assert ce["copy_engine_link_bytes_per_pass"] >= ce["link_bytes_per_pass"] and 1.0 <= ce["slot_overhead"] < 1.05
        assert m["working_set"]["allocations"] >= 1 and m["raw_peer_copy_GBps"] > 0
        pf = m["speculative_prefetch_depth4"]
        assert "skipped" not in pf, (mode, pf)
        assert pf["depth_k"] == 4 and pf["pages_per_step"] > 0 and pf["pipelined"]["pages"] > 0 and pf["ms_submit_to_landed"] > 0
        assert m["fp16_pages"]["inbound_GBps_per_compute_gpu"] > 0
    layout = "cfg3" if n == 2 else "cfg4"
    rx = d["roofline_xgmi"]
    assert rx["bound"] == "xgmi" and rx["layout"].startswith(layout) and rx["links"] == n - 1 and rx["engine"] in ENGINES
    assert abs(rx["peak_nominal_per_direction"] - (n - 1) * 153.6) < 0.1 and "ONE-GPU DRY RUN" in rx["note"]
    assert rx["engines_bit_identical"] is True
    r = d["roofline"]["xgmi"]
    assert r["frac"] == rx["frac"] and r["links"] == n - 1 and "one_gpu_dry_run" in r
    # the remote figure stands at the top level beside `value` (replica scaling of the local path)
    assert d["value_remote_fetch_blocks_per_s"] == x[layout][rx["engine"]]["blocks_per_s_whole_job"] > 0
    assert d["value_remote_fetch_GBps_inbound"] == rx["achieved"] and "replicas" in d["value_note"]


def test_bench_gpus_2_remote_pool_dry_run_on_one_gpu(record_property):
    """BASELINE configs[2]: 1 compute + 1 remote-HBM pool, speculative prefetch depth 4 (every peer on GPU 0)."""
    d, took = run_bench(2)
    record_property("seconds", round(took, 1))
    print(f"bench.py --gpus 2 (one-GPU dry run): {took:.1f} s")
    check_line(d, 2, 3, 1)


def test_bench_gpus_8_one_plus_seven_pool_dry_run_on_one_gpu(record_property):
    """BASELINE configs[3]: 1 compute + 7 pooled HBM, pages striped page % 7 (every peer on GPU 0)."""
    d, took = run_bench(8)
    record_property("seconds", round(took, 1))
    print(f"bench.py --gpus 8 (one-GPU dry run): {took:.1f} s")
    check_line(d, 8, 3, 1)
