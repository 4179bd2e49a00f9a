#!/usr/bin/env python3
"""Copy one round-6 evidence run (profiles/tools/r6_final.sh <run>) from gpurun_out/ into profiles/ under a tag:
    python profiles/publish_r06.py r06a r6a
Files: <tag>_summary.json (collect_r04.sh: the driver's command traced per dispatch, PMC), <tag>_kernel_trace_driver_cmd.csv (the
per-dispatch rows of the dominant kernel), <tag>_pmc.json (what bench.py's roofline.traffic reads), the bench lines, the PMC of
k_attend_mx4 and of the batched tensor codec (SQ counters, memory side), the MXFP4 attention table, striped pools, the connector step by
pool format and phase by phase, the access-miss tool, the accuracy table, the predictor's launches per dispatch.  <tag>_pmc.json also carries
the SHA-256 of the headline kernel's instructions as built (tests/test_build_guards.py checks the newest such file against the build:
bench.py's roofline.traffic is read from it and must not outlive the kernel it was measured on)."""
import collections, csv, glob, json, os, shutil, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, run = sys.argv[1:3]
g, P = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")
prof = os.path.join(g, f"prof_{run}")


def one_line(src, dst):
    s = open(src).read()
    s = s[s.index('{"metric'):]
    line = [ln for ln in s.strip().splitlines() if ln.startswith('{"metric')][-1]
    json.loads(line)
    open(dst, "w").write(line + "\n")


summ = json.load(open(os.path.join(prof, "summary.json")))
json.dump(summ, open(os.path.join(P, f"{tag}_summary.json"), "w"), indent=1)
# the per-dispatch rows of the dominant kernel (the file the averages in _summary.json were taken from)
inst = summ["traced_run"]["kernel_instance"]
rows = []
for f in glob.glob(os.path.join(prof, "trace_driver_cmd", "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if inst.rstrip(">") in r["Kernel_Name"]:
            rows.append((int(r["Start_Timestamp"]), int(r["Dispatch_Id"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["VGPR_Count"], r["Grid_Size_X"]))
rows.sort()
with open(os.path.join(P, f"{tag}_kernel_trace_driver_cmd.csv"), "w") as f:
    f.write("launch_index,dispatch_id,start_ns,duration_ns,vgpr_count,grid_size_x\n")
    for i, (st, did, dur, vg, gs) in enumerate(rows):
        f.write(f"{i},{did},{st},{dur},{vg},{gs}\n")
for f in glob.glob(os.path.join(prof, "trace_driver_cmd", "**", "*kernel_stats.csv"), recursive=True):
    shutil.copy(f, os.path.join(P, f"{tag}_kernel_stats_driver_cmd.csv"))
k = next(v for kk, v in summ["pmc"].items() if kk.startswith("k_fetch_decompress<2, 0, false"))
sys.path.insert(0, ROOT)
from tests.test_build_guards import headline_kernel_hash
hk_name, hk_sha, hk_n = headline_kernel_hash()
json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) on `bench.py --gpus 1 --steps 6 --warmup 2 --no-variants --no-extras` (profiles/collect_r04.sh)",
           "kernel": hk_name, "kernel_instructions": hk_n, "kernel_instructions_sha256": hk_sha,
           "units": "KiB; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B read requests at 64 B)",
           "pmc": {"k_fetch_decompress<2, 0, false, 0>": k, **{kk: v for kk, v in summ["pmc"].items() if not kk.startswith("k_fetch_decompress<2, 0, false")}}},
          open(os.path.join(P, f"{tag}_pmc.json"), "w"), indent=1)
one_line(os.path.join(prof, "bench_driver_cmd_traced.json"), os.path.join(P, f"{tag}_bench_driver_cmd_traced.json"))
one_line(os.path.join(prof, "bench_driver_cmd_unprofiled.json"), os.path.join(P, f"{tag}_bench_driver_cmd_unprofiled.json"))
for src, dst in ((f"{run}_bench_n1.json", "bench.json"), (f"{run}_bench_driver_cmd.json", "bench_driver_cmd.json"),
                 (f"{run}_bench_n2fake.json", "bench_2ranks_one_gpu.json"), (f"{run}_bench_n8fake.json", "bench_8ranks_one_gpu.json")):
    try:
        one_line(os.path.join(g, src), os.path.join(P, f"{tag}_{dst}"))
    except Exception as e:
        print("skipped", src, repr(e))
for src, dst in ((f"pmc_{run}_mx4/summary.json", "mx4_pmc.json"), (f"pmcmem_{run}_mx4/summary.json", "mx4_mem_pmc.json")):
    if os.path.exists(os.path.join(g, src)):
        shutil.copy(os.path.join(g, src), os.path.join(P, f"{tag}_{dst}"))
# the batched tensor codec: both kernels, SQ and memory side, in one file
tc = {}
for kern, label in (("tcm", "k_tcm_fused<0, true> (compress, fp32 source)"), ("tdm", "k_tdm_fused<0, true> (decompress, fp32 output)")):
    one = {}
    for d in (f"pmcmem_{run}_{kern}", f"pmc_{run}_{kern}"):
        fn = os.path.join(g, d, "summary.json")
        if os.path.exists(fn):
            one.update(json.load(open(fn)))
    if one:
        tc[label] = one
if tc:
    tc["workload"] = "4096 tensors x 131 072 fp32 elements (the reference's call size), N(0,1): profiles/tools/tcb_bench.py; counters are per launch (mean)"
    json.dump(tc, open(os.path.join(P, f"{tag}_tensor_codec_batched_pmc.json"), "w"), indent=1)
for src, dst in ((f"{run}_mx4_bench.txt", "mx4_bench.txt"), (f"{run}_conn_step_schemes.txt", "connector_step_by_scheme.txt"), (f"{run}_access_miss.txt", "access_miss.txt"),
                 (f"{run}_tcb_bench.json", "tensor_codec_batched.json"), (f"{run}_striped.txt", "striped_attention.txt"), (f"{run}_conn_step.txt", "connector_step.txt"),
                 (f"{run}_kv_accuracy.txt", "kv_format_accuracy.txt"), (f"{run}_pred_trace.txt", "predictor_trace.txt")):
    if os.path.exists(os.path.join(g, src)):
        shutil.copy(os.path.join(g, src), os.path.join(P, f"{tag}_{dst}"))
print("published", tag)
