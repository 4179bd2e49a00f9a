#!/usr/bin/env python3
"""Copy one round-5 evidence run (profiles/tools/r5_final.sh <run>) from gpurun_out/ into profiles/ under a tag:
    python profiles/publish_r05.py r05a r5a
Files: <tag>_summary.json (collect_r04.sh: the driver's command traced per dispatch, PMC), <tag>_kernel_trace_driver_cmd.csv (the
per-dispatch rows of the dominant kernel), <tag>_pmc.json (what bench.py's roofline.traffic reads), the bench lines, the PMC of
k_attend_mx4 (SQ counters, memory side), the MXFP4 attention table, the connector step by pool format, the access-miss tool."""
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
json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) on `bench.py --gpus 1 --steps 6 --warmup 2 --no-variants --no-extras` (profiles/collect_r04.sh)",
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
for src, dst in ((f"{run}_mx4_bench.txt", "mx4_bench.txt"), (f"{run}_conn_step_schemes.txt", "connector_step_by_scheme.txt"), (f"{run}_access_miss.txt", "access_miss.txt")):
    if os.path.exists(os.path.join(g, src)):
        shutil.copy(os.path.join(g, src), os.path.join(P, f"{tag}_{dst}"))
print("published", tag)
