#!/usr/bin/env python3
"""Copy one round-4 evidence run (profiles/tools/r4_final.sh <run>) from gpurun_out/ into profiles/ under a tag:
    python profiles/publish_r04.py r04a r4a
Files: <tag>_summary.json (collect_r04.sh: the driver's command traced per dispatch, PMC), <tag>_kernel_trace_driver_cmd.csv (the
per-dispatch rows of the dominant kernel), <tag>_pmc.json (what bench.py's roofline.traffic reads), the bench lines, the PMC of
the whole-record INT4 kernel and of the tensor codec at 2.5 GiB."""
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
for src, dst in ((f"pmc_{run}_int4_wg8/summary.json", "int4_wg8_pmc.json"), (f"pmcmem_{run}_int4_wg8/summary.json", "int4_wg8_mem_pmc.json")):
    if os.path.exists(os.path.join(g, src)):
        shutil.copy(os.path.join(g, src), os.path.join(P, f"{tag}_{dst}"))
# tensor codec at 2.5 GiB: per-kernel FETCH_SIZE / WRITE_SIZE means and kernel times
tc = {"source": "rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE | --kernel-trace --stats (three runs) on profiles/tools/tc_bench.py 1342177280 (2.5 GiB fp16 source)", "kernels": {}}
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    vals = collections.defaultdict(list)
    for f in glob.glob(os.path.join(g, f"pmc_{run}_tc_{ctr}", "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            nm = r["Kernel_Name"]
            if "::k_t" in nm and r["Counter_Name"] == ctr:
                a = nm.index("::k_t") + 2
                vals[nm[a:nm.find("(", a)]].append(float(r["Counter_Value"]))
    for nm, v in vals.items():
        tc["kernels"].setdefault(nm, {})[ctr + "_KiB_mean"] = sum(v) / len(v)
for f in glob.glob(os.path.join(g, f"trace_{run}_tc", "**", "*kernel_stats.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        nm = r["Name"]
        if "::k_t" in nm:
            a = nm.index("::k_t") + 2
            tc["kernels"].setdefault(nm[a:nm.find("(", a)], {})["avg_us"] = round(float(r["AverageNs"]) / 1e3, 1)
for nm, d in tc["kernels"].items():
    if "FETCH_SIZE_KiB_mean" in d and "WRITE_SIZE_KiB_mean" in d:
        d["hbm_bytes"] = int(d["FETCH_SIZE_KiB_mean"] * 2048 + d["WRITE_SIZE_KiB_mean"] * 1024)
if tc["kernels"]:
    json.dump(tc, open(os.path.join(P, f"{tag}_tensor_codec_pmc.json"), "w"), indent=1)
if os.path.exists(os.path.join(g, f"{run}_conn_step.txt")):
    shutil.copy(os.path.join(g, f"{run}_conn_step.txt"), os.path.join(P, f"{tag}_connector_step.txt"))
print("published", tag)
