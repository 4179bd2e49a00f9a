#!/usr/bin/env python3
"""Condense a profiles/collect_r02.sh output directory into one JSON (kernel stats + PMC means per kernel)."""
import collections, csv, glob, json, os, sys
out_dir = sys.argv[1]
res = {"kernel_stats": {}, "pmc": {}}
for tag in ("trace_ascalled", "trace_full"):
    for f in glob.glob(os.path.join(out_dir, tag, "**", "*kernel_stats.csv"), recursive=True):
        rows = list(csv.DictReader(open(f)))
        res["kernel_stats"][tag] = [{k: r[k] for k in r if k in ("Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs")} for r in rows[:12]]
for d in sorted(glob.glob(os.path.join(out_dir, "pmc_*"))):
    if not os.path.isdir(d):
        continue
    agg = collections.defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            name = r.get("Kernel_Name", "")
            if "k_fetch_decompress" not in name:
                continue
            agg[r.get("Counter_Name", "")].append(float(r.get("Counter_Value", 0)))
    res["pmc"][os.path.basename(d)] = {k: {"launches": len(v), "mean": sum(v) / len(v)} for k, v in agg.items() if v}
print(json.dumps(res, indent=1))
