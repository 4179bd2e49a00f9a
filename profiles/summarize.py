#!/usr/bin/env python3
"""Condense a gpurun_out/prof_* directory (rocprofv3 --kernel-trace --stats and
separate --pmc FETCH_SIZE / WRITE_SIZE passes of bench.py, see profiles/tools/prof.sh /
profiles/README.md) into small tracked files:

    profiles/<tag>_kernel_stats.csv   per-kernel Calls / Avg / Min / Max ns
    profiles/<tag>_pmc.json           per-launch HBM traffic of the dominant kernel

HBM byte accounting follows MI355X_MICROARCH.md "HBM": FETCH_SIZE / WRITE_SIZE are
in KiB; on gfx950 FETCH_SIZE counts exactly half of a wide coalesced streaming
read (16 B/lane), so it is doubled; WRITE_SIZE is exact for 16 B/lane stores.
"""
import collections
import csv
import glob
import json
import os
import sys


def main(src, tag):
    here = os.path.dirname(os.path.abspath(__file__))
    stats = glob.glob(os.path.join(src, "trace", "**", "*kernel_stats.csv"), recursive=True)
    rows = list(csv.DictReader(open(stats[0]))) if stats else []
    with open(os.path.join(here, f"{tag}_kernel_stats.csv"), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name(truncated)", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
        for r in rows:
            w.writerow([r["Name"][:96], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"],
                        r["MinNs"], r["MaxNs"], r["StdDev"]])
    pmc = {}
    for counter, sub in (("FETCH_SIZE", "pmc_fetch"), ("WRITE_SIZE", "pmc_write")):
        files = glob.glob(os.path.join(src, sub, "**", "*counter_collection.csv"), recursive=True)
        agg = collections.defaultdict(list)
        for fn in files:
            for r in csv.DictReader(open(fn)):
                if r["Counter_Name"] == counter:
                    agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
        for k, v in agg.items():
            if "speckv" in k:
                short = k.split("(anonymous namespace)::")[-1].split("(")[0]
                pmc.setdefault(short, {})[counter + "_KiB_mean"] = sum(v) / len(v)
                pmc[short][counter + "_launches"] = len(v)
    for k, d in pmc.items():
        fetch = d.get("FETCH_SIZE_KiB_mean", 0.0) * 1024 * 2      # gfx950: x2 for wide coalesced reads
        write = d.get("WRITE_SIZE_KiB_mean", 0.0) * 1024
        d["hbm_read_bytes_per_launch_corrected"] = fetch
        d["hbm_write_bytes_per_launch"] = write
        d["hbm_traffic_bytes_per_launch"] = fetch + write
    avg = {r["Name"].split("(anonymous namespace)::")[-1].split("(")[0]: float(r["AverageNs"]) for r in rows if "speckv" in r["Name"]}
    json.dump({"source": os.path.basename(src.rstrip("/")), "command": "bench.py --steps 20 --warmup 3 (trace) / --steps 5 --warmup 1 (pmc)",
               "avg_duration_ns": avg, "pmc": pmc}, open(os.path.join(here, f"{tag}_pmc.json"), "w"), indent=1)
    print(json.dumps({"avg_duration_ns": avg, "pmc": pmc}, indent=1))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
