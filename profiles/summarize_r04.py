#!/usr/bin/env python3
"""Condense a profiles/collect_r04.sh directory into one JSON whose headline numbers can be checked against the bench
line of the SAME process:

  * the bench prints, per variant, the 0-based index range of its timed dispatches of k_fetch_decompress<2, 0, false, 0>
    (`roofline.launches`); this script sorts the rocprofv3 per-dispatch kernel trace by start time, keeps that template
    instance, and averages exactly those dispatches: avg_us_as_called, avg_us_ramped, avg_us_sustained;
  * beside each it puts the bench's own HIP-event average (which also contains the gaps between back-to-back launches)
    and the relative difference -- VERDICT r3: within 2 %;
  * frac = algorithmic bytes per launch / that average / 8 TB/s, from the trace alone;
  * PMC: FETCH_SIZE x 1024 x 2 (gfx950 tallies a 128-B request as 64 B: MI355X_MICROARCH.md, HBM) + WRITE_SIZE x 1024.

    python profiles/summarize_r04.py gpurun_out/prof_<tag>  > profiles/<tag>_summary.json
"""
import collections, csv, glob, json, os, sys

HBM_PEAK = 8000.0e9


def last_json_line(path):
    try:
        lines = [ln for ln in open(path).read().splitlines() if ln.startswith('{"metric"')]
        return json.loads(lines[-1]) if lines else None
    except OSError:
        return None


def dispatches(trace_dir, instance):
    rows = []
    for f in glob.glob(os.path.join(trace_dir, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if instance in r["Kernel_Name"]:
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r["Dispatch_Id"])))
    rows.sort()
    return rows


def main(out_dir):
    res = {"command": "python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline"}
    traced = last_json_line(os.path.join(out_dir, "bench_driver_cmd_traced.json"))
    plain = last_json_line(os.path.join(out_dir, "bench_driver_cmd_unprofiled.json"))
    if plain:
        r = plain["roofline"]
        res["unprofiled_run"] = {"value": plain["value"], "ms_per_step": plain["ms_per_step"], "frac": r["frac"], "frac_as_called": r.get("frac_as_called"),
                                 "avg_launch_ms": r["avg_launch_ms"], "avg_launch_ms_as_called": r.get("avg_launch_ms_as_called"),
                                 "variants": {k: {"avg_launch_ms": v["avg_launch_ms"], "frac_hbm": v["frac_hbm"]} for k, v in plain["variants"].items()}}
    if traced:
        r = traced["roofline"]
        inst = r["launches"]["kernel_instance"]
        # (match on the instance's leading template arguments: a later argument with a default may have joined the list)
        rows = dispatches(os.path.join(out_dir, "trace_driver_cmd"), inst.rstrip(">").rsplit(", false", 1)[0] if inst.count("false") > 1 else inst.rstrip(">"))
        alg = r["algorithmic_bytes_per_launch"]
        t = {"kernel_instance": inst, "dispatches_of_instance_in_trace": len(rows), "algorithmic_bytes_per_launch": alg,
             "bench_line_of_this_process": {"frac": r["frac"], "frac_as_called": r["frac_as_called"], "avg_launch_ms": r["avg_launch_ms"]}}
        for name, v in traced["variants"].items():
            lo, hi = v["launches"]
            sel = rows[lo:hi + 1]
            if len(sel) != hi - lo + 1:
                t[name] = {"error": f"trace holds {len(rows)} dispatches, variant wants [{lo}, {hi}]"}
                continue
            durs = [e - s for s, e, _ in sel]
            avg = sum(durs) / len(durs)
            span = (sel[-1][1] - sel[0][0]) / len(sel)           # start of first to end of last: includes the gaps, like HIP events
            t[name] = {"launch_index_range": [lo, hi], "dispatch_id_range": [sel[0][2], sel[-1][2]], "n": len(sel),
                       f"avg_us_{name}": round(avg / 1e3, 2), "min_us": round(min(durs) / 1e3, 2), "max_us": round(max(durs) / 1e3, 2),
                       "avg_us_first_start_to_last_end": round(span / 1e3, 2),
                       "frac_from_trace": round(alg / (avg * 1e-9) / HBM_PEAK, 4),
                       "bench_hip_event_avg_us": round(v["avg_launch_ms"] * 1e3, 2), "bench_frac_hbm": v["frac_hbm"],
                       "trace_vs_hip_events": round(avg / 1e3 / (v["avg_launch_ms"] * 1e3) - 1.0, 4)}
        res["traced_run"] = t
        for f in glob.glob(os.path.join(out_dir, "trace_driver_cmd", "**", "*kernel_stats.csv"), recursive=True):
            res["kernel_stats_all_launches"] = [{k: row[k] for k in ("Name", "Calls", "AverageNs", "MinNs", "MaxNs", "Percentage")}
                                                 for row in list(csv.DictReader(open(f)))[:6] if "speckv" in row["Name"]]
    pmc = {}
    for tag, ctr in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
        vals = collections.defaultdict(list)
        for f in glob.glob(os.path.join(out_dir, tag, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                if r.get("Counter_Name") == ctr and "speckv" in r.get("Kernel_Name", ""):
                    nm = r["Kernel_Name"]
                    if "::k_" in nm:
                        a = nm.index("::k_") + 2
                        b = nm.find("(", a)
                        nm = nm[a:b if b > 0 else None]
                    vals[nm].append(float(r["Counter_Value"]))
        for nm, v in vals.items():
            pmc.setdefault(nm, {})[ctr + "_KiB_mean"] = sum(v) / len(v)
            pmc[nm]["launches_" + ctr] = len(v)
    for nm, d in pmc.items():
        if "FETCH_SIZE_KiB_mean" in d and "WRITE_SIZE_KiB_mean" in d:
            d["hbm_traffic_bytes_per_launch"] = int(d["FETCH_SIZE_KiB_mean"] * 1024 * 2 + d["WRITE_SIZE_KiB_mean"] * 1024)
            d["correction"] = "FETCH_SIZE x 1024 x 2 (gfx950: 128-B requests tallied at 64 B) + WRITE_SIZE x 1024"
    res["pmc"] = pmc
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main(sys.argv[1])
