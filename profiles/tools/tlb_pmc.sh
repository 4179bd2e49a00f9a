#!/bin/bash
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
cd /tmp
OUT=$R/gpurun_out/${PMC_TAG:-tlb_pmc}
rm -rf $OUT; mkdir -p $OUT
for i in 1 2 3 4 5 6; do
  rocprofv3 --pmc ${PMC_SET} --output-format csv -d $OUT/p$i -- python3 $R/profiles/tools/footprint.py 8192 80 40 > $OUT/p$i.log 2>&1
done
python3 - $OUT <<'PY'
import csv, glob, sys, os, collections, json
out = sys.argv[1]
res = []
for i in range(1, 7):
    rows = []
    for fn in glob.glob(os.path.join(out, f"p{i}", "**", "*counter_collection.csv"), recursive=True):
        rows += [r for r in csv.DictReader(open(fn)) if "k_fetch_decompress" in r["Kernel_Name"]]
    agg = collections.defaultdict(list)
    dur = {}
    for r in rows:
        agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    d = sorted(dur.values())
    d = d[len(d) // 4:]                      # drop the cold first quarter
    e = {k: sum(v) / len(v) for k, v in agg.items()}
    e["kernel_us_mean"] = sum(d) / len(d)
    e["launches"] = len(dur)
    res.append(e)
    print(i, {k: round(v, 1) for k, v in e.items()})
json.dump(res, open(os.path.join(out, "summary.json"), "w"), indent=1)
PY
