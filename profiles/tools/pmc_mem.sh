#!/bin/bash
# memory-side PMC of one kernel: pmc_mem.sh <name-substring> <tag> -- <script args>
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
cd /tmp
KN=$1; TAG=$2; shift 3
OUT=$R/gpurun_out/$TAG
rm -rf $OUT; mkdir -p $OUT
CMD="$*"
run() { name=$1; shift; timeout 300 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- python3 $R/$CMD > $OUT/$name.log 2>&1; }
run m1 FETCH_SIZE
run m2 WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
run m3 TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
run m4 TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TA_BUSY_avr TCP_GATE_EN1_sum
python3 - $OUT "$KN" <<'PY'
import csv, sys, glob, collections, os, json
out, kn_sub = sys.argv[1], sys.argv[2]
res = {}
for fn in sorted(glob.glob(os.path.join(out, "m*", "**", "*counter_collection.csv"), recursive=True)):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(fn)):
        if kn_sub in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in sorted(agg.items()):
        res[k] = {"launches": len(v), "mean": sum(v) / len(v)}
json.dump(res, open(os.path.join(out, "summary.json"), "w"), indent=1)
for k, v in res.items(): print(k, v)
PY
grep -il "error\|invalid" $OUT/*.log | head
