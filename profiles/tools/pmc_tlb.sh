#!/bin/bash
# address-translation PMC of one kernel: pmc_tlb.sh <name-substring> <tag> -- <script args>
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
cd /tmp
KN=$1; TAG=$2; shift 3
OUT=$R/gpurun_out/$TAG
rm -rf $OUT; mkdir -p $OUT
CMD="$*"
run() { name=$1; shift; timeout 300 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- python3 $R/$CMD > $OUT/$name.log 2>&1; }
run t1 TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum
run t2 TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RD_UNCACHED_32B_sum TCC_TAG_STALL_sum
run t3 TCC_EA0_RDREQ_LEVEL_sum TCC_BUSY_avr GRBM_GUI_ACTIVE
python3 - $OUT "$KN" <<'PY'
import csv, sys, glob, collections, os, json
out, kn_sub = sys.argv[1], sys.argv[2]
res = {}
for fn in sorted(glob.glob(os.path.join(out, "t*", "**", "*counter_collection.csv"), recursive=True)):
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
