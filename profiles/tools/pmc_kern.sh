#!/bin/bash
# PMC of one kernel: pmc_kern.sh <name-substring> <outdir-tag> -- <python script and args>
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
cd /tmp
KN=$1; TAG=$2; shift 3
OUT=$R/gpurun_out/$TAG
rm -rf $OUT; mkdir -p $OUT
run() { name=$1; shift; timeout 300 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- python3 $R/$CMD > $OUT/$name.log 2>&1; }
CMD="$*"
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVES GRBM_GUI_ACTIVE
run sq2 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_SMEM
run sq3 SQ_WAIT_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/$CMD > $OUT/trace.log 2>&1
python3 - $OUT "$KN" <<'PY'
import csv, sys, glob, collections, os, json
out, kn_sub = sys.argv[1], sys.argv[2]
res = {}
for fn in sorted(glob.glob(os.path.join(out, "sq*", "**", "*counter_collection.csv"), recursive=True)):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(fn)):
        if kn_sub in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in sorted(agg.items()):
        res[k] = {"launches": len(v), "mean": sum(v) / len(v)}
for fn in glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True):
    for r in csv.DictReader(open(fn)):
        if kn_sub in r["Name"]:
            res["trace"] = {"calls": r["Calls"], "avg_ns": r["AverageNs"]}
json.dump(res, open(os.path.join(out, "summary.json"), "w"), indent=1)
for k, v in res.items(): print(k, v)
PY
