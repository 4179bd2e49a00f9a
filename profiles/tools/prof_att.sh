#!/bin/bash
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
cd /tmp
OUT=$R/gpurun_out/${1:-att_a}
mkdir -p $OUT
export SPLITS=${SPLITS:-8}
run() { name=$1; shift
  timeout 120 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- python3 $R/profiles/tools/attend_bench.py > $OUT/$name.log 2>&1
}
timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/profiles/tools/attend_bench.py > $OUT/trace.log 2>&1
run fetch FETCH_SIZE TCC_EA_RDREQ_sum TCC_EA_RDREQ_32B_sum
run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCP_TCC_READ_REQ_sum
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAVES GRBM_GUI_ACTIVE
run sq2 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM SQ_INSTS_MFMA SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS
run sq3 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VALU_MFMA_MOPS_F8 SQ_INST_LEVEL_VMEM SQ_WAIT_INST_LDS TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr TCP_TCC_READ_REQ_LATENCY_sum
for f in $(find $OUT/trace -name "*kernel_stats.csv"); do grep -E "attend|qk_scores" $f | cut -c1-160; done
python3 - $OUT <<'PY'
import csv, sys, glob, collections, os
out = sys.argv[1]
for fn in sorted(glob.glob(os.path.join(out, "*", "**", "*counter_collection.csv"), recursive=True)):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(fn)):
        kn = r["Kernel_Name"]
        if "k_attend_fp8" in kn:
            agg[(kn[:34], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        print(f"{k[0]:36s} {k[1]:32s} n={len(v)} mean={sum(v)/len(v):.5g}")
PY
grep -h "rror" $OUT/*.log | sort | uniq -c | head -8
