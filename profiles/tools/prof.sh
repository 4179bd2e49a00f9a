#!/bin/bash
# rocprofv3 passes for the bench (kernel trace + two PMC passes)
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
cd /tmp
OUT=$R/gpurun_out/prof_r01
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras > $OUT/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extras > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extras > $OUT/pmc_write.log 2>&1
find $OUT -name "*.csv" | head -20
for f in $(find $OUT/trace -name "*kernel_stats.csv"); do echo "== $f"; head -12 $f; done
for f in $(find $OUT/pmc_fetch $OUT/pmc_write -name "*counter_collection.csv"); do echo "== $f"; head -3 $f; python3 - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(list)
for r in rows:
    agg[(r.get("Kernel_Name","")[:60], r.get("Counter_Name",""))].append(float(r.get("Counter_Value",0)))
for k, v in agg.items():
    print(k, "n=", len(v), "mean=", sum(v)/len(v))
PY
done
tail -2 $OUT/trace.log
