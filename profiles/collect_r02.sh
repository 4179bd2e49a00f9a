#!/bin/bash
# Round-2 evidence: kernel trace + PMC passes, each in its own rocprofv3 run (never --pmc together with a trace domain).
# usage (on the GPU box): bash profiles/collect_r02.sh <tag>
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
cd /tmp
OUT=$R/gpurun_out/${1:-prof_r02}
rm -rf $OUT
mkdir -p $OUT
# 1. the bench as the driver calls it (as-called figure only: W warm-ups + K steps) and the full default bench
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_ascalled -- python3 $R/bench.py --steps 20 --warmup 5 --no-variants --no-extras --no-cpu-baseline > $OUT/bench_ascalled.json 2> $OUT/trace_ascalled.log
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_full -- python3 $R/bench.py --no-cpu-baseline > $OUT/bench_full.json 2> $OUT/trace_full.log
# 2. PMC on the dominant kernel at three footprints (FETCH_SIZE and WRITE_SIZE need separate passes; hit/miss a third)
for cfg in "4096 32" "16384 32" "8192 80"; do
  tag=$(echo $cfg | tr ' ' x)
  for pmc in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"; do
    name=$(echo $pmc | tr ' ' '+')
    rocprofv3 --pmc $pmc --output-format csv -d $OUT/pmc_${tag}_${name} -- python3 $R/profiles/tools/footprint.py $cfg 6 > $OUT/pmc_${tag}_${name}.log 2>&1
  done
done
python3 $R/profiles/summarize_r02.py $OUT > $OUT/summary.json
tail -c 1500 $OUT/summary.json
