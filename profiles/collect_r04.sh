#!/bin/bash
# Round-4 headline evidence (VERDICT r3 "Next" #1): the driver's exact bench command under `rocprofv3 --kernel-trace`
# (per-dispatch CSV), so that roofline.avg_launch_ms of THAT run can be recomputed from profiles/ dispatch by dispatch,
# plus the two PMC passes (separate runs; PMC is never combined with a trace domain).
#   usage (on the GPU box):  bash profiles/collect_r04.sh <tag>       -> gpurun_out/prof_<tag>/
# The program stands directly behind `rocprofv3 ... --` (python3 bench.py ...): no env / bash -c hop.
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
cd /tmp
OUT=$R/gpurun_out/prof_${1:-r04}
rm -rf $OUT
mkdir -p $OUT
# 0. the same command with no profiler attached (what the driver runs), extras and CPU baseline off to keep it short
python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $OUT/bench_driver_cmd_unprofiled.json 2> $OUT/bench_driver_cmd_unprofiled.err
# 1. the same command under the kernel trace: variants on (as_called, ramped, sustained all in one process)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_driver_cmd -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $OUT/bench_driver_cmd_traced.json 2> $OUT/trace_driver_cmd.log
# 2. PMC: HBM traffic of the dominant kernel, FETCH_SIZE and WRITE_SIZE in separate passes
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $R/bench.py --gpus 1 --steps 6 --warmup 2 --no-variants --no-extras --no-cpu-baseline > $OUT/pmc_fetch.json 2> $OUT/pmc_fetch.log
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $R/bench.py --gpus 1 --steps 6 --warmup 2 --no-variants --no-extras --no-cpu-baseline > $OUT/pmc_write.json 2> $OUT/pmc_write.log
python3 $R/profiles/summarize_r04.py $OUT > $OUT/summary.json
cat $OUT/summary.json | head -c 3000
