#!/bin/bash
# Round-6 evidence run (on the GPU box): bash profiles/tools/r6_final.sh <tag>      e.g. r6a
#   headline: profiles/collect_r04.sh (the driver's command plain and under rocprofv3 --kernel-trace, PMC passes; same recipe since round 4)
#   default bench line, the driver's command line, 2 and 8 ranks on the one GPU started by bench.py ITSELF (--gpus N, no launcher)
#   GPU suite, smoke; PMC of k_attend_mx4 (tile-planar records: SQ counters, memory side) and of the batched tensor codec;
#   MXFP4 attention table, striped pools, connector step by format, access miss, predictor trace
cd $GRAFT_REPO_ROOT
TAG=${1:-r6a}
export PYTHONFAULTHANDLER=1
O=gpurun_out
bash profiles/collect_r04.sh $TAG > $O/${TAG}_collect.log 2>&1; echo "collect rc=$?"
t0=$(date +%s)
timeout 1200 python bench.py > $O/${TAG}_bench_n1.json 2> $O/${TAG}_bench_n1.err; echo "n1 rc=$? wall=$(( $(date +%s) - t0 ))s"
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/${TAG}_bench_driver_cmd.json 2> /dev/null; echo "driver cmd rc=$?"
SPECKV_BENCH_SINGLE_GPU_TEST=1 timeout 600 python bench.py --gpus 2 --steps 10 --warmup 2 > $O/${TAG}_bench_n2fake.json 2> $O/${TAG}_bench_n2fake.err; echo "n2fake (self-spawned) rc=$?"
SPECKV_BENCH_SINGLE_GPU_TEST=1 SPECKV_BENCH_WATCHDOG_S=900 SPECKV_XGMI_TIMEOUT_S=600 timeout 1200 python bench.py --gpus 8 --steps 10 --warmup 2 > $O/${TAG}_bench_n8fake.json 2> $O/${TAG}_bench_n8fake.err; echo "n8fake (self-spawned) rc=$?"
timeout 1500 python -m pytest tests -x -q -m gpu -p no:cacheprovider > $O/soak_${TAG}.log 2>&1; echo "suite rc=$?"; tail -2 $O/soak_${TAG}.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
bash profiles/tools/pmc_kern.sh k_attend_mx4 pmc_${TAG}_mx4 -- profiles/tools/mx4_bench.py single > /dev/null 2>&1; echo "pmc mx4 sq rc=$?"
bash profiles/tools/pmc_mem.sh k_attend_mx4 pmcmem_${TAG}_mx4 -- profiles/tools/mx4_bench.py single > /dev/null 2>&1; echo "pmc mx4 mem rc=$?"
for k in tcm tdm; do
  bash profiles/tools/pmc_kern.sh "k_${k}_fused<0, true>" pmc_${TAG}_${k} -- profiles/tools/tcb_bench.py > /dev/null 2>&1; echo "pmc $k sq rc=$?"
  bash profiles/tools/pmc_mem.sh "k_${k}_fused<0, true>" pmcmem_${TAG}_${k} -- profiles/tools/tcb_bench.py > /dev/null 2>&1; echo "pmc $k mem rc=$?"
done
timeout 300 python profiles/tools/tcb_bench.py 2>&1 | grep -v amdgpu.ids > $O/${TAG}_tcb_bench.json
timeout 300 python profiles/tools/mx4_bench.py both 2>&1 | grep -v amdgpu.ids > $O/${TAG}_mx4_bench.txt
for c in "256 8192" "256 4096" "256 2048" "256 1024"; do timeout 200 python profiles/tools/mx4_bench.py batch $c 2>&1 | grep -v amdgpu.ids >> $O/${TAG}_mx4_bench.txt; done
timeout 600 python profiles/tools/striped_extra.py 2>&1 | grep -v amdgpu.ids > $O/${TAG}_striped.txt
timeout 300 python profiles/tools/conn_step_schemes.py 2>&1 | grep -v amdgpu.ids > $O/${TAG}_conn_step_schemes.txt
timeout 300 python profiles/tools/conn_step.py mxfp4 2>&1 | grep -v amdgpu.ids > $O/${TAG}_conn_step.txt
timeout 300 python profiles/tools/access_miss.py 2050 2>&1 | grep -v amdgpu.ids > $O/${TAG}_access_miss.txt
timeout 300 python profiles/tools/kv_accuracy.py 2>&1 | grep -v amdgpu.ids > $O/${TAG}_kv_accuracy.txt
# the predictor's two launches, per dispatch (is the single prediction's first launch itself >= 10 us?)
export TMPDIR=/tmp; ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/pred_${TAG} -- python3 $GRAFT_REPO_ROOT/profiles/tools/pred_trace.py > $GRAFT_REPO_ROOT/$O/${TAG}_pred_trace.log 2>&1 )
python3 profiles/tools/pred_trace_summary.py $(find $O/pred_${TAG} -name "*kernel_trace.csv" | head -1) > $O/${TAG}_pred_trace.txt 2>&1; tail -3 $O/${TAG}_pred_trace.log >> $O/${TAG}_pred_trace.txt
echo done
