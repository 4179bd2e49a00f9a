#!/bin/bash
# Round-4 evidence run (on the GPU box): bash profiles/tools/r4_final.sh <tag>      e.g. r4a
#   headline: profiles/collect_r04.sh (the driver's command plain and under rocprofv3 --kernel-trace, PMC passes)
#   default bench line, the driver's command line, 2 and 8 ranks on the one GPU started by bench.py ITSELF (--gpus N, no launcher)
#   GPU suite twice, smoke, PMC of the kernels that changed this round
cd $GRAFT_REPO_ROOT
TAG=${1:-r4a}
export PYTHONFAULTHANDLER=1
O=gpurun_out
bash profiles/collect_r04.sh $TAG > $O/${TAG}_collect.log 2>&1; echo "collect rc=$?"
t0=$(date +%s)
timeout 900 python bench.py > $O/${TAG}_bench_n1.json 2> $O/${TAG}_bench_n1.err; echo "n1 rc=$? wall=$(( $(date +%s) - t0 ))s"
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/${TAG}_bench_driver_cmd.json 2> /dev/null; echo "driver cmd rc=$?"
SPECKV_BENCH_SINGLE_GPU_TEST=1 timeout 600 python bench.py --gpus 2 --steps 10 --warmup 2 > $O/${TAG}_bench_n2fake.json 2> $O/${TAG}_bench_n2fake.err; echo "n2fake (self-spawned) rc=$?"
SPECKV_BENCH_SINGLE_GPU_TEST=1 SPECKV_BENCH_WATCHDOG_S=900 SPECKV_XGMI_TIMEOUT_S=600 timeout 1200 python bench.py --gpus 8 --steps 10 --warmup 2 > $O/${TAG}_bench_n8fake.json 2> $O/${TAG}_bench_n8fake.err; echo "n8fake (self-spawned) rc=$?"
for i in 1 2; do timeout 900 python -m pytest tests -x -q -m gpu -p no:cacheprovider > $O/soak_${TAG}_$i.log 2>&1; echo "soak $i rc=$?"; tail -1 $O/soak_${TAG}_$i.log; done
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
# PMC: the whole-record INT4 kernel (SQ counters, memory side), the tensor codec's kernels at 2.5 GiB (FETCH_SIZE / WRITE_SIZE)
bash profiles/tools/pmc_kern.sh k_attend_int4_wg8 pmc_${TAG}_int4_wg8 -- profiles/tools/int4_bench.py 32768 80 > /dev/null 2>&1; echo "pmc int4 sq rc=$?"
bash profiles/tools/pmc_mem.sh k_attend_int4_wg8 pmcmem_${TAG}_int4_wg8 -- profiles/tools/int4_bench.py 32768 80 > /dev/null 2>&1; echo "pmc int4 mem rc=$?"
export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 600 rocprofv3 --pmc $c --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc_${TAG}_tc_$c -- python3 $GRAFT_REPO_ROOT/profiles/tools/tc_bench.py 1342177280 > $GRAFT_REPO_ROOT/$O/pmc_${TAG}_tc_$c.log 2>&1); echo "pmc tensor codec $c rc=$?"
done
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/trace_${TAG}_tc -- python3 $GRAFT_REPO_ROOT/profiles/tools/tc_bench.py 1342177280 > $GRAFT_REPO_ROOT/$O/trace_${TAG}_tc.log 2>&1); echo "trace tensor codec rc=$?"
timeout 300 python profiles/tools/conn_step.py 2>&1 | grep -v amdgpu.ids > $O/${TAG}_conn_step.txt
echo done
