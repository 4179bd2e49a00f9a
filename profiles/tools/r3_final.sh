#!/bin/bash
# Round-3 evidence run (on the GPU box): bash profiles/tools/r3_final.sh <tag>      e.g. r3a
cd $GRAFT_REPO_ROOT
TAG=${1:-r3a}
export PYTHONFAULTHANDLER=1
t0=$(date +%s)
timeout 900 python bench.py > gpurun_out/${TAG}_bench_n1.json 2> gpurun_out/${TAG}_bench_n1.err
echo "n1 rc=$? wall=$(( $(date +%s) - t0 ))s"; tail -2 gpurun_out/${TAG}_bench_n1.err
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_driver_cmd.json 2> /dev/null
echo "driver cmd rc=$?"
SPECKV_BENCH_SINGLE_GPU_TEST=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 10 --warmup 2 > gpurun_out/${TAG}_bench_n2fake.json 2> gpurun_out/${TAG}_bench_n2fake.err
echo "n2fake rc=$?"
SPECKV_BENCH_SINGLE_GPU_TEST=1 SPECKV_BENCH_WATCHDOG_S=900 SPECKV_XGMI_TIMEOUT_S=600 timeout 1200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 8 --steps 10 --warmup 2 > gpurun_out/${TAG}_bench_n8fake.json 2> gpurun_out/${TAG}_bench_n8fake.err
echo "n8fake rc=$?"
for i in 1 2; do timeout 900 python -m pytest tests -x -q -m gpu -p no:cacheprovider > gpurun_out/soak_${TAG}_$i.log 2>&1; echo "soak $i rc=$?"; tail -1 gpurun_out/soak_${TAG}_$i.log; done
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 300 python profiles/tools/conn_step.py 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_conn_step.txt
timeout 600 python profiles/tools/striped_bench.py 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_striped.txt
bash profiles/collect_r02.sh prof_${TAG} > /dev/null 2>&1
echo "collect rc=$?"
