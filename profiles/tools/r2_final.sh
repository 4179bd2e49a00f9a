#!/bin/bash
cd $GRAFT_REPO_ROOT
export PYTHONFAULTHANDLER=1
t0=$(date +%s)
timeout 900 python bench.py > gpurun_out/r2h_bench_n1.json 2> gpurun_out/r2h_bench_n1.err
echo "n1 rc=$? wall=$(( $(date +%s) - t0 ))s"; tail -2 gpurun_out/r2h_bench_n1.err
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r2h_bench_driver_cmd.json 2> /dev/null
echo "driver cmd rc=$?"
SPECKV_BENCH_SINGLE_GPU_TEST=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 10 --warmup 2 > gpurun_out/r2h_bench_n2fake.json 2> gpurun_out/r2h_bench_n2fake.err
echo "n2fake rc=$?"
for i in 1 2 3; do timeout 600 python -m pytest tests -x -q -m gpu -p no:cacheprovider > gpurun_out/soak_r2h_$i.log 2>&1; echo "soak $i rc=$?"; tail -1 gpurun_out/soak_r2h_$i.log; done
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 300 python profiles/tools/conn_step.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r2h_conn_step.txt
timeout 600 python profiles/tools/batch_rule_ab.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r2h_batch_rule_ab.txt
bash profiles/collect_r02.sh prof_r02h > /dev/null 2>&1
echo "collect rc=$?"
