#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r2aw_tests.log 2>&1; echo "pytest rc=$?"; tail -1 gpurun_out/r2aw_tests.log
for R in 1 2; do
  timeout -k 10 400 python tools/simscale.py $R c3 > gpurun_out/r2z_sim_$R.log 2>&1; echo "sim $R rc=$?"
  grep -E "^R=|last rank" gpurun_out/r2z_sim_$R.log | cut -c1-470
done
export PFP_BENCH_BACKEND=gloo PFP_BENCH_ONE_GPU=1
timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 2 --warmup 1 > gpurun_out/r2aw_2ranks.log 2>&1; echo "2 ranks rc=$?"
grep '^{"metric"' gpurun_out/r2aw_2ranks.log | cut -c1-330
