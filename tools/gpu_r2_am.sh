#!/bin/bash
mkdir -p gpurun_out
export PFP_BENCH_BACKEND=gloo PFP_BENCH_ONE_GPU=1
timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 2 --warmup 1 > gpurun_out/r2am_2ranks.log 2>&1; echo "rc=$?"
grep '^{"metric"' gpurun_out/r2am_2ranks.log | cut -c1-900
tail -3 gpurun_out/r2am_2ranks.log | cut -c1-300
