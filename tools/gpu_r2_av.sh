#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_distributed.py tests/test_gpu_fuzz.py -m gpu -q -x > gpurun_out/r2av_tests.log 2>&1; echo "dist tests rc=$?"; tail -3 gpurun_out/r2av_tests.log | cut -c1-300
PFP_DEBUG=1 timeout -k 10 600 python -m pytest tests/test_distributed.py -m gpu -q -x > gpurun_out/r2av_tests_d.log 2>&1; echo "dist tests debug rc=$?"; tail -1 gpurun_out/r2av_tests_d.log | cut -c1-300
for R in 2; do
  timeout -k 10 400 python tools/simscale.py $R c3 > gpurun_out/r2av_sim_$R.log 2>&1; echo "sim rc=$?"
  grep -E "^R=|last rank" gpurun_out/r2av_sim_$R.log | cut -c1-450
done
