#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_distributed.py tests/test_gpu_fuzz.py -m gpu -q -x > gpurun_out/r2aa_tests.log 2>&1; echo "tests rc=$?"; tail -2 gpurun_out/r2aa_tests.log
PFP_KEYSONLY=1 PFP_DEBUG=1 timeout -k 10 600 python -m pytest tests/test_distributed.py -m gpu -q -x > gpurun_out/r2aa_tests_ko.log 2>&1; echo "tests keysonly rc=$?"; tail -2 gpurun_out/r2aa_tests_ko.log
for R in 1 2; do
  PFP_TRACE_ROUNDS=1 timeout -k 10 400 python tools/simscale.py $R c3 > gpurun_out/r2aa_sim_$R.log 2>&1; echo "sim $R rc=$?"
  grep "doubling N=" gpurun_out/r2aa_sim_$R.log | grep -v "N=78\|N=157" | awk '!s[$0]++' | tail -14 | cut -c1-160
  grep -E "^R=|last rank|traced total|radix|pivot" gpurun_out/r2aa_sim_$R.log | cut -c1-420
done
