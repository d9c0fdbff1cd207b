#!/bin/bash
# round-2 first GPU call: parity suite under the checking pool, then plain, then the bench lines
mkdir -p gpurun_out
run() { # name, timeout, cmd...
  local name=$1 to=$2; shift 2
  echo "=== $name" | tee -a gpurun_out/r2a_summary.txt
  timeout -k 10 $to "$@" > gpurun_out/$name.log 2>&1
  local rc=$?
  echo "rc=$rc" | tee -a gpurun_out/r2a_summary.txt
  tail -5 gpurun_out/$name.log | tee -a gpurun_out/r2a_summary.txt
  if [ $rc -ge 124 ]; then echo "killed/timeout: stopping" | tee -a gpurun_out/r2a_summary.txt; exit $rc; fi
}
rm -f gpurun_out/r2a_summary.txt
export PFP_POOL_DEBUG=1
run r2a_pooldebug_tests 1000 python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_fullsize.py
unset PFP_POOL_DEBUG
run r2a_tests 900 python -m pytest tests -m gpu -q
run r2a_bench_c3 600 python bench.py --steps 5 --warmup 2
run r2a_bench_c2 600 python bench.py --steps 5 --warmup 2 --workload c2
