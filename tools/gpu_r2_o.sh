#!/bin/bash
mkdir -p gpurun_out
S=gpurun_out/r2o_summary.txt
run() { local name=$1 to=$2; shift 2
  echo "=== $name" | tee -a $S
  timeout -k 10 $to "$@" > gpurun_out/$name.log 2>&1
  local rc=$?
  echo "rc=$rc" | tee -a $S
  tail -4 gpurun_out/$name.log | cut -c1-400 | tee -a $S
  if [ $rc -ge 124 ]; then echo "killed/timeout: stopping" | tee -a $S; exit $rc; fi
}
rm -f $S
run r2o_tests_big 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "large_hard_groups or mid_size or golden"
run r2o_bench_huge 600 python bench.py --steps 2 --warmup 1 --workload huge --no-cpu-baseline
run r2o_bench_huge_s 600 python bench.py --steps 2 --warmup 1 --workload huge_s --no-cpu-baseline
