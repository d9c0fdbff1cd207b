#!/bin/bash
mkdir -p gpurun_out
S=gpurun_out/r2s_summary.txt
run() { local name=$1 to=$2; shift 2
  echo "=== $name" | tee -a $S
  timeout -k 10 $to "$@" > gpurun_out/$name.log 2>&1
  local rc=$?
  echo "rc=$rc" | tee -a $S
  tail -2 gpurun_out/$name.log | cut -c1-300 | tee -a $S
  if [ $rc -ge 124 ]; then echo "killed/timeout: stopping" | tee -a $S; exit $rc; fi
}
rm -f $S
run r2s_test_north_star 900 python -m pytest tests/test_gpu_fullsize.py -m gpu -q -x
run r2s_bench_huge_s 600 python bench.py --steps 3 --warmup 1 --workload huge_s --no-cpu-baseline
