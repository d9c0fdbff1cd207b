#!/bin/bash
mkdir -p gpurun_out
S=gpurun_out/r2h_summary.txt
run() { local name=$1 to=$2; shift 2
  echo "=== $name" | tee -a $S
  timeout -k 10 $to "$@" > gpurun_out/$name.log 2>&1
  local rc=$?
  echo "rc=$rc" | tee -a $S
  tail -8 gpurun_out/$name.log | cut -c1-1500 | tee -a $S
  if [ $rc -ge 124 ]; then echo "killed/timeout: stopping" | tee -a $S; exit $rc; fi
}
rm -f $S
run r2h_tests 1100 python -m pytest tests -m gpu -q -x
run r2h_bench_c3 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline
export PFP_TRACE_ROUNDS=1
run r2h_wide31 600 python tools/check_wide.py --workload wide31
run r2h_wide 900 python tools/check_wide.py --workload wide
