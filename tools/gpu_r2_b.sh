#!/bin/bash
mkdir -p gpurun_out
run() { local name=$1 to=$2; shift 2
  echo "=== $name" | tee -a gpurun_out/r2b_summary.txt
  timeout -k 10 $to "$@" > gpurun_out/$name.log 2>&1
  local rc=$?
  echo "rc=$rc" | tee -a gpurun_out/r2b_summary.txt
  tail -15 gpurun_out/$name.log | cut -c1-400 | tee -a gpurun_out/r2b_summary.txt
  if [ $rc -ge 124 ]; then echo "killed/timeout: stopping" | tee -a gpurun_out/r2b_summary.txt; exit $rc; fi
}
rm -f gpurun_out/r2b_summary.txt
run r2b_tests 1100 python -m pytest tests -m gpu -q -x
