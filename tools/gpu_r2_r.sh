#!/bin/bash
mkdir -p gpurun_out
S=gpurun_out/r2r_summary.txt
run() { local name=$1 to=$2; shift 2
  echo "=== $name" | tee -a $S
  timeout -k 10 $to "$@" > gpurun_out/$name.log 2>&1
  local rc=$?
  echo "rc=$rc" | tee -a $S
  tail -2 gpurun_out/$name.log | cut -c1-300 | tee -a $S
  if [ $rc -ge 124 ]; then echo "killed/timeout: stopping" | tee -a $S; exit $rc; fi
}
rm -f $S
run r2r_bench_huge_s 600 python bench.py --steps 3 --warmup 1 --workload huge_s --no-cpu-baseline
run r2r_bench_huge 600 python bench.py --steps 3 --warmup 1 --workload huge --no-cpu-baseline
run r2r_bench_c3 400 python bench.py --steps 5 --warmup 2 --no-cpu-baseline
