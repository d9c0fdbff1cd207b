#!/bin/bash
mkdir -p gpurun_out
S=gpurun_out/r2u_summary.txt
run() { local name=$1 to=$2; shift 2
  echo "=== $name" | tee -a $S
  timeout -k 10 $to "$@" > gpurun_out/$name.log 2>&1
  local rc=$?
  echo "rc=$rc" | tee -a $S
  tail -1 gpurun_out/$name.log | cut -c1-200 | tee -a $S
  if [ $rc -ge 124 ]; then echo "killed/timeout: stopping" | tee -a $S; exit $rc; fi
}
rm -f $S
run r2u_bench_c2 400 python bench.py --steps 5 --warmup 2 --workload c2 --no-cpu-baseline
run r2u_bench_big 600 python bench.py --steps 3 --warmup 1 --workload big --no-cpu-baseline
run r2u_bench_c4s 400 python bench.py --steps 5 --warmup 2 --workload c4s --no-cpu-baseline
run r2u_bench_c5s 400 python bench.py --steps 5 --warmup 2 --workload c5s --no-cpu-baseline
