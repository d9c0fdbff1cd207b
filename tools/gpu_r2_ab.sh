#!/bin/bash
mkdir -p gpurun_out
S=gpurun_out/r2ab_summary.txt
run() { local name=$1 to=$2; shift 2
  echo "=== $name" | tee -a $S
  timeout -k 10 $to "$@" > gpurun_out/$name.log 2>&1
  local rc=$?
  echo "rc=$rc" | tee -a $S
  tail -1 gpurun_out/$name.log | cut -c1-200 | tee -a $S
  if [ $rc -ge 124 ]; then echo "killed/timeout: stopping" | tee -a $S; exit $rc; fi
}
rm -f $S
run r2ab_tests 900 python -m pytest tests -m gpu -q -x
run r2ab_bench_c3 400 python bench.py --steps 5 --warmup 2 --no-cpu-baseline
run r2ab_bench_huge 500 python bench.py --steps 3 --warmup 1 --workload huge --no-cpu-baseline
run r2ab_bench_big 500 python bench.py --steps 3 --warmup 1 --workload big --no-cpu-baseline
for R in 1 2; do
  run r2ab_sim_$R 400 python tools/simscale.py $R c3
  grep -E "^R=|last rank" gpurun_out/r2ab_sim_$R.log | cut -c1-450 | tee -a $S
done
