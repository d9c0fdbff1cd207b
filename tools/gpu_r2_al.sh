#!/bin/bash
mkdir -p gpurun_out
S=gpurun_out/r2al_summary.txt
run() { local name=$1 to=$2; shift 2
  echo "=== $name" | tee -a $S
  timeout -k 10 $to "$@" > gpurun_out/$name.log 2>&1
  local rc=$?
  echo "rc=$rc" | tee -a $S
  tail -1 gpurun_out/$name.log | cut -c1-200 | tee -a $S
  if [ $rc -ge 124 ]; then echo "killed/timeout: stopping" | tee -a $S; exit $rc; fi
}
rm -f $S
run r2al_tests_debug 900 env PFP_DEBUG=1 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_distributed.py -m gpu -q -x
run r2al_tests 900 python -m pytest tests -m gpu -q -x
run r2al_bench_c3 400 python bench.py --steps 5 --warmup 2 --no-cpu-baseline
for V in "X=1" "PFP_NO_MIDSEG=1"; do
  env $V timeout -k 10 400 python bench.py --steps 3 --warmup 1 --workload huge --no-cpu-baseline --no-host-boundary > gpurun_out/r2al_huge.log 2>&1
  echo "rc=$? $V" | tee -a $S
  python3 tools/benchsum.py gpurun_out/r2al_huge.log | sed -n 1,1p | cut -c1-300 | tee -a $S
  python3 tools/benchsum.py gpurun_out/r2al_huge.log | grep -E "seg_small|seg_mid|segmented|radix_sort_pairs<u64,u32>" | tee -a $S
done
for R in 2; do
  run r2al_sim_$R 400 python tools/simscale.py $R c3
  grep -E "^R=|last rank" gpurun_out/r2al_sim_$R.log | cut -c1-450 | tee -a $S
done
