#!/bin/bash
mkdir -p gpurun_out
S=gpurun_out/r2w_summary.txt
run() { local name=$1 to=$2; shift 2
  echo "=== $name" | tee -a $S
  timeout -k 10 $to "$@" > gpurun_out/$name.log 2>&1
  local rc=$?
  echo "rc=$rc" | tee -a $S
  tail -2 gpurun_out/$name.log | cut -c1-250 | tee -a $S
  if [ $rc -ge 124 ]; then echo "killed/timeout: stopping" | tee -a $S; exit $rc; fi
}
rm -f $S
run r2w_tests_keysonly 900 env PFP_KEYSONLY=1 PFP_DEBUG=1 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_distributed.py -m gpu -q -x
run r2w_tests 900 python -m pytest tests -m gpu -q -x
run r2w_bench_c3 400 python bench.py --steps 5 --warmup 2 --no-cpu-baseline
run r2w_bench_huge_s 600 python bench.py --steps 3 --warmup 1 --workload huge_s --no-cpu-baseline
run r2w_bench_c2 400 python bench.py --steps 5 --warmup 2 --workload c2 --no-cpu-baseline
