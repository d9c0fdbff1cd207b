#!/bin/bash
mkdir -p gpurun_out
S=gpurun_out/r2an_summary.txt
run() { local name=$1 to=$2; shift 2
  echo "=== $name" | tee -a $S
  timeout -k 10 $to "$@" > gpurun_out/$name.log 2>&1
  local rc=$?
  echo "rc=$rc" | tee -a $S
  tail -1 gpurun_out/$name.log | cut -c1-200 | tee -a $S
  if [ $rc -ge 124 ]; then echo "killed/timeout: stopping" | tee -a $S; exit $rc; fi
}
rm -f $S
run r2an_tests_debug 900 env PFP_DEBUG=1 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_distributed.py -m gpu -q -x
run r2an_tests 900 python -m pytest tests -m gpu -q -x
PFP_TRACE_ROUNDS=1 timeout -k 10 300 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-host-boundary > gpurun_out/r2an_trace.log 2>&1
grep "doubling N=1288" gpurun_out/r2an_trace.log | awk '!s[$0]++' | cut -c1-150 | tee -a $S
run r2an_bench_c3 400 python bench.py --steps 5 --warmup 2 --no-cpu-baseline
run r2an_bench_huge 500 python bench.py --steps 3 --warmup 1 --workload huge --no-cpu-baseline
run r2an_sim_2 400 python tools/simscale.py 2 c3
grep -E "^R=|last rank" gpurun_out/r2an_sim_2.log | cut -c1-450 | tee -a $S
