#!/bin/bash
mkdir -p gpurun_out
S=gpurun_out/r2at_summary.txt
rm -f $S
timeout -k 10 900 env PFP_DEBUG=1 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_distributed.py -m gpu -q -x > gpurun_out/r2at_tests_debug.log 2>&1; echo "debug tests rc=$?" | tee -a $S; tail -1 gpurun_out/r2at_tests_debug.log | tee -a $S
timeout -k 10 900 python -m pytest tests -m gpu -q -x > gpurun_out/r2at_tests.log 2>&1; echo "tests rc=$?" | tee -a $S; tail -1 gpurun_out/r2at_tests.log | tee -a $S
for W in c5s c3 c2 c4s; do
  PFP_TRACE_ROUNDS=1 timeout -k 10 300 python bench.py --steps 5 --warmup 2 --workload $W --no-cpu-baseline --no-host-boundary > gpurun_out/r2at_$W.log 2>&1; echo "rc=$? $W" | tee -a $S
  python3 tools/benchsum.py gpurun_out/r2at_$W.log | sed -n 1,1p | cut -c1-250 | tee -a $S
done
grep "doubling N=6336" gpurun_out/r2at_c5s.log | awk '!s[$0]++' | head -16 | cut -c1-150 | tee -a $S
timeout -k 10 500 python bench.py --steps 3 --warmup 1 --workload huge --no-cpu-baseline --no-host-boundary > gpurun_out/r2at_huge.log 2>&1; echo "rc=$? huge" | tee -a $S
python3 tools/benchsum.py gpurun_out/r2at_huge.log | sed -n 1,1p | cut -c1-250 | tee -a $S
