#!/bin/bash
mkdir -p gpurun_out
for KO in 0 1; do
  PFP_KEYSONLY=$KO PFP_TRACE_ROUNDS=1 timeout -k 10 400 python bench.py --steps 1 --warmup 0 --workload huge --no-cpu-baseline --no-host-boundary > gpurun_out/r2x_trace_huge_$KO.log 2>&1
  echo "rc=$? keysonly=$KO"
  grep "doubling N=17" gpurun_out/r2x_trace_huge_$KO.log | awk '!s[$0]++' | head -14
  python3 tools/benchsum.py gpurun_out/r2x_trace_huge_$KO.log | sed -n 1,12p
done
