#!/bin/bash
mkdir -p gpurun_out
for V in "X=1" "PFP_NO_FINISHER=1" "X=2" "PFP_NO_FINISHER=1"; do
for W in c3 c5s; do
  env $V timeout -k 10 300 python bench.py --steps 5 --warmup 2 --workload $W --no-cpu-baseline --no-host-boundary > gpurun_out/r2au_$W.log 2>&1
  echo "$V $W $(python3 tools/benchsum.py gpurun_out/r2au_$W.log | sed -n 1,1p | cut -c20-250)"
done
done
