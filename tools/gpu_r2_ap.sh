#!/bin/bash
mkdir -p gpurun_out
for V in "X=1" "PFP_KEYBITS=39" "PFP_KEYBITS=43"; do
  env $V timeout -k 10 300 python bench.py --steps 5 --warmup 2 --workload c2 --no-cpu-baseline --no-host-boundary > gpurun_out/r2ap_c2.log 2>&1
  echo "rc=$? $V"
  python3 tools/benchsum.py gpurun_out/r2ap_c2.log | sed -n 1,4p | cut -c1-260
done
