#!/bin/bash
mkdir -p gpurun_out
for KO in 0 1; do
  PFP_KEYSONLY=$KO timeout -k 10 300 python bench.py --steps 5 --warmup 2 --workload c2 --no-cpu-baseline --no-host-boundary > gpurun_out/r2y_c2_$KO.log 2>&1
  echo "rc=$? keysonly=$KO"
  python3 tools/benchsum.py gpurun_out/r2y_c2_$KO.log | sed -n 1,9p
done
