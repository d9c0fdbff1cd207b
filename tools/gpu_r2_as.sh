#!/bin/bash
mkdir -p gpurun_out
PFP_TRACE_ROUNDS=1 timeout -k 10 300 python bench.py --steps 3 --warmup 1 --workload c5s --no-cpu-baseline --no-host-boundary > gpurun_out/r2as_c5s.log 2>&1; echo rc=$?
grep "doubling" gpurun_out/r2as_c5s.log | awk '!s[$0]++' | head -30 | cut -c1-150
python3 tools/benchsum.py gpurun_out/r2as_c5s.log | sed -n 1,16p | cut -c1-250
