#!/bin/bash
mkdir -p gpurun_out
PFP_TRACE_HOST=1 timeout -k 10 400 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r2aj_c3.log 2>&1; echo rc=$?
grep -a "file to files\|host boundary" gpurun_out/r2aj_c3.log | tail -3
python3 tools/benchsum.py gpurun_out/r2aj_c3.log | grep -E "host"  | cut -c1-700
