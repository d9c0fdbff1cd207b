#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q -x > gpurun_out/r2ah_tests.log 2>&1; echo "tests rc=$?"; tail -1 gpurun_out/r2ah_tests.log
PFP_TRACE_HOST=1 timeout -k 10 400 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r2ah_c3.log 2>&1; echo rc=$?
grep "host boundary" gpurun_out/r2ah_c3.log | tail -1
python3 tools/benchsum.py gpurun_out/r2ah_c3.log | grep -E "^gpurun|host"  | cut -c1-600
