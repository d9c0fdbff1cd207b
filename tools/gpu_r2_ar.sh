#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 900 python tools/check_wide.py --workload wide --flags 0 > gpurun_out/r2ar_wide.log 2>&1; echo "rc=$?"
tail -1 gpurun_out/r2ar_wide.log | cut -c1-1500
