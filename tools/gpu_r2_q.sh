#!/bin/bash
mkdir -p gpurun_out
PFP_TRACE_ROUNDS=1 timeout -k 10 300 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-host-boundary > gpurun_out/r2q_trace_c3.log 2>&1
echo rc=$?
grep "doubling" gpurun_out/r2q_trace_c3.log | head -40
