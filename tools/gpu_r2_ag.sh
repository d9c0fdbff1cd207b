#!/bin/bash
mkdir -p gpurun_out
PFP_TRACE_HOST=1 timeout -k 10 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r2ag.log 2>&1; echo rc=$?
grep "host boundary" gpurun_out/r2ag.log | head -5
nproc; grep -i hugepage /sys/kernel/mm/transparent_hugepage/enabled 2>/dev/null; cat /sys/kernel/mm/transparent_hugepage/enabled
