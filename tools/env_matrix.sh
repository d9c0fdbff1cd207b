#!/bin/bash
# usage: tools/env_matrix.sh [first [count]]  - (18 variants x ~110 s: more than one 20-minute gpurun call, run it as "0 7" and "7 6") the switches first .. first+count-1 of the list (default: all)
fail=0
first=${1:-0}; count=${2:-1000}; idx=-1
for e in "X=1" "PFP_NO_FINFLAG=1" "PFP_KEYBITS=63" "PFP_KEYBITS=23" "PFP_PIVOT_CAP=0" "PFP_PIVOT_CAP=16" "PFP_NO_SMALLSEG=1" "PFP_FORCE_IDX64=1" "PFP_POOL_DEBUG=1" "PFP_BIG_BUDGET=5000" "PFP_KEYSONLY=1" "PFP_NO_FINISHER=1" "PFP_PREC_DIRECT=1" "PFP_PARSE_PIVOT_MIN=64" "PFP_OWN_SORT=1" "PFP_WINDOW_HASH=kr" "PFP_PARSE_DENSITY=1" "PFP_PARSE_DENSITY=3"; do
  idx=$((idx+1)); if [ $idx -lt $first ] || [ $idx -ge $((first+count)) ]; then continue; fi
  if env $e PFP_DEBUG=1 timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_distributed.py -m gpu -x -q > gpurun_out/env_$e.log 2>&1; then
    echo "$e: $(tail -n 1 gpurun_out/env_$e.log)"
  else
    echo "$e: FAILED"; tail -n 15 gpurun_out/env_$e.log; fail=1
  fi
done
exit $fail
