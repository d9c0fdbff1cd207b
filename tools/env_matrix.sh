#!/bin/bash
# usage: tools/env_matrix.sh [first [count]]  - the switches first .. first+count-1 of the list (default: all)
fail=0
first=${1:-0}; count=${2:-1000}; idx=-1
for e in "X=1" "PFP_NO_FINFLAG=1" "PFP_NO_PAYLOAD=1" "PFP_KEYBITS=63" "PFP_KEYBITS=23" "PFP_SEGSORT=0" "PFP_LAZY_RATIO=0" "PFP_LAZY_RATIO=1000000" "PFP_PIVOT_CAP=0" "PFP_PIVOT_CAP=16" "PFP_NO_RUNKEYS=1" "PFP_HARD_MODE=1" "PFP_HARD_MODE=2" "PFP_HARD_MODE=3" "PFP_EAGER_PIVOT_RANKS=1" "PFP_SEG_MINAVG=2" "PFP_NO_SMALLSEG=1" "PFP_DENSE_SA=1" "PFP_FORCE_IDX64=1" "PFP_POOL_DEBUG=1" "PFP_BIG_BUDGET=5000" "PFP_BIG_BY_RANK=1" "PFP_KEYSONLY=1" "PFP_NO_BIGSIDE=1" "PFP_NO_FINISHER=1"; do
  idx=$((idx+1)); if [ $idx -lt $first ] || [ $idx -ge $((first+count)) ]; then continue; fi
  if env $e PFP_DEBUG=1 timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_distributed.py -m gpu -x -q > gpurun_out/env_$e.log 2>&1; then
    echo "$e: $(tail -n 1 gpurun_out/env_$e.log)"
  else
    echo "$e: FAILED"; tail -n 15 gpurun_out/env_$e.log; fail=1
  fi
done
exit $fail
