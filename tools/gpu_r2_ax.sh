#!/bin/bash
mkdir -p gpurun_out
for e in "X=1" "PFP_FORCE_IDX64=1" "PFP_DENSE_SA=1" "PFP_HARD_MODE=3" "PFP_POOL_DEBUG=1" "PFP_KEYSONLY=1"; do
  if env $e PFP_DEBUG=1 timeout -k 10 400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_distributed.py -m gpu -x -q > gpurun_out/env_$e.log 2>&1; then
    echo "$e: $(tail -n 1 gpurun_out/env_$e.log)"
  else
    echo "$e: FAILED"; tail -n 15 gpurun_out/env_$e.log
  fi
done
