#!/bin/bash
mkdir -p gpurun_out
for V in "X=1" "PFP_HARD_MODE=1" "PFP_HARD_MODE=2" "PFP_HARD_MODE=3" "PFP_DENSE_SA=1"; do
  env $V PFP_DEBUG=1 timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -q -x > gpurun_out/r2aq_tests.log 2>&1; echo "tests $V rc=$?"; tail -1 gpurun_out/r2aq_tests.log
done
timeout -k 10 600 python -m pytest tests/test_gpu_fullsize.py tests/test_distributed.py -m gpu -q -x > gpurun_out/r2aq_tests2.log 2>&1; echo "tests2 rc=$?"; tail -1 gpurun_out/r2aq_tests2.log
for W in c4s c3; do
timeout -k 10 400 python bench.py --steps 5 --warmup 2 --workload $W --no-cpu-baseline --no-host-boundary > gpurun_out/r2aq_$W.log 2>&1
python3 tools/benchsum.py gpurun_out/r2aq_$W.log | grep -E "^gpurun|hard_groups" | cut -c1-300
done
