#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -q -x > gpurun_out/r2ao_tests.log 2>&1; echo "tests rc=$?"; tail -1 gpurun_out/r2ao_tests.log
timeout -k 10 400 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-host-boundary > gpurun_out/r2ao_c3.log 2>&1
python3 tools/benchsum.py gpurun_out/r2ao_c3.log | grep -E "^gpurun|expand" | cut -c1-300
