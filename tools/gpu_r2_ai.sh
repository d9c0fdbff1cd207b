#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_stage_tools.py tests/test_abi_and_host.py -m gpu -q -x -k "driver or stage or file or c_driver" > gpurun_out/r2ai_tests.log 2>&1; echo "tests rc=$?"; tail -1 gpurun_out/r2ai_tests.log
timeout -k 10 400 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r2ai_c3.log 2>&1; echo rc=$?
python3 tools/benchsum.py gpurun_out/r2ai_c3.log | grep -E "^gpurun|host"  | cut -c1-700
