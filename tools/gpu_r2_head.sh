#!/bin/bash
# what the driver runs at round end: GPU tests, smoke(), the default bench
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/head_tests.log 2>&1; echo "pytest rc=$?"; tail -1 gpurun_out/head_tests.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/head_smoke.log 2>&1; echo "smoke rc=$?"; tail -1 gpurun_out/head_smoke.log
timeout -k 10 600 python bench.py > gpurun_out/head_bench.log 2>&1; echo "bench rc=$?"; grep '^{"metric"' gpurun_out/head_bench.log | cut -c1-400
