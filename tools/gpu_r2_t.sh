#!/bin/bash
# rocprofv3 kernel trace of the headline workload: per-kernel stats + busy fraction of the last chain
mkdir -p gpurun_out
export TMPDIR=/tmp
D=$PWD/gpurun_out/r2t_prof
rm -rf $D; mkdir -p $D
( cd /tmp && timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $D/k -o c3 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-host-boundary > $D/k.log 2>&1 ); echo rc=$?
F=$(find $D/k -name "c3_kernel_stats.csv" | head -1)
python3 tools/kstats.py $F 60 > gpurun_out/r2t_c3_kernel_stats.txt 2>&1
T=$(find $D/k -name "c3_kernel_trace.csv" | head -1)
python3 tools/busy.py $T | tee gpurun_out/r2t_c3_busy.txt
find $D -name "*.db" -delete; find $D -name "*kernel_trace.csv" -delete
tail -1 $D/k.log | cut -c1-200
