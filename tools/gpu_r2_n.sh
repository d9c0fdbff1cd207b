#!/bin/bash
# rocprofv3 kernel trace of the 12.6 GB workloads (every kernel, incl. the library's), one chain each
mkdir -p gpurun_out
S=gpurun_out/r2n_summary.txt
rm -f $S
export TMPDIR=/tmp
D=$PWD/gpurun_out/r2n_prof
rm -rf $D; mkdir -p $D
for W in huge huge_s; do
  echo "=== rocprof kernel stats $W" | tee -a $S
  ( cd /tmp && timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $D/$W -o $W -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --workload $W --no-cpu-baseline --no-host-boundary > $D/$W.log 2>&1 ); rc=$?
  echo rc=$rc | tee -a $S
  if [ $rc -ge 124 ]; then exit $rc; fi
  F=$(find $D/$W -name "${W}_kernel_stats.csv" | head -1)
  python3 tools/kstats.py $F 60 > gpurun_out/r2n_${W}_kernel_stats.txt 2>&1; head -45 gpurun_out/r2n_${W}_kernel_stats.txt | tee -a $S
  find $D/$W -name "*.db" -delete; find $D/$W -name "*kernel_trace.csv" -delete
done
du -sh $D | tee -a $S
