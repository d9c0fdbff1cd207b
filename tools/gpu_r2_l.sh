#!/bin/bash
mkdir -p gpurun_out
S=gpurun_out/r2l_summary.txt
rm -f $S
for R in 1 2 4 8; do
  echo "=== simscale R=$R" | tee -a $S
  timeout -k 10 500 python tools/simscale.py $R c3 > gpurun_out/r2l_sim_$R.log 2>&1; rc=$?
  echo rc=$rc | tee -a $S
  head -4 gpurun_out/r2l_sim_$R.log | cut -c1-700 | tee -a $S
  if [ $rc -ge 124 ]; then exit $rc; fi
done
echo "=== simscale R=4 allgather dedup" | tee -a $S
timeout -k 10 500 python tools/simscale.py 4 c3 allgather > gpurun_out/r2l_sim_4ag.log 2>&1; echo rc=$? | tee -a $S
head -3 gpurun_out/r2l_sim_4ag.log | cut -c1-700 | tee -a $S
