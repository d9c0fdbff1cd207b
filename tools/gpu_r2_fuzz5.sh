#!/bin/bash
mkdir -p gpurun_out
S=gpurun_out/r2fuzz5_summary.txt
rm -f $S
i=0
for V in "X=1" "PFP_DEBUG=1" "PFP_FORCE_IDX64=1" "PFP_KEYSONLY=1"; do
  i=$((i+1))
  env $V FUZZ_MAXN=150000 timeout -k 10 90 python tools/fuzz.py $((800+i)) 1500 > gpurun_out/r2fuzz5_$i.log 2>&1; rc=$?
  echo "$V fuzz (dist incl. -s -e) rc=$rc: $(grep -a MISMATCH gpurun_out/r2fuzz5_$i.log | head -2 | cut -c1-200) last: $(tail -1 gpurun_out/r2fuzz5_$i.log | cut -c1-100)" | tee -a $S
done
