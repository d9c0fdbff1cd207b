#!/bin/bash
# fewer, larger fuzz texts (up to 6 MB): the paths that only start at scale (segmented sorts from 2^20 elements, keys-only first round, side sorts)
mkdir -p gpurun_out
S=gpurun_out/r2fuzz3_summary.txt
rm -f $S
i=0
for V in "X=1" "PFP_KEYSONLY=1" "PFP_DEBUG=1" "PFP_NO_SMALLSEG=1" "PFP_HARD_MODE=2" "PFP_FORCE_IDX64=1" "PFP_NO_BIGSIDE=1 PFP_SEG_MINAVG=2"; do
  i=$((i+1))
  env $V FUZZ_MAXN=6000000 timeout -k 10 150 python tools/fuzz.py $((500+i)) 400 > gpurun_out/r2fuzz3_$i.log 2>&1; rc=$?
  echo "$V fuzz(<=6MB) rc=$rc: $(grep -c . gpurun_out/r2fuzz3_$i.log) lines; $(grep -a MISMATCH gpurun_out/r2fuzz3_$i.log | head -3 | cut -c1-200); last: $(tail -1 gpurun_out/r2fuzz3_$i.log | cut -c1-120)" | tee -a $S
done
