#!/bin/bash
# the fuzz drivers under the remaining switches (small texts, many trials)
mkdir -p gpurun_out
S=gpurun_out/r2fuzz4_summary.txt
rm -f $S
i=0
for V in "PFP_NO_BIGSIDE=1" "PFP_HARD_MODE=1" "PFP_DENSE_SA=1" "PFP_BIG_BUDGET=100" "PFP_BIG_BY_RANK=1" "PFP_NO_SMALLSEG=1" "PFP_SEGSORT=0" "PFP_PIVOT_CAP=16" "PFP_PIVOT_CAP=0" "PFP_NO_PAYLOAD=1" "PFP_NO_FINFLAG=1" "PFP_LAZY_RATIO=0" "PFP_EAGER_PIVOT_RANKS=1" "PFP_KEYBITS=23 PFP_KEYSONLY=1" "PFP_KEYBITS=63 PFP_KEYSONLY=1" "PFP_KEYSONLY=1 PFP_NO_FINISHER=1 PFP_NO_SMALLSEG=1"; do
  i=$((i+1))
  env $V FUZZ_MAXN=150000 timeout -k 10 60 python tools/fuzz.py $((600+i)) 1000 > gpurun_out/r2fuzz4_$i.log 2>&1; rc=$?
  echo "$V fuzz rc=$rc: $(grep -a MISMATCH gpurun_out/r2fuzz4_$i.log | head -2 | cut -c1-160) last: $(tail -1 gpurun_out/r2fuzz4_$i.log | cut -c1-100)" | tee -a $S
  env $V timeout -k 10 40 python tools/fuzz_sa.py $((700+i)) 800 > gpurun_out/r2fuzz4sa_$i.log 2>&1; rc=$?
  echo "$V fuzz_sa rc=$rc: $(grep -a mismatch gpurun_out/r2fuzz4sa_$i.log | head -2 | cut -c1-160) last: $(tail -1 gpurun_out/r2fuzz4sa_$i.log | cut -c1-100)" | tee -a $S
done
