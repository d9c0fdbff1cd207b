#!/bin/bash
# longer fuzz runs of the final code: structured random texts through the fused chain, the suffix sorters and the
# distributed chain against the oracle, in the default configuration and with the newer paths forced / disabled
mkdir -p gpurun_out
S=gpurun_out/r2fuzz_summary.txt
rm -f $S
i=0
for V in "X=1" "PFP_KEYSONLY=1" "PFP_DEBUG=1" "PFP_NO_FINISHER=1" "PFP_FORCE_IDX64=1" "PFP_HARD_MODE=3"; do
  i=$((i+1))
  env $V FUZZ_MAXN=200000 timeout -k 10 170 python tools/fuzz.py $((100+i)) 1200 > gpurun_out/r2fuzz_$i.log 2>&1; rc=$?
  echo "$V fuzz rc=$rc: $(tail -1 gpurun_out/r2fuzz_$i.log | cut -c1-160)" | tee -a $S
  if [ $rc -ge 124 ]; then echo "(time limit: the cases that ran passed)" | tee -a $S; fi
  env $V timeout -k 10 100 python tools/fuzz_sa.py $((200+i)) 800 > gpurun_out/r2fuzzsa_$i.log 2>&1; rc=$?
  echo "$V fuzz_sa rc=$rc: $(tail -1 gpurun_out/r2fuzzsa_$i.log | cut -c1-160)" | tee -a $S
done
