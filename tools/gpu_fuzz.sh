#!/bin/bash
# usage (GPU box): [FUZZ_SEED=s] tools/gpu_fuzz.sh <tag> <trials> [VAR=VALUE ...]   - tools/fuzz.py and tools/fuzz_sa.py under every given
# environment (one variant per argument; "X=1" = defaults), 240 s each at most; one summary line per run in gpurun_out/<tag>_fuzz.txt
tag=$1; n=$2; shift 2
mkdir -p gpurun_out
S=gpurun_out/${tag}_fuzz.txt
: > $S
for e in "$@"; do
  for drv in fuzz fuzz_sa; do
    env $e timeout -k 10 240 python tools/$drv.py ${FUZZ_SEED:-7} $n > gpurun_out/${tag}_${drv}_$e.log 2>&1; rc=$?
    echo "$e $drv rc=$rc: $(grep -c MISMATCH gpurun_out/${tag}_${drv}_$e.log) mismatch lines; last: $(tail -n 1 gpurun_out/${tag}_${drv}_$e.log | cut -c1-160)" | tee -a $S
    if [ $rc -ge 124 ]; then echo "time limit: stopping here" | tee -a $S; exit 0; fi
  done
done
