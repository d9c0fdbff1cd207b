#!/bin/bash
mkdir -p gpurun_out
S=gpurun_out/r2f_summary.txt
run() { local name=$1 to=$2; shift 2
  echo "=== $name" | tee -a $S
  timeout -k 10 $to "$@" > gpurun_out/$name.log 2>&1
  local rc=$?
  echo "rc=$rc" | tee -a $S
  tail -6 gpurun_out/$name.log | cut -c1-500 | tee -a $S
  if [ $rc -ge 124 ]; then echo "killed/timeout: stopping" | tee -a $S; exit $rc; fi
}
rm -f $S
run r2f_tests 1100 python -m pytest tests -m gpu -q -x
run r2f_bench_c3 600 python bench.py --steps 5 --warmup 2
run r2f_wide 900 python tools/check_wide.py
export TMPDIR=/tmp
D=$PWD/gpurun_out/r2f_prof
rm -rf $D; mkdir -p $D
( cd /tmp && run_dummy=1 )
echo "=== rocprof kernel stats c3" | tee -a $S
( cd /tmp && timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $D/k -o c3 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-host-boundary > $D/k.log 2>&1 ); echo rc=$? | tee -a $S
python3 tools/kstats.py $D/k/*/c3_kernel_stats.csv 60 > gpurun_out/r2f_c3_kernel_stats.txt 2>&1; head -30 gpurun_out/r2f_c3_kernel_stats.txt | tee -a $S
echo "=== pmc fetch" | tee -a $S
( cd /tmp && timeout -k 10 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $D/f -o f -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-host-boundary > $D/f.log 2>&1 ); echo rc=$? | tee -a $S
echo "=== pmc write" | tee -a $S
( cd /tmp && timeout -k 10 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $D/w -o w -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-host-boundary > $D/w.log 2>&1 ); echo rc=$? | tee -a $S
python3 tools/pmc_summary.py $D/f/*/f_counter_collection.csv $D/w/*/w_counter_collection.csv > gpurun_out/r2f_c3_pmc_traffic.txt 2>&1
python3 tools/pmc_to_json.py $D/f/*/f_counter_collection.csv $D/w/*/w_counter_collection.csv c3 gpurun_out/r2f_c3_pmc_traffic.json >> $S 2>&1
head -40 gpurun_out/r2f_c3_pmc_traffic.txt | tee -a $S
rm -rf $D/k/*/*.db $D/f/*/*.db $D/w/*/*.db 2>/dev/null
du -sh $D | tee -a $S
