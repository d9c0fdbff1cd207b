#!/bin/bash
# round-2 record: full bench lines, rocprofv3 kernel stats + busy fraction, PMC traffic, virtual-rank compute
mkdir -p gpurun_out
S=gpurun_out/r2z_summary.txt
run() { local name=$1 to=$2; shift 2
  echo "=== $name" | tee -a $S
  timeout -k 10 $to "$@" > gpurun_out/$name.log 2>&1
  local rc=$?
  echo "rc=$rc" | tee -a $S
  tail -1 gpurun_out/$name.log | cut -c1-200 | tee -a $S
  if [ $rc -ge 124 ]; then echo "killed/timeout: stopping" | tee -a $S; exit $rc; fi
}
rm -f $S
run r2z_bench_c3 500 python bench.py
run r2z_bench_c2 400 python bench.py --steps 5 --warmup 2 --workload c2 --no-cpu-baseline
run r2z_bench_huge 500 python bench.py --steps 3 --warmup 1 --workload huge --no-cpu-baseline
run r2z_bench_huge_s 500 python bench.py --steps 3 --warmup 1 --workload huge_s --no-cpu-baseline
run r2z_bench_c4s 300 python bench.py --steps 5 --warmup 2 --workload c4s --no-cpu-baseline
run r2z_bench_c5s 300 python bench.py --steps 5 --warmup 2 --workload c5s --no-cpu-baseline
export TMPDIR=/tmp
D=$PWD/gpurun_out/r2z_prof
rm -rf $D; mkdir -p $D
echo "=== rocprof kernel stats c3" | tee -a $S
( cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $D/k -o c3 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-host-boundary > $D/k.log 2>&1 ); rc=$?; echo rc=$rc | tee -a $S
if [ $rc -ge 124 ]; then exit $rc; fi
python3 tools/kstats.py $(find $D/k -name "c3_kernel_stats.csv" | head -1) 60 > gpurun_out/r2z_c3_kernel_stats.txt 2>&1
python3 tools/busy.py $(find $D/k -name "c3_kernel_trace.csv" | head -1) 40 > gpurun_out/r2z_c3_last_chain.txt 2>&1; tail -1 gpurun_out/r2z_c3_last_chain.txt | tee -a $S
echo "=== pmc fetch" | tee -a $S
( cd /tmp && timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $D/f -o f -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-host-boundary > $D/f.log 2>&1 ); rc=$?; echo rc=$rc | tee -a $S
if [ $rc -ge 124 ]; then exit $rc; fi
echo "=== pmc write" | tee -a $S
( cd /tmp && timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $D/w -o w -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-host-boundary > $D/w.log 2>&1 ); rc=$?; echo rc=$rc | tee -a $S
if [ $rc -ge 124 ]; then exit $rc; fi
python3 tools/pmc_summary.py $(find $D/f -name "f_counter_collection.csv" | head -1) $(find $D/w -name "w_counter_collection.csv" | head -1) > gpurun_out/r2z_c3_pmc_traffic.txt 2>&1
python3 tools/pmc_to_json.py $(find $D/f -name "f_counter_collection.csv" | head -1) $(find $D/w -name "w_counter_collection.csv" | head -1) c3 gpurun_out/r2z_c3_pmc_traffic.json >> $S 2>&1
head -12 gpurun_out/r2z_c3_pmc_traffic.txt | tee -a $S
find $D -name "*.db" -delete; find $D -name "*kernel_trace.csv" -delete; find $D -name "*counter_collection.csv" -delete
for R in 1 2; do
  run r2z_sim_$R 400 python tools/simscale.py $R c3
done
