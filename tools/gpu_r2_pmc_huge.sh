#!/bin/bash
# PMC traffic (FETCH_SIZE / WRITE_SIZE, separate passes) and kernel stats of the 12.6 GB workload with -s
mkdir -p gpurun_out
export TMPDIR=/tmp
D=$PWD/gpurun_out/r2pmc_huge
rm -rf $D; mkdir -p $D
( cd /tmp && timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $D/k -o h -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --workload huge_s --no-cpu-baseline --no-host-boundary > $D/k.log 2>&1 ); rc=$?; echo "stats rc=$rc"
if [ $rc -ge 124 ]; then exit $rc; fi
python3 tools/kstats.py $(find $D/k -name "h_kernel_stats.csv" | head -1) 50 > gpurun_out/r2pmc_huge_s_kernel_stats.txt 2>&1
python3 tools/busy.py $(find $D/k -name "h_kernel_trace.csv" | head -1) 45 > gpurun_out/r2pmc_huge_s_last_chain.txt 2>&1; tail -1 gpurun_out/r2pmc_huge_s_last_chain.txt
( cd /tmp && timeout -k 10 500 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $D/f -o f -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --workload huge_s --no-cpu-baseline --no-host-boundary > $D/f.log 2>&1 ); rc=$?; echo "fetch rc=$rc"
if [ $rc -ge 124 ]; then exit $rc; fi
( cd /tmp && timeout -k 10 500 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $D/w -o w -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --workload huge_s --no-cpu-baseline --no-host-boundary > $D/w.log 2>&1 ); rc=$?; echo "write rc=$rc"
if [ $rc -ge 124 ]; then exit $rc; fi
python3 tools/pmc_summary.py $(find $D/f -name "f_counter_collection.csv" | head -1) $(find $D/w -name "w_counter_collection.csv" | head -1) > gpurun_out/r2pmc_huge_s_pmc_traffic.txt 2>&1
python3 tools/pmc_to_json.py $(find $D/f -name "f_counter_collection.csv" | head -1) $(find $D/w -name "w_counter_collection.csv" | head -1) huge_s gpurun_out/r2pmc_huge_s_pmc_traffic.json 2>&1 | tail -2
head -30 gpurun_out/r2pmc_huge_s_pmc_traffic.txt
find $D -name "*.db" -delete; find $D -name "*kernel_trace.csv" -delete; find $D -name "*counter_collection.csv" -delete
