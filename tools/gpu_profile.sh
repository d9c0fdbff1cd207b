#!/bin/bash
# usage (on the GPU box, from the repo root): tools/gpu_profile.sh <tag> <workload> [extra bench.py flags]
#   rocprofv3 --kernel-trace --stats of `python3 bench.py --workload W`, then FETCH_SIZE and WRITE_SIZE in separate
#   --pmc passes (kernel trace only, as the pool's rules ask), summaries under gpurun_out/<tag>_<W>_*.
tag=$1; wl=$2; shift 2
mkdir -p gpurun_out
export TMPDIR=/tmp
D=$PWD/gpurun_out/${tag}_prof_$wl
rm -rf $D; mkdir -p $D
B="python3 $GRAFT_REPO_ROOT/bench.py --workload $wl --no-cpu-baseline --no-host-boundary --north-star off $*"
( cd /tmp && timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $D/k -o k -- $B --steps 5 --warmup 2 > $D/k.log 2>&1 ); rc=$?; echo "stats rc=$rc"
if [ $rc -ge 124 ]; then exit $rc; fi
tail -1 $D/k.log > gpurun_out/${tag}_${wl}_bench_under_rocprof.json
python3 tools/kstats.py $(find $D/k -name "k_kernel_stats.csv" | head -1) 60 > gpurun_out/${tag}_${wl}_kernel_stats.txt 2>&1
python3 tools/busy.py $(find $D/k -name "k_kernel_trace.csv" | head -1) 45 > gpurun_out/${tag}_${wl}_last_chain.txt 2>&1; tail -1 gpurun_out/${tag}_${wl}_last_chain.txt
( cd /tmp && timeout -k 10 500 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $D/f -o f -- $B --steps 1 --warmup 1 > $D/f.log 2>&1 ); rc=$?; echo "fetch rc=$rc"
if [ $rc -ge 124 ]; then exit $rc; fi
( cd /tmp && timeout -k 10 500 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $D/w -o w -- $B --steps 1 --warmup 1 > $D/w.log 2>&1 ); rc=$?; echo "write rc=$rc"
if [ $rc -ge 124 ]; then exit $rc; fi
python3 tools/pmc_summary.py $(find $D/f -name "f_counter_collection.csv" | head -1) $(find $D/w -name "w_counter_collection.csv" | head -1) > gpurun_out/${tag}_${wl}_pmc_traffic.txt 2>&1
python3 tools/pmc_to_json.py $(find $D/f -name "f_counter_collection.csv" | head -1) $(find $D/w -name "w_counter_collection.csv" | head -1) $wl gpurun_out/${tag}_${wl}_pmc_traffic.json 2>&1 | tail -2
head -24 gpurun_out/${tag}_${wl}_pmc_traffic.txt
find $D -name "*.db" -delete; find $D -name "*kernel_trace.csv" -delete; find $D -name "*counter_collection.csv" -delete
