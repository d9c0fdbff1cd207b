#!/bin/bash
mkdir -p gpurun_out
S=gpurun_out/r2i_summary.txt
run() { local name=$1 to=$2; shift 2
  echo "=== $name" | tee -a $S
  timeout -k 10 $to "$@" > gpurun_out/$name.log 2>&1
  local rc=$?
  echo "rc=$rc" | tee -a $S
  tail -8 gpurun_out/$name.log | cut -c1-1500 | tee -a $S
  if [ $rc -ge 124 ]; then echo "killed/timeout: stopping" | tee -a $S; exit $rc; fi
}
rm -f $S
export PFP_TRACE_ROUNDS=1
run r2i_wide 900 python tools/check_wide.py --workload wide
unset PFP_TRACE_ROUNDS
export PFP_BENCH_BACKEND=gloo PFP_BENCH_ONE_GPU=1
run r2i_bench_2ranks 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 2 --warmup 1
