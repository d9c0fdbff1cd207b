#!/bin/bash
mkdir -p gpurun_out
for V in "X=1" "PFP_KEYBITS=39" "PFP_KEYBITS=43"; do
  env $V timeout -k 10 400 python bench.py --steps 2 --warmup 1 --workload huge --no-cpu-baseline --no-host-boundary > gpurun_out/r2ae_huge.log 2>&1
  echo "rc=$? $V"
  python3 tools/benchsum.py gpurun_out/r2ae_huge.log | sed -n 1,1p | cut -c1-300
  python3 tools/benchsum.py gpurun_out/r2ae_huge.log | grep -E "seg_small|segmented|radix_sort_pairs<u64,u32>|pivot"
done
