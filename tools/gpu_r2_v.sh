#!/bin/bash
# rocPRIM onesweep configurations for the first-round sort (library built with -DPFP_SORTCFG_EXPERIMENT)
mkdir -p gpurun_out
S=gpurun_out/r2v_summary.txt
rm -f $S
for W in c2 c3; do
for CFG in 0 1 2 3 4 5 6 8; do
  PFP_SORTCFG=$CFG timeout -k 10 200 python bench.py --steps 3 --warmup 1 --workload $W --no-cpu-baseline --no-host-boundary > gpurun_out/r2v_${W}_$CFG.log 2>&1; rc=$?
  if [ $rc -ge 124 ]; then echo "cfg $CFG killed" | tee -a $S; exit $rc; fi
  python3 - "$W" "$CFG" gpurun_out/r2v_${W}_$CFG.log <<'PY' | tee -a $S
import json, sys
w, cfg, f = sys.argv[1:4]
for line in open(f):
    if line.startswith('{"metric"'):
        j = json.loads(line)
        srt = [k for k in j['kernels'] if k['kernel'].startswith('rocprim::radix_sort_pairs<u64,u32>')]
        print(w, 'cfg', cfg, 'ms/step', j['ms_per_step'], 'sort ms', srt[0]['ms_per_step'] if srt else None, 'perm_ok', j['verified'].get('bwt_is_permutation_of_text_plus_eos'), 'digests', (j['verified'].get('outputs_match_reference_digests_whole_text') or {}))
        break
else:
    print(w, 'cfg', cfg, 'no result')
PY
done
done
