#!/bin/bash
# usage (GPU box, repo root): tools/pmc_kernel.sh <kernel-name-substring> <workload> "<COUNTER ...>" ["<COUNTER ...>" ...]
#   one rocprofv3 --pmc pass per counter group (kernel trace only) over tools/one_chain.py <workload> 2; prints the named
#   kernel's counters of its last dispatch.
k=$1; wl=$2; shift 2
export TMPDIR=/tmp
D=$PWD/gpurun_out/pmc_kernel; rm -rf $D; mkdir -p $D
i=0
for grp in "$@"; do
  i=$((i+1))
  ( cd /tmp && timeout -k 10 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $D/g$i -o g -- python3 $GRAFT_REPO_ROOT/tools/one_chain.py $wl 2 > $D/g$i.log 2>&1 ) || { echo "group $i failed"; tail -3 $D/g$i.log; exit 1; }
  python3 - "$k" $(find $D/g$i -name "g_counter_collection.csv" | head -1) <<'P'
import csv, sys, collections
k, f = sys.argv[1], sys.argv[2]
rows = [r for r in csv.DictReader(open(f)) if k in r['Kernel_Name']]
last = max(int(r['Dispatch_Id']) for r in rows) if rows else None
for r in rows:
    if int(r['Dispatch_Id']) == last:
        print('%-28s %16.0f   (grid %s wg %s vgpr %s sgpr %s lds %s)' % (r['Counter_Name'], float(r['Counter_Value']), r.get('Grid_Size'), r.get('Workgroup_Size'), r.get('VGPR_Count'), r.get('SGPR_Count'), r.get('LDS_Block_Size')))
P
done
find $D -name "*.db" -delete
