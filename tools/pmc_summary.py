#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files per kernel: average counter value per dispatch.
usage: pmc_summary.py <counter_collection.csv> [...]   (FETCH_SIZE / WRITE_SIZE are in KiB)"""
import csv, re, sys, collections
def short(n):
    m = re.search(r'pfp::(\w+)', n)
    if m and 'rocprim' not in n[:40]: return 'pfp::' + m.group(1)
    for k in ['onesweep_iteration', 'onesweep_global_offsets', 'partition_impl', 'scan_impl', 'block_sort']:
        if k in n: return 'rocprim::' + k
    return n[:50]
for f in sys.argv[1:]:
    agg = collections.defaultdict(lambda: [0, 0.0])
    cname = None
    for r in csv.DictReader(open(f)):
        cname = r['Counter_Name']
        a = agg[short(r['Kernel_Name'])]; a[0] += 1; a[1] += float(r['Counter_Value'])
    print(f"# {f}: {cname}")
    for k, (c, v) in sorted(agg.items(), key=lambda x: -x[1][1])[:16]:
        print(f"{k:42s} dispatches={c:5d}  total={v*1024/1e9:10.3f} GB  avg/dispatch={v*1024/c/1e6:10.2f} MB")
