#!/usr/bin/env python3
"""Summarise a rocprofv3 *_kernel_stats.csv with short kernel names (used to write profiles/*.md)."""
import csv, os, re, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from rocnames import sort_types
rows = list(csv.DictReader(open(sys.argv[1])))
def short(n):
    m = re.search(r'pfp::(\w+)(<\d+>)?', n)
    if m and 'rocprim' not in n[:40]: return 'pfp::' + m.group(1) + (m.group(2) or '')
    for k in ['onesweep_iteration', 'onesweep_global_offsets', 'partition_impl', 'scan_impl', 'block_sort', 'radix_sort_single', 'lookback_scan_state', 'init_lookback']:
        if k in n:
            kv = sort_types(n)
            key = ('%s%s' % (kv[0], ',' + kv[1] if kv[1] else ' keys')) if kv else ('u64key' if re.search(r'onesweep_config<[^>]*unsigned long, unsigned int', n) else ('u32key' if 'onesweep' in n else ''))
            return 'rocprim::' + k + (' ' + key if key else '')
    return n[:70]
tot = sum(float(r['TotalDurationNs']) for r in rows)
agg = {}
for r in rows:
    a = agg.setdefault(short(r['Name']), [0, 0.0]); a[0] += int(r['Calls']); a[1] += float(r['TotalDurationNs'])
print(f"{'kernel':48s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>10s} {'%':>6s}")
for s, (c, t) in sorted(agg.items(), key=lambda x: -x[1][1])[:int(sys.argv[2]) if len(sys.argv) > 2 else 30]:
    print(f"{s:48s} {c:7d} {t/1e6:10.2f} {t/c/1e3:10.1f} {100*t/tot:6.2f}")
print(f"total kernel time ms: {tot/1e6:.2f}")
