#!/usr/bin/env python3
"""rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE counter_collection.csv (separate passes) -> the JSON that
bench.py reads for roofline.traffic.
usage: pmc_to_json.py <fetch.csv> <write.csv> <workload> <out.json>
Per kernel (bench.py's names): raw FETCH/WRITE bytes per chain execution and the corrected HBM bytes
2*FETCH + WRITE (gfx950: FETCH_SIZE reports half of the bytes of wide coalesced streaming reads,
MI355X_MICROARCH.md 'HBM'; exact for WRITE_SIZE).  A chain = one pfp_bigbwt_dev call; the number of
chains in the profiled run is the number of pfp::line_terms_kernel dispatches (the dictionary index: one per chain in every mode)."""
import collections, csv, json, os, re, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from rocnames import sort_types

def bench_name(n):
    m = re.search(r'pfp::(\w+)', n)
    if m and 'rocprim' not in n[:40]:
        k = m.group(1)
        # (round 4: bench.py's trace labels are the launched kernels' own names - no mapping table)
        return 'pfp::' + k
    if 'onesweep' in n or 'radix_sort' in n or 'block_sort' in n:
        kv = sort_types(n)
        if kv and kv[1] is None:
            return f'rocprim::radix_sort_keys<{kv[0]}>'
        if kv and kv[1]:
            return f'rocprim::radix_sort_pairs<{kv[0]},{kv[1]}>'
        return 'rocprim::radix_sort_pairs<u64,u32>' if re.search(r'unsigned long, unsigned int|unsigned long,unsigned int', n) else 'rocprim::radix_sort_pairs<u32,u32>'
    return None

def load(f):
    agg, disp = collections.defaultdict(float), collections.Counter()
    for r in csv.DictReader(open(f)):
        k = bench_name(r['Kernel_Name'])
        if k is None:
            continue
        agg[k] += float(r['Counter_Value']) * 1024.0
        disp[k] += 1
    return agg, disp

fetch, disp = load(sys.argv[1])
write, _ = load(sys.argv[2])
chains = max(1, disp.get('pfp::line_terms_kernel', 1))
out = dict(workload=sys.argv[3], chains_in_profiled_run=chains,
           source="rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only), tools/pmc_to_json.py",
           correction="traffic = 2*FETCH_SIZE + WRITE_SIZE (KiB*1024); the x2 is calibrated for 16 B/lane streaming loads, gathers are over-corrected",
           kernels={})
for k in sorted(set(fetch) | set(write), key=lambda k: -(2 * fetch.get(k, 0) + write.get(k, 0))):
    f, w = fetch.get(k, 0.0) / chains, write.get(k, 0.0) / chains
    out['kernels'][k] = dict(fetch_bytes_per_chain_raw=f, write_bytes_per_chain_raw=w, hbm_bytes_per_chain_corrected=2 * f + w,
                             dispatches_per_chain=disp.get(k, 0) / chains)
json.dump(out, open(sys.argv[4], 'w'), indent=1)
print(json.dumps({k: v for k, v in list(out['kernels'].items())[:6]}, indent=1))
