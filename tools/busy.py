#!/usr/bin/env python3
"""GPU busy fraction of the last chain in a rocprofv3 kernel trace: python tools/busy.py <*_kernel_trace.csv>

The last chain = from the last pfp::kr_scan_kernel (or kr_flag_kernel) dispatch to the last kernel of the trace that belongs to the
library or to rocPRIM before the next non-library kernel burst.  Prints span, summed kernel time, union of the busy
intervals and the number of dispatches - the gap between span and union is launch latency and host round trips."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows), key=lambda x: x[0])
starts = [k for k, e in enumerate(ev) if "kr_scan_kernel" in e[2] or "kr_flag_kernel" in e[2]]
if not starts:
    sys.exit("no pfp::kr_scan_kernel or kr_flag_kernel in the trace")
first = starts[-1]
chain = ev[first:]
# cut at the first torch kernel after the chain (the bench's checks)
cut = len(chain)
for k, e in enumerate(chain):
    if "at::native" in e[2] or "at::cuda" in e[2]:
        cut = k
        break
chain = chain[:cut]
span = chain[-1][1] - chain[0][0]
summed = sum(e[1] - e[0] for e in chain)
union = 0
cur_s, cur_e = chain[0][0], chain[0][1]
for s, e, _ in chain[1:]:
    if s > cur_e:
        union += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
union += cur_e - cur_s
import re
agg = {}
for st, en, name in chain:
    m = re.search(r"pfp::(\w+)", name)
    key = "pfp::" + m.group(1) if m and "rocprim" not in name[:40] else re.sub(r"<.*", "", name.replace("void ", ""))[:60]
    a = agg.setdefault(key, [0, 0]); a[0] += 1; a[1] += en - st
for k, (cnt, ns) in sorted(agg.items(), key=lambda x: -x[1][1])[:int(sys.argv[2]) if len(sys.argv) > 2 else 25]:
    print(f"  {k:62s} {cnt:5d} {ns / 1e6:9.3f} ms")
print(f"dispatches {len(chain)}  span {span / 1e6:.2f} ms  kernel time {summed / 1e6:.2f} ms  busy (union) {union / 1e6:.2f} ms  "
      f"busy fraction {union / span:.3f}  idle {(span - union) / 1e6:.2f} ms")
