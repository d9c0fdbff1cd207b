"""virtual-rank profile of the distributed chain on ONE GPU: R contexts, one rank's compute phases timed (no wire).

    python tools/simscale.py R [workload] [dedup]

Rank r holds variant r of the workload (bench.py's weak-scaling collection); dist.simulate serves the collectives by
concatenation, so the sum of the timed library calls of one rank is that rank's compute per step; R = 1 is the
same chain on one shard."""
import importlib
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

pkg = entry.load_package()
D = importlib.import_module("bigbwt_amd.dist")
synth = importlib.import_module("bigbwt_amd.synth")
dev = torch.device('cuda', 0)
R = int(sys.argv[1]); name = sys.argv[2] if len(sys.argv) > 2 else 'c3'
dedup = sys.argv[3] if len(sys.argv) > 3 else 'alltoall'
wl = synth.WORKLOADS[name]
texts = [synth.workload_text_torch(dev, name, variant=r) for r in range(R)]
torch.cuda.empty_cache()
ctxs = [pkg.Context(0) for _ in range(R)]
acc = {}


per_phase, last_snap, tracing = {}, {}, [False]


def wrap(obj, meth):
    f = getattr(obj, meth)

    def g(*a, **k):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r = f(*a, **k)
        torch.cuda.synchronize(); acc[meth] = acc.get(meth, 0.0) + (time.perf_counter() - t0) * 1e3
        if tracing[0]:      # SIMSCALE_PHASES=1: the kernels of this call alone (difference of the cumulative trace)
            snap = {x['name']: (x['total_ms'], x['launches']) for x in obj.kernel_trace()}
            d = per_phase.setdefault(meth, {})
            for nm, (ms, ln) in snap.items():
                ms0, ln0 = last_snap.get(nm, (0.0, 0))
                if ln > ln0:
                    o = d.get(nm, (0.0, 0)); d[nm] = (o[0] + ms - ms0, o[1] + ln - ln0)
            last_snap.clear(); last_snap.update(snap)
        return r
    setattr(obj, meth, g)


for m_ in ('dist_parse_plan', 'dist_propose_triggers2', 'dist_decide_density', 'dist_local_parse2', 'dist_export_local', 'dist_partition_words', 'dist_export_partition',
           'dist_owner_dedup', 'dist_export_owned', 'dist_global_sort', 'dist_global_sort_distinct', 'dist_global_finish', 'dist_merge',
           'pack5_dev', 'sample_runs_dev', 'dist_sample_runs', 'dist_parse_sort', 'dist_set_parse_sa'):
    wrap(ctxs[R - 1], m_)
for it in range(2):
    acc.clear()
    if it == 1:
        ctxs[R - 1].set_kernel_trace(True)
        tracing[0] = os.environ.get("SIMSCALE_PHASES") == "1"
    torch.cuda.synchronize(); t0 = time.perf_counter()
    res = D.simulate(ctxs, texts, wl['w'], wl['p'], wl['flags'], dedup=dedup, trim=set(range(R - 1)) if R > 2 else False)   # the timed rank keeps its pool warm, as a real rank does
    torch.cuda.synchronize(); el = (time.perf_counter() - t0) * 1e3
print('R=%d workload=%s dedup=%s flags=%d total wall (all ranks one after the other) %.1f ms' % (R, name, dedup, wl['flags'], el))
print('last rank compute ms:', {k: round(v, 2) for k, v in acc.items()}, 'sum %.1f' % sum(acc.values()))
ms_ = ctxs[R - 1].mem_stats()
print('last rank device memory: peak in use %.2f GB (%.1f bytes per byte of its shard, %.1f per byte of the union dictionary)' % (
    ms_['peak'] / 1e9, ms_['peak'] / texts[R - 1].numel(), ms_['peak'] / max(res[R - 1]['stats']['glob']['dict_bytes'], 1)))
st = res[R - 1]['stats']
print('stats', {k: st[k] for k in ('phrases_total', 'shard_bytes', 'sa_shares', 'parse_shares', 'dedup', 'parse_density')}, st['glob'])
kt = ctxs[R - 1].kernel_trace()
rows = sorted(kt, key=lambda r: -r['total_ms'])
for r in rows[:int(os.environ.get("SIMSCALE_ROWS", "12"))]:
    print('   %-48s %8.2f ms %5d launches' % (r['name'], r['total_ms'], r['launches']))
print('   traced total %.1f ms' % sum(r['total_ms'] for r in rows))
if tracing[0]:
    for meth, d in sorted(per_phase.items(), key=lambda kv: -sum(v[0] for v in kv[1].values())):
        tot = sum(v[0] for v in d.values())
        if tot < 1.0:
            continue
        print('  -- %s: %.1f ms of kernels (%.1f ms wall)' % (meth, tot, acc.get(meth, 0.0)))
        for nm, (ms, ln) in sorted(d.items(), key=lambda kv: -kv[1][0])[:14]:
            print('       %-48s %8.2f ms %5d launches' % (nm, ms, ln))
