"""virtual-rank profile of the distributed chain on ONE GPU: R contexts, phases timed per rank (compute only, no wire)"""
import sys, os, time, torch, importlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry, bench
pkg = entry.load_package()
D = importlib.import_module("bigbwt_amd.dist")
dev = torch.device('cuda', 0)
R = int(sys.argv[1]); name = sys.argv[2] if len(sys.argv) > 2 else 'c2'
wl = bench.WORKLOADS[name]
texts = [bench.make_text(dev, wl, 2, variant=r) for r in range(R)]
ctxs = [pkg.Context(0) for _ in range(R)]
# time the methods of rank 0's context
acc = {}
def wrap(obj, meth):
    f = getattr(obj, meth)
    def g(*a, **k):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r = f(*a, **k)
        torch.cuda.synchronize(); acc[meth] = acc.get(meth, 0.0) + (time.perf_counter() - t0) * 1e3
        return r
    setattr(obj, meth, g)
for m_ in ('dist_propose_triggers', 'dist_local_parse', 'dist_export_local', 'dist_global_sort', 'dist_global_finish', 'dist_merge'):
    wrap(ctxs[R - 1], m_)
for it in range(2):
    acc.clear()
    if it == 1: ctxs[R - 1].set_kernel_trace(True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    res = D.simulate(ctxs, texts, wl['w'], wl['p'], 0)
    torch.cuda.synchronize(); el = (time.perf_counter() - t0) * 1e3
print('R=%d workload=%s total wall (all ranks sequential) %.1f ms' % (R, name, el))
print('last rank phases ms:', {k: round(v, 2) for k, v in acc.items()}, 'sum %.1f' % sum(acc.values()))
print('stats', res[R - 1]['stats'])

kt = ctxs[R - 1].kernel_trace()
rows = sorted(kt, key=lambda r: -r['total_ms'])
for r in rows[:14]: print('   %-48s %8.2f ms %5d launches' % (r['name'], r['total_ms'], r['launches']))
print('   traced total %.1f ms' % sum(r['total_ms'] for r in rows))
