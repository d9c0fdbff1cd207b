import sys, os, time, importlib.util
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'oracle'))
spec = importlib.util.spec_from_file_location('bigbwt_amd', os.path.join(ROOT, 'big-bwt_amd', '__init__.py'), submodule_search_locations=[os.path.join(ROOT,'big-bwt_amd')])
m = importlib.util.module_from_spec(spec); sys.modules['bigbwt_amd'] = m; spec.loader.exec_module(m)
import oracle as O, numpy as np
ctx = m.Context(0)
def check(name, text, w, p):
    text = np.frombuffer(bytes(text), dtype=np.uint8)
    ends, used = ctx.scan(text, w, p)
    oe = O.scan(text, w, p)
    print(name, 'scan', len(ends), len(oe), np.array_equal(ends, oe), flush=True)
    ps = ctx.parse(text, w, p, want_sai=True)
    op = O.parse(text, w, p)
    for k in ('dict','occ','parse','last'):
        print('  parse', k, np.array_equal(ps[k], op[k]), flush=True)
    print('  parse sai', np.array_equal(m.unpack5(ps['sai']), op['sai']))
    for flags in (0, 1, 6):
        t0 = time.time(); g = ctx.bigbwt(text, w, p, flags); t1 = time.time()
        o = O.bigbwt(text, w, p, flags)
        ok = np.array_equal(g['bwt'], o['bwt'])
        msg = f'bwt={ok}'
        if flags & 1: msg += f" sa={np.array_equal(m.unpack5(g['sa']), o['sa'])}"
        if flags & 2: msg += f" ssa={np.array_equal(m.unpack5(g['ssa']).reshape(-1,2), o['ssa'])}"
        if flags & 4: msg += f" esa={np.array_equal(m.unpack5(g['esa']).reshape(-1,2), o['esa'])}"
        print('  bigbwt flags', flags, msg, '%.3fs' % (t1-t0), ctx.stats(), flush=True)
check('KAT1', b'CCGATTACAT!GATTACAT!GATTAGATA', 4, 11)
check('KATQ1', b'GATTACAT!GATTACAT!GATTAGATA', 4, 11)
check('GEN1e5x4', O.gen_fasta(100000, 4, 0.001, 7), 10, 100)
check('GEN1e6x4', O.gen_fasta(1000000, 4, 0.001, 7), 10, 100)
