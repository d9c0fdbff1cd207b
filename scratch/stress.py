import sys, os, time, importlib.util
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'oracle'))
spec = importlib.util.spec_from_file_location('bigbwt_amd', os.path.join(ROOT, 'big-bwt_amd', '__init__.py'), submodule_search_locations=[os.path.join(ROOT,'big-bwt_amd')])
m = importlib.util.module_from_spec(spec); sys.modules['bigbwt_amd'] = m; spec.loader.exec_module(m)
import oracle as O, numpy as np
ctx = m.Context(0)
text = O.gen_fasta(1000000, 4, 0.001, 7)
o = {f: O.bigbwt(text, 10, 100, f) for f in (0,1,6)}
for it in range(6):
    for flags in (0, 1, 6):
        g = ctx.bigbwt(text, 10, 100, flags)
        ok = np.array_equal(g['bwt'], o[flags]['bwt'])
        st = ctx.stats()
        extra = ''
        if not ok:
            bad = np.nonzero(g['bwt'] != o[flags]['bwt'])[0]
            extra = f' nbad={len(bad)} first={bad[:10]} got={g["bwt"][bad[:10]]} exp={o[flags]["bwt"][bad[:10]]}'
        if flags & 1:
            sa_ok = np.array_equal(m.unpack5(g['sa']), o[flags]['sa'])
            extra += f' sa={sa_ok}'
        print(it, flags, ok, st['hard_groups'], st['hard_chars'], extra, flush=True)
