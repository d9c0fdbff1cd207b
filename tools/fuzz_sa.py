import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import __graft_entry__ as entry
from test_gpu_fuzz import gen
pkg = entry.load_package(); O = entry.load_oracle()
ctx = pkg.Context(0)
rng = np.random.default_rng(int(sys.argv[1])); ntr = int(sys.argv[2])
bad = 0; n1 = n2 = n3 = 0
for it in range(ntr):
    t = gen(rng, O)
    w = int(rng.choice([4, 10])); p = int(rng.choice([10, 20, 100]))
    try:
        pr = O.parse(t, w, p)
    except Exception:
        continue
    d = pr["dict"]
    a = ctx.gsacak(d); b, _ = O.gsacak(d, want_lcp=False)
    n1 += 1
    if not np.array_equal(a, b):
        bad += 1; print("GSACAK mismatch it", it, len(d), flush=True); np.save("/root/repo/gpurun_out/fuzzsa_bad_%d.npy" % it, t)
    s = np.concatenate([pr["parse"], np.zeros(1, np.uint32)])
    if len(s) > 2:
        n2 += 1
        if not np.array_equal(ctx.sacak_int(s, int(s.max()) + 1), O.sacak_int(s)):
            bad += 1; print("SACAK_INT mismatch it", it, flush=True)
    tt = np.concatenate([t[: 20000], np.zeros(1, np.uint8)])
    n3 += 1
    if not np.array_equal(ctx.sacak(tt), O.sacak(tt)):
        bad += 1; print("SACAK mismatch it", it, flush=True)
print("gsacak", n1, "sacak_int", n2, "sacak", n3, "bad", bad)
