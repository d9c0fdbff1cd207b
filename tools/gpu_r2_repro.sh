#!/bin/bash
mkdir -p gpurun_out
cat > /tmp/repro.py <<'PY'
import sys, os, numpy as np
sys.path.insert(0, os.getcwd())
import __graft_entry__ as entry
pkg = entry.load_package(); O = entry.load_oracle()
ctx = pkg.Context(0)
t = np.load("tools/fuzzcases/keysonly_sa_536.npy")
print("n", len(t), "distinct bytes", len(set(t.tolist())))
for w, p in ((4, 10), (4, 20), (4, 100), (10, 10), (10, 20), (10, 100)):
    want = O.bigbwt(t, w, p, O.FLAG_SA)
    try:
        got = ctx.bigbwt(t, w, p, pkg.FLAG_SA)
        eq = np.array_equal(got["bwt"], want["bwt"])
        print("bwt equal", eq, "sa equal", np.array_equal(pkg.unpack5(got["sa"]), want["sa"]), ctx.stats()["dict_size"], ctx.stats()["n_words"])
        if not eq:
            d = np.nonzero(got["bwt"] != want["bwt"])[0]
            print("first diffs", d[:10], len(d))
    except pkg.PfpError as ex:
        print("ERR", str(ex)[:600])
    pr = O.parse(t, w, p)
    d = pr["dict"]
    try:
        a = ctx.gsacak(d); b, _ = O.gsacak(d, want_lcp=False)
        bad = np.nonzero(a != b)[0]
        print("gsacak len", len(d), "mismatches", len(bad), bad[:10], a[bad[:6]], b[bad[:6]])
        if len(bad):
            i, j = int(a[bad[0]]), int(b[bad[0]])
            print("ours  ", bytes(d[i:i+60]))
            print("theirs", bytes(d[j:j+60]))
    except pkg.PfpError as ex:
        print("ERR gsacak", str(ex)[:600])
PY
for V in "PFP_KEYSONLY=1" "PFP_KEYSONLY=1 PFP_DEBUG=1" "PFP_KEYSONLY=1 PFP_NO_FINISHER=1" "PFP_KEYSONLY=1 PFP_NO_BIGSIDE=1" "X=1"; do
  echo "== $V"; env $V PFP_TRACE_ROUNDS=1 timeout -k 10 120 python /tmp/repro.py 2>&1 | grep -v amdgpu.ids | tail -14 | cut -c1-300
done
