#!/bin/bash
mkdir -p gpurun_out
S=gpurun_out/r2fuzz2_summary.txt
rm -f $S
python3 - <<'PY' 2>&1 | grep -v amdgpu.ids | tee -a $S
import os, sys, numpy as np
os.environ["PFP_KEYSONLY"] = "1"
sys.path.insert(0, os.getcwd())
import __graft_entry__ as entry
pkg = entry.load_package(); O = entry.load_oracle()
ctx = pkg.Context(0)
for f in ("keysonly_sa_536", "keysonly_sa_164", "keysonly_686"):
    t = np.load("tools/fuzzcases/%s.npy" % f)
    for w, p in ((4, 10), (4, 20), (10, 10), (10, 20), (10, 100)):
        try:
            pr = O.parse(t, w, p)
        except Exception:
            continue
        d = pr["dict"]
        a = ctx.gsacak(d); b, _ = O.gsacak(d, want_lcp=False)
        ok = np.array_equal(a, b)
        try:
            want = O.bigbwt(t, w, p, O.FLAG_SA); got = ctx.bigbwt(t, w, p, pkg.FLAG_SA)
            ok2 = np.array_equal(got["bwt"], want["bwt"]) and np.array_equal(pkg.unpack5(got["sa"]), want["sa"])
        except RuntimeError:
            ok2 = None
        print(f, w, p, "gsacak", ok, "chain", ok2)
PY
i=0
for V in "PFP_KEYSONLY=1" "PFP_KEYSONLY=1 PFP_DEBUG=1" "X=1"; do
  i=$((i+1))
  env $V FUZZ_MAXN=200000 timeout -k 10 200 python tools/fuzz.py $((300+i)) 1500 > gpurun_out/r2fuzz2_$i.log 2>&1; rc=$?
  echo "$V fuzz rc=$rc: $(tail -1 gpurun_out/r2fuzz2_$i.log | cut -c1-160)" | tee -a $S
  env $V timeout -k 10 150 python tools/fuzz_sa.py $((400+i)) 1500 > gpurun_out/r2fuzz2sa_$i.log 2>&1; rc=$?
  echo "$V fuzz_sa rc=$rc: $(tail -1 gpurun_out/r2fuzz2sa_$i.log | cut -c1-160)" | tee -a $S
done
