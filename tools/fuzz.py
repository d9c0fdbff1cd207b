"""structured random inputs through the fused chain, the suffix sorters and the distributed chain vs the oracle"""
import sys, os, importlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import __graft_entry__ as entry
pkg = entry.load_package(); O = entry.load_oracle()
D = importlib.import_module("bigbwt_amd.dist")
ctx = pkg.Context(0)
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
ntr = int(sys.argv[2]) if len(sys.argv) > 2 else 100
def gen():
    kind = rng.integers(0, 7)
    n = int(rng.integers(200, int(os.environ.get("FUZZ_MAXN", "60000"))))
    if kind == 0:      # periodic
        unit = rng.integers(3, 256, size=int(rng.integers(1, 40))).astype(np.uint8)
        t = np.tile(unit, n // len(unit) + 1)[:n].copy()
        for _ in range(int(rng.integers(0, 6))): t[rng.integers(0, n)] = rng.integers(3, 256)
    elif kind == 1:    # tiny alphabet
        t = rng.choice(np.frombuffer(b"AB", dtype=np.uint8), size=n)
    elif kind == 2:    # runs
        t = np.repeat(rng.choice(np.frombuffer(b"ACGTN", dtype=np.uint8), size=n // 50 + 1), rng.integers(1, 120, size=n // 50 + 1))[:n].copy()
    elif kind == 3:    # all byte values
        t = rng.integers(3, 256, size=n).astype(np.uint8)
    elif kind == 4:    # near-identical copies
        base = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=max(50, n // 8))
        parts = []
        for c in range(8):
            b = base.copy()
            for _ in range(int(rng.integers(0, 5))): b[rng.integers(0, len(b))] = ord("ACGT"[rng.integers(0, 4)])
            parts.append(b)
        t = np.concatenate(parts)
    elif kind == 5:    # long exact repeats
        blk = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=int(rng.integers(500, 9000)))
        t = np.concatenate([blk, rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=100), blk, blk[: len(blk) // 2]])
    else:              # fasta-like with N block and lower case
        t = O.gen_fasta(max(600, n // 3), 3, 0.01, int(rng.integers(1, 1 << 30)), n_blocks=[(100, max(60, n // 10))])
        m = rng.random(len(t)) < 0.1
        t = np.where(m & (t >= 65), t | 32, t).astype(np.uint8)
    return np.ascontiguousarray(t, dtype=np.uint8), int(kind)
import time
timing = os.environ.get("FUZZ_TIMING") is not None      # seconds per iteration (oracle / chain / distributed) on stdout
bad = 0; compared = 0; dist_runs = 0; skipped = 0
for it in range(ntr):
    t_it = time.perf_counter()
    t, kind = gen()
    w = int(rng.choice([4, 5, 10, 17])); p = int(rng.choice([10, 11, 20, 100]))
    flags = int(rng.choice([0, 1, 6]))
    if it and it % 20 == 0: print("progress it=%d compared=%d bad=%d" % (it, compared, bad), flush=True)
    try:
        want = O.bigbwt(t, w, p, flags)
    except Exception as ex:
        skipped += 1
        continue
    t_or = time.perf_counter()
    ok = True; msg = ""
    try:
        ctx.set_max_phrase(int(rng.choice([0, 700, 32768])) if len(t) > 3000 else 0)
        got = ctx.bigbwt(t, w, p, flags)
        if timing: print("it=%d kind=%d n=%d w=%d p=%d flags=%d: oracle %.2f s, chain %.2f s" % (it, kind, len(t), w, p, flags, t_or - t_it, time.perf_counter() - t_or), flush=True)
        compared += 1
        ok = np.array_equal(got["bwt"], want["bwt"])
        if flags & 1: ok = ok and np.array_equal(pkg.unpack5(got["sa"]), want["sa"])
        if flags & 6: ok = ok and np.array_equal(pkg.unpack5(got["ssa"]).reshape(-1, 2), want["ssa"]) and np.array_equal(pkg.unpack5(got["esa"]).reshape(-1, 2), want["esa"])
        if ok and it % 3 == 0 and len(t) > 2000:
            R = int(rng.choice([2, 3]))
            cuts = [0] + sorted(int(x) for x in rng.choice(np.arange(400, len(t) - 400), size=R - 1, replace=False)) + [len(t)]
            ctxs = [pkg.Context(0) for _ in range(R)]
            try:
                for c in ctxs: c.set_max_phrase(2000)
                shards = [torch.from_numpy(t[cuts[r]:cuts[r + 1]].copy()).cuda() for r in range(R)]
                try:
                    res = D.simulate(ctxs, shards, w, p, flags, halo=1 << 16)
                    dist_runs += 1
                    bw = torch.cat([r["bwt"] for r in res]).cpu().numpy()
                    ok = np.array_equal(bw, want["bwt"]); msg = "dist R=%d shares=%s" % (R, res[0]["stats"]["sa_shares"])
                    if ok and flags == 1: ok = np.array_equal(torch.cat([r["sa"] for r in res]).cpu().numpy().astype(np.uint64)[1:], want["sa"])
                    if ok and flags == 6:       # the ranks' pieces of .ssa / .esa, in rank order, are the files
                        for key in ("ssa", "esa"):
                            pieces = torch.cat([r[key] for r in res]).cpu().numpy()
                            ok = ok and np.array_equal(pkg.unpack5(pieces).reshape(-1, 2), want[key])
                            offs = [r[key + "_off"] for r in res]
                            ok = ok and offs == [int(x) for x in np.cumsum([0] + [len(r[key]) for r in res[:-1]])]
                except pkg.PfpError as ex:
                    # a shard too small for its halo / without a phrase boundary is refused on every rank (the rank that
                    # failed names the reason, the others the step and its code): not a mismatch
                    refused = "halo" in str(ex) or "PFP_ESHORT" in str(ex) or ("local parse failed" in str(ex) and "PFP_ELIMIT" in str(ex))
                    if not refused: ok = False; msg = "dist " + str(ex)
            finally:
                for c in ctxs: c.close()
    except pkg.PfpError as ex:
        if "PFP_ESHORT" in str(ex) or "fewer than 2 phrases" in str(ex): continue
        ok = False; msg = str(ex)
    if not ok:
        bad += 1
        np.save("/root/repo/gpurun_out/fuzz_bad_%d.npy" % it, t)
        print("MISMATCH it=%d kind=%d n=%d w=%d p=%d flags=%d %s" % (it, kind, len(t), w, p, flags, msg), flush=True)
print("trials", ntr, "compared", compared, "dist", dist_runs, "skipped", skipped, "bad", bad)
