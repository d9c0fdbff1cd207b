"""random texts through pfp_bigbwt_files_multi (the native multi-GPU host, rank threads on one device: PFP_MULTI_LOOPBACK=1)
against the oracle's files.      python tools/fuzz_multi.py [seed] [trials]
A refusal (a shard without a complete phrase, a halo shorter than a phrase) must be a PfpError, never a hang or a wrong file."""
import importlib
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["PFP_MULTI_LOOPBACK"] = "1"
import __graft_entry__ as entry  # noqa: E402

pkg = entry.load_package()
O = entry.load_oracle()
pfpmod = importlib.import_module("bigbwt_amd.pfp")
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
ntr = int(sys.argv[2]) if len(sys.argv) > 2 else 100
ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)


def gen():
    kind = int(rng.integers(0, 5))
    n = int(rng.integers(3000, int(os.environ.get("FUZZ_MAXN", "120000"))))
    if kind == 0:      # near-identical copies
        base = rng.choice(ACGT, size=max(200, n // 8))
        parts = []
        for _ in range(8):
            b = base.copy()
            for _ in range(int(rng.integers(0, 8))):
                b[rng.integers(0, len(b))] = ACGT[rng.integers(0, 4)]
            parts.append(b)
        t = np.concatenate(parts)
    elif kind == 1:    # random bytes
        t = rng.integers(3, 256, size=n).astype(np.uint8)
    elif kind == 2:    # runs
        t = np.repeat(rng.choice(np.frombuffer(b"ACGTN", dtype=np.uint8), size=n // 40 + 1), rng.integers(1, 90, size=n // 40 + 1))[:n].copy()
    elif kind == 3:    # long exact repeats
        blk = rng.choice(ACGT, size=int(rng.integers(500, 9000)))
        t = np.concatenate([blk, rng.choice(ACGT, size=100), blk, blk[: len(blk) // 2], rng.choice(ACGT, size=n // 4)])
    else:              # fasta-like
        t = O.gen_fasta(max(900, n // 3), 3, 0.01, int(rng.integers(1, 1 << 30)), n_blocks=[(100, max(60, n // 20))])
    return np.ascontiguousarray(t, dtype=np.uint8), kind


bad = compared = refused = 0
tmp = tempfile.mkdtemp(prefix="pfpfm", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
for it in range(ntr):
    t, kind = gen()
    w = int(rng.choice([4, 5, 10, 17]))
    p = int(rng.choice([10, 11, 20, 100]))
    flags = int(rng.choice([0, 1, 2, 4, 6]))
    ranks = int(rng.integers(2, 7))
    halo = int(rng.choice([256, 4096, 1 << 16, 1 << 20]))
    if it and it % 25 == 0:
        print("progress it=%d compared=%d refused=%d bad=%d" % (it, compared, refused, bad), flush=True)
    try:
        want = O.bigbwt(t, w, p, flags)
    except Exception:
        continue
    base = os.path.join(tmp, "f")
    try:
        st = pfpmod.bigbwt_files_multi(t, base, [0] * ranks, w, p, flags, halo=halo)
    except pkg.PfpError as ex:
        refused += 1
        if ex.code not in (-5, -8, -6):      # ELIMIT (halo), ESHORT (no phrase in a shard), EFORMAT
            bad += 1
            print("MISMATCH it=%d unexpected error %s (kind=%d n=%d w=%d p=%d flags=%d ranks=%d halo=%d)" % (it, ex, kind, len(t), w, p, flags, ranks, halo), flush=True)
        continue
    ok = np.array_equal(np.fromfile(base + ".bwt", dtype=np.uint8), want["bwt"])
    if ok and flags & 1:
        ok = np.array_equal(pkg.unpack5(np.fromfile(base + ".sa", dtype=np.uint8)), want["sa"])
    if ok and flags & 2:
        ok = np.array_equal(pkg.unpack5(np.fromfile(base + ".ssa", dtype=np.uint8)).reshape(-1, 2), want["ssa"])
    if ok and flags & 4:
        ok = np.array_equal(pkg.unpack5(np.fromfile(base + ".esa", dtype=np.uint8)).reshape(-1, 2), want["esa"])
    compared += 1
    if not ok:
        bad += 1
        np.save(os.path.join(ROOT, "gpurun_out", "fuzz_multi_bad_%d.npy" % it), t)
        print("MISMATCH it=%d kind=%d n=%d w=%d p=%d flags=%d ranks=%d halo=%d shares=%d" % (it, kind, len(t), w, p, flags, ranks, halo, st["sa_shares"]), flush=True)
    for ext in (".bwt", ".sa", ".ssa", ".esa"):
        if os.path.exists(base + ext):
            os.unlink(base + ext)
print("trials %d compared %d refused %d bad %d" % (ntr, compared, refused, bad))
sys.exit(1 if bad else 0)
