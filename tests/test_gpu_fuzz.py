"""Structured random inputs (periodic text, two-letter alphabets, runs, all byte values, near-identical
copies, long exact repeats, soft-masked FASTA with an N block) through the fused chain and the
distributed chain, against the oracle.  The generator found the one real bug of round 1: an extra
trigger (pfp_set_max_phrase) equal to the text's first window turned the end-of-string byte of the BWT
into the reference's first-window quirk; tests/golden/runs_first_window.bin.gz is
that input."""
import gzip
import importlib
import os

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def gen(rng, O):
    kind = int(rng.integers(0, 7))
    n = int(rng.integers(200, 40000))
    dna = np.frombuffer(b"ACGT", dtype=np.uint8)
    if kind == 0:      # periodic
        unit = rng.integers(3, 256, size=int(rng.integers(1, 40))).astype(np.uint8)
        t = np.tile(unit, n // len(unit) + 1)[:n].copy()
        for _ in range(int(rng.integers(0, 6))):
            t[rng.integers(0, n)] = rng.integers(3, 256)
    elif kind == 1:    # two letters
        t = rng.choice(np.frombuffer(b"AB", dtype=np.uint8), size=n)
    elif kind == 2:    # runs
        t = np.repeat(rng.choice(np.frombuffer(b"ACGTN", dtype=np.uint8), size=n // 50 + 1), rng.integers(1, 120, size=n // 50 + 1))[:n].copy()
    elif kind == 3:    # all byte values
        t = rng.integers(3, 256, size=n).astype(np.uint8)
    elif kind == 4:    # near-identical copies
        base = rng.choice(dna, size=max(50, n // 8))
        parts = []
        for _ in range(8):
            b = base.copy()
            for _ in range(int(rng.integers(0, 5))):
                b[rng.integers(0, len(b))] = dna[rng.integers(0, 4)]
            parts.append(b)
        t = np.concatenate(parts)
    elif kind == 5:    # long exact repeats
        blk = rng.choice(dna, size=int(rng.integers(500, 9000)))
        t = np.concatenate([blk, rng.choice(dna, size=100), blk, blk[: len(blk) // 2]])
    else:              # FASTA with an N block and lower case
        t = O.gen_fasta(max(600, n // 3), 3, 0.01, int(rng.integers(1, 1 << 30)), n_blocks=[(100, max(60, n // 10))])
        m = rng.random(len(t)) < 0.1
        t = np.where(m & (t >= 65), t | 32, t).astype(np.uint8)
    return np.ascontiguousarray(t, dtype=np.uint8)


def check(pkg, O, ctx, t, w, p, flags, max_phrase):
    want = O.bigbwt(t, w, p, flags)
    ctx.set_max_phrase(max_phrase)
    got = ctx.bigbwt(t, w, p, flags)
    assert np.array_equal(got["bwt"], want["bwt"])
    if flags & 1:
        assert np.array_equal(pkg.unpack5(got["sa"]), want["sa"])
    if flags & 6:
        assert np.array_equal(pkg.unpack5(got["ssa"]).reshape(-1, 2), want["ssa"])
        assert np.array_equal(pkg.unpack5(got["esa"]).reshape(-1, 2), want["esa"])
    return want


@pytest.mark.gpu
def test_extra_trigger_never_fires_on_the_first_window(pkg, O, ctx):
    with gzip.open(os.path.join(ROOT, "tests", "golden", "runs_first_window.bin.gz"), "rb") as fh:
        t = np.frombuffer(fh.read(), dtype=np.uint8)
    for mp in (0, 700, 32768):
        check(pkg, O, ctx, t, 10, 100, 0, mp)
    ctx.set_max_phrase(32768)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [11, 12, 13])
def test_structured_random_inputs_match_oracle(pkg, O, wctx, seed):
    ctx = wctx
    bits = 64 if "idx64" in os.environ.get("PYTEST_CURRENT_TEST", "") else 0
    d = importlib.import_module("bigbwt_amd.dist")
    rng = np.random.default_rng(seed)
    done = dist_done = 0
    try:
        for it in range(60):
            t = gen(rng, O)
            w = int(rng.choice([4, 5, 10, 17]))
            p = int(rng.choice([10, 11, 20, 100]))
            flags = int(rng.choice([0, 1, 6]))
            mp = int(rng.choice([0, 700, 32768])) if len(t) > 3000 else 0
            try:
                want = check(pkg, O, ctx, t, w, p, flags, mp)
            except (pkg.PfpError, RuntimeError) as ex:
                if "PFP_ESHORT" in str(ex) or "fewer than 2 phrases" in str(ex) or "orc_bigbwt" in str(ex):
                    continue        # one-phrase parses: the reference aborts on them too
                raise
            done += 1
            if it % 4 == 0 and len(t) > 2000 and flags in (0, 1):
                R = int(rng.choice([2, 3]))
                cuts = [0] + sorted(int(x) for x in rng.choice(np.arange(400, len(t) - 400), size=R - 1, replace=False)) + [len(t)]
                ctxs = [pkg.Context(0) for _ in range(R)]
                try:
                    for c in ctxs:
                        c.set_max_phrase(2000)
                        c.set_index_bits(bits)
                    shards = [torch.from_numpy(t[cuts[r]:cuts[r + 1]].copy()).cuda() for r in range(R)]
                    try:
                        res = d.simulate(ctxs, shards, w, p, flags, halo=1 << 16)
                    except pkg.PfpError as ex:
                        if "halo" in str(ex) or "PFP_ESHORT" in str(ex):
                            continue
                        raise
                    assert np.array_equal(torch.cat([r["bwt"] for r in res]).cpu().numpy(), want["bwt"])
                    if flags:
                        assert np.array_equal(torch.cat([r["sa"] for r in res]).cpu().numpy().astype(np.uint64)[1:], want["sa"])
                    dist_done += 1
                finally:
                    for c in ctxs:
                        c.close()
    finally:
        ctx.set_max_phrase(32768)
    assert done >= 40 and dist_done >= 3


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["keysonly_sa_536", "keysonly_sa_164", "keysonly_686"])
def test_keys_only_first_round_on_small_dictionaries(pkg, O, ctx, monkeypatch, case):
    """Inputs the fuzz drivers found (tools/fuzz.py, tools/fuzz_sa.py with PFP_KEYSONLY=1): near-constant periodic
    texts whose first-round keys fill the word up to bit 64.  rocPRIM sorts up to 2^20 elements by merging with a
    comparator mask of (1 << end_bit) - 1, i.e. a shift by 64 there, and ordered the words by their index bits; the
    library now sorts the whole word for small inputs (prims.hip sort_keys_db)."""
    import os
    t = np.load(os.path.join(os.path.dirname(__file__), "golden", "fuzzcases", case + ".npy"))
    monkeypatch.setenv("PFP_KEYSONLY", "1")
    checked = 0
    for w, p in ((4, 10), (10, 10), (10, 20), (10, 100)):
        try:
            pr = O.parse(t, w, p)
        except Exception:
            continue
        a = ctx.gsacak(pr["dict"])
        b, _ = O.gsacak(pr["dict"], want_lcp=False)
        assert np.array_equal(a, b), (w, p)
        try:
            want = O.bigbwt(t, w, p, O.FLAG_SA)
        except RuntimeError:
            continue
        got = ctx.bigbwt(t, w, p, pkg.FLAG_SA)
        assert np.array_equal(got["bwt"], want["bwt"]) and np.array_equal(pkg.unpack5(got["sa"]), want["sa"]), (w, p)
        checked += 1
    assert checked
