"""GPU parity tests proper: the HIP path, called through the C ABI, against (a) the golden
vectors the real reference produced and (b) the CPU oracle on the same seeded inputs; at larger
sizes through size-independent properties of a BWT / suffix array."""
import hashlib
import os
import subprocess

import numpy as np
import pytest

from textgen import make_text

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


@pytest.mark.parametrize("idx", range(15))
def test_bigbwt_matches_reference_golden(golden, O, pkg, wctx, idx):
    """pfp_bigbwt == bigbwt -w W -p M [-S | -s -e] (bit-exact .bwt/.sa/.ssa/.esa)."""
    if idx >= len(golden):
        pytest.skip("no such case")
    c = golden[idx]
    text = make_text(c["spec"], O)
    ctx = wctx
    for flags in (0, 1, 6):
        r = c["runs"][str(flags)]
        got = ctx.bigbwt(text, c["w"], c["p"], flags)
        assert ctx.stats()["index_bits"] == (64 if "idx64" in os.environ.get("PYTEST_CURRENT_TEST", "") else 32)
        assert len(got["bwt"]) == r["bwt_len"]
        assert sha(got["bwt"]) == r["bwt_sha256"], (c["name"], flags, "bwt")
        if flags & 1:
            assert sha(got["sa"]) == r["sa_sha256"], (c["name"], "sa")
        if flags & 2:
            assert sha(got["ssa"]) == r["ssa_sha256"], (c["name"], "ssa")
        if flags & 4:
            assert sha(got["esa"]) == r["esa_sha256"], (c["name"], "esa")


@pytest.mark.parametrize("idx", range(15))
def test_stage_files_match_reference_golden(golden, O, pkg, wctx, idx):
    """pfp_parse / pfp_bwtparse / pfp_merge reproduce every temp file of the reference."""
    if idx >= len(golden):
        pytest.skip("no such case")
    c = golden[idx]
    text = make_text(c["spec"], O)
    r = c["runs"]["6"]
    ctx = wctx
    ps = ctx.parse(text, c["w"], c["p"], want_sai=True)
    for k in ("dict", "occ", "parse", "last", "sai"):
        assert sha(ps[k]) == r[k + "_sha256"], (c["name"], k)
    ilist, bwlast, bwsai = ctx.bwtparse(ps["parse"], ps["last"], ps["occ"], ps["sai"])
    assert sha(ilist) == r["ilist_sha256"] and sha(bwlast) == r["bwlast_sha256"] and sha(bwsai) == r["bwsai_sha256"]
    out = ctx.merge(ps["dict"], ps["occ"], ilist, bwlast, bwsai, c["w"], pkg.FLAG_SSA | pkg.FLAG_ESA)
    assert sha(out["bwt"]) == r["bwt_sha256"] and sha(out["ssa"]) == r["ssa_sha256"] and sha(out["esa"]) == r["esa_sha256"]
    out = ctx.merge(ps["dict"], ps["occ"], ilist, bwlast, bwsai, c["w"], pkg.FLAG_SA)
    assert sha(out["sa"]) == c["runs"]["1"]["sa_sha256"]
    out = ctx.merge(ps["dict"], ps["occ"], ilist, bwlast, None, c["w"], 0)
    assert sha(out["bwt"]) == c["runs"]["0"]["bwt_sha256"]


@pytest.mark.parametrize("fast_hash", [True, False])
@pytest.mark.parametrize("max_phrase", [0, 700, 3000])
def test_outputs_do_not_depend_on_the_parse(golden, O, pkg, ctx, max_phrase, fast_hash):
    """fused chain with adaptive extra triggers (giant-phrase splitting) on/off, cutting by its own window hash (round 4:
    pfp_set_window_hash) or by the reference's Karp-Rabin hash: identical .bwt/.sa/.ssa/.esa (the Q1
    case `kat_q1`, where the reference's first window triggers, included); the staged pfp_parse keeps the reference's parse."""
    try:
        ctx.set_max_phrase(max_phrase)
        ctx.set_window_hash(fast_hash)
        used = 0
        for c in golden:
            if c["n"] > 200000:
                continue
            text = make_text(c["spec"], O)
            for flags in (0, 1, 6):
                r = c["runs"][str(flags)]
                got = ctx.bigbwt(text, c["w"], c["p"], flags)
                used += ctx.stats()["extra_triggers"]
                assert sha(got["bwt"]) == r["bwt_sha256"], (c["name"], flags, max_phrase)
                if flags & 1:
                    assert sha(got["sa"]) == r["sa_sha256"]
                if flags & 2:
                    assert sha(got["ssa"]) == r["ssa_sha256"] and sha(got["esa"]) == r["esa_sha256"]
            ps = ctx.parse(text, c["w"], c["p"])
            assert sha(ps["dict"]) == c["runs"]["6"]["dict_sha256"]
        if not os.environ.get("PFP_PARSE_DENSITY"):          # (at a pinned higher density no phrase of these texts reaches 700 bytes)
            assert (used > 0) == (max_phrase > 0)
    finally:
        ctx.set_max_phrase(1 << 15)
        ctx.set_window_hash(True)


def test_phrase_length_follows_repetitiveness(O, pkg, ctx):
    """pfp_set_parse_density(0), the default: the fused chain cuts a collection of many near-identical copies twice as often as
    -p says (shorter phrases: a smaller dictionary, the outputs unchanged) and a single genome exactly as often; 1 pins -p"""
    coll = O.gen_fasta(120000, 40, 0.002, 31)         # 40 copies, 4.9 MB (the choice wants a thousand sampled cuts)
    single = O.gen_fasta(800000, 1, 0.0, 32)
    want_coll, want_single = O.bigbwt(coll, 10, 100, O.FLAG_SSA | O.FLAG_ESA), O.bigbwt(single, 10, 100, 0)
    try:
        seen = {}
        for dens in (0.0, 1.0, 2.0, 0.5):
            ctx.set_parse_density(dens)
            got = ctx.bigbwt(coll, 10, 100, pkg.FLAG_SSA | pkg.FLAG_ESA)
            st = ctx.stats()
            seen[dens] = (st["parse_density"], st["n_phrases"])
            assert np.array_equal(got["bwt"], want_coll["bwt"]), dens
            assert np.array_equal(pkg.unpack5(got["ssa"]).reshape(-1, 2), want_coll["ssa"]), dens
            assert np.array_equal(pkg.unpack5(got["esa"]).reshape(-1, 2), want_coll["esa"]), dens
        assert abs(seen[0.0][0] - 100 / 48) < 1e-6 and seen[1.0][0] == 1.0 and seen[2.0][0] == 2.0 and seen[0.5][0] == 0.5
        assert 1.6 < seen[2.0][1] / seen[1.0][1] < 2.4 and seen[0.0][1] >= seen[2.0][1] and seen[0.5][1] < seen[1.0][1]
        ctx.set_parse_density(0.0)
        got = ctx.bigbwt(single, 10, 100, 0)
        st = ctx.stats()
        assert st["parse_density"] == 1.0 and np.array_equal(got["bwt"], want_single["bwt"])
        ctx.set_parse_density(1.0)
        ctx.bigbwt(single, 10, 100, 0)
        assert ctx.stats()["n_phrases"] == st["n_phrases"]          # the nominal cuts of the doubled scan ARE the scan at 1 / p
    finally:
        ctx.set_parse_density(0.0)


def test_window_hash_parses_differently_but_as_densely(O, ctx):
    """the fused chain's window hash is not Karp-Rabin (different phrases) and cuts about as often (1 / p)"""
    text = O.gen_fasta(400000, 3, 0.001, 9)
    counts = {}
    try:
        for fast in (True, False):
            ctx.set_window_hash(fast)
            ctx.set_max_phrase(0)
            ctx.bigbwt(text, 10, 100, 0)
            st = ctx.stats()
            counts[fast] = (st["n_phrases"], st["n_words"], st["dict_size"])
    finally:
        ctx.set_window_hash(True)
        ctx.set_max_phrase(1 << 15)
    assert counts[True] != counts[False]
    assert 0.8 < counts[True][0] / counts[False][0] < 1.25, counts


def test_scan_matches_oracle(O, ctx):
    """K1/K2 against KR_window::addchar restated on the CPU, many (w,p) incl. even/odd/huge p."""
    text = O.gen_fasta(50000, 2, 0.001, 21)
    for w, p in [(4, 11), (5, 10), (8, 64), (10, 100), (12, 200), (16, 37), (17, 1000), (18, 50), (25, 33), (10, 1 << 33)]:
        ends, used = ctx.scan(text, w, p)
        want = O.scan(text, w, p)
        assert used == len(text)
        assert np.array_equal(ends, want), (w, p)


def test_scan_random_bytes_all_windows(O, ctx):
    """window hash over the full byte range 3..255 (Barrett reduction corner values)"""
    rng = np.random.default_rng(5)
    text = rng.integers(3, 256, size=200000, dtype=np.uint8)
    text[1000:1100] = 255
    text[5000:5100] = 3
    for w in (4, 7, 10, 13, 17):
        ends, _ = ctx.scan(text, w, 10)
        assert np.array_equal(ends, O.scan(text, w, 10)), w


def test_suffix_sorters_match_oracle(O, wctx):
    ctx = wctx
    rng = np.random.default_rng(7)
    s = np.concatenate([rng.integers(1, 50, size=30000), [0]]).astype(np.uint32)
    assert np.array_equal(ctx.sacak_int(s), O.sacak_int(s))
    rep = np.concatenate([np.tile(rng.integers(1, 5, size=100), 200), [0]]).astype(np.uint32)   # deep LCPs
    assert np.array_equal(ctx.sacak_int(rep), O.sacak_int(rep))
    t = np.concatenate([O.gen_fasta(20000, 3, 0.001, 9), [0]]).astype(np.uint8)
    assert np.array_equal(ctx.sacak(t), O.sacak(t))
    d = O.parse(O.gen_fasta(30000, 3, 0.002, 10), 10, 100)["dict"]
    sa, _ = O.gsacak(d, want_lcp=False)
    assert np.array_equal(ctx.gsacak(d), sa)
    ex = np.frombuffer(b"banana\x01anaba\x01anan\x01\x00", dtype=np.uint8)   # gsa/README.md:76-104
    assert np.array_equal(ctx.gsacak(ex), O.gsacak(ex, want_lcp=False)[0])
    # the -DM64 entry points (gsa/gsacak.h:42-60): the same arrays in 64-bit entries
    assert np.array_equal(ctx.gsacak64(d), sa.astype(np.uint64))
    assert np.array_equal(ctx.sacak64(t), O.sacak(t).astype(np.uint64))
    assert np.array_equal(ctx.sacak_int64(s), O.sacak_int(s).astype(np.uint64))
    # gsacak's optional LCP and DA outputs (gsa/gsacak.h:96-105, gsa/README.md:76-104)
    osa, olcp = O.gsacak(d, want_lcp=True)
    word_of = np.concatenate([[0], np.cumsum(d == 1)[:-1]])          # index of the string every position lies in
    for wide in (False, True):
        gsa, glcp, gda = ctx.gsacak_lcp_da(d, wide)
        assert np.array_equal(gsa.astype(np.uint32), osa) and np.array_equal(glcp.astype(np.int32), olcp)
        assert np.array_equal(gda.astype(np.int64), word_of[osa])
    esa, elcp, eda = ctx.gsacak_lcp_da(ex)
    assert elcp.tolist() == O.gsacak(ex, want_lcp=True)[1].tolist()


def test_lcp_of_long_repeats(O, wctx):
    """gsacak's LCP output on a collection full of long exact repeats - a 150 K run of one letter (149 999 ... 1 between
    neighbouring suffixes), the same 30 K string twice plus a prefix of it, a 80 K periodic string: nearly every pair shares
    more than the 2 KB one thread compares, and goes through the text-order pass (pipeline.hip lcp_long_kernel)"""
    rng = np.random.default_rng(5)
    r = rng.integers(65, 70, size=30000).astype(np.uint8)
    coll = np.concatenate([np.full(150000, ord("N"), np.uint8), [1], r, [1], np.tile(np.frombuffer(b"ACGT", np.uint8), 20000), [1], r, [1],
                           r[:20000], [1, 0]]).astype(np.uint8)
    osa, olcp = O.gsacak(coll, want_lcp=True)
    assert (olcp >= 2048).sum() > 250000 and olcp.max() == 149999
    for wide in (False, True):
        gsa, glcp, gda = wctx.gsacak_lcp_da(coll, wide)
        assert np.array_equal(gsa.astype(np.uint32), osa)
        assert np.array_equal(glcp.astype(np.int64), olcp.astype(np.int64))


def test_error_behaviour(pkg, ctx, O):
    text = O.gen_fasta(5000, 1, 0, 3)
    for kw, code in [(dict(w=3), -1), (dict(p=9), -1), (dict(flags=pkg.FLAG_SA | pkg.FLAG_SSA), -1)]:
        with pytest.raises(pkg.PfpError) as ei:
            ctx.bigbwt(text, **kw)
        assert ei.value.code == code
    with pytest.raises(pkg.PfpError) as ei:      # no trigger at all -> one phrase (bwtparse.c:244 asserts n>1)
        ctx.bigbwt(b"ACGTACGTAC", 10, 100)
    assert ei.value.code == -8
    with pytest.raises(pkg.PfpError) as ei:
        ctx.bigbwt(b"", 10, 100)
    assert ei.value.code == -8
    # still usable afterwards
    assert len(ctx.bigbwt(text, 10, 100)["bwt"]) == len(text) + 1


def test_special_byte_truncates_like_reference(O, ctx):
    """the reference stops reading at the first byte <= 2 and builds the BWT of the prefix."""
    text = O.gen_fasta(20000, 1, 0, 4).copy()
    text[12345] = 1
    got = ctx.bigbwt(text, 10, 100)
    want = O.bigbwt(text[:12345], 10, 100)
    assert np.array_equal(got["bwt"], want["bwt"]) and len(got["bwt"]) == 12346


def test_many_snp_variants_do_not_collide(O, pkg, wctx):
    """48 copies at 1 % SNP rate: thousands of phrase variants that differ in two bytes.  A weak
    phrase hash (high chunk byte reaching only the top byte of the term) produced verified
    collisions on exactly this shape; the dedup must stay exact and need no reseed."""
    text = O.gen_fasta(100000, 48, 0.01, 77)
    ctx = wctx
    got = ctx.bigbwt(text, 10, 100, pkg.FLAG_SA)
    assert ctx.stats()["hash_reseeds"] == 0
    want = O.bigbwt(text, 10, 100, O.FLAG_SA)
    assert np.array_equal(got["bwt"], want["bwt"]) and np.array_equal(pkg.unpack5(got["sa"]), want["sa"])


def test_mid_size_against_oracle(O, pkg, wctx):
    """~24 MB, 8 near-identical copies: every output against the oracle."""
    ctx = wctx
    text = O.gen_fasta(3000000, 8, 0.001, 31)
    got = ctx.bigbwt(text, 10, 100, pkg.FLAG_SSA | pkg.FLAG_ESA)
    want = O.bigbwt(text, 10, 100, O.FLAG_SSA | O.FLAG_ESA)
    assert np.array_equal(got["bwt"], want["bwt"])
    assert np.array_equal(pkg.unpack5(got["ssa"]).reshape(-1, 2), want["ssa"])
    assert np.array_equal(pkg.unpack5(got["esa"]).reshape(-1, 2), want["esa"])


@pytest.mark.parametrize("direct", ["1", "0"])
def test_position_records_per_slot_or_per_position(golden, O, pkg, wctx, monkeypatch, direct):
    """the merge's 16-byte records computed by every slot itself (what a rank's share of the multi-GPU chain and a dictionary
    too large for one record per position get) or written per position and gathered: same files, all three flag sets"""
    monkeypatch.setenv("PFP_PREC_DIRECT", direct)
    for c in golden[:8]:
        text = make_text(c["spec"], O)
        for flags in (0, 1, 6):
            r = c["runs"][str(flags)]
            got = wctx.bigbwt(text, c["w"], c["p"], flags)
            assert sha(got["bwt"]) == r["bwt_sha256"], (c["name"], flags, "bwt")
            if flags & 1:
                assert sha(got["sa"]) == r["sa_sha256"], (c["name"], "sa")
            if flags & 2:
                assert sha(got["ssa"]) == r["ssa_sha256"] and sha(got["esa"]) == r["esa_sha256"], (c["name"], "ssa/esa")


@pytest.mark.parametrize("env", [{}, {"PFP_BIG_BUDGET": "6000"}, {"PFP_BIG_BUDGET": "100"}, {"PFP_BIG_CAP": "2"}])
def test_large_hard_groups_without_a_dominating_char(O, pkg, wctx, monkeypatch, env):
    """1200 copies of a short random sequence, a few of them mutated, small window: suffixes of 5+ characters are shared
    by several words with ~1200 occurrences each and different preceding chars - hard groups of thousands of
    occurrences where no char dominates.  They are merged by one device-wide sort per chunk of groups (chunked by
    PFP_BIG_BUDGET occurrences; a group beyond the budget - every group with PFP_BIG_BUDGET=100 - takes the per-occurrence ranking kernel;
    PFP_BIG_CAP=2: the queue of such groups overflows and the pass is redone with one that fits)."""
    rng = np.random.default_rng(5)
    base = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=1500)]
    copies = np.tile(base, (1200, 1))
    for c_ in rng.integers(0, 1200, size=40):
        copies[c_, rng.integers(0, 1500)] = ord("N")
    text = copies.reshape(-1).copy()
    for k_, v_ in env.items():
        monkeypatch.setenv(k_, v_)
    for flags, oflags in ((0, 0), (pkg.FLAG_SA, O.FLAG_SA), (pkg.FLAG_SSA | pkg.FLAG_ESA, O.FLAG_SSA | O.FLAG_ESA)):
        got = wctx.bigbwt(text, 4, 11, flags)
        assert wctx.stats()["hard_big_groups"] > (2 if "PFP_BIG_CAP" in env else 0)      # (PFP_BIG_CAP=2: the queue did overflow)
        want = O.bigbwt(text, 4, 11, oflags)
        assert np.array_equal(got["bwt"], want["bwt"])
        if flags & pkg.FLAG_SA:
            assert np.array_equal(pkg.unpack5(got["sa"]), want["sa"])
        if flags & pkg.FLAG_SSA:
            assert np.array_equal(pkg.unpack5(got["ssa"]).reshape(-1, 2), want["ssa"])
            assert np.array_equal(pkg.unpack5(got["esa"]).reshape(-1, 2), want["esa"])


def test_out_of_device_memory_is_an_error_not_a_leak(O, pkg, monkeypatch):
    """PFP_TEST_POOL_LIMIT: the context's pool refuses requests beyond 32 .. 128 MB of live bytes, as a full device would.  A 12 MB text
    fails with PFP_ENOMEM somewhere inside the chain - whichever stage it reaches, at several limits - nothing stays allocated,
    and the same context then builds a small text's BWT bit-exact."""
    big = O.gen_fasta(3000000, 4, 0.002, 5)
    small = O.gen_fasta(40000, 3, 0.002, 6)
    want = O.bigbwt(small, 10, 100, O.FLAG_SSA | O.FLAG_ESA)
    for limit in list(range(26, 170, 9)):          # (every few MB another allocation site is the one that fails)
        monkeypatch.setenv("PFP_TEST_POOL_LIMIT", str(limit << 20))
        c = pkg.Context(0)
        try:
            for flags in (0, pkg.FLAG_SA, pkg.FLAG_SSA | pkg.FLAG_ESA):
                with pytest.raises(pkg.PfpError) as ei:
                    c.bigbwt(big, 10, 100, flags)
                assert ei.value.code == -7, (limit, flags, str(ei.value))          # PFP_ENOMEM
                assert c.mem_stats()["live"] == 0, (limit, flags)
            got = c.bigbwt(small, 10, 100, pkg.FLAG_SSA | pkg.FLAG_ESA)
            assert np.array_equal(got["bwt"], want["bwt"])
            assert np.array_equal(pkg.unpack5(got["ssa"]).reshape(-1, 2), want["ssa"])
        finally:
            c.close()


def test_hash_collisions_are_found_and_cured(O, pkg, monkeypatch):
    """PFP_TEST_HASH_BITS=8: the first attempt of the phrase dedup keeps 8 bits of every hash, so different phrases collide; the byte
    verification of every occurrence finds it (newscan.cpp:282 compares strings on every hit), the hash is reseeded and the result
    is exact.  -8: every attempt collides and the call fails with PFP_ECOLLISION like the reference (newscan.cpp:282-286), not wrong."""
    text = O.gen_fasta(200000, 4, 0.002, 77)
    want = O.bigbwt(text, 10, 100, O.FLAG_SSA | O.FLAG_ESA)
    monkeypatch.setenv("PFP_TEST_HASH_BITS", "8")
    c = pkg.Context(0)
    try:
        got = c.bigbwt(text, 10, 100, pkg.FLAG_SSA | pkg.FLAG_ESA)
        assert c.stats()["hash_reseeds"] >= 1
        assert np.array_equal(got["bwt"], want["bwt"])
        assert np.array_equal(pkg.unpack5(got["ssa"]).reshape(-1, 2), want["ssa"])
        monkeypatch.setenv("PFP_TEST_HASH_BITS", "-8")
        with pytest.raises(pkg.PfpError) as ei:
            c.bigbwt(text, 10, 100, 0)
        assert ei.value.code == -4          # PFP_ECOLLISION
        monkeypatch.delenv("PFP_TEST_HASH_BITS")
        assert np.array_equal(c.bigbwt(text, 10, 100, 0)["bwt"], want["bwt"])      # the context is usable afterwards
    finally:
        c.close()


def test_trigger_dense_text_overflows_the_scan_buffer(O, pkg, wctx):
    """a text of period 7 one of whose windows is a trigger: 300 K phrase ends in 2.1 MB where the fused scan kernel keeps room
    for 4 n / p + 64 K of them - the pass reports the true count and is repeated with room for all (scan.hip: cap_hint); one word,
    300 K occurrences, every group hard"""
    unit = np.frombuffer(b"AAAGCTA", dtype=np.uint8)
    assert len(O.scan(np.tile(unit, 60), 10, 100)) >= 50
    text = np.tile(unit, 300000)
    assert len(O.scan(text, 10, 100)) > 4 * len(text) // 100 + 65536

    def check(text):
        for flags, oflags in ((0, 0), (pkg.FLAG_SA, O.FLAG_SA), (pkg.FLAG_SSA | pkg.FLAG_ESA, O.FLAG_SSA | O.FLAG_ESA)):
            got = wctx.bigbwt(text, 10, 100, flags)
            assert wctx.stats()["n_phrases"] > 4 * len(text) // 100 + 65536
            want = O.bigbwt(text, 10, 100, oflags)
            assert np.array_equal(got["bwt"], want["bwt"])
            if flags & pkg.FLAG_SA:
                assert np.array_equal(pkg.unpack5(got["sa"]), want["sa"])
            if flags & pkg.FLAG_SSA:
                assert np.array_equal(pkg.unpack5(got["ssa"]).reshape(-1, 2), want["ssa"])
                assert np.array_equal(pkg.unpack5(got["esa"]).reshape(-1, 2), want["esa"])
    try:
        wctx.set_window_hash(False)          # that text is dense in Karp-Rabin triggers
        check(text)
        wctx.set_window_hash(True)           # the chain's own window hash: look for a period-7 text one of whose windows it cuts
        rng = np.random.default_rng(5)
        for _ in range(400):
            # (a random tail gives the reference's hash - the oracle's parse - something to cut as well)
            cand = np.concatenate([np.tile(np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, 7)], 300000), O.gen_fasta(20000, 1, 0.0, 77)[8:]])
            try:
                wctx.bigbwt(cand[:70000], 10, 100, 0)
            except pkg.PfpError:          # (no window of this period cuts at all: PFP_ESHORT like the reference, bwtparse.c:244)
                continue
            if wctx.stats()["n_phrases"] > 9000:
                check(cand)
                break
        else:
            pytest.fail("no trigger-dense periodic text found for the window hash")
    finally:
        wctx.set_window_hash(True)


def test_steady_state_calls_do_not_reach_the_driver(O, pkg):
    """the context's pool: a repeated call of the same size is served from cached blocks (no hipMalloc, no trim),
    and the device-format buffers handed out go back into the pool with pfp_dev_free"""
    import torch
    text = O.gen_fasta(300000, 6, 0.002, 23)
    dev = torch.device("cuda:0")
    t = torch.from_numpy(text).to(dev)
    bwt = torch.empty(len(text) + 1 + 16, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    c = pkg.Context(0)
    try:
        flags = pkg.FLAG_SSA | pkg.FLAG_ESA
        def call():
            used, outs = c.bigbwt_formats_dev(t.data_ptr(), len(text), bwt.data_ptr(), 10, 100, flags)
            got = {k: c.fetch_dev(ptr, nb) for k, (ptr, nb) in outs.items()}
            for ptr, _ in outs.values():
                c.dev_free(ptr)
            return used, got
        used, first = call()
        call()
        before = c.pool_counters()
        used2, again = call()
        after = c.pool_counters()
        assert after == before, (before, after)
        assert used == used2 == len(text) and all(np.array_equal(first[k], again[k]) for k in first)
        want = O.bigbwt(text, 10, 100, O.FLAG_SSA | O.FLAG_ESA)
        assert np.array_equal(bwt[: len(text) + 1].cpu().numpy(), want["bwt"])
        assert np.array_equal(pkg.unpack5(again["ssa"]).reshape(-1, 2), want["ssa"])
        assert np.array_equal(pkg.unpack5(again["esa"]).reshape(-1, 2), want["esa"])
    finally:
        c.close()


def _check_bwt_sa_properties(text, bwt, sa_vals, rng):
    n = len(text)
    assert len(bwt) == n + 1
    # BWT is a permutation of text + EOS
    hist_t = np.bincount(text, minlength=256); hist_t[0] += 1
    assert np.array_equal(np.bincount(bwt, minlength=256), hist_t)
    # SA[1..n] is a permutation of 0..n-1 and BWT[i] = T[SA[i]-1]
    sa = np.concatenate([[n], sa_vals]).astype(np.int64)
    chk = np.zeros(n + 1, dtype=bool); chk[sa] = True
    assert chk.all()
    prev = np.where(sa > 0, text[np.maximum(sa - 1, 0)], 0)
    assert np.array_equal(prev.astype(np.uint8), bwt)
    # sampled sortedness of adjacent suffixes
    tz = np.concatenate([text, np.zeros(1, np.uint8)])
    for i in rng.integers(0, n, size=3000):
        a, b = int(sa[i]), int(sa[i + 1])
        k = 0
        while tz[a + k] == tz[b + k]:
            k += 1
        assert tz[a + k] < tz[b + k]


def test_large_properties_full_sa(O, pkg, ctx):
    """~64 MB non-repetitive + N block (the shape of BASELINE config 2), checked through
    size-independent properties: permutation, BWT[i]=T[SA[i]-1], sampled order."""
    text = O.gen_fasta_fast(63000000, 1, 0, 2, n_blocks=[(20000000, 4000000), (50000000, 10000)])
    got = ctx.bigbwt(text, 10, 100, pkg.FLAG_SA)
    _check_bwt_sa_properties(text, got["bwt"], pkg.unpack5(got["sa"]), np.random.default_rng(1))
    plain = ctx.bigbwt(text, 10, 100, 0)
    assert np.array_equal(plain["bwt"], got["bwt"])


def test_device_resident_entry_point(O, pkg, ctx):
    import torch
    text = O.gen_fasta(400000, 6, 0.002, 17)
    dev = torch.device("cuda:0")
    t = torch.from_numpy(text).to(dev)
    bwt = torch.empty(len(text) + 1 + 16, dtype=torch.uint8, device=dev)
    sa = torch.empty(len(text) + 1, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    used = ctx.bigbwt_dev(t.data_ptr(), len(text), bwt.data_ptr(), sa.data_ptr(), 10, 100, pkg.FLAG_SA)
    assert used == len(text)
    want = O.bigbwt(text, 10, 100, O.FLAG_SA)
    assert np.array_equal(bwt[: len(text) + 1].cpu().numpy(), want["bwt"])
    assert np.array_equal(sa[1:].cpu().numpy().astype(np.uint64), want["sa"])
    assert int(sa[0]) == len(text)


def test_c_driver_end_to_end(golden, O, tmp_path):
    """the C bigbwt binary: outputs, -k temp files, --sum, -c"""
    c = {x["name"]: x for x in golden}["gen_small"]
    text = make_text(c["spec"], O)
    f = tmp_path / "t.fa"
    f.write_bytes(text.tobytes())
    exe = os.path.join(ROOT, "big-bwt_amd", "bigbwt")
    out = subprocess.run([exe, "-w", "10", "-p", "100", "-s", "-e", "--sum", "-c", str(f)], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "BWTs match" in out.stdout and c["runs"]["6"]["bwt_sha256"] in out.stdout
    for ext in ("bwt", "ssa", "esa"):
        assert sha(np.fromfile(str(f) + "." + ext, dtype=np.uint8)) == c["runs"]["6"][ext + "_sha256"]
    assert not os.path.exists(str(f) + ".dict")
    out = subprocess.run([exe, "-S", "-k", str(f)], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert sha(np.fromfile(str(f) + ".sa", dtype=np.uint8)) == c["runs"]["1"]["sa_sha256"]
    for ext in ("dict", "occ", "parse", "last", "sai", "ilist", "bwlast", "bwsai"):
        assert sha(np.fromfile(str(f) + "." + ext, dtype=np.uint8)) == c["runs"]["6"][ext + "_sha256"], ext


@pytest.mark.parametrize("where", ["memory_fs", "tmp_path"])
def test_c_driver_outputs_through_mapped_files(golden, O, tmp_path, where):
    """the .bwt and .sa whose sizes the text fixes are copied from HBM straight into the mapped pages of their files
    (pipeline.hip MappedOut: files of 64 MB and more in a memory file system; PFP_MAP_MIN_BYTES brings the small ones of this
    test in): same bytes as the pinned-buffer path, in /dev/shm and in a directory that may not be a memory file system (the
    ordinary path there)"""
    import shutil, tempfile
    c = {x["name"]: x for x in golden}["gen_small"]
    text = make_text(c["spec"], O)
    d = tempfile.mkdtemp(dir="/dev/shm") if where == "memory_fs" else str(tmp_path)
    try:
        f = os.path.join(d, "t.fa")
        with open(f, "wb") as fh:
            fh.write(text.tobytes())
        exe = os.path.join(ROOT, "big-bwt_amd", "bigbwt")
        env = dict(os.environ, PFP_MAP_MIN_BYTES="1", PFP_TRACE_HOST="1")
        for flag, run in (("-S", "1"), ("-s", "6"), ("-e", "6")):
            out = subprocess.run([exe, "-w", "10", "-p", "100", flag, f], capture_output=True, text=True, env=env)
            assert out.returncode == 0, out.stdout + out.stderr
            if where == "memory_fs":
                assert "2 of them straight" in out.stderr, out.stderr
            assert sha(np.fromfile(f + ".bwt", dtype=np.uint8)) == c["runs"][run]["bwt_sha256"]
            ext = {"-S": "sa", "-s": "ssa", "-e": "esa"}[flag]
            assert sha(np.fromfile(f + "." + ext, dtype=np.uint8)) == c["runs"][run][ext + "_sha256"]
        if where == "memory_fs":
            # a piece the runtime refuses to register: the whole file takes the pwrite path, same bytes (PFP_MAP_PIECE_MB=2 and a
            # 5 MB text: piece 1 exists for the .sa)
            big = np.tile(text, 5_000_000 // len(text) + 1)[:5_000_000]
            with open(f, "wb") as fh:
                fh.write(big.tobytes())
            want = O.bigbwt(big, 10, 100, 1)
            for fail in ("0", "1"):
                env2 = dict(env, PFP_TEST_MAP_FAIL=fail, PFP_MAP_PIECE_MB="2")
                out = subprocess.run([exe, "-w", "10", "-p", "100", "-S", f], capture_output=True, text=True, env=env2)
                assert out.returncode == 0, out.stdout + out.stderr
                assert "2 of them straight" not in out.stderr
                assert np.array_equal(np.fromfile(f + ".bwt", dtype=np.uint8), want["bwt"])
                sa5 = np.fromfile(f + ".sa", dtype=np.uint8).reshape(-1, 5)
                got_sa = sum(sa5[:, k].astype(np.uint64) << np.uint64(8 * k) for k in range(5))
                assert np.array_equal(got_sa, want["sa"])
        # a text too short for the parameters: the early file does not stay behind
        with open(f, "wb") as fh:
            fh.write(b"ACGT")
        for ext in ("bwt", "sa"):
            if os.path.exists(f + "." + ext):
                os.remove(f + "." + ext)
        out = subprocess.run([exe, "-w", "10", "-p", "100", "-S", f], capture_output=True, text=True, env=env)
        if out.returncode != 0:
            assert not os.path.exists(f + ".bwt") and not os.path.exists(f + ".sa")
    finally:
        if where == "memory_fs":
            shutil.rmtree(d, ignore_errors=True)


@pytest.mark.parametrize("n", [1, 2, 63, 64, 65, 2047, 2049, 4096, 4097, 50000, 300001, 2_500_000, 9_000_000])
def test_first_round_sort_against_numpy(ctx, n):
    """csrc/radix.hip (the hand-written first-round sort: partition passes + bucket sorts in LDS) against numpy's stable sort:
    pairs and keys only, bit ranges as the suffix sorter passes them, uniform / skewed / constant keys (oversize buckets go
    to the library, more than 4096 of them send the whole array there)"""
    rng = np.random.default_rng(n)
    for lo, hi, kind in ((0, 48, "uniform"), (0, 40, "skew"), (0, 64, "uniform"), (28, 64, "keysonly"), (0, 36, "const"), (0, 17, "few"), (3, 11, "uniform")):
        if kind == "uniform":
            keys = rng.integers(0, 1 << 63, n, dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, n, dtype=np.uint64)
        elif kind == "skew":          # half of the elements share one key, a tenth a second one
            keys = rng.integers(0, 1 << 40, n, dtype=np.uint64)
            keys[rng.random(n) < 0.5] = np.uint64(0x123456789)
            keys[rng.random(n) < 0.1] = np.uint64(0xFFFFFFFFFF)
        elif kind == "const":
            keys = np.full(n, 0x5A5A5A5A5, dtype=np.uint64)
        elif kind == "few":
            keys = rng.integers(0, 7, n, dtype=np.uint64) << np.uint64(10)
        else:                          # (key << 28 | index): the keys-only form of the suffix sorter
            keys = (rng.integers(0, 1 << 36, n, dtype=np.uint64) << np.uint64(28)) | np.arange(n, dtype=np.uint64)
        mask = np.uint64(((1 << hi) - 1) ^ ((1 << lo) - 1))
        order = np.argsort(keys & mask, kind="stable")
        if kind == "keysonly":
            got = keys.copy()
            ctx.debug_msd_sort(got, None, lo, hi)
            assert np.array_equal(got, keys[order]), (n, lo, hi, kind)
        else:
            got, vals = keys.copy(), np.arange(n, dtype=np.uint32)
            ctx.debug_msd_sort(got, vals, lo, hi)
            assert np.array_equal(vals, order.astype(np.uint32)), (n, lo, hi, kind)
            assert np.array_equal(got, keys[order]), (n, lo, hi, kind)
