"""Full-size parity (-m gpu): the HIP path on the benchmark workloads themselves, against digests the
real reference produced for the same texts in the build container (tests/golden/golden_full.json, made
by tests/golden/make_golden_full.py), and SURVEY.md section 4's GEN(12e6,16,1e-3,42) digests."""
import hashlib
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def sha_dev(t):
    h = hashlib.sha256()
    for s in range(0, t.numel(), 1 << 28):
        h.update(t[s:s + (1 << 28)].cpu().numpy().tobytes())
    return h.hexdigest()


def run_dev(pkg, ctx, text, w, p, flags):
    """device-resident chain + reference file formats, all through the C ABI"""
    import torch
    n = text.numel()
    bwt = torch.empty(n + 17, dtype=torch.uint8, device=text.device)
    sa = torch.empty(n + 1, dtype=torch.int64, device=text.device) if flags else None
    used = ctx.bigbwt_dev(text.data_ptr(), n, bwt.data_ptr(), sa.data_ptr() if flags else None, w, p, flags)
    assert used == n
    out = {"bwt": sha_dev(bwt[: n + 1])}
    if flags & pkg.FLAG_SA:
        packed = torch.empty(5 * n + 16, dtype=torch.uint8, device=text.device)
        ctx.pack5_dev(sa.data_ptr() + 8, n, packed.data_ptr())
        out["sa"] = sha_dev(packed[: 5 * n])
    for key, flag, run_end in (("ssa", pkg.FLAG_SSA, False), ("esa", pkg.FLAG_ESA, True)):
        if flags & flag:
            k = ctx.sample_runs_dev(bwt.data_ptr(), sa.data_ptr(), n + 1, 0, -1, -1, run_end)
            pairs = torch.empty(10 * k + 16, dtype=torch.uint8, device=text.device)
            assert ctx.sample_runs_dev(bwt.data_ptr(), sa.data_ptr(), n + 1, 0, -1, -1, run_end, pairs.data_ptr(), k) == k
            out[key] = sha_dev(pairs[: 10 * k])
    return out


def run_formats(pkg, ctx, text, w, p, flags):
    """the same through pfp_bigbwt_formats_dev (SA values stay inside the call, the files come back as device buffers)"""
    import torch
    n = text.numel()
    bwt = torch.empty(n + 17, dtype=torch.uint8, device=text.device)
    used, outs = ctx.bigbwt_formats_dev(text.data_ptr(), n, bwt.data_ptr(), w, p, flags)
    assert used == n
    res = {"bwt": sha_dev(bwt[: n + 1])}
    for key, (ptr, nbytes) in outs.items():
        res[key] = hashlib.sha256(ctx.fetch_dev(ptr, nbytes).tobytes()).hexdigest()
        ctx.dev_free(ptr)
    return res


def check(got, g):
    for key in ("bwt", "sa", "ssa", "esa"):
        if key + "_sha256" in g:
            assert got[key] == g[key + "_sha256"], key


def test_survey_gen16_digests(O, pkg, ctx, golden_full):
    """GEN(12e6,16,1e-3,42), 195 MB: .bwt b53bfd57..., .ssa b16eb8c4..., .esa dc6dc812..."""
    import torch
    g = golden_full["gen16"]
    assert g["bwt_sha256"].startswith("b53bfd57") and g["ssa_sha256"].startswith("b16eb8c4") and g["esa_sha256"].startswith("dc6dc812")
    text = O.gen_fasta(12_000_000, 16, 1e-3, 42)
    assert hashlib.sha256(text.tobytes()).hexdigest() == g["text_sha256"]
    check(run_dev(pkg, ctx, torch.from_numpy(text).cuda(), g["w"], g["p"], g["flags"]), g)


@pytest.mark.parametrize("name", ["small", "c2", "c2r", "c3", "c4s", "c5s"])
def test_benchmark_workloads_match_reference_digests(pkg, ctx, synth, golden_full, name):
    """BASELINE configs[1] (c2), configs[2] (c3, -s -e) at full size; configs[3]/[4] flag sets (-S; -w 12 -p 200 -s)
    on 16-copy stand-ins; c2r = configs[1] with a chromosome's repeat structure (interspersed families, satellite arrays of
    17 K near-identical monomers, microsatellites: word families in the thousands, round 4)"""
    import torch
    if name not in golden_full:
        pytest.skip("no reference digest committed for this workload")
    g = golden_full[name]
    text = synth.workload_text_torch(torch.device("cuda", 0), name)
    assert text.numel() == g["n"] and sha_dev(text) == g["text_sha256"]          # torch evaluator == numpy evaluator
    check(run_dev(pkg, ctx, text, g["w"], g["p"], g["flags"]), g)
    if g["flags"]:
        got = run_formats(pkg, ctx, text, g["w"], g["p"], g["flags"])
        assert set(got) == {"bwt"} | {k for k, bit in (("sa", 1), ("ssa", 2), ("esa", 4)) if g["flags"] & bit}
        check(got, g)


@pytest.mark.parametrize("name,ranks", [("c4s", 4), ("c5s", 3), ("c3", 2), ("c2", 3), ("c2r", 2)])
def test_benchmark_workloads_on_several_ranks(pkg, synth, golden_full, tmp_path, monkeypatch, name, ranks):
    """pfp_bigbwt_files_multi (the host of `bigbwt -G N`): BASELINE configs[3] / configs[4] flag sets on their 0.2 GB stand-ins and
    configs[2] and configs[1] (non-repetitive, an 18 Mb run of N across a rank boundary) at full size, split over rank threads that
    share this card (device copies in place of RCCL): the FILES against the reference's digests"""
    import importlib
    if name not in golden_full:
        pytest.skip("no reference digest committed for this workload")
    monkeypatch.setenv("PFP_MULTI_LOOPBACK", "1")
    pfpmod = importlib.import_module("bigbwt_amd.pfp")
    g = golden_full[name]
    text = synth.workload_text_np(name)
    assert len(text) == g["n"]
    base = "/dev/shm/pfp_multi_%s_%d" % (name, os.getpid())
    try:
        st = pfpmod.bigbwt_files_multi(text, base, [0] * ranks, g["w"], g["p"], g["flags"])
        assert st["ranks"] == ranks and st["n"] == g["n"]
        # (c2r: the satellite arrays' families outlast a share's pivot rounds - the range reports "needs doubling", every rank
        #  then sorts the whole dictionary and the output is cut in equal slices: the fallback of DESIGN.md section 6)
        assert st["sa_shares"] == (ranks if name != "c2r" else st["sa_shares"]) and st["sa_shares"] in (1, ranks)
        if name not in ("c2", "c2r"):          # (the copies of a collection: the parse's suffix array is sorted in shares too)
            assert st["parse_shares"] == ranks
        for key, ext, bit in (("bwt", ".bwt", 0), ("sa", ".sa", 1), ("ssa", ".ssa", 2), ("esa", ".esa", 4)):
            if bit == 0 or g["flags"] & bit:
                h = hashlib.sha256()
                with open(base + ext, "rb") as fh:
                    for blk in iter(lambda: fh.read(1 << 24), b""):
                        h.update(blk)
                assert os.path.getsize(base + ext) == g[key + "_bytes"] and h.hexdigest() == g[key + "_sha256"], (name, ext)
    finally:
        for ext in (".bwt", ".sa", ".ssa", ".esa"):
            if os.path.exists(base + ext):
                os.unlink(base + ext)


def test_north_star_workload_on_8_ranks(pkg, ctx, synth, golden_full, monkeypatch):
    """The 12.6 GB workload (-s) through the multi-GPU chain: 8 rank threads of pfp_bigbwt_files_multi sharing this card (device
    copies in place of RCCL) - byte-range scan with halos, hash-partitioned dedup, the suffix array of the 1.7 GB union dictionary
    in 8 key ranges, 8 output ranges written with pwrite.  The files against the digests of the reference's 42-minute run."""
    import importlib
    import torch
    if "huge_s" not in golden_full:
        pytest.skip("no reference digest committed for this workload")
    g = golden_full["huge_s"]
    dev = torch.device("cuda", 0)
    free, _total = torch.cuda.mem_get_info(dev)
    vfs = os.statvfs("/dev/shm")
    if free < 230 * (1 << 30) or vfs.f_bavail * vfs.f_frsize < 16 * (1 << 30):
        pytest.skip("needs about 230 GB of free device memory and 16 GB in /dev/shm")
    ctx.pool_trim()
    text = synth.workload_text_torch(dev, "huge_s").cpu().numpy()
    torch.cuda.empty_cache()
    assert len(text) == g["n"]
    monkeypatch.setenv("PFP_MULTI_LOOPBACK", "1")
    pfpmod = importlib.import_module("bigbwt_amd.pfp")
    base = "/dev/shm/pfp_multi_huge_%d" % os.getpid()
    try:
        st = pfpmod.bigbwt_files_multi(text, base, [0] * 8, g["w"], g["p"], g["flags"])
        del text
        assert st["ranks"] == 8 and st["n"] == g["n"] and st["sa_shares"] == 8 and st["parse_shares"] == 8
        for key, ext in (("bwt", ".bwt"), ("ssa", ".ssa")):
            h = hashlib.sha256()
            with open(base + ext, "rb") as fh:
                for blk in iter(lambda: fh.read(1 << 26), b""):
                    h.update(blk)
            assert os.path.getsize(base + ext) == g[key + "_bytes"] and h.hexdigest() == g[key + "_sha256"], ext
    finally:
        for ext in (".bwt", ".ssa"):
            if os.path.exists(base + ext):
                os.unlink(base + ext)


def test_north_star_12_6_gb_with_sampled_sa(pkg, ctx, synth, golden_full):
    """The north star's workload on one GPU: 1024 mutated copies, 12.6 GB, -w 10 -p 100 -s.  The reference took 42
    minutes for it in the build container (tests/golden/make_golden_huge.py); its .bwt and .ssa digests against
    the device results of pfp_bigbwt_formats_dev (the SA values stay inside the call: 8 bytes per run boundary)."""
    import torch
    if "huge_s" not in golden_full:
        pytest.skip("no reference digest committed for this workload")
    g = golden_full["huge_s"]
    dev = torch.device("cuda", 0)
    free, _total = torch.cuda.mem_get_info(dev)
    if free < 230 * (1 << 30):
        pytest.skip("needs about 230 GB of free device memory")
    ctx.pool_trim()
    text = synth.workload_text_torch(dev, "huge_s")
    torch.cuda.empty_cache()
    assert text.numel() == g["n"] and sha_dev(text) == g["text_sha256"]
    try:
        got = run_formats(pkg, ctx, text, g["w"], g["p"], g["flags"])
        assert set(got) == {"bwt", "ssa"}
        check(got, g)
        assert ctx.stats()["hard_big_groups"] > 0 and ctx.stats()["hard_minor_groups"] > 0      # every hard-group path ran
    finally:
        del text
        ctx.pool_trim()
        torch.cuda.empty_cache()


@pytest.mark.parametrize("name", ["c1", "c1_gen"])
def test_baseline_configs0_with_the_sacak_check(O, pkg, ctx, synth, golden_full, name):
    """BASELINE configs[0] (the yeast-sized plumbing case): `bigbwt -w 10 -p 100 --sum -c` - the C driver's own whole-text SACA-K
    BWT (simplebwt.c:28-100 -> pfp_sacak) must equal the .bwt ("BWTs match", bigbwt:177-194) and both the reference's digest.
    c1 = the synth-family text, c1_gen = BASELINE.md section 5's literal GEN(12.1e6,1,0,1)."""
    import subprocess
    g = golden_full[name]
    assert g["check_bwts_match"]
    text = synth.workload_text_np("c1") if name == "c1" else O.gen_fasta(12_100_000, 1, 0.0, 1)
    assert len(text) == g["n"] and hashlib.sha256(text.tobytes()).hexdigest() == g["text_sha256"]
    f = "/dev/shm/pfp_c1_%s_%d" % (name, os.getpid())
    try:
        with open(f, "wb") as fh:
            fh.write(text.tobytes())
        exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "big-bwt_amd", "bigbwt")
        out = subprocess.run([exe, "-w", str(g["w"]), "-p", str(g["p"]), "--sum", "-c", f], capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stdout + out.stderr
        assert "BWTs match" in out.stdout and g["bwt_sha256"] in out.stdout, out.stdout
        with open(f + ".bwt", "rb") as fh:
            assert hashlib.sha256(fh.read()).hexdigest() == g["bwt_sha256"]
    finally:
        for ext in ("", ".bwt", ".Bwt", ".log"):
            if os.path.exists(f + ext):
                os.unlink(f + ext)


def test_full_sa_with_values_beyond_4g(pkg, ctx, synth, golden_full):
    """BASELINE configs[3]'s flag set (-w 10 -p 100 -S) where the 5-byte integers matter: 512 copies, 6.3 GB > 2^32 bytes, two
    thousand million SA values with a non-zero fifth byte (utils.c:112-129).  The 31.5 GB .sa of pfp_bigbwt_formats_dev and the
    .bwt against the digests of the reference's files (tests/golden/make_golden_huge.py big_S: 25 minutes of one core)."""
    import torch
    if "big_S" not in golden_full:
        pytest.skip("no reference digest committed for this workload")
    g = golden_full["big_S"]
    assert g["sa_values_over_4g"] > 10**9 and g["sa_bytes"] == 5 * g["n"]
    dev = torch.device("cuda", 0)
    free, _total = torch.cuda.mem_get_info(dev)
    if free < 230 * (1 << 30):
        pytest.skip("needs about 230 GB of free device memory")
    ctx.pool_trim()
    text = synth.workload_text_torch(dev, "big_S")
    torch.cuda.empty_cache()
    assert text.numel() == g["n"] and sha_dev(text) == g["text_sha256"]
    try:
        n = text.numel()
        bwt = torch.empty(n + 17, dtype=torch.uint8, device=dev)
        used, outs = ctx.bigbwt_formats_dev(text.data_ptr(), n, bwt.data_ptr(), g["w"], g["p"], g["flags"])
        del text
        assert used == n and set(outs) == {"sa"}
        assert sha_dev(bwt[: n + 1]) == g["bwt_sha256"]
        ptr, nbytes = outs["sa"]
        assert nbytes == g["sa_bytes"]
        h = hashlib.sha256()
        top = 0
        for off in range(0, nbytes, 5 << 26):          # (pieces of whole 5-byte values)
            piece = ctx.fetch_dev(ptr + off, min(5 << 26, nbytes - off))
            h.update(piece.tobytes())
            top = max(top, int(piece[4::5].max()))
        ctx.dev_free(ptr)
        assert top == g["sa_top_byte_max"] and h.hexdigest() == g["sa_sha256"]
    finally:
        ctx.pool_trim()
        torch.cuda.empty_cache()


def test_full_sa_beyond_4g_on_8_ranks(pkg, ctx, synth, golden_full, monkeypatch):
    """the same 6.3 GB -S job through the multi-GPU chain: 8 rank threads of pfp_bigbwt_files_multi on this card, every rank
    packs and pwrites its range of the 31.5 GB .sa (values above 2^32 in most ranges)"""
    import importlib
    import torch
    if "big_S" not in golden_full:
        pytest.skip("no reference digest committed for this workload")
    g = golden_full["big_S"]
    dev = torch.device("cuda", 0)
    free, _total = torch.cuda.mem_get_info(dev)
    vfs = os.statvfs("/dev/shm")
    if free < 230 * (1 << 30) or vfs.f_bavail * vfs.f_frsize < 48 * (1 << 30):
        pytest.skip("needs about 230 GB of free device memory and 48 GB in /dev/shm")
    ctx.pool_trim()
    text = synth.workload_text_torch(dev, "big_S").cpu().numpy()
    torch.cuda.empty_cache()
    assert len(text) == g["n"]
    monkeypatch.setenv("PFP_MULTI_LOOPBACK", "1")
    pfpmod = importlib.import_module("bigbwt_amd.pfp")
    base = "/dev/shm/pfp_multi_bigS_%d" % os.getpid()
    try:
        st = pfpmod.bigbwt_files_multi(text, base, [0] * 8, g["w"], g["p"], g["flags"])
        del text
        assert st["ranks"] == 8 and st["n"] == g["n"]
        for key, ext in (("bwt", ".bwt"), ("sa", ".sa")):
            h = hashlib.sha256()
            with open(base + ext, "rb") as fh:
                for blk in iter(lambda: fh.read(1 << 26), b""):
                    h.update(blk)
            assert os.path.getsize(base + ext) == g[key + "_bytes"] and h.hexdigest() == g[key + "_sha256"], ext
    finally:
        for ext in (".bwt", ".sa"):
            if os.path.exists(base + ext):
                os.unlink(base + ext)


@pytest.mark.parametrize("name", ["big_w12", "huge_w12"])
def test_configs4_flag_set_on_12_6_gb(pkg, ctx, synth, golden_full, name):
    """BASELINE configs[4]'s flag set (-w 12 -p 200 -s) at the north star's size: the 12.6 GB, 1024-copy text.  Parsed as -p 200
    says, its dictionary is 3.0 GB - beyond the reference's 32-bit merger (its digests come from pfbwtNT64.x: 33 minutes, 51 GB of
    host memory) and beyond what one GPU holds (PFP_ENOMEM in round 4's first attempt).  The fused chain halves the phrase length of
    such a collection by itself (pfp_set_parse_density: 1.7 GB of dictionary) and fits.  .bwt and .ssa against the reference's
    digests (they equal the -w 10 -p 100 ones: the outputs do not depend on the parse)."""
    import torch
    if name not in golden_full:
        pytest.skip("no reference digest committed for this workload")
    g = golden_full[name]          # (big_w12: the same flag set on 512 copies, 6.3 GB - the reference's 32-bit merger, 21 minutes)
    assert g["w"] == 12 and g["p"] == 200 and g["bwt_sha256"] == golden_full["huge_s" if name == "huge_w12" else "big_S"]["bwt_sha256"]
    dev = torch.device("cuda", 0)
    free, _total = torch.cuda.mem_get_info(dev)
    if free < 260 * (1 << 30):
        pytest.skip("needs about 260 GB of free device memory")
    ctx.pool_trim()
    text = synth.workload_text_torch(dev, name)
    torch.cuda.empty_cache()
    assert text.numel() == g["n"] and sha_dev(text) == g["text_sha256"]
    try:
        got = run_formats(pkg, ctx, text, g["w"], g["p"], g["flags"])
        assert set(got) == {"bwt", "ssa"}
        check(got, g)
        st = ctx.stats()
        assert st["parse_density"] > 2.0 and st["dict_size"] < (1 << 31), st
    finally:
        del text
        ctx.pool_trim()
        torch.cuda.empty_cache()


def test_pack_and_sample_exports_on_slices(O, pkg, ctx):
    """pfp_pack5_dev / pfp_sample_runs_dev: slices with one halo byte on each side concatenate to the whole file"""
    import torch
    text = O.gen_fasta(40000, 3, 0.002, 11)
    want = O.bigbwt(text, 10, 100, 6)
    n = len(text)
    dev = torch.device("cuda", 0)
    bwt = torch.from_numpy(want["bwt"]).to(dev)
    full = O.bigbwt(text, 10, 100, 1)["sa"]
    sa = torch.from_numpy(np.concatenate([[n], full]).astype(np.int64)).to(dev)
    packed = torch.empty(5 * n, dtype=torch.uint8, device=dev)
    ctx.pack5_dev(sa.data_ptr() + 8, n, packed.data_ptr())
    assert np.array_equal(packed.cpu().numpy(), O.pack5(full))
    for odd in (1, 3, 7):          # short, unaligned packs
        ctx.pack5_dev(sa.data_ptr() + 8 * odd, odd, packed.data_ptr())
        assert np.array_equal(packed[: 5 * odd].cpu().numpy(), O.pack5(sa[odd: 2 * odd].cpu().numpy().astype(np.uint64)))
    for run_end, key in ((False, "ssa"), (True, "esa")):
        cuts = [0, 1, 17, 4096, 4097, n // 3, n // 3 + 1, n - 5, n, n + 1]
        parts = []
        for lo, hi in zip(cuts[:-1], cuts[1:]):
            left = -1 if lo == 0 else int(want["bwt"][lo - 1])
            right = -1 if hi == n + 1 else int(want["bwt"][hi])
            k = ctx.sample_runs_dev(bwt.data_ptr() + lo, sa.data_ptr() + 8 * lo, hi - lo, lo, left, right, run_end)
            buf = torch.empty(10 * k + 16, dtype=torch.uint8, device=dev)
            assert ctx.sample_runs_dev(bwt.data_ptr() + lo, sa.data_ptr() + 8 * lo, hi - lo, lo, left, right, run_end, buf.data_ptr(), k) == k
            parts.append(buf[: 10 * k].cpu().numpy())
        got = pkg.unpack5(np.concatenate(parts)).reshape(-1, 2)
        assert np.array_equal(got, want[key]), key
    with pytest.raises(pkg.PfpError):
        buf = torch.empty(64, dtype=torch.uint8, device=dev)
        ctx.sample_runs_dev(bwt.data_ptr(), sa.data_ptr(), n + 1, 0, -1, -1, False, buf.data_ptr(), 1)


@pytest.mark.gpu
def test_dictionary_over_4gib_takes_the_wide_build():
    """a 4.33 GB non-repetitive text: dictionary of 4.78 GB, so positions and slots are 64 bits wide (the reference switches
    to its -DM64 executables there, bigbwt:109-151).  Checked by size-independent properties (tools/check_wide.py: the BWT is
    a permutation of text + EOS; a second parse with another window and modulus gives the same BWT byte for byte) - no
    reference digests exist for this size (the reference needs > 64 GiB of host memory for it).  Needs most of the GPU."""
    import json
    import subprocess
    import sys
    import torch
    free, _total = torch.cuda.mem_get_info(0)
    if free < (245 << 30):
        pytest.skip(f"needs ~235 GB of free device memory, {free >> 30} GB free")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "check_wide.py")], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    r = json.loads(out.stdout.strip().splitlines()[-1])
    assert r["all_ok"] and r["index_bits"] == 64 and r["dict_over_4GiB"] and r["second_parse"]["index_bits"] == 64, r
    assert r["peak_device_bytes"] <= 215e9, r["peak_device_bytes"]          # round 2: 274 GB
    assert r["warm_pool"]["MBps"] >= 2000, r["warm_pool"]


@pytest.mark.gpu
def test_dictionary_between_2_and_4_gib_on_one_gpu():
    """200 copies at 3 % SNPs: a 2.4 GB dictionary (>= 2^31, < 2^32 bytes) on ONE GPU.  The narrow build has no spare bit in a
    position for the sorter's settled flag there (doubling rounds only: 2.0 s); by size the chain now takes the wide build where
    its ~96 bytes per dictionary byte are free (pivot rounds: 1.25 s) - the reference switches to its 64-bit executables at
    2^31 - 4 too (bigbwt:130).  All three selections give the same BWT; the automatic one is bounded in time."""
    import json
    import subprocess
    import sys
    import torch
    free, _total = torch.cuda.mem_get_info(torch.device("cuda", 0))
    if free < 250 * (1 << 30):
        pytest.skip("needs about 250 GB of free device memory")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "check_width_agree.py"), "wide31"], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    j = json.loads(out.stdout.strip().splitlines()[-1])
    r = j["runs"]
    assert j["agree"] and (1 << 31) <= r["0"]["dict_bytes"] < (1 << 32)
    assert r["0"]["index_bits"] == 64 and r["32"]["index_bits"] == 32 and r["64"]["index_bits"] == 64
    assert r["0"]["s_warm"] <= 1.6, r          # (1.25 s measured; the narrow build's doubling rounds take 2.0 s)


@pytest.mark.gpu
def test_union_dictionary_between_2_and_4_gib_across_ranks():
    """Multi-GPU chain, union dictionary of 2.4 GB (>= 2^31 bytes, < 2^32): the 32-bit build has no spare bit in a position for the
    sorter's settled flag there and a share of the suffix array cannot fall back on doubling rounds, so the shares take the wide
    build.  (Before round 3's fix the flag was granted by the share's size and positions >= 2^31 lost their top bit: "a dictionary
    word was claimed by no share".)  Two virtual ranks x 100 copies at 3 % SNPs against the single-GPU chain on the concatenation."""
    import json
    import subprocess
    import sys
    import torch
    free, _total = torch.cuda.mem_get_info(torch.device("cuda", 0))
    if free < 200 * (1 << 30):
        pytest.skip("needs about 200 GB of free device memory")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "check_range31.py"), "2", "100"], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    j = json.loads(out.stdout.strip().splitlines()[-1])
    assert j["bwt_equal"] and j["complete"] and j["index_bits_multi"] == 64 and (1 << 31) <= j["union_dict_bytes"] < (1 << 32)

