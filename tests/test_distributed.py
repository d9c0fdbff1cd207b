"""Multi-rank path.  CPU: the collective plumbing of big-bwt_amd/dist.py under torch.distributed
(gloo, world_size 2, 127.0.0.1).  GPU: the whole distributed chain with R virtual ranks on one
card, bit for bit against the oracle."""
import os
import socket
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _toy_phases(rank, size):
    """same protocol as dist.phases: yields ('allgather', tensor), gets the list back"""
    a = torch.arange(3 + 2 * rank, dtype=torch.uint8) + 10 * rank          # different lengths per rank
    got = yield ("allgather", a)
    b = torch.tensor([sum(int(t.sum()) for t in got) + rank], dtype=torch.int64)
    got2 = yield ("allgather", b)
    return torch.cat(got).tolist(), [int(t.item()) for t in got2]


def _worker(rank, size, port, q):
    sys.path.insert(0, ROOT)
    import __graft_entry__ as entry
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=size)
    entry.load_package()
    import importlib
    d = importlib.import_module("bigbwt_amd.dist")
    # 1. variable-length allgather
    t = torch.arange(5 + 3 * rank, dtype=torch.int32) * (rank + 1)
    parts = d.all_gather_var(t)
    ok = len(parts) == size and all(p.numel() == 5 + 3 * r and int(p.sum()) == int((torch.arange(5 + 3 * r) * (r + 1)).sum())
                                    for r, p in enumerate(parts))
    # 2. the driver loop of dist.run on a toy generator (no GPU needed)
    gen = _toy_phases(rank, size)
    reply = None
    try:
        while True:
            kind, payload = gen.send(reply)
            reply = d.all_gather_var(payload)
    except StopIteration as fin:
        res = fin.value
    # 3. slice bounds tile the output exactly
    bounds = [d.slice_bounds(1001, r, size) for r in range(size)]
    ok &= bounds[0][0] == 0 and bounds[-1][1] == 1001 and all(bounds[i][1] == bounds[i + 1][0] for i in range(size - 1))
    q.put((rank, ok, res))
    dist.barrier()
    dist.destroy_process_group()


def test_allgather_plumbing_gloo_world2():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want_cat = list(range(0, 3)) + [10 + i for i in range(5)]
    for rank, ok, res in out:
        assert ok
        assert res[0] == want_cat
        assert res[1] == [sum(want_cat) + 0, sum(want_cat) + 1]


def _pieces(res, key):
    """concatenate the ranks' pieces of one output file after checking that their offsets tile it"""
    off = 0
    parts = []
    for r in res:
        assert r[key + "_off"] == off, (key, r[key + "_off"], off)
        parts.append(r[key].cpu().numpy())
        off += r[key].numel()
    return np.concatenate(parts)


def _check_files(pkg, res, want, flags, n):
    assert res[0]["lo"] == 0 and res[-1]["hi"] == n + 1 and all(res[k]["hi"] == res[k + 1]["lo"] for k in range(len(res) - 1))
    assert np.array_equal(torch.cat([r["bwt"] for r in res]).cpu().numpy(), want["bwt"])
    if flags & pkg.FLAG_SA:
        assert np.array_equal(pkg.unpack5(_pieces(res, "sa5")), want["sa"])
    if flags & pkg.FLAG_SSA:
        assert np.array_equal(pkg.unpack5(_pieces(res, "ssa")).reshape(-1, 2), want["ssa"])
    if flags & pkg.FLAG_ESA:
        assert np.array_equal(pkg.unpack5(_pieces(res, "esa")).reshape(-1, 2), want["esa"])


@pytest.mark.gpu
@pytest.mark.parametrize("width", [0, 64], ids=["idx32", "idx64"])
@pytest.mark.parametrize("R", [2, 3, 8])
@pytest.mark.parametrize("cfg", [(10, 100, 1), (10, 100, 6), (12, 200, 2), (12, 200, 4)],
                         ids=["c4_w10_p100_S", "w10_p100_s_e", "c5_w12_p200_s", "w12_p200_e"])
def test_baseline_flag_sets_across_ranks(O, pkg, R, cfg, width, tmp_path):
    """BASELINE configs[3] (-w 10 -p 100 -S) and configs[4] (-w 12 -p 200 -s) flag sets, plus -s -e / -e, with 2, 3 and 8
    virtual ranks: every rank's pieces of .bwt/.sa/.ssa/.esa - written with pfp_pwrite_dev at their offsets -
    give the oracle's files; both index widths."""
    import importlib
    d = importlib.import_module("bigbwt_amd.dist")
    w, p, flags = cfg
    text = O.gen_fasta(60000, 8, 0.003, 59)
    n = len(text)
    want = O.bigbwt(text, w, p, flags)
    cuts = [0] + [n * (r + 1) // R + (5 * r - 2) for r in range(R - 1)] + [n]
    ctxs = [pkg.Context(0) for _ in range(R)]
    try:
        for c in ctxs:
            c.set_index_bits(width)
        shards = [torch.from_numpy(text[cuts[r]:cuts[r + 1]].copy()).cuda() for r in range(R)]
        res = d.simulate(ctxs, shards, w, p, flags, halo=8192)
        assert {r["stats"]["glob"]["index_bits"] for r in res} == {64 if width else 32}
        _check_files(pkg, res, want, flags, n)
        base = str(tmp_path / "t")
        for ext in (".bwt", ".sa", ".ssa", ".esa"):      # leftovers of an earlier, longer run under the same name must not survive
            with open(base + ext, "wb") as f:
                f.write(b"\xee" * (6 * n + 77))
        for r in range(R - 1, -1, -1):          # any order: every piece lands at its own offset (one rank creates the files first)
            d.write_outputs(ctxs[r], base, res[r], create=(r == R - 1))
        assert np.array_equal(np.fromfile(base + ".bwt", dtype=np.uint8), want["bwt"])
        if flags & pkg.FLAG_SA:
            assert np.array_equal(np.fromfile(base + ".sa", dtype=np.uint8), O.pack5(want["sa"]))
        if flags & pkg.FLAG_SSA:
            assert np.array_equal(np.fromfile(base + ".ssa", dtype=np.uint8), O.pack5(want["ssa"].reshape(-1)))
        if flags & pkg.FLAG_ESA:
            assert np.array_equal(np.fromfile(base + ".esa", dtype=np.uint8), O.pack5(want["esa"].reshape(-1)))
        # -S together with -s / -e is refused on every rank, before anything is exchanged (bigbwt:59-61)
        for bad in (3, 5, 7):
            with pytest.raises(pkg.PfpError):
                d.simulate(ctxs, shards, w, p, bad, halo=8192)
    finally:
        for c in ctxs:
            c.close()


@pytest.mark.gpu
@pytest.mark.parametrize("shard_sa", [True, False], ids=["sa_by_key_range", "sa_replicated"])
@pytest.mark.parametrize("R", [2, 3, 5])
@pytest.mark.parametrize("nblock", [True, False], ids=["n_block", "no_n_block"])
def test_distributed_chain_matches_oracle(O, pkg, R, shard_sa, nblock):
    """R virtual ranks on one GPU: text shards -> halo -> local parse -> union dictionary ->
    suffix array of the dictionary (sharded by key range, or replicated) -> output slices;
    concatenated slices must equal the oracle's files."""
    import importlib
    d = importlib.import_module("bigbwt_amd.dist")
    dev = torch.device("cuda:0")
    # N block: giant phrase -> shared extra triggers; its 2 kB all-N phrases are a long exact repeat inside
    # the dictionary, which a key range may be unable to settle alone (-> replicated fallback)
    text = O.gen_fasta(150000, 6, 0.002, 41, n_blocks=[(40000, 60000)] if nblock else [])
    n = len(text)
    cuts = [0] + [n * (r + 1) // R + (7 * r - 3) for r in range(R - 1)] + [n]     # uneven, unaligned shards
    ctxs = [pkg.Context(0) for _ in range(R)]
    try:
        for flags in (0, pkg.FLAG_SA, pkg.FLAG_SSA | pkg.FLAG_ESA):
            shards = [torch.from_numpy(text[cuts[r]:cuts[r + 1]].copy()).to(dev) for r in range(R)]
            for c in ctxs:
                c.set_max_phrase(2000)
            # the replicated-sort variant also takes the simple union dedup (allgather of whole local dictionaries), the
            # sharded one the hash-partitioned all-to-all
            res = d.simulate(ctxs, shards, 10, 100, flags, halo=4096, shard_sa=shard_sa, dedup="alltoall" if shard_sa else "allgather")
            assert res[0]["stats"]["dedup"] == ("alltoall" if shard_sa else "allgather")
            # (the N block is a giant phrase for the reference's hash - shared extra triggers; under the window hash one of its
            #  line-break windows may cut it already, then nothing needs splitting: PFP_WINDOW_HASH=kr in the env matrix pins the first)
            assert res[0]["stats"]["extra_triggers"] >= (1 if nblock and os.environ.get("PFP_WINDOW_HASH") == "kr" else 0)
            shares = {r["stats"]["sa_shares"] for r in res}
            # debugging switches that turn pivot rounds off make every range fall back to the replicated sort
            hobbled = any(os.environ.get(k) for k in ("PFP_NO_FINFLAG", "PFP_PIVOT_CAP"))
            must_shard = shard_sa and not nblock and not hobbled
            assert shares == ({R} if must_shard else ({1} if not shard_sa else shares & {1, R})) and len(shares) == 1
            assert all(res[k]["hi"] == res[k + 1]["lo"] for k in range(R - 1))
            bwt = torch.cat([r["bwt"] for r in res]).cpu().numpy()
            want = O.bigbwt(text, 10, 100, flags)
            assert res[0]["lo"] == 0 and res[-1]["hi"] == n + 1
            assert np.array_equal(bwt, want["bwt"]), (R, flags)
            _check_files(pkg, res, want, flags, n)
            if flags & pkg.FLAG_SA:
                sa = torch.cat([r["sa"] for r in res]).cpu().numpy().astype(np.uint64)
                assert sa[0] == n and np.array_equal(sa[1:], want["sa"])
            assert sum(r["stats"]["local"]["phrases"] for r in res) == res[0]["stats"]["phrases_total"]
    finally:
        for c in ctxs:
            c.close()


@pytest.mark.gpu
@pytest.mark.parametrize("R", [2, 3, 8])
def test_parse_suffix_array_in_shares(O, pkg, R):
    """the replicated parse's suffix array sorted in R key ranges (pfp_dist_parse_sort: first sort, pivot round with the member whose
    next rare symbol is farthest, comparison finisher - no ranks of other ranges) and all-gathered, against every rank sorting the whole
    parse inside the merge: the same files, and the shares really are used on a collection of copies"""
    import importlib
    d = importlib.import_module("bigbwt_amd.dist")
    text = O.gen_fasta(150000, 12, 0.002, 61)
    n = len(text)
    dev = torch.device("cuda:0")
    want = O.bigbwt(text, 10, 100, O.FLAG_SSA | O.FLAG_ESA)
    for shard_parse in (True, False):
        ctxs = [pkg.Context(0) for _ in range(R)]
        try:
            shards = [torch.from_numpy(text[n * r // R: n * (r + 1) // R].copy()).to(dev) for r in range(R)]
            res = d.simulate(ctxs, shards, 10, 100, pkg.FLAG_SSA | pkg.FLAG_ESA, halo=8192, shard_parse=shard_parse)
            assert res[0]["stats"]["parse_shares"] == (R if shard_parse else 1)
            bwt = np.concatenate([x["bwt"].cpu().numpy() for x in res])
            assert np.array_equal(bwt, want["bwt"])
            ssa = np.concatenate([x["ssa"].cpu().numpy() for x in res])
            assert np.array_equal(pkg.unpack5(ssa).reshape(-1, 2), want["ssa"])
        finally:
            for c in ctxs:
                c.close()


@pytest.mark.gpu
@pytest.mark.parametrize("R", [4, 6])
def test_many_ranks_on_a_tiny_dictionary(O, pkg, R):
    """a dictionary of ~120 suffixes split over 4-6 key ranges (shares of a few dozen slots, emit counts from 1 up)"""
    import importlib
    d = importlib.import_module("bigbwt_amd.dist")
    t = np.frombuffer((b"ACGTACGTTTGACA" * 40 + b"GGGTTTAAACCC" * 30) * 3, dtype=np.uint8).copy()
    cuts = [len(t) * r // R for r in range(R + 1)]
    ctxs = [pkg.Context(0) for _ in range(R)]
    try:
        for c in ctxs:
            c.set_window_hash(False)          # (the shards of this 2.8 KB text are cut where the reference's hash cuts: under another hash one holds no complete phrase)
        shards = [torch.from_numpy(t[cuts[r]:cuts[r + 1]].copy()).cuda() for r in range(R)]
        res = d.simulate(ctxs, shards, 4, 11, pkg.FLAG_SA, halo=256)
        want = O.bigbwt(t, 4, 11, O.FLAG_SA)
        assert np.array_equal(torch.cat([r["bwt"] for r in res]).cpu().numpy(), want["bwt"])
        assert np.array_equal(torch.cat([r["sa"] for r in res]).cpu().numpy().astype(np.uint64)[1:], want["sa"])
        assert sum(r["hi"] - r["lo"] for r in res) == len(t) + 1
    finally:
        for c in ctxs:
            c.close()


@pytest.mark.gpu
def test_sharded_sort_falls_back_when_a_range_cannot_finish(O, pkg, monkeypatch):
    """a dictionary word with a long exact repeat (an N run cut into 20 kB phrases) leaves a group that
    pivot rounds cannot settle inside one key range: all ranks fall back to the replicated sort"""
    import importlib
    d = importlib.import_module("bigbwt_amd.dist")
    dev = torch.device("cuda:0")
    text = O.gen_fasta(120000, 2, 0.001, 47, n_blocks=[(20000, 80000)])
    n = len(text)
    ctxs = [pkg.Context(0) for _ in range(2)]
    try:
        for c in ctxs:
            c.set_max_phrase(0)          # no extra triggers: the N block stays one giant phrase
        shards = [torch.from_numpy(text[: n // 2].copy()).to(dev), torch.from_numpy(text[n // 2:].copy()).to(dev)]
        res = d.simulate(ctxs, shards, 10, 100, 0, halo=1 << 17)
        bwt = torch.cat([r["bwt"] for r in res]).cpu().numpy()
        assert np.array_equal(bwt, O.bigbwt(text, 10, 100, 0)["bwt"])
        assert res[0]["stats"]["sa_shares"] in (1, 2)       # 1 when the fallback was taken
    finally:
        for c in ctxs:
            c.close()


@pytest.mark.gpu
def test_distributed_halo_too_small_is_reported(O, pkg):
    import importlib
    d = importlib.import_module("bigbwt_amd.dist")
    dev = torch.device("cuda:0")
    text = O.gen_fasta(60000, 2, 0.0, 43)
    ctxs = [pkg.Context(0), pkg.Context(0)]
    try:
        shards = [torch.from_numpy(text[:50000].copy()).to(dev), torch.from_numpy(text[50000:].copy()).to(dev)]
        with pytest.raises(pkg.PfpError):
            d.simulate(ctxs, shards, 10, 100, 0, halo=12)      # 12 bytes cannot hold a phrase boundary
        # an error on ONE rank (a byte <= 2 inside its shard) stops every rank at the same step: nobody is left
        # waiting in the next collective
        bad = text.copy()
        bad[55000] = 1
        shards = [torch.from_numpy(bad[:50000].copy()).to(dev), torch.from_numpy(bad[50000:].copy()).to(dev)]
        gens = [d.phases(ctxs[r], shards[r], r, 2, 10, 100, 0, 4096) for r in range(2)]
        replies, raised = [None, None], [None, None]
        for _ in range(40):
            reqs = []
            for r in range(2):
                if raised[r] is None:
                    try:
                        reqs.append(gens[r].send(replies[r])[1])
                    except pkg.PfpError as ex:
                        raised[r] = ex
            if all(x is not None for x in raised):
                break
            assert len(reqs) == 2, "one rank stopped while the other went on to the next collective"
            replies = [list(reqs), list(reqs)]
        assert all(x is not None for x in raised) and "rank 1" in str(raised[0]) and raised[1].code == -6
    finally:
        for c in ctxs:
            c.close()


def _gpu_worker(rank, size, port, q):
    sys.path.insert(0, ROOT)
    import __graft_entry__ as entry
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=size)
    pkg = entry.load_package()
    O = entry.load_oracle()
    import importlib
    d = importlib.import_module("bigbwt_amd.dist")
    dev = torch.device("cuda:0")
    text = O.gen_fasta(120000, 4, 0.002, 47)
    n = len(text)
    lo, hi = n * rank // size, n * (rank + 1) // size
    shard = torch.from_numpy(text[lo:hi].copy()).to(dev)
    ctx = pkg.Context(0)
    res = d.run(ctx, shard, 10, 100, pkg.FLAG_SA, halo=8192)
    # a halo too small on one rank only: both ranks must leave the chain at the same collective (no deadlock)
    try:
        d.run(ctx, shard, 10, 100, 0, halo=12 if rank == 0 else 8192)
        failed = None
    except pkg.PfpError as ex:
        failed = str(ex)
    q.put((rank, res["lo"], res["hi"], res["bwt"].cpu().numpy(), res["sa"].cpu().numpy(), failed))
    dist.barrier()
    ctx.close()
    dist.destroy_process_group()


@pytest.mark.gpu
def test_distributed_run_two_processes_one_gpu(O):
    """dist.run under real torch.distributed with two PROCESSES (gloo rendezvous, both on cuda:0):
    everything of the N>1 path except the RCCL transport itself."""
    import torch.multiprocessing as mp
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    port = _free_port()
    procs = [mpc.Process(target=_gpu_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = sorted([q.get(timeout=300) for _ in range(2)], key=lambda x: x[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    text = O.gen_fasta(120000, 4, 0.002, 47)
    want = O.bigbwt(text, 10, 100, O.FLAG_SA)
    bwt = np.concatenate([o[3] for o in out])
    sa = np.concatenate([o[4] for o in out]).astype(np.uint64)
    assert out[0][1] == 0 and out[1][2] == len(text) + 1
    assert np.array_equal(bwt, want["bwt"])
    assert sa[0] == len(text) and np.array_equal(sa[1:], want["sa"])
    assert all(o[5] is not None for o in out), "a rank-local failure must stop every rank"


def _nccl_worker(rank, size, port, q):
    sys.path.insert(0, ROOT)
    import __graft_entry__ as entry
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=size)          # RCCL over xGMI
    pkg = entry.load_package()
    O = entry.load_oracle()
    import importlib
    d = importlib.import_module("bigbwt_amd.dist")
    text = O.gen_fasta(120000, 4, 0.002, 47)
    n = len(text)
    lo, hi = n * rank // size, n * (rank + 1) // size
    shard = torch.from_numpy(text[lo:hi].copy()).to(torch.device("cuda", rank))
    ctx = pkg.Context(rank)
    res = d.run(ctx, shard, 12, 200, pkg.FLAG_SSA, halo=8192)
    q.put((rank, res["lo"], res["hi"], res["bwt"].cpu().numpy(), res["ssa"].cpu().numpy(), res["ssa_off"]))
    dist.barrier()
    ctx.close()
    dist.destroy_process_group()


@pytest.mark.gpu
def test_distributed_run_over_rccl():
    """the N>1 path on the nccl (= RCCL) backend, one process per GPU; needs two GPUs, skips itself on a one-GPU box"""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs 2 GPUs (the driver's 8-GPU node runs it)")
    import torch.multiprocessing as mp
    sys.path.insert(0, ROOT)
    import __graft_entry__ as entry
    O = entry.load_oracle()
    pkg = entry.load_package()
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    port = _free_port()
    procs = [mpc.Process(target=_nccl_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = sorted([q.get(timeout=600) for _ in range(2)], key=lambda x: x[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    text = O.gen_fasta(120000, 4, 0.002, 47)
    want = O.bigbwt(text, 12, 200, O.FLAG_SSA)
    assert np.array_equal(np.concatenate([o[3] for o in out]), want["bwt"])
    assert out[0][5] == 0 and out[1][5] == len(out[0][4])
    assert np.array_equal(pkg.unpack5(np.concatenate([o[4] for o in out])).reshape(-1, 2), want["ssa"])


# ----------------------------------------------------------------------------- the C driver's -G N mode
EXE = os.path.join(ROOT, "big-bwt_amd", "bigbwt")


def _sha(a):
    import hashlib
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def test_c_driver_multi_gpu_mode_fails_loudly_without_a_gpu(tmp_path):
    """bigbwt -G 2 without a GPU: the library call says so (native ranks) / every rank process says so (PFP_MULTI_PYTHON=1) and
    the driver reports the failure (exit code 1, no output files): no CPU path behind the multi-GPU mode either."""
    import subprocess
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    f = tmp_path / "t.txt"
    f.write_bytes(b"ACGT" * 1000)
    out = subprocess.run([EXE, "-G", "2", str(f)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 1
    assert "on 2 GPUs" in out.stdout and "pfp_bigbwt_files_multi" in out.stdout and "no usable HIP device" in out.stdout
    assert not os.path.exists(str(f) + ".bwt")
    out = subprocess.run([EXE, "-G", "2", str(f)], capture_output=True, text=True, timeout=300, env=dict(os.environ, PFP_MULTI_PYTHON="1"))
    assert out.returncode == 1
    assert "on 2 GPUs" in out.stdout and "2 ranks of dist_main.py: exit code" in out.stdout
    assert "no GPU" in out.stderr and not os.path.exists(str(f) + ".bwt")
    bad = subprocess.run([EXE, "-G", "2", "-k", str(f)], capture_output=True, text=True)
    assert bad.returncode == 2 and "not with -k" in bad.stdout


MULTI_CASES = [("gen_small", 2, ["-s", "-e"], "6"), ("gen_nblock", 3, ["-S"], "1"), ("gen_1e6x4", 2, [], "0"), ("gen_1e6x4", 5, ["-s"], "6"),
               ("gen_p101_w12", 4, ["-e"], "6"), ("n_run", 3, ["-S"], "1"), ("kat1", 2, ["-s", "-e"], "6")]


@pytest.mark.gpu
@pytest.mark.parametrize("name,G,opts,run", MULTI_CASES)
def test_c_driver_multi_gpu_mode(golden, O, tmp_path, name, G, opts, run):
    """`bigbwt -G N`: N rank threads inside the library (pfp_bigbwt_files_multi; here all on cuda:0, exchanging through device
    copies instead of RCCL: PFP_MULTI_LOOPBACK=1) each read a byte range of the file and write their ranges of the outputs; the
    files must be the reference's (golden digests), --sum and -c work as for one GPU."""
    import subprocess
    from textgen import make_text
    c = {x["name"]: x for x in golden}[name]
    f = tmp_path / "t.fa"
    f.write_bytes(make_text(c["spec"], O).tobytes())
    for ext in (".bwt", ".sa", ".ssa", ".esa"):          # stale, longer files from "an earlier run"
        (tmp_path / ("t.fa" + ext)).write_bytes(b"x" * 5_000_000)
    env = dict(os.environ, PFP_MULTI_LOOPBACK="1")
    cmd = [EXE, "-G", str(G), "-w", str(c["w"]), "-p", str(c["p"]), "--halo", "65536", "--sum", "-c", "-v"] + opts + [str(f)]
    out = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600)
    if len(make_text(c["spec"], O)) < 64 * G:          # a text too short to give every rank a phrase: refused, not wrong
        assert out.returncode in (0, 1), out.stdout + out.stderr
        if out.returncode == 1:
            assert "Error executing command line" in out.stdout
            return
    assert out.returncode == 0, out.stdout + out.stderr
    want = c["runs"][run]
    assert "BWTs match" in out.stdout and want["bwt_sha256"] in out.stdout and f"{G} ranks: chain" in out.stdout
    for ext in ["bwt"] + [{"-s": "ssa", "-e": "esa", "-S": "sa"}[o] for o in opts]:
        assert _sha(np.fromfile(str(f) + "." + ext, dtype=np.uint8)) == want[ext + "_sha256"], ext
    log = open(str(f) + ".log").read()
    assert f"Ranks: {G}" in log and "distinct words" in log


@pytest.mark.gpu
@pytest.mark.parametrize("opts,run", [(["-s", "-e"], "6"), (["-S"], "1")])
def test_c_driver_multi_gpu_replicated_fallback(golden, O, tmp_path, opts, run):
    """no pivot rounds (PFP_PIVOT_CAP=0: what a dictionary with long exact repeats comes to): no share of the suffix array can
    finish alone, every rank sorts the whole dictionary and the output is split into equal slices - the native host's fallback"""
    import subprocess
    from textgen import make_text
    c = {x["name"]: x for x in golden}["gen_small"]
    f = tmp_path / "t.fa"
    f.write_bytes(make_text(c["spec"], O).tobytes())
    env = dict(os.environ, PFP_MULTI_LOOPBACK="1", PFP_PIVOT_CAP="0")
    out = subprocess.run([EXE, "-G", "3", "-w", str(c["w"]), "-p", str(c["p"]), "--halo", "65536", "-c", "-v"] + opts + [str(f)],
                         capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "BWTs match" in out.stdout and "in 1 share(s)" in out.stdout
    for ext in ["bwt"] + [{"-s": "ssa", "-e": "esa", "-S": "sa"}[o] for o in opts]:
        assert _sha(np.fromfile(str(f) + "." + ext, dtype=np.uint8)) == c["runs"][run][ext + "_sha256"], ext


@pytest.mark.gpu
def test_c_driver_multi_gpu_over_rccl(golden, O, tmp_path):
    """the same over RCCL, one rank per GPU; needs two GPUs, skips itself on a one-GPU box"""
    import subprocess
    from textgen import make_text
    if torch.cuda.device_count() < 2:
        pytest.skip("needs 2 GPUs (the driver's 8-GPU node runs it)")
    G = min(torch.cuda.device_count(), 8)
    c = {x["name"]: x for x in golden}["gen_1e6x4"]
    f = tmp_path / "t.fa"
    f.write_bytes(make_text(c["spec"], O).tobytes())
    for opts, run in ((["-s", "-e"], "6"), (["-S"], "1")):
        out = subprocess.run([EXE, "-G", str(G), "-w", str(c["w"]), "-p", str(c["p"]), "--halo", "65536", "-c", "-v"] + opts + [str(f)],
                             capture_output=True, text=True, timeout=900)
        assert out.returncode == 0, out.stdout + out.stderr
        assert "BWTs match" in out.stdout
        for ext in ["bwt"] + [{"-s": "ssa", "-e": "esa", "-S": "sa"}[o] for o in opts]:
            assert _sha(np.fromfile(str(f) + "." + ext, dtype=np.uint8)) == c["runs"][run][ext + "_sha256"], ext


@pytest.mark.gpu
def test_rccl_transport_on_one_device(pkg):
    """the native host's RCCL transport where only one GPU exists: librccl through the library's own dlopen table, a communicator
    from ncclCommInitAll over cuda:0, and every exchange shape of the chain as a self send / recv - the grouped all-to-all, the
    all-gather as ONE ncclAllGather and in its grouped send / recv form, empty contributions.  (What two GPUs add is the wire;
    symbols, enum values and group semantics are settled here.)"""
    import importlib
    pfpmod = importlib.import_module("bigbwt_amd.pfp")
    pfpmod.multi_rccl_selftest(0)
    # ... and a rank's error path: a failure between ncclGroupStart and ncclGroupEnd with a send already queued - the group is
    # closed, every communicator aborted (ncclCommAbort), the call comes back (peers must not hang in a receive)
    pfpmod.multi_rccl_selftest(0, inject_failure=True)
    pfpmod.multi_rccl_selftest(0)          # a fresh communicator works afterwards


def test_launcher_of_the_python_ranks_never_counts_devices(tmp_path):
    """`bigbwt -G 0` counts the devices through HIP; under PFP_MULTI_PYTHON=1 the driver forks and execs the rank processes and
    must not have initialised HIP before that, so the combination is refused before anything touches the GPU (CPU test)"""
    import subprocess
    f = tmp_path / "t.fa"
    f.write_bytes(b"ACGT" * 1000)
    out = subprocess.run([EXE, "-G", "0", str(f)], capture_output=True, text=True, timeout=120, env=dict(os.environ, PFP_MULTI_PYTHON="1"))
    assert out.returncode == 2 and "pass -G N" in out.stderr, out.stdout + out.stderr


@pytest.mark.gpu
def test_c_driver_python_ranks(golden, O, tmp_path):
    """PFP_MULTI_PYTHON=1: the same through N processes of dist_main.py under torch.distributed (gloo, one GPU)"""
    import subprocess
    from textgen import make_text
    c = {x["name"]: x for x in golden}["gen_small"]
    f = tmp_path / "t.fa"
    f.write_bytes(make_text(c["spec"], O).tobytes())
    env = dict(os.environ, PFP_MULTI_PYTHON="1", PFP_DIST_BACKEND="gloo", PFP_DIST_ONE_GPU="1")
    out = subprocess.run([EXE, "-G", "2", "-w", str(c["w"]), "-p", str(c["p"]), "--halo", "65536", "--sum", "-c", "-v", "-s", "-e", str(f)],
                         capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    want = c["runs"]["6"]
    assert "BWTs match" in out.stdout and want["bwt_sha256"] in out.stdout and "2 ranks (gloo)" in out.stdout
    for ext in ("bwt", "ssa", "esa"):
        assert _sha(np.fromfile(str(f) + "." + ext, dtype=np.uint8)) == want[ext + "_sha256"], ext


@pytest.mark.gpu
def test_c_driver_multi_gpu_fasta_and_failure(tmp_path):
    """-f with -G: the ranks read byte ranges of the filtered sequences; a text the chain refuses (a byte <= 2) or a halo that
    does not reach the last phrase boundary ends every rank and the driver with an error instead of a hang or a partial file set"""
    import json
    import subprocess
    with open(os.path.join(ROOT, "tests", "golden", "golden_fasta.json")) as fh:
        cases = json.load(fh)["cases"]
    c = max(cases, key=lambda x: len(x["raw_hex"]))
    f = tmp_path / "in.fa"
    f.write_bytes(bytes.fromhex(c["raw_hex"]))
    env = dict(os.environ, PFP_MULTI_LOOPBACK="1")
    out = subprocess.run([EXE, "-f", "-G", "2", "-w", str(c["w"]), "-p", str(c["p"]), "-c", str(f)], capture_output=True, text=True, env=env,
                         timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "BWTs match" in out.stdout
    assert _sha(np.fromfile(str(f) + ".bwt", dtype=np.uint8)) == c["bwt_sha256"]
    g = tmp_path / "bad.txt"
    g.write_bytes(b"ACGTTGCA" * 5000 + b"\x01" + b"ACGTTGCA" * 5000)
    out = subprocess.run([EXE, "-G", "2", str(g)], capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 1 and "local parse" in out.stdout, out.stdout + out.stderr
    rng = np.random.default_rng(3)
    h = tmp_path / "halo.txt"
    h.write_bytes(np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=400000)].tobytes())
    out = subprocess.run([EXE, "-G", "2", "--halo", "12", str(h)], capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 1 and "halo" in out.stdout, out.stdout + out.stderr
    same = subprocess.run([EXE, "-G", "2", str(h)], capture_output=True, text=True, timeout=600)     # RCCL wants one rank per GPU
    if torch.cuda.device_count() < 2:
        assert same.returncode == 1 and "does not exist" in same.stdout, same.stdout + same.stderr
    allg = subprocess.run([EXE, "-G", "0", "-c", str(h)], capture_output=True, text=True, timeout=600)     # -G 0: every visible GPU
    assert allg.returncode == 0 and "BWTs match" in allg.stdout, allg.stdout + allg.stderr
    if torch.cuda.device_count() > 1:
        assert "on %d GPUs" % torch.cuda.device_count() in allg.stdout


@pytest.mark.gpu
def test_multi_gpu_out_of_memory_on_the_ranks_ends_the_call(golden, O, pkg, tmp_path, monkeypatch):
    """PFP_TEST_POOL_LIMIT on every rank's pool: whichever library step runs out of memory first (a different one at every limit), all
    rank threads leave together with PFP_ENOMEM - no hang, no partial file set taken for a result - and where the limit is large enough
    the files are right"""
    from textgen import make_text
    import importlib
    monkeypatch.setenv("PFP_MULTI_LOOPBACK", "1")
    pfpmod = importlib.import_module("bigbwt_amd.pfp")
    c = {x["name"]: x for x in golden}["gen_1e6x4"]
    text = make_text(c["spec"], O)
    seen = set()
    for limit_mb in (1, 3, 6, 10, 16, 24, 40, 400):
        monkeypatch.setenv("PFP_TEST_POOL_LIMIT", str(limit_mb << 20))
        base = str(tmp_path / ("o%d" % limit_mb))
        try:
            pfpmod.bigbwt_files_multi(text, base, [0] * 3, c["w"], c["p"], 6, halo=1 << 16)
            seen.add("ok")
            want = c["runs"]["6"]
            assert _sha(np.fromfile(base + ".bwt", dtype=np.uint8)) == want["bwt_sha256"]
            assert _sha(np.fromfile(base + ".ssa", dtype=np.uint8)) == want["ssa_sha256"]
        except pkg.PfpError as ex:
            assert ex.code == -7, str(ex)
            seen.add("enomem")
    assert seen == {"ok", "enomem"}


@pytest.mark.gpu
def test_multi_gpu_entry_point_from_python(golden, O, pkg, tmp_path, monkeypatch):
    """pfp_bigbwt_files_multi called directly (ctypes): 8 rank threads on one device, every flag set"""
    from textgen import make_text
    import importlib
    monkeypatch.setenv("PFP_MULTI_LOOPBACK", "1")
    pfpmod = importlib.import_module("bigbwt_amd.pfp")
    c = {x["name"]: x for x in golden}["gen_1e6x4"]
    text = make_text(c["spec"], O)
    for flags in (0, 1, 6):
        base = str(tmp_path / ("m%d" % flags))
        st = pfpmod.bigbwt_files_multi(text, base, [0] * 8, c["w"], c["p"], flags, halo=1 << 16)
        assert st["ranks"] == 8 and st["n"] == len(text) and st["n_words"] > 0
        want = c["runs"][str(flags)]
        assert _sha(np.fromfile(base + ".bwt", dtype=np.uint8)) == want["bwt_sha256"]
        if flags & 1:
            assert _sha(np.fromfile(base + ".sa", dtype=np.uint8)) == want["sa_sha256"]
        if flags & 2:
            assert _sha(np.fromfile(base + ".ssa", dtype=np.uint8)) == want["ssa_sha256"]
            assert _sha(np.fromfile(base + ".esa", dtype=np.uint8)) == want["esa_sha256"]
    with pytest.raises(pkg.PfpError) as ei:
        pfpmod.bigbwt_files_multi(text, str(tmp_path / "bad"), [0, 0], c["w"], c["p"], 3)
    assert ei.value.code == -1


@pytest.mark.gpu
def test_bench_two_ranks_rehearsal():
    """`python bench.py --gpus 2` with no launcher around it: bench starts its two ranks itself (here both on cuda:0 over gloo,
    the one-GPU rehearsal of what the round-end driver runs on an 8-GPU node) and rank 0 prints ONE JSON line for the job"""
    import json
    import subprocess
    env = dict(os.environ, PFP_BENCH_BACKEND="gloo", PFP_BENCH_ONE_GPU="1")
    env.pop("RANK", None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--workload", "small"],
                         capture_output=True, text=True, env=env, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["scaling"] == "weak" and j["value"] > 0 and j["steps"] == 1
    assert j["verified"]["bwt_is_permutation_of_text_plus_eos"] is True
    assert j["rccl"]["ranks"] == 2 and j["rccl"]["backend"] == "gloo"
