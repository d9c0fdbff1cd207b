"""Multi-GPU prefix-free-parsing BWT: one process per GPU, torch.distributed (RCCL over xGMI).

SURVEY.md 8(e): the text is sharded by byte range (the reference's pscan.hpp:114-165 does the same
across threads), phrases are independent units, the final BWT is sharded by output range (as
pfthreads.hpp:456-493 does).  Per rank:

    tails      allgather     last `halo` bytes of every shard (phrases that straddle a boundary)
    local      compute       scan halo+shard, own the phrases ending in the shard, local dedup
    dicts/occ  allgatherv    local dictionaries (O(|D|), small for repetitive collections)
    global     compute       dedup of the union -> global dictionary (replicated); its suffix array
                             sharded by key range: every rank sorts the suffixes whose first-round
                             key lies in its part of the key space (same splitters everywhere, no
                             exchange) and holds one contiguous range of SA(D)
    words      allgather     slot of every word's first suffix (4 B per word) + how many BWT
                             positions each range emits -> word ranks, output offsets;
                             own parse translated to global lexicographic ranks
    parse      allgatherv    parse symbols, last, sai  (O(P) = O(n/p))
    merge      compute       BWT of the parse (replicated), then the BWT / SA positions that the
                             held range of SA(D) emits (one contiguous slice per rank)
A range whose groups cannot be settled without other ranks' ranks (a long exact repeat inside the
dictionary) makes all ranks fall back to the replicated sort and equal output slices.

The algorithm is written once as a generator that yields collective requests; `run` drives it with
torch.distributed, `simulate` drives R of them in one process (tests: R virtual ranks on one GPU).
"""
import torch

from . import pfp

DEFAULT_HALO = 1 << 20


# ----------------------------------------------------------------------------- collectives
def all_gather_var(t, group=None):
    """all-gather 1-D tensors of different lengths (same dtype/device) -> list, one per rank.
    nccl (RCCL) backend: one all_gather_into_tensor on device tensors; gloo: list all_gather on CPU tensors.
    The path is chosen from the backend alone, never from a rank-local failure (every rank must issue the
    same collectives)."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    backend = dist.get_backend(group)
    if t.is_cuda and backend == "gloo":
        # gloo has no device all_gather: stage through the host (tests that run several ranks on one GPU)
        return [x.to(t.device) for x in all_gather_var(t.cpu(), group)]
    n = torch.tensor([t.numel()], dtype=torch.int64, device=t.device)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n, group=group)
    sizes = [int(s.item()) for s in sizes]
    mx = max(max(sizes), 1)
    if t.numel() == mx:
        buf = t.contiguous()
    else:
        buf = torch.zeros(mx, dtype=t.dtype, device=t.device)
        buf[: t.numel()] = t
    if backend == "nccl":
        flat = torch.empty(world * mx, dtype=t.dtype, device=t.device)
        dist.all_gather_into_tensor(flat, buf, group=group)      # one RCCL all-gather into one buffer, no per-rank copies
        return [flat[r * mx: r * mx + s] for r, s in enumerate(sizes)]
    outs = [torch.empty(mx, dtype=t.dtype, device=t.device) for _ in range(world)]
    dist.all_gather(outs, buf, group=group)
    return [o[:s] for o, s in zip(outs, sizes)]


def slice_bounds(n_out, rank, size):
    """output range [lo,hi) of rank `rank`: equal split of the n+1 BWT positions"""
    return (n_out * rank) // size, (n_out * (rank + 1)) // size


class _Step:
    """Runs one rank-local compute step inside the generator and lets all ranks agree on its outcome before
    anybody enters the next collective: a rank whose step failed (halo too small, a byte <= 2 in its shard,
    out of memory ...) would otherwise leave the others blocked in an all-gather forever."""

    def __init__(self, rank, dev):
        self.rank, self.dev, self.err = rank, dev, None

    def run(self, fn):
        try:
            return fn()
        except pfp.PfpError as ex:
            self.err = ex
            return None

    def status(self):
        code = self.err.code if self.err is not None else 0
        return torch.tensor([code], dtype=torch.int64, device=self.dev)

    def check(self, gathered, what):
        bad = [(r, int(t[0])) for r, t in enumerate(gathered) if int(t[0]) != 0]
        mine, self.err = self.err, None
        if not bad:
            return
        if mine is not None:
            raise mine
        r, code = bad[0]
        raise pfp.PfpError(code, f"{what} failed on rank {r}; every rank stops here")


# ----------------------------------------------------------------------------- the algorithm
def phases(ctx, shard, rank, size, w=10, p=100, flags=0, halo=DEFAULT_HALO, shard_sa=True):
    """Generator.  `shard`: 1-D uint8 device tensor, this rank's byte range of the text.
    Yields ('allgather', tensor) and receives the list of all ranks' tensors.
    Returns dict(bwt=slice tensor, sa=slice tensor|None, lo, hi, n_total, stats) plus, as the flags ask, the
    rank's pieces of the reference's output files as device byte tensors with their file offsets:
    sa5 / sa5_off (.sa, 5-byte ints, pfbwt.cpp:159-160), ssa / ssa_off, esa / esa_off (.ssa/.esa pairs,
    pfbwt.cpp:605-676)."""
    dev = shard.device
    n_shard = shard.numel()
    want_sai = bool(flags)
    step = _Step(rank, dev)
    # --- halo: the tail of every shard travels to its right neighbour
    tail = shard[-min(halo, n_shard):].contiguous()
    tails = yield ("allgather", tail)
    lens = yield ("allgather", torch.tensor([n_shard], dtype=torch.int64, device=dev))
    shard_sizes = [int(x.item()) for x in lens]
    n_total = sum(shard_sizes)
    goff = sum(shard_sizes[:rank])
    left = tails[rank - 1] if rank > 0 else shard[:0]
    local = torch.cat([left, shard]).contiguous() if rank > 0 else shard.contiguous()
    torch.cuda.synchronize(dev)
    # --- one trigger set for all ranks: the reference's plus the union of every rank's proposals for
    #     splitting giant phrases (N runs); the outputs do not depend on the parse
    mine = step.run(lambda: ctx.dist_propose_triggers(local.data_ptr(), local.numel(), w, p)) or []
    prop = torch.full((10,), -1, dtype=torch.int64, device=dev)
    if mine:
        prop[: len(mine)] = torch.tensor(mine, dtype=torch.int64, device=dev)
    if rank == 0 and n_shard >= w:
        # the text's first window must not become an extra trigger (it would turn the end-of-string byte of
        # the BWT into the reference's first-window quirk): rank 0 announces its hash
        h0 = 0
        for b in shard[:w].tolist():
            h0 = (h0 * 256 + b) % 1999999973          # newscan.cpp:168-202
        prop[8] = h0
    prop[9] = step.status()[0]
    props = yield ("allgather", prop)
    step.check([t[9:10] for t in props], "trigger proposal")
    banned = {int(t[8]) for t in props if int(t[8]) >= 0}
    # at most 32 extra triggers, taken round-robin over the ranks so that no rank's giant phrases are left out
    per_rank = [[int(v) for v in t[:8].tolist() if v >= 0 and int(v) not in banned] for t in props]
    extra = []
    for k in range(8):
        for lst in per_rank:
            if k < len(lst) and lst[k] not in extra and len(extra) < 32:
                extra.append(lst[k])
    extra.sort()

    def local_parse():
        info = ctx.dist_local_parse(local.data_ptr(), local.numel(), left.numel(), w, p, rank == 0, rank == size - 1, goff, flags,
                                    extra)
        # the next rank re-derives my last phrase boundary from the last tail.numel() bytes of my shard
        if rank < size - 1 and info["last_trigger"] - (w - 1) < local.numel() - tail.numel():
            raise pfp.PfpError(-5, f"rank {rank}: last phrase boundary lies outside the {tail.numel()}-byte halo; raise `halo`")
        return info
    info = step.run(local_parse)
    st = yield ("allgather", step.status())
    step.check(st, "local parse")
    d_dict = torch.empty(info["dict_bytes"], dtype=torch.uint8, device=dev)
    d_occ = torch.empty(info["words"], dtype=torch.int32, device=dev)
    d_last = torch.empty(info["phrases"], dtype=torch.uint8, device=dev)
    d_sai = torch.empty(info["phrases"], dtype=torch.int64, device=dev) if want_sai else None
    ctx.dist_export_local(d_dict.data_ptr(), d_occ.data_ptr(), d_last.data_ptr(), d_sai.data_ptr() if want_sai else None)
    del local
    # --- global dictionary from the union of the local ones
    dicts = yield ("allgather", d_dict)
    occs = yield ("allgather", d_occ)
    union = torch.cat(dicts).contiguous()
    union_occ = torch.cat(occs).contiguous()
    word_base = sum(o.numel() for o in occs[:rank])
    d_sym = torch.empty(info["phrases"], dtype=torch.int32, device=dev)
    wslot = torch.zeros(union_occ.numel(), dtype=torch.int64, device=dev)
    torch.cuda.synchronize(dev)
    parts = size if shard_sa else 1
    ginfo = step.run(lambda: ctx.dist_global_sort(union.data_ptr(), union.numel(), union_occ.data_ptr(), union_occ.numel(),
                                                  rank if parts > 1 else 0, parts, wslot.data_ptr()))
    ok = ginfo is not None
    status = yield ("allgather", torch.tensor([1 if (ok and ginfo["complete"]) else 0, ginfo["emits"] if ok else 0, int(step.status()[0])],
                                              dtype=torch.int64, device=dev))
    step.check([s[2:3] for s in status], "dictionary suffix sort")
    if parts > 1 and not all(int(s[0]) for s in status):
        # some range needs ranks it does not hold: every rank sorts everything (the replicated path)
        parts = 1
        wslot.zero_()
        torch.cuda.synchronize(dev)          # the library works on its own stream, not on torch's
        ginfo = step.run(lambda: ctx.dist_global_sort(union.data_ptr(), union.numel(), union_occ.data_ptr(), union_occ.numel(), 0, 1,
                                                      wslot.data_ptr()))
        st = yield ("allgather", step.status())
        step.check(st, "replicated dictionary suffix sort")
    emits = [int(s[1]) for s in status]
    wslots = yield ("allgather", wslot[: ginfo["words"]].contiguous())
    if parts > 1:
        wslot_all = torch.cat(wslots).contiguous()
        torch.cuda.synchronize(dev)
        step.run(lambda: ctx.dist_global_finish(wslot_all.data_ptr(), parts, word_base, d_sym.data_ptr()))
        del wslot_all
    else:
        step.run(lambda: ctx.dist_global_finish(wslot.data_ptr(), 1, word_base, d_sym.data_ptr()))
    st = yield ("allgather", step.status())
    step.check(st, "word ranking")
    del union, union_occ, dicts, occs, d_dict, d_occ, wslot, wslots
    # --- the whole parse everywhere
    syms = yield ("allgather", d_sym)
    lasts = yield ("allgather", d_last)
    sym_all = torch.cat(syms).contiguous()
    last_all = torch.cat(lasts).contiguous()
    sai_all = None
    if want_sai:
        sais = yield ("allgather", d_sai)
        sai_all = torch.cat(sais).contiguous()
    n_out = n_total + 1
    if parts > 1:
        if sum(emits) != n_out:
            raise pfp.PfpError(-6, f"the ranges of SA(D) emit {sum(emits)} positions, text length + 1 is {n_out}")
        lo = sum(emits[:rank])
        hi = lo + emits[rank]
    else:
        lo, hi = slice_bounds(n_out, rank, size)
    bwt = torch.empty(hi - lo + 16, dtype=torch.uint8, device=dev)
    sa = torch.empty(hi - lo + 1, dtype=torch.int64, device=dev) if flags else None
    torch.cuda.synchronize(dev)
    step.run(lambda: ctx.dist_merge(sym_all.data_ptr(), sym_all.numel(), last_all.data_ptr(), sai_all.data_ptr() if want_sai else None,
                                    flags, n_total, lo, hi, bwt.data_ptr(), sa.data_ptr() if flags else None))
    # --- the reference's output formats.  Run sampling needs one byte of halo from either neighbour:
    #     the BWT byte just before and just after this rank's slice (slices may be empty).
    cnt = hi - lo
    edge = torch.tensor([cnt, int(bwt[0]) if cnt and step.err is None else 0, int(bwt[cnt - 1]) if cnt and step.err is None else 0,
                         int(step.status()[0])], dtype=torch.int64, device=dev)
    edges = yield ("allgather", edge)
    step.check([e[3:4] for e in edges], "merge")
    out = dict(bwt=bwt[:cnt], sa=sa[:cnt] if flags else None, lo=lo, hi=hi, n_total=n_total)
    if flags & pfp.FLAG_SA:
        # .sa holds SA[1..n] (SA[0] = n is not written: pfbwt.cpp:158-162)
        first = 1 if lo == 0 else 0
        k = max(cnt - first, 0)
        sa5 = torch.empty(5 * k + 16, dtype=torch.uint8, device=dev)
        if k:
            ctx.pack5_dev(sa.data_ptr() + 8 * first, k, sa5.data_ptr())
        out["sa5"], out["sa5_off"] = sa5[: 5 * k], 5 * (lo + first - 1)
    if flags & (pfp.FLAG_SSA | pfp.FLAG_ESA):
        lefts = [int(e[2]) for e in edges[:rank] if int(e[0]) > 0]
        rights = [int(e[1]) for e in edges[rank + 1:] if int(e[0]) > 0]
        lb, rb = (lefts[-1] if lefts else -1), (rights[0] if rights else -1)
        counts = {}
        for key, flag, run_end in (("ssa", pfp.FLAG_SSA, False), ("esa", pfp.FLAG_ESA, True)):
            if flags & flag:
                k = ctx.sample_runs_dev(bwt.data_ptr(), sa.data_ptr(), cnt, lo, lb, rb, run_end) if cnt else 0
                buf = torch.empty(10 * k + 16, dtype=torch.uint8, device=dev)
                if k:
                    ctx.sample_runs_dev(bwt.data_ptr(), sa.data_ptr(), cnt, lo, lb, rb, run_end, buf.data_ptr(), k)
                out[key] = buf[: 10 * k]
                counts[key] = k
        ks = yield ("allgather", torch.tensor([counts.get("ssa", 0), counts.get("esa", 0)], dtype=torch.int64, device=dev))
        for j, key in enumerate(("ssa", "esa")):
            if key in out:
                out[key + "_off"] = 10 * sum(int(t[j]) for t in ks[:rank])
    out["stats"] = dict(local=info, glob=ginfo, phrases_total=int(sym_all.numel()), shard_bytes=n_shard, extra_triggers=len(extra),
                        sa_shares=parts)
    return out


def write_outputs(ctx, path, res):
    """Every rank writes its pieces of <path>.bwt / .sa / .ssa / .esa at their file offsets (the reference's
    threads pwrite their ranges the same way, pfthreads.hpp:369-376).  Device buffers are streamed to the
    file through the library's pinned staging buffers (pfp_pwrite_dev, C host code)."""
    ctx.pwrite_dev(path + ".bwt", res["lo"], res["bwt"].data_ptr(), res["bwt"].numel())
    for key, ext in (("sa5", ".sa"), ("ssa", ".ssa"), ("esa", ".esa")):
        if key in res:
            ctx.pwrite_dev(path + ext, res[key + "_off"], res[key].data_ptr(), res[key].numel())


def run(ctx, shard, w=10, p=100, flags=0, halo=DEFAULT_HALO, group=None, shard_sa=True):
    """Drive `phases` with torch.distributed (backend nccl == RCCL on ROCm; gloo works for CPU tests of the plumbing)."""
    import torch.distributed as dist
    rank, size = dist.get_rank(group), dist.get_world_size(group)
    gen = phases(ctx, shard, rank, size, w, p, flags, halo, shard_sa)
    reply = None
    try:
        while True:
            kind, payload = gen.send(reply)
            assert kind == "allgather"
            reply = all_gather_var(payload, group)
            if payload.is_cuda:
                torch.cuda.synchronize(payload.device)
    except StopIteration as fin:
        return fin.value


def simulate(ctxs, shards, w=10, p=100, flags=0, halo=DEFAULT_HALO, shard_sa=True):
    """Run R virtual ranks in one process (ctxs[r], shards[r] may all live on one GPU): every
    collective is served by plain concatenation.  Used by the GPU tests to check the distributed
    chain bit for bit against the single-GPU chain."""
    size = len(shards)
    gens = [phases(ctxs[r], shards[r], r, size, w, p, flags, halo, shard_sa) for r in range(size)]
    replies = [None] * size
    results = [None] * size
    live = set(range(size))
    while live:
        reqs = {}
        for r in sorted(live):
            try:
                kind, payload = gens[r].send(replies[r])
                assert kind == "allgather"
                reqs[r] = payload
            except StopIteration as fin:
                results[r] = fin.value
        live = set(reqs)
        if live:
            assert len(live) == size, "ranks fell out of step"
            gathered = [reqs[r] for r in range(size)]
            replies = [list(gathered) for _ in range(size)]
    return results
