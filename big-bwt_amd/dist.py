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
    Works with the nccl (RCCL) backend on device tensors and with gloo on CPU tensors."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    if t.is_cuda and dist.get_backend(group) == "gloo":
        # gloo has no device all_gather: stage through the host (tests that run several ranks on one GPU)
        return [x.to(t.device) for x in all_gather_var(t.cpu(), group)]
    n = torch.tensor([t.numel()], dtype=torch.int64, device=t.device)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n, group=group)
    sizes = [int(s.item()) for s in sizes]
    mx = max(max(sizes), 1)
    if sizes.count(mx) == world and t.numel() == mx:
        buf = t.contiguous()
    else:
        buf = torch.zeros(mx, dtype=t.dtype, device=t.device)
        buf[: t.numel()] = t
    flat = torch.empty(world * mx, dtype=t.dtype, device=t.device)
    try:
        dist.all_gather_into_tensor(flat, buf, group=group)      # one RCCL all-gather into one buffer, no per-rank copies
    except (RuntimeError, NotImplementedError, AttributeError):
        outs = [torch.empty(mx, dtype=t.dtype, device=t.device) for _ in range(world)]
        dist.all_gather(outs, buf, group=group)
        return [o[:s] for o, s in zip(outs, sizes)]
    return [flat[r * mx: r * mx + s] for r, s in enumerate(sizes)]


def slice_bounds(n_out, rank, size):
    """output range [lo,hi) of rank `rank`: equal split of the n+1 BWT positions"""
    return (n_out * rank) // size, (n_out * (rank + 1)) // size


# ----------------------------------------------------------------------------- the algorithm
def phases(ctx, shard, rank, size, w=10, p=100, flags=0, halo=DEFAULT_HALO, shard_sa=True):
    """Generator.  `shard`: 1-D uint8 device tensor, this rank's byte range of the text.
    Yields ('allgather', tensor) and receives the list of all ranks' tensors.
    Returns dict(bwt=slice tensor, sa=slice tensor|None, lo, hi, n_total, stats)."""
    dev = shard.device
    n_shard = shard.numel()
    want_sai = bool(flags)
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
    mine = ctx.dist_propose_triggers(local.data_ptr(), local.numel(), w, p)
    prop = torch.full((9,), -1, dtype=torch.int64, device=dev)
    if mine:
        prop[: len(mine)] = torch.tensor(mine, dtype=torch.int64, device=dev)
    if rank == 0 and n_shard >= w:
        # the text's first window must not become an extra trigger (it would turn the end-of-string byte of
        # the BWT into the reference's first-window quirk): rank 0 announces its hash
        h0 = 0
        for b in shard[:w].tolist():
            h0 = (h0 * 256 + b) % 1999999973          # newscan.cpp:168-202
        prop[8] = h0
    props = yield ("allgather", prop)
    banned = {int(t[8]) for t in props if int(t[8]) >= 0}
    extra = sorted({int(v) for t in props for v in t[:8].tolist() if v >= 0} - banned)[:32]
    info = ctx.dist_local_parse(local.data_ptr(), local.numel(), left.numel(), w, p, rank == 0, rank == size - 1, goff, want_sai,
                                extra)
    if rank < size - 1:
        # the next rank re-derives my last phrase boundary from the last tail.numel() bytes of my shard
        if info["last_trigger"] - (w - 1) < local.numel() - tail.numel():
            raise pfp.PfpError(-5, f"rank {rank}: last phrase boundary lies outside the {tail.numel()}-byte halo; raise `halo`")
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
    wslot = torch.zeros(union_occ.numel(), dtype=torch.int32, device=dev)
    torch.cuda.synchronize(dev)
    parts = size if shard_sa else 1
    ginfo = ctx.dist_global_sort(union.data_ptr(), union.numel(), union_occ.data_ptr(), union_occ.numel(), rank if parts > 1 else 0,
                                 parts, wslot.data_ptr())
    status = yield ("allgather", torch.tensor([1 if ginfo["complete"] else 0, ginfo["emits"]], dtype=torch.int64, device=dev))
    if parts > 1 and not all(int(s[0]) for s in status):
        # some range needs ranks it does not hold: every rank sorts everything (the replicated path)
        parts = 1
        wslot.zero_()
        ginfo = ctx.dist_global_sort(union.data_ptr(), union.numel(), union_occ.data_ptr(), union_occ.numel(), 0, 1, wslot.data_ptr())
    emits = [int(s[1]) for s in status]
    wslots = yield ("allgather", wslot[: ginfo["words"]].contiguous())
    if parts > 1:
        wslot_all = torch.cat(wslots).contiguous()
        torch.cuda.synchronize(dev)
        ctx.dist_global_finish(wslot_all.data_ptr(), parts, word_base, d_sym.data_ptr())
        del wslot_all
    else:
        ctx.dist_global_finish(wslot.data_ptr(), 1, word_base, d_sym.data_ptr())
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
    ctx.dist_merge(sym_all.data_ptr(), sym_all.numel(), last_all.data_ptr(), sai_all.data_ptr() if want_sai else None, flags,
                   n_total, lo, hi, bwt.data_ptr(), sa.data_ptr() if flags else None)
    stats = dict(local=info, glob=ginfo, phrases_total=int(sym_all.numel()), shard_bytes=n_shard, extra_triggers=len(extra),
                 sa_shares=parts)
    return dict(bwt=bwt[: hi - lo], sa=sa[: hi - lo] if flags else None, lo=lo, hi=hi, n_total=n_total, stats=stats)


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
