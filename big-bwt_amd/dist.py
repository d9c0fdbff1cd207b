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


def all_to_all_var(parts, group=None):
    """parts[r] goes to rank r (1-D tensors, same dtype/device, any lengths) -> list: what every rank sent here.
    nccl (RCCL): one all_to_all_single with the split sizes the ranks announce first; gloo (CPU tests, ranks
    sharing one GPU): every rank's concatenated send buffer is all-gathered and the own pieces are cut out."""
    import torch.distributed as dist
    world, me = dist.get_world_size(group), dist.get_rank(group)
    backend = dist.get_backend(group)
    dev, dtype = parts[0].device, parts[0].dtype
    send_sizes = torch.tensor([p.numel() for p in parts], dtype=torch.int64, device=dev)
    if backend != "nccl":
        tables = all_gather_var(send_sizes, group)                     # tables[s][r] = elements s sends to r
        bufs = all_gather_var(torch.cat(parts) if parts else torch.empty(0, dtype=dtype, device=dev), group)
        out = []
        for s_ in range(world):
            off = int(tables[s_][:me].sum())
            out.append(bufs[s_][off: off + int(tables[s_][me])])
        return out
    recv_sizes = torch.empty_like(send_sizes)
    dist.all_to_all_single(recv_sizes, send_sizes, group=group)
    rs, ss = [int(x) for x in recv_sizes.tolist()], [int(x) for x in send_sizes.tolist()]
    recv = torch.empty(sum(rs), dtype=dtype, device=dev)
    dist.all_to_all_single(recv, torch.cat(parts).contiguous(), output_split_sizes=rs, input_split_sizes=ss, group=group)
    return list(torch.split(recv, rs))


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
        if self.err is not None:          # an earlier step of this rank already failed: do nothing until the ranks have agreed
            return None
        try:
            return fn()
        except pfp.PfpError as ex:
            self.err = ex
        except torch.cuda.OutOfMemoryError as ex:      # a torch.empty between two collectives
            self.err = pfp.PfpError(-7, f"rank {self.rank}: {ex}")
        except RuntimeError as ex:                     # HIP errors surfaced by torch
            self.err = pfp.PfpError(-3, f"rank {self.rank}: {ex}")
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


def _s64(x):
    """an unsigned 64-bit value as the int64 a tensor holds"""
    x &= 0xFFFFFFFFFFFFFFFF
    return x - (1 << 64) if x >> 63 else x


# ----------------------------------------------------------------------------- the algorithm
def phases(ctx, shard, rank, size, w=10, p=100, flags=0, halo=DEFAULT_HALO, shard_sa=True, dedup="alltoall", shard_parse=True):
    """Generator.  `shard`: 1-D uint8 device tensor, this rank's byte range of the text.
    Yields ('allgather', tensor) and receives the list of all ranks' tensors, or ('alltoall', [tensor per rank])
    and receives what every rank sent to this one.
    dedup: "alltoall" - every distinct word is owned by the rank its hash points at (all-to-all of words, all-to-all
    of the answers, allgatherv of distinct words only); "allgather" - every rank gathers all local dictionaries and
    deduplicates the union itself (the simple form, kept for comparison).
    Returns dict(bwt=slice tensor, sa=slice tensor|None, lo, hi, n_total, stats) plus, as the flags ask, the
    rank's pieces of the reference's output files as device byte tensors with their file offsets:
    sa5 / sa5_off (.sa, 5-byte ints, pfbwt.cpp:159-160), ssa / ssa_off, esa / esa_off (.ssa/.esa pairs,
    pfbwt.cpp:605-676)."""
    dev = shard.device
    n_shard = shard.numel()
    want_sai = bool(flags)
    if (flags & pfp.FLAG_SA) and (flags & (pfp.FLAG_SSA | pfp.FLAG_ESA)):      # bigbwt:59-61, pfbwt.cpp:298; every rank alike
        raise pfp.PfpError(-1, "You can either compute the full SA or a sample of it, not both (bigbwt:59-61)")
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
    # --- the collection's parse plan (round 4: the window hash and the phrase length by repetitiveness of the single-GPU chain,
    #     agreed between the ranks): rank 0 holds the text's first window and makes the plan, everybody receives it
    plan_t = torch.zeros(6, dtype=torch.int64, device=dev)          # plan[0..3], first window's hash under it, status
    if rank == 0:
        made = step.run(lambda: ctx.dist_parse_plan(bytes(shard[: min(n_shard, 32)].tolist()), w, p, size))
        if made is not None:
            pl, fh = made
            plan_t[:5] = torch.tensor([_s64(x) for x in pl] + [_s64(fh)], dtype=torch.int64, device=dev)
    plan_t[5] = step.status()[0]
    plans = yield ("allgather", plan_t)
    step.check([t[5:6] for t in plans], "parse plan")
    plan = [int(v) & 0xFFFFFFFFFFFFFFFF for v in plans[0][:4].tolist()]
    first_hash = int(plans[0][4]) & 0xFFFFFFFFFFFFFFFF
    # --- one trigger set for all ranks: the plan's plus the union of every rank's proposals for splitting giant phrases (N runs);
    #     with a candidate density in the plan also a sample of every rank's cuts: the samples are gathered and every rank settles
    #     the density alike (the outputs do not depend on the parse)
    if plan[3]:
        scap = max(4096, local.numel() // max(1, p) // 2 + 4096)
        d_sample = torch.empty(scap, dtype=torch.int64, device=dev)
        got = step.run(lambda: ctx.dist_propose_triggers2(local.data_ptr(), local.numel(), left.numel(), w, p, plan, d_sample.data_ptr(), scap))
        n_sample = got[1] if got is not None else 0
        st = yield ("allgather", step.status())
        step.check(st, "sampling the cuts")
        samples = yield ("allgather", d_sample[:n_sample].contiguous())
        allsmp = torch.cat(samples).contiguous()
        torch.cuda.synchronize(dev)
        settled = step.run(lambda: ctx.dist_decide_density(allsmp.data_ptr(), allsmp.numel(), p, plan))
        st = yield ("allgather", step.status())
        step.check(st, "parse density")
        plan = settled
        del samples, allsmp, d_sample
    got = step.run(lambda: ctx.dist_propose_triggers2(local.data_ptr(), local.numel(), left.numel(), w, p, plan, 0, 0))
    mine = got[0] if got is not None else []
    prop = torch.full((10,), -1, dtype=torch.int64, device=dev)
    if mine:
        prop[: len(mine)] = torch.tensor(mine, dtype=torch.int64, device=dev)
    if rank == 0 and first_hash != 0xFFFFFFFFFFFFFFFF:
        # the text's first window must not become an extra trigger (it would turn the end-of-string byte of
        # the BWT into the reference's first-window quirk): rank 0 announces its hash
        prop[8] = first_hash
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
        info = ctx.dist_local_parse2(local.data_ptr(), local.numel(), left.numel(), w, p, rank == 0, rank == size - 1, goff, flags,
                                     plan, extra)
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
    # --- global dictionary
    d_sym = torch.empty(info["phrases"], dtype=torch.int32, device=dev)
    parts = size if shard_sa else 1
    word_base = 0
    if dedup == "alltoall":
        # exchange A: words travel to the owner of their hash class, come back as global ids; only
        # distinct words are gathered
        del d_dict, d_occ
        counts = step.run(lambda: ctx.dist_partition_words(size)) or [(0, 0)] * size
        sendb = torch.empty(sum(b for _, b in counts) + 1, dtype=torch.uint8, device=dev)
        sendo = torch.empty(sum(k for k, _ in counts) + 1, dtype=torch.int32, device=dev)
        if step.err is None:
            step.run(lambda: ctx.dist_export_partition(sendb.data_ptr(), sendo.data_ptr()))
        st = yield ("allgather", step.status())
        step.check(st, "word partition")
        wsplit, bsplit = [k for k, _ in counts], [b for _, b in counts]
        recvb = yield ("alltoall", list(torch.split(sendb[: sum(bsplit)], bsplit)))
        recvo = yield ("alltoall", list(torch.split(sendo[: sum(wsplit)], wsplit)))
        rb, ro = torch.cat(recvb).contiguous(), torch.cat(recvo).contiguous()
        pid = torch.empty(ro.numel() + 1, dtype=torch.int32, device=dev)
        torch.cuda.synchronize(dev)
        own = step.run(lambda: ctx.dist_owner_dedup(rb.data_ptr(), rb.numel(), ro.data_ptr(), ro.numel(), pid.data_ptr())) or (0, 0)
        ownb = torch.empty(own[1] + 1, dtype=torch.uint8, device=dev)
        owno = torch.empty(own[0] + 1, dtype=torch.int32, device=dev)
        if step.err is None:
            step.run(lambda: ctx.dist_export_owned(ownb.data_ptr(), owno.data_ptr()))
        owned = yield ("allgather", torch.tensor([own[0], int(step.status()[0])], dtype=torch.int64, device=dev))
        step.check([t[1:2] for t in owned], "owner dedup")
        base = sum(int(t[0]) for t in owned[:rank])
        answers = yield ("alltoall", list(torch.split(pid[: ro.numel()] + base, [t.numel() for t in recvo])))
        gid_sent = torch.cat(answers).contiguous()          # owner order = the order this rank exported its words in
        dict_parts = yield ("allgather", ownb[: own[1]].contiguous())
        occ_parts = yield ("allgather", owno[: own[0]].contiguous())
        union = torch.cat(dict_parts).contiguous()
        union_occ = torch.cat(occ_parts).contiguous()
        wslot = torch.zeros(union_occ.numel(), dtype=torch.int64, device=dev)
        torch.cuda.synchronize(dev)
        sort_call = lambda part, nparts: ctx.dist_global_sort_distinct(union.data_ptr(), union.numel(), union_occ.data_ptr(), union_occ.numel(),
                                                                       gid_sent.data_ptr(), part, nparts, wslot.data_ptr())
        del recvb, recvo, rb, ro, sendb, sendo, dict_parts, occ_parts
    else:
        dicts = yield ("allgather", d_dict)
        occs = yield ("allgather", d_occ)
        union = torch.cat(dicts).contiguous()
        union_occ = torch.cat(occs).contiguous()
        word_base = sum(o.numel() for o in occs[:rank])
        wslot = torch.zeros(union_occ.numel(), dtype=torch.int64, device=dev)
        torch.cuda.synchronize(dev)
        sort_call = lambda part, nparts: ctx.dist_global_sort(union.data_ptr(), union.numel(), union_occ.data_ptr(), union_occ.numel(),
                                                              part, nparts, wslot.data_ptr())
        del dicts, occs, d_dict, d_occ
    ginfo = step.run(lambda: sort_call(rank if parts > 1 else 0, parts))
    ok = ginfo is not None
    status = yield ("allgather", torch.tensor([1 if (ok and ginfo["complete"]) else 0, ginfo["emits"] if ok else 0, int(step.status()[0])],
                                              dtype=torch.int64, device=dev))
    step.check([s[2:3] for s in status], "dictionary suffix sort")
    if parts > 1 and not all(int(s[0]) for s in status):
        # some range needs ranks it does not hold: every rank sorts everything (the replicated path)
        parts = 1
        wslot.zero_()
        torch.cuda.synchronize(dev)          # the library works on its own stream, not on torch's
        ginfo = step.run(lambda: sort_call(0, 1))
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
    del union, union_occ, wslot, wslots
    # --- the whole parse everywhere
    syms = yield ("allgather", d_sym)
    lasts = yield ("allgather", d_last)
    sym_all = torch.cat(syms).contiguous()
    last_all = torch.cat(lasts).contiguous()
    sai_all = None
    if want_sai:
        sais = yield ("allgather", d_sai)
        sai_all = torch.cat(sais).contiguous()
    # --- the suffix array of the parse in shares (key ranges, like the dictionary's): a rank sorts its range with the first sort,
    #     a pivot round and the comparison finisher - no ranks of other ranges needed; if any range would need a doubling round,
    #     nobody sets anything and every rank sorts the whole parse inside the merge as before
    if size > 1 and shard_parse and sym_all.numel() >= 2:
        P_all = sym_all.numel()
        share = torch.empty(P_all + 1, dtype=torch.int32, device=dev)
        torch.cuda.synchronize(dev)
        pinfo = step.run(lambda: ctx.dist_parse_sort(sym_all.data_ptr(), P_all, rank, size, share.data_ptr()))
        okp = pinfo is not None and pinfo["complete"]
        pst = yield ("allgather", torch.tensor([pinfo["entries"] if okp else 0, 1 if okp else 0, int(step.status()[0]), pinfo["slot_base"] if okp else 0],
                                               dtype=torch.int64, device=dev))
        step.check([t[2:3] for t in pst], "parse suffix sort")
        # the shares must tile [0, P + 1): every one starts where those before it end (a gap in one and an overlap in another of
        # the same size would pass a check of the sum alone)
        starts_ok = all(int(t[3]) == sum(int(u[0]) for u in pst[:r]) for r, t in enumerate(pst))
        if all(int(t[1]) for t in pst) and starts_ok and sum(int(t[0]) for t in pst) == P_all + 1:
            pieces = yield ("allgather", share[: pinfo["entries"]].contiguous())
            sa_parse = torch.cat(pieces).contiguous()
            torch.cuda.synchronize(dev)
            step.run(lambda: ctx.dist_set_parse_sa(sa_parse.data_ptr(), sa_parse.numel()))
            del pieces, sa_parse
            parse_shares = size
        else:
            parse_shares = 1
        del share
    else:
        parse_shares = 1
    n_out = n_total + 1
    if parts > 1:
        if sum(emits) != n_out:
            raise pfp.PfpError(-6, f"the ranges of SA(D) emit {sum(emits)} positions, text length + 1 is {n_out}")
        lo = sum(emits[:rank])
        hi = lo + emits[rank]
    else:
        lo, hi = slice_bounds(n_out, rank, size)
    bwt = torch.empty(hi - lo + 16, dtype=torch.uint8, device=dev)
    # -S: the slice of SA values; -s / -e: the values stay in the library (8 bytes per run boundary of the slice)
    sa = torch.empty(hi - lo + 1, dtype=torch.int64, device=dev) if (flags & pfp.FLAG_SA) else None
    torch.cuda.synchronize(dev)
    step.run(lambda: ctx.dist_merge(sym_all.data_ptr(), sym_all.numel(), last_all.data_ptr(), sai_all.data_ptr() if want_sai else None,
                                    flags, n_total, lo, hi, bwt.data_ptr(), sa.data_ptr() if sa is not None else None))
    # --- the reference's output formats.  Run sampling needs one byte of halo from either neighbour:
    #     the BWT byte just before and just after this rank's slice (slices may be empty).
    cnt = hi - lo
    edge = torch.tensor([cnt, int(bwt[0]) if cnt and step.err is None else 0, int(bwt[cnt - 1]) if cnt and step.err is None else 0,
                         int(step.status()[0])], dtype=torch.int64, device=dev)
    edges = yield ("allgather", edge)
    step.check([e[3:4] for e in edges], "merge")
    out = dict(bwt=bwt[:cnt], sa=sa[:cnt] if sa is not None else None, lo=lo, hi=hi, n_total=n_total)
    # every allocation / library call below runs through `step`: a rank that fails here (out of memory, a limit) still
    # enters the next all-gather with its status, and all ranks stop together
    if flags & pfp.FLAG_SA:
        # .sa holds SA[1..n] (SA[0] = n is not written: pfbwt.cpp:158-162)
        first = 1 if lo == 0 else 0
        k = max(cnt - first, 0)

        def pack():
            sa5 = torch.empty(5 * k + 16, dtype=torch.uint8, device=dev)
            if k:
                ctx.pack5_dev(sa.data_ptr() + 8 * first, k, sa5.data_ptr())
            return sa5
        sa5 = step.run(pack)
        st = yield ("allgather", step.status())
        step.check(st, "5-byte packing")
        out["sa5"], out["sa5_off"] = sa5[: 5 * k], 5 * (lo + first - 1)
    if flags & (pfp.FLAG_SSA | pfp.FLAG_ESA):
        lefts = [int(e[2]) for e in edges[:rank] if int(e[0]) > 0]
        rights = [int(e[1]) for e in edges[rank + 1:] if int(e[0]) > 0]
        lb, rb = (lefts[-1] if lefts else -1), (rights[0] if rights else -1)
        counts = {}

        def sample(run_end):
            # the slice's own edge is a run start (end) unless the neighbour's adjacent byte is the same
            drop = bool(cnt) and ((rb >= 0 and rb == int(edges[rank][2])) if run_end else (lb >= 0 and lb == int(edges[rank][1])))
            k = ctx.dist_sample_runs(run_end, drop) if cnt else 0
            buf = torch.empty(10 * k + 16, dtype=torch.uint8, device=dev)
            if k:
                ctx.dist_sample_runs(run_end, drop, buf.data_ptr(), k)
            return buf[: 10 * k], k
        for key, flag, run_end in (("ssa", pfp.FLAG_SSA, False), ("esa", pfp.FLAG_ESA, True)):
            if flags & flag:
                got = step.run(lambda: sample(run_end))
                out[key], counts[key] = got if got is not None else (None, 0)
        ks = yield ("allgather", torch.tensor([counts.get("ssa", 0), counts.get("esa", 0), int(step.status()[0])], dtype=torch.int64, device=dev))
        step.check([t[2:3] for t in ks], "run sampling")
        for j, key in enumerate(("ssa", "esa")):
            if key in out:
                out[key + "_off"] = 10 * sum(int(t[j]) for t in ks[:rank])
        out["sampled_total"] = {key: sum(int(t[j]) for t in ks) for j, key in enumerate(("ssa", "esa")) if key in out}
    out["stats"] = dict(local=info, glob=ginfo, phrases_total=int(sym_all.numel()), shard_bytes=n_shard, extra_triggers=len(extra),
                        parse_density=(__import__("struct").unpack("<d", __import__("struct").pack("<Q", plan[2]))[0] if plan[0] else 1.0),
                        sa_shares=parts, parse_shares=parse_shares, dedup=dedup)
    return out


def output_sizes(res):
    """final byte size of every output file of the run `res` is a piece of (the same on every rank)"""
    n = res["n_total"]
    sizes = {".bwt": n + 1}
    if "sa5" in res:
        sizes[".sa"] = 5 * n
    for key in ("ssa", "esa"):
        if key in res:
            sizes["." + key] = 10 * res["sampled_total"][key]
    return sizes


def create_outputs(path, res):
    """ONE rank, before anybody writes: create every output file with its final size.  The reference opens its outputs
    "wb" (pfbwt.cpp:132-141): whatever an earlier, longer run left under the same name is gone."""
    import os
    for ext, size in output_sizes(res).items():
        with open(path + ext, "wb") as f:
            f.truncate(size)
            os.fsync(f.fileno())


def write_outputs(ctx, path, res, group=None, create=None):
    """Every rank writes its pieces of <path>.bwt / .sa / .ssa / .esa at their file offsets (the reference's
    threads pwrite their ranges the same way, pfthreads.hpp:369-376).  Device buffers are streamed to the
    file through the library's pinned staging buffers (pfp_pwrite_dev, C host code).
    Under torch.distributed rank 0 first creates the files at their final sizes (create_outputs) and all ranks meet
    at a barrier; `create` = True / False overrides that choice (simulated ranks: the caller creates once)."""
    import torch.distributed as dist
    ranked = dist.is_available() and dist.is_initialized()
    if create is None:
        create = (dist.get_rank(group) == 0) if ranked else True
    if create:
        create_outputs(path, res)
    if ranked:
        dist.barrier(group)
    ctx.pwrite_dev(path + ".bwt", res["lo"], res["bwt"].data_ptr(), res["bwt"].numel())
    for key, ext in (("sa5", ".sa"), ("ssa", ".ssa"), ("esa", ".esa")):
        if key in res:
            ctx.pwrite_dev(path + ext, res[key + "_off"], res[key].data_ptr(), res[key].numel())


def run(ctx, shard, w=10, p=100, flags=0, halo=DEFAULT_HALO, group=None, shard_sa=True, dedup="alltoall", shard_parse=True):
    """Drive `phases` with torch.distributed (backend nccl == RCCL on ROCm; gloo works for CPU tests of the plumbing)."""
    import torch.distributed as dist
    rank, size = dist.get_rank(group), dist.get_world_size(group)
    gen = phases(ctx, shard, rank, size, w, p, flags, halo, shard_sa, dedup, shard_parse)
    reply = None
    import time
    log = []          # (kind, bytes this rank contributed, bytes it received, ms) per collective, in order
    try:
        while True:
            kind, payload = gen.send(reply)
            t0 = time.perf_counter()
            if kind == "allgather":
                reply = all_gather_var(payload, group)
                sent = payload.numel() * payload.element_size()
            else:
                assert kind == "alltoall"
                reply = all_to_all_var(payload, group)
                sent = sum(x.numel() * x.element_size() for x in payload)
            if shard.is_cuda:
                torch.cuda.synchronize(shard.device)
            log.append((kind, sent, sum(x.numel() * x.element_size() for x in reply), (time.perf_counter() - t0) * 1e3))
    except StopIteration as fin:
        res = fin.value
        res["stats"]["collectives"] = dict(
            backend=dist.get_backend(group), ranks=size, count=len(log), ms=round(sum(x[3] for x in log), 3),
            bytes_sent=sum(x[1] for x in log), bytes_received=sum(x[2] for x in log),
            largest=[dict(kind=k, bytes_sent=b, bytes_received=r, ms=round(m, 3)) for k, b, r, m in sorted(log, key=lambda x: -x[2])[:6]])
        return res


def simulate(ctxs, shards, w=10, p=100, flags=0, halo=DEFAULT_HALO, shard_sa=True, dedup="alltoall", trim=False, shard_parse=True):
    """Run R virtual ranks in one process (ctxs[r], shards[r] may all live on one GPU): every
    collective is served by plain concatenation.  Used by the GPU tests to check the distributed
    chain bit for bit against the single-GPU chain."""
    size = len(shards)
    gens = [phases(ctxs[r], shards[r], r, size, w, p, flags, halo, shard_sa, dedup, shard_parse) for r in range(size)]
    replies = [None] * size
    results = [None] * size
    live = set(range(size))
    while live:
        reqs, kinds = {}, set()
        for r in sorted(live):
            try:
                kind, payload = gens[r].send(replies[r])
                kinds.add(kind)
                reqs[r] = payload
            except StopIteration as fin:
                results[r] = fin.value
            if trim is True or (trim and r in trim):   # many virtual ranks on one card: a rank's cached device blocks must
                ctxs[r].pool_trim()                    # not starve the next one (trim = True, or the set of ranks to trim)
        live = set(reqs)
        if live:
            assert len(live) == size and len(kinds) == 1, "ranks fell out of step"
            if kinds == {"allgather"}:
                gathered = [reqs[r] for r in range(size)]
                replies = [list(gathered) for _ in range(size)]
            else:
                replies = [[reqs[s_][r] for s_ in range(size)] for r in range(size)]
    return results
