#!/usr/bin/env python3
"""bench.py -- headline benchmark: input MB/s to bit-exact .bwt (+ sampled SA files) on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c3|c2|huge|...]

A "step" is one complete pass of the hot path (window scan -> phrase dedup -> dictionary suffix sort
-> BWT of the parse -> merge -> run sampling / 5-byte packing) over one synthetic input that is already
resident in HBM when the timed region starts, leaving the finished reference-format outputs (.bwt
bytes, and the .ssa/.esa/.sa bytes the workload's flags ask for) in HBM.  Default workload at N = 1:
BASELINE.json configs[2], the largest single-GPU configuration (64 mutated copies of a yeast-sized
genome, ~0.79 GB, -w 10 -p 100, BWT + -s -e).  `--workload c2` is configs[1] (chr1-shaped, 253 MB, BWT
only), `--workload huge` / `huge_s` the north star's >= 10 GB repetitive input on one GPU.
One process per GPU.  For N > 1 the ranks build ONE BWT of a collection of N such inputs (rank r holds
variant r of the same base sequences - a pangenome slice per GPU, weak scaling) through
big-bwt_amd/dist.py; `--multi independent` lets every rank build the BWT of its own text instead.

Rank 0 prints ONE JSON line.  `roofline` = the kernel with the largest share of the step, timed live
with HIP events on the library's own stream; `roofline_passes` = SURVEY.md 8(d)'s per-pass algorithmic
bytes over the measured pass times; `cpu_baseline` = the real reference (oracle/_ref: pscan.x -t N ->
bwtparse -t N -> pfbwt[NT].x, N = host cores) timed on this host on a bounded prefix of the same text.
"""
import argparse
import hashlib
import importlib
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md); measured copy peak ~6.3 TB/s


def sha_dev(t):
    """sha256 of a device byte tensor (copied out in 256 MB pieces)"""
    h = hashlib.sha256()
    for s in range(0, t.numel(), 1 << 28):
        h.update(t[s:s + (1 << 28)].cpu().numpy().tobytes())
    return h.hexdigest()


def cpu_baseline(text_host, w, p, flags, O, threads):
    """the real reference on a prefix of the text: pscan.x -t N -> bwtparse -t N -> pfbwt.x -t N (pfbwtNT.x
    when -s/-e is asked for, as bigbwt:132,141 does)"""
    if O.have_ref():
        r = O.run_ref(text_host.tobytes(), w, p, flags, threads=threads, want_intermediates=False)
        secs = sum(r["seconds"].values())
        kind = "reference"
        outs = {k: np.frombuffer(r[k], dtype=np.uint8) for k in ("bwt", "sa", "ssa", "esa") if k in r}
        detail = {k: round(v, 3) for k, v in r["seconds"].items()}
        cores = threads if threads > 0 else 1
    else:
        t0 = time.time()
        o = O.bigbwt(text_host, w, p, flags)
        secs = time.time() - t0
        kind, detail, cores = "port", {}, 1
        outs = {"bwt": o["bwt"]}
    return dict(value=round(len(text_host) / secs / 1e6, 3), unit="MB/s", cores=cores, kind=kind,
                sample=f"{len(text_host)} bytes of the same text (the first, = all of it unless --cpu-sample-mb says otherwise), same flags, -t {threads}; stage seconds: {detail}",
                seconds=round(secs, 3)), outs


def pass_rows(st, w, flags, n_slice, P_merge, D_slots, R, t_hash, t_formats, ms_per_step):
    """algorithmic bytes of the scan, phrase-hash and merge passes over their measured times"""
    P, H, nn = st["n_phrases"], st["hard_chars"], st["n"]
    b_scan = nn + 8 * P
    b_hash = nn + w * P + 8 * P
    b_merge = n_slice + 12 * D_slots + 5 * (P_merge + 1) + 8 * H
    if flags & 1:
        b_merge += 10 * n_slice
    b_merge_moved = b_merge
    if flags & 6:
        b_merge += 10 * (R.get("ssa", 0) + R.get("esa", 0)) + 5 * n_slice
        # (8d's "5 n gathered from bwsai" is what a merge that computes every SA value reads; the sparse SA reads 16 bytes per RUN)
        b_merge_moved += 10 * (R.get("ssa", 0) + R.get("esa", 0)) + 16 * max(R.get("ssa", 0), R.get("esa", 0))
    t_merge = st["ms_merge"] + t_formats

    def row(nbytes, ms):
        if ms <= 0:
            return None
        g = nbytes / (ms * 1e-3) / 1e9
        return dict(algo_bytes=int(nbytes), ms=round(ms, 3), achieved_GBps=round(g, 1), frac=round(g / HBM_PEAK_GBS, 4))
    passes = {"scan (K1+K2: n + 8P)": row(b_scan, st["ms_scan"]),
              "phrase hash (n + wP + 8P)": row(b_hash, t_hash),
              "merge ((n+1) + 12|D| + 5(P+1) + 8H [+ SA terms]; incl. run sampling / packing)": row(b_merge, t_merge),
              "merge, counting only what the sparse SA moves (16 B per run instead of 8d's 5 n)": row(b_merge_moved, t_merge),
              "scan + merge": row(b_scan + b_merge, st["ms_scan"] + t_merge),
              "end to end floor (2n)": row(2 * nn, ms_per_step)}
    bad = [k for k, v in passes.items() if v and v["frac"] > 1.0]
    assert not bad, f"pass fraction above 1 ({bad}): the timed figure is not the work"
    return passes


def north_star_leg(pkg, synth, ctx, dev, steps=2, with_cli=True):
    """the north star's own workload, driver-visible: >= 10 GB of repetitive FASTA (1024 mutated copies, 12.6 GB) -> .bwt +
    .ssa on ONE GPU, `steps` timed steps after one warm-up, outputs compared with the digests of the real reference's files
    (tests/golden/golden_full.json: 42 min of oracle/_ref on one core in the build container), pass fractions as above."""
    name = "huge_s"
    wl = synth.WORKLOADS[name]
    w, p, flags = wl["w"], wl["p"], wl["flags"]
    t_gen = time.perf_counter()
    text = synth.workload_text_torch(dev, name)
    n = text.numel()
    torch.cuda.empty_cache()
    torch.cuda.synchronize()
    gen_s = time.perf_counter() - t_gen
    # the drop-in first, while this process holds nothing but the text: the child then starts on an idle card (memory a process
    # has just freed is scrubbed by the driver before it is handed out again - ~30 ms per GB - and a child started right after
    # this process gave back a 142 GB pool spent 6 s of its "text in" waiting for that)
    cli = None
    if with_cli:
        try:
            gold0 = json.load(open(os.path.join(ROOT, "tests", "golden", "golden_full.json"))).get(name)
            cli = cli_leg(text, w, p, flags, gold0 if gold0 and gold0["n"] == n else None)
        except Exception as ex:
            cli = {"error": f"{type(ex).__name__}: {ex}"}
    outs = {}
    bwt = torch.empty(n + 1 + 16, dtype=torch.uint8, device=dev)

    def step():
        for ptr, _ in outs.values():
            ctx.dev_free(ptr)
        outs.clear()
        used, o = ctx.bigbwt_formats_dev(text.data_ptr(), n, bwt.data_ptr(), w, p, flags)
        outs.update(o)
        return used
    # two warm-up steps: the pool's cached blocks settle into the chain's request order only with the second (after one, the timed
    # steps still went to the driver three times - at ~30 ms per GB)
    step()
    step()
    torch.cuda.synchronize()
    pc0 = ctx.pool_counters()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    pc1 = ctx.pool_counters()
    ctx.set_kernel_trace(True)
    ctx.set_profiling(True)
    step()
    torch.cuda.synchronize()
    st = ctx.stats()
    kt = {r["name"]: r["total_ms"] for r in ctx.kernel_trace()}
    ctx.set_profiling(False)
    ctx.set_kernel_trace(False)
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "golden_full.json"))).get(name)
    digests = None
    if gold and gold["n"] == n:
        digests = {"text": sha_dev(text) == gold["text_sha256"], "bwt": sha_dev(bwt[: n + 1]) == gold["bwt_sha256"]}
        for key in ("ssa", "esa"):
            if key in outs and key + "_sha256" in gold:
                digests[key] = hashlib.sha256(ctx.fetch_dev(*outs[key]).tobytes()).hexdigest() == gold[key + "_sha256"]
    R = {k: outs[k][1] // 10 for k in ("ssa", "esa") if k in outs}
    t_formats = sum(kt.get(k, 0.0) for k in ("pfp::run_count_kernel", "pfp::run_place_kernel", "pfp::bitmap_place_kernel", "pfp::pack5_kernel"))
    passes = pass_rows(st, w, flags, st["n"] + 1, st["n_phrases"], st["dict_size"], R, kt.get("pfp::phrase_hash_kernel", 0.0), t_formats, ms)
    mem = ctx.mem_stats()
    top = sorted(kt.items(), key=lambda x: -x[1])[:8]
    res = dict(workload=wl["desc"], bytes=n, steps=steps, ms_per_step=round(ms, 2), value=round(n / ms / 1e3, 1), unit="MB/s",
               outputs=["bwt"] + sorted(outs), outputs_match_reference_digests=digests, roofline_passes=passes,
               phases_ms={k: round(st[k], 2) for k in ("ms_scan", "ms_phrases", "ms_sa_dict", "ms_sa_parse", "ms_merge")},
               top_kernels_ms={k: round(v, 2) for k, v in top}, words=st["n_words"], dict_bytes=st["dict_size"], phrases=st["n_phrases"], runs=R,
               parse_density=st.get("parse_density"),
               peak_device_bytes=mem["peak"], driver_allocations_in_timed_steps=pc1["driver_allocs"] - pc0["driver_allocs"],
               text_generation_s=round(gen_s, 2))
    for ptr, _ in outs.values():
        ctx.dev_free(ptr)
    del bwt
    if with_cli:
        res["cli_file_to_file"] = cli
    del text
    return res


def cli_leg(text, w, p, flags, gold, ctx=None):
    """the drop-in itself: `bigbwt` (the C driver) in a cold process on a file in /dev/shm - process wall time and, from
    PFP_TRACE_HOST, where it goes (text in: mmap -> pinned chunks -> HBM; chain: on a cold pool, i.e. including every
    hipMalloc; files out: HBM -> pinned chunks -> pwrite); the rest is process start, context and teardown.  The calling
    process gives its cached device memory back first (the child needs the card)."""
    import shutil
    import subprocess
    import tempfile
    n = int(text.numel())
    tmpd = tempfile.mkdtemp(prefix="pfpbench_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    try:
        fn = os.path.join(tmpd, "t.fa")
        with open(fn, "wb") as fh:
            for s0 in range(0, n, 1 << 28):
                fh.write(text[s0:s0 + (1 << 28)].cpu().numpy().tobytes())
        if ctx is not None:
            ctx.pool_trim()
        torch.cuda.empty_cache()
        cmd = [os.path.join(ROOT, "big-bwt_amd", "bigbwt"), "-w", str(w), "-p", str(p)]
        cmd += [f for f, bit in (("-S", 1), ("-s", 2), ("-e", 4)) if flags & bit] + [fn]
        # untimed, so that the child meets what a user's run meets - a file that has been lying in the page cache and a card whose
        # free memory is clean: one read pass over the file just written (the first reader of fresh tmpfs pages gets 6 GB/s, later
        # ones 40: profiles/r04_cli_probe_mapped.txt, first run of every series), and a pause in which the driver finishes scrubbing
        # the device memory this process has just given back (without it the child's cold pool costs 3.6 s instead of 0.75 for 12.6 GB)
        settle = float(os.environ.get("PFP_BENCH_CLI_SETTLE", "10" if n > (2 << 30) else "3"))
        t_settle = time.perf_counter()
        with open(fn, "rb", buffering=0) as fh:
            buf = bytearray(1 << 26)
            while fh.readinto(buf):
                pass
        time.sleep(max(0.0, settle - (time.perf_counter() - t_settle)))
        import re

        def one_run():
            for ext in ("bwt", "sa", "ssa", "esa", "log"):
                if os.path.exists(fn + "." + ext):
                    os.remove(fn + "." + ext)
            t1 = time.perf_counter()
            pr = subprocess.run(cmd, capture_output=True, text=True, env=dict(os.environ, PFP_TRACE_HOST="1"))
            cli_s = time.perf_counter() - t1
            inner = [ln for ln in pr.stdout.splitlines() if ln.startswith("Total construction time")]
            split = None
            for ln in pr.stderr.splitlines():
                if "file to files" in ln:
                    m = re.search(r"text in ([0-9.]+) ms, chain ([0-9.]+) ms.*files out ([0-9.]+) ms", ln)
                    if m:
                        ti, ch, fo = (float(x) / 1e3 for x in m.groups())
                        split = dict(text_in_s=round(ti, 3), chain_on_cold_pool_s=round(ch, 3), files_out_s=round(fo, 3),
                                     start_context_teardown_s=round(cli_s - ti - ch - fo, 3))
            return pr, cli_s, inner, split
        # A second cold process, 8 s later, when the first one's pool cost more than the whole warm chain: device memory that any
        # process has freed within the last seconds costs ~30 ms per GB to get (tools/microbench/alloc.hip: 3.6-4.1 s for 128 GB in
        # back-to-back processes), memory that has lain free for a while a tenth of that (the standalone probe's runs, 5 s apart:
        # 0.6-0.7 s of chain each), and the pause above is not always enough.  Both attempts are reported; the better one is the figure.
        attempts = []
        best = None
        for k in range(2):
            if k:
                time.sleep(8.0)          # (what the first child gave back is being cleaned too)
            r = one_run()
            attempts.append(dict(seconds_process=round(r[1], 3), split=r[3]))
            if best is None or (r[0].returncode == 0 and r[1] < best[1]):
                best = r
            if r[0].returncode != 0 or r[3] is None or r[3]["chain_on_cold_pool_s"] <= 0.05 + 1.2e-10 * n:
                break
        pr, cli_s, inner, split = best
        ok = pr.returncode == 0 and os.path.getsize(fn + ".bwt") == n + 1
        if ok and gold is not None:
            h = hashlib.sha256()
            with open(fn + ".bwt", "rb") as fh:
                for blk in iter(lambda: fh.read(1 << 26), b""):
                    h.update(blk)
            ok = h.hexdigest() == gold["bwt_sha256"]
        return dict(MBps_process=round(n / cli_s / 1e6, 1), seconds_process=round(cli_s, 3),
                    seconds_construction=float(inner[0].split(":")[1]) if inner else None, outputs_ok=bool(ok), split=split, attempts=attempts,
                    note="bigbwt (C driver), cold process, file in /dev/shm: parallel pread into pinned chunks -> H2D -> chain -> outputs copied from HBM "
                         "straight into the mapped, registered pages of their files (a helper allocates and registers them beside the input and the chain: "
                         "1.6-1.9 s for 12.6 GB, the critical path; DESIGN.md section 5); device memory another process has just freed costs ~30 ms per GB "
                         "when it is handed out again (tools/microbench/alloc.hip), so the child is started PFP_BENCH_CLI_SETTLE seconds (default 10 / 3) after "
                         "this process gave its memory back and after one untimed read pass over the input file; the C driver alone on an idle card: "
                         "12.6 GB in 2.1-2.7 s (profiles/r04_cli_probe_mapped.txt; 3.1-3.3 s through pinned buffers and pwrite)")
    finally:
        shutil.rmtree(tmpd, ignore_errors=True)


def launch_ranks(n):
    """`python bench.py --gpus N` without RANK in the environment: run `python -m torch.distributed.run --nproc-per-node N
    bench.py <same arguments>` as a child (this parent has not touched the GPU and never will), pass its output
    through and return its exit code."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default=None)
    ap.add_argument("--cpu-sample-mb", type=float, default=0.0, help="prefix of the text the reference is timed on (0, the default: the whole text, ~100 s at -t 16)")
    ap.add_argument("--cpu-threads", type=int, default=16, help="-t N of the reference's threaded parser / merge (a one-GPU box gives 16 host cores per GPU)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-host-boundary", action="store_true")
    ap.add_argument("--multi", default="collection", choices=["collection", "independent"],
                    help="N>1: one BWT of a sharded collection (RCCL exchanges) or one independent text per GPU")
    ap.add_argument("--north-star", dest="north_star", default="auto", choices=["auto", "on", "off"],
                    help="N=1, default workload: also time 2 steps of the >= 10 GB north-star workload (huge_s) with its digest check")
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ:
        # no launcher around us: start the N ranks ourselves (one process per GPU) BEFORE anything touches the GPU, as
        # child processes - a process that has initialised HIP must never exec - and relay rank 0's JSON line
        sys.exit(launch_ranks(args.gpus))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    if world > 1:
        import torch.distributed as dist_mod
        dist = dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # PFP_BENCH_BACKEND=gloo PFP_BENCH_ONE_GPU=1: rehearsal of the N>1 path on a one-GPU box (all ranks on cuda:0)
        dist.init_process_group(backend=os.environ.get("PFP_BENCH_BACKEND", "nccl"), rank=rank, world_size=world)
    assert world == args.gpus, f"--gpus {args.gpus} but the launcher started {world} rank(s)"
    if os.environ.get("PFP_BENCH_ONE_GPU"):
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    pkg = entry.load_package()          # fails loudly if libpfpgpu.so is missing
    synth = importlib.import_module("bigbwt_amd.synth")
    collection = world > 1 and args.multi == "collection"
    wl_name = args.workload or "c3"
    wl = synth.WORKLOADS[wl_name]
    w, p, flags = wl["w"], wl["p"], wl["flags"]
    O = entry.load_oracle() if (rank == 0 and not args.no_cpu_baseline and world == 1) else None

    # rank r holds variant r of the same base sequences (collection) or its own text (independent: other seed)
    variant = rank if world > 1 else 0
    text = synth.workload_text_torch(dev, wl_name, variant=variant)
    n = text.numel()
    bwt = torch.empty(n + 1 + 16, dtype=torch.uint8, device=dev)
    torch.cuda.empty_cache()          # the generator's temporaries go back to the driver
    torch.cuda.synchronize()

    ctx = pkg.Context(local_rank)
    dist_mod_pfp = importlib.import_module("bigbwt_amd.dist") if collection else None
    last = {}
    outbuf = {}          # reference-format outputs in HBM: sa5 / ssa / esa (sized by the first warm-up step)

    def step(sizing=False):
        """one pass of the hot path: device text -> .bwt bytes in `bwt` and, as the flags ask, the .sa / .ssa / .esa bytes in
        device buffers of the library (SA values never leave the call: pfp_bigbwt_formats_dev)"""
        if collection:
            last["r"] = dist_mod_pfp.run(ctx, text, w, p, flags)
            return n
        for ptr, _ in outbuf.values():
            ctx.dev_free(ptr)
        outbuf.clear()
        if flags:
            used, outs = ctx.bigbwt_formats_dev(text.data_ptr(), n, bwt.data_ptr(), w, p, flags)
            outbuf.update(outs)
            for k, (_, nb) in outs.items():
                last[k + "_bytes"] = nb
        else:
            used = ctx.bigbwt_dev(text.data_ptr(), n, bwt.data_ptr(), None, w, p, 0)
        last["n_used"] = used
        return used

    def barrier():
        if dist is not None:
            dist.barrier()

    step(sizing=True)
    for _ in range(max(0, args.warmup - 1)):
        step()
    barrier(); torch.cuda.synchronize()
    pc0 = ctx.pool_counters()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize(); barrier()
    elapsed = time.perf_counter() - t0
    pc1 = ctx.pool_counters()
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        tot = torch.tensor([float(n)], dtype=torch.float64, device=dev)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        total_bytes = float(tot.item())
    else:
        total_bytes = float(n)
    ms_per_step = elapsed / args.steps * 1e3
    value = total_bytes * args.steps / elapsed / 1e6
    # per-kernel device times: HIP events around every kernel on the library's own stream, over TRACE_STEPS further steps
    # of the same input right after the timed ones (two events per launch inside the timed region cost ~1 ms per step)
    TRACE_STEPS = 2
    ctx.set_kernel_trace(True)
    t1 = time.perf_counter()
    for _ in range(TRACE_STEPS):
        step()
    torch.cuda.synchronize()
    traced_ms_per_step = (time.perf_counter() - t1) / TRACE_STEPS * 1e3
    ktable = ctx.kernel_trace()
    ctx.set_kernel_trace(False)

    # per-phase breakdown of one extra (untimed) profiled step
    ctx.set_profiling(True)
    t1 = time.perf_counter()
    step()
    torch.cuda.synchronize()
    prof_ms = (time.perf_counter() - t1) * 1e3
    st = ctx.stats()
    ctx.set_profiling(False)
    mem = ctx.mem_stats()

    # ---- correctness of what was just measured (outside the timed region)
    def hist(t):
        h = torch.zeros(256, dtype=torch.int64, device=dev)
        for s in range(0, t.numel(), 1 << 28):
            h += torch.bincount(t[s:s + (1 << 28)].to(torch.int64), minlength=256)
        return h
    hist_t = hist(text)
    if collection:
        res = last["r"]
        hist_b = hist(res["bwt"])
        dist.all_reduce(hist_t); dist.all_reduce(hist_b)
        hist_t[0] += 1
        dstats = res["stats"]
    else:
        hist_t[0] += 1
        hist_b = hist(bwt[: n + 1])
    permutation_ok = bool(torch.equal(hist_t, hist_b))

    out = None
    if rank == 0:
        # ---- the measured device buffers against the reference's digests for this exact text (tests/golden/golden_full.json,
        #      made by running oracle/_ref on the same synthetic text in the build container)
        digests = None
        if world == 1:
            try:
                gold = json.load(open(os.path.join(ROOT, "tests", "golden", "golden_full.json"))).get(wl_name)
            except Exception:
                gold = None
            if gold and gold["n"] == n:
                digests = {"text": sha_dev(text) == gold["text_sha256"], "bwt": sha_dev(bwt[: n + 1]) == gold["bwt_sha256"]}
                for key in ("sa", "ssa", "esa"):
                    if key in outbuf and key + "_sha256" in gold:
                        digests[key] = hashlib.sha256(ctx.fetch_dev(*outbuf[key]).tobytes()).hexdigest() == gold[key + "_sha256"]

        # ---- roofline: per-kernel device time measured live (HIP events on the ctx stream, timed steps)
        rows = []
        for r in ktable:
            if r["total_ms"] <= 0 or r["launches"] == 0:
                continue
            gbs = r["algo_bytes"] / (r["total_ms"] * 1e-3) / 1e9
            rows.append(dict(kernel=r["name"], ms_per_step=round(r["total_ms"] / TRACE_STEPS, 3),
                             launches_per_step=round(r["launches"] / TRACE_STEPS, 1),
                             us_per_launch=round(r["total_ms"] / r["launches"] * 1e3, 2),
                             algo_bytes_per_launch=int(r["algo_bytes"] / r["launches"]),
                             achieved_GBps=round(gbs, 1), frac=round(gbs / HBM_PEAK_GBS, 4)))
        rows.sort(key=lambda x: -x["ms_per_step"])
        dom = rows[0]
        roofline = dict(bound="hbm", kernel=dom["kernel"], achieved=dom["achieved_GBps"], peak=HBM_PEAK_GBS, unit="GB/s",
                        frac=dom["frac"], traffic=None, us_per_launch=dom["us_per_launch"],
                        launches_per_step=dom["launches_per_step"], algo_bytes_per_launch=dom["algo_bytes_per_launch"],
                        share_of_step=round(dom["ms_per_step"] / ms_per_step, 3))
        # (the dominant launch may be the library's LSD sort, whose fraction by algorithmic bytes is low by construction - five
        #  to eight passes over the data; the largest HAND-WRITTEN kernel is reported next to it)
        own = next((x for x in rows if x["kernel"].startswith("pfp::")), None)
        if own is not None and own is not dom:
            roofline["largest_own_kernel"] = dict(kernel=own["kernel"], achieved=own["achieved_GBps"], frac=own["frac"], us_per_launch=own["us_per_launch"],
                                                  launches_per_step=own["launches_per_step"], share_of_step=round(own["ms_per_step"] / ms_per_step, 3))
        # HBM traffic of that kernel from the committed PMC passes of the same workload, per launch like `achieved`
        for rnd in ("r04",):          # (same round only: the kernels of earlier rounds are not this code)
            try:
                pmc = json.load(open(os.path.join(ROOT, "profiles", f"{rnd}_{wl_name}_pmc_traffic.json")))
                base = dom["kernel"].split(" [")[0]         # a library sort's label carries its call site: "... [what for]"
                if world == 1 and base in pmc["kernels"]:
                    tot = pmc["kernels"][base]["hbm_bytes_per_chain_corrected"]
                    src = f"profiles/{rnd}_{wl_name}_pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE, WRITE_SIZE in separate passes; 2*FETCH+WRITE)"
                    if base != dom["kernel"]:
                        # the counters see the library's kernels of ALL call sites of this sort type: this site's part by its
                        # share of the type's algorithmic bytes (every site moves n * 2 * (key + value) per digit pass)
                        same = [x for x in rows if x["kernel"].split(" [")[0] == base]
                        allb = sum(x["algo_bytes_per_launch"] * x["launches_per_step"] for x in same)
                        tot *= dom["algo_bytes_per_launch"] * dom["launches_per_step"] / max(1, allb)
                        src += "; the sort type's bytes apportioned to this call site by algorithmic bytes"
                    roofline["traffic"] = int(tot / dom["launches_per_step"])
                    roofline["traffic_source"] = src
                    break
            except Exception:
                pass

        # ---- per-pass fractions with SURVEY.md 8(d)'s algorithmic-byte formulas over the synced phase times of the profiled step
        # (N > 1: THIS RANK's share of every pass over this rank's phase timers - its shard and own phrases for the scan
        #  and the hash, its slice of the BWT, the range of SA(D) it holds and the whole parse it reads for the merge)
        R = {k: last.get(k + "_bytes", 0) // 10 for k in ("ssa", "esa")}
        n_slice, P_merge, D_ = st["n"] + 1, st["n_phrases"], st["dict_size"]
        if collection:
            res = last["r"]
            n_slice, P_merge, D_ = res["hi"] - res["lo"], dstats["phrases_total"], dstats["glob"]["slots"]
            R = {k: (res[k].numel() // 10 if k in res and res[k] is not None else 0) for k in ("ssa", "esa")}
        ktime = {r["kernel"]: r["ms_per_step"] for r in rows}
        t_hash = ktime.get("pfp::phrase_hash_kernel", 0.0)
        t_formats = sum(ktime.get(k, 0.0) for k in ("pfp::run_count_kernel", "pfp::run_place_kernel", "pfp::bitmap_place_kernel", "pfp::pack5_kernel"))
        passes = pass_rows(st, w, flags, n_slice, P_merge, D_, R, t_hash, t_formats, ms_per_step)

        scan_row = next((x for x in rows if x["kernel"] in ("pfp::kr_flag_kernel", "pfp::kr_scan_kernel")), None)
        # the host-buffer entry point (H2D of the text + D2H of the outputs included): reported, never `value`
        host_boundary, host_ok = None, None
        if world == 1 and not args.no_host_boundary and n <= (2 << 30):
            host_text = text.cpu().numpy()
            ctx.bigbwt(host_text[: 1 << 20], w, p, flags)          # warm the pinned staging buffers
            t1 = time.perf_counter()
            hb = ctx.bigbwt(host_text, w, p, flags)
            host_s = time.perf_counter() - t1
            host_boundary = dict(MBps=round(n / host_s / 1e6, 1), seconds=round(host_s, 4),
                                 frac_of_device_resident=round((n / host_s / 1e6) / value, 3),
                                 note="pfp_bigbwt: host text in, host .bwt/.ssa/.esa out (PCIe inclusive)")
            host_ok = bool(np.array_equal(hb["bwt"], bwt[: n + 1].cpu().numpy()))
            del hb, host_text
        # the C `bigbwt` driver, file to files (process start, HIP initialisation, page-cache reads and writes included)
        cli = None
        if world == 1 and not args.no_host_boundary and n <= (2 << 30):
            cli = cli_leg(text, w, p, flags, gold if digests is not None else None)
        cpu = None
        parity_sample = None
        if O is not None:
            # N = this process's share of the host: a one-GPU box gives 16 cores per GPU (the affinity mask shows all of the node's)
            affinity = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
            threads = max(1, min(args.cpu_threads, affinity))
            sample_bytes = n if args.cpu_sample_mb <= 0 else int(min(n, args.cpu_sample_mb * 1e6))
            host = text[:sample_bytes].cpu().numpy()
            cpu, ref = cpu_baseline(host, w, p, flags, O, threads)
            cpu["threads_passed"] = threads
            cpu["host_affinity_count"] = affinity
            cpu["cores_note"] = (f"-t {threads} of the {affinity} hardware threads this process may run on (a one-GPU box shares its host: 16 cores per GPU); "
                                 "with -s / -e the reference's last stage is single-threaded whatever -t says (bigbwt:132,141)")
            got = ctx.bigbwt(host, w, p, flags)           # the same sample through the HIP path: bit-exact?
            parity_sample = {k: bool(np.array_equal(got[k], ref[k])) for k in ref if k in got}
        out = {
            "metric": "input MB/s to .bwt (bit-exact vs ref)", "value": round(value, 2), "unit": "MB/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8",
            "data": "synthetic",
            "config": {"workload": wl["desc"], "name": wl_name, "bytes_per_gpu": n, "w": w, "p": p, "flags": flags,
                       "phrases": dstats["phrases_total"] if collection else st["n_phrases"], "words": st["n_words"], "dict_bytes": st["dict_size"],
                       "parse_density": st.get("parse_density"),          # the fused chain cut with probability parse_density / p (pfpgpu.h: pfp_set_parse_density)
                       "outputs_in_timed_step": ["bwt"] + [k for k in ("sa", "ssa", "esa") if k in outbuf or (collection and ("sa5" if k == "sa" else k) in last.get("r", {}))],
                       "parallelism": ("1 GPU" if world == 1 else
                                       (f"{world} shards of one collection over RCCL: halo allgather, hash-partitioned all-to-all phrase dedup, "
                                        f"allgatherv of the distinct words and of the parse, suffix array of the global dictionary sharded "
                                        f"by key range, every rank emits (and samples / packs) the BWT range its share of SA(D) produces") if collection else
                                       f"{world} independent texts (one per GPU), no collective")},
            "roofline": roofline,
            "roofline_passes": passes,
            "kernels": rows[:24],
            "scan_pass_k1": scan_row,
            "cpu_baseline": cpu,
            "host_buffer_boundary": host_boundary,
            "cli_file_to_file": cli,
            "phases_ms": dict({k: round(st[k], 3) for k in ("ms_scan", "ms_phrases", "ms_sa_dict", "ms_sa_parse", "ms_merge", "ms_total")},
                              formats=round(t_formats, 3), profiled_step=round(prof_ms, 3), traced_step=round(traced_ms_per_step, 3)),
            "sa_rounds": {"dict": st["sa_rounds_dict"], "parse": st["sa_rounds_parse"]},
            "merge_stats": {k: st[k] for k in ("hard_groups", "hard_chars", "hard_big_groups", "hard_max_chars", "hard_max_members", "hard_minor_groups", "hard_minor_chars", "extra_triggers", "index_bits")},
            "runs": R,
            "rccl": (dict(dstats["collectives"], ranks=world) if collection else None),
            "device_memory": {"peak_bytes_in_use": mem["peak"], "held_from_driver": mem["held"],
                              "driver_allocations_in_timed_steps": pc1["driver_allocs"] - pc0["driver_allocs"],
                              "pool_trims_in_timed_steps": pc1["trims"] - pc0["trims"]},
            "verified": {"outputs_match_reference_digests_whole_text": digests,
                         "bwt_is_permutation_of_text_plus_eos": permutation_ok,
                         "bit_exact_vs_reference_on_cpu_sample": parity_sample,
                         "host_and_device_entry_points_agree": host_ok},
        }
    if rank == 0 and world == 1 and args.north_star != "off" and (args.north_star == "on" or args.workload is None):
        # the north star's own >= 10 GB workload next to the headline configuration (BASELINE configs[2]): ~15 s of GPU time
        del text, bwt
        torch.cuda.empty_cache()
        if torch.cuda.get_device_properties(dev).total_memory >= (200 << 30):
            try:
                out["north_star"] = north_star_leg(pkg, synth, ctx, dev, with_cli=not args.no_host_boundary)
            except Exception as ex:          # reported, never hidden: the headline line above stands on its own
                out["north_star"] = {"error": f"{type(ex).__name__}: {ex}"}
        else:
            out["north_star"] = {"skipped": "needs ~200 GB of device memory"}
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()
    if rank == 0:
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
