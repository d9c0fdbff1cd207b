"""One rank of `bigbwt -G N`: the C driver (host/bigbwt.c) starts N of these through `python -m torch.distributed.run`.

    dist_main.py --text FILE --base NAME [-w W] [-p M] [--flags F] [--halo H] [-v]

Rank r reads bytes [n*r/N, n*(r+1)/N) of FILE (the reference's threaded parser splits the input the same way,
pscan.hpp:114-165), runs the distributed chain (dist.run) and writes its pieces of NAME.bwt / .sa / .ssa / .esa at their
file offsets (dist.write_outputs).  Rank 0 appends the counts the reference logs (newscan.cpp:396-402) to NAME.log.
Environment: PFP_DIST_BACKEND (default nccl = RCCL), PFP_DIST_ONE_GPU=1 puts every rank on device 0 (rehearsal of
the N>1 path on a one-GPU box, with PFP_DIST_BACKEND=gloo).  Any failure on any rank ends every rank with a non-zero
exit code (the status exchange in dist.phases), which the launcher and then the C driver pass on.
"""
import argparse
import importlib.util
import os
import sys
import time


def load_package():
    """this directory's name carries a hyphen: register it as `bigbwt_amd` (as __graft_entry__.load_package does)"""
    if "bigbwt_amd" in sys.modules:
        return sys.modules["bigbwt_amd"]
    here = os.path.dirname(os.path.abspath(__file__))
    spec = importlib.util.spec_from_file_location("bigbwt_amd", os.path.join(here, "__init__.py"), submodule_search_locations=[here])
    mod = importlib.util.module_from_spec(spec)
    sys.modules["bigbwt_amd"] = mod
    spec.loader.exec_module(mod)
    return mod


def shard_range(n, rank, size):
    return n * rank // size, n * (rank + 1) // size


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--text", required=True, help="the bytes to transform (the input, or the sequences the driver filtered out of a FASTA file)")
    ap.add_argument("--base", required=True, help="outputs are BASE.bwt, BASE.sa ...")
    ap.add_argument("-w", type=int, default=10)
    ap.add_argument("-p", type=int, default=100)
    ap.add_argument("--flags", type=int, default=0)
    ap.add_argument("--halo", type=int, default=0)
    ap.add_argument("-v", action="store_true")
    a = ap.parse_args(argv)

    import numpy as np
    import torch
    import torch.distributed as dist
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    local = 0 if os.environ.get("PFP_DIST_ONE_GPU") else int(os.environ.get("LOCAL_RANK", rank))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if not torch.cuda.is_available() or local >= torch.cuda.device_count():
        print(f"bigbwt: rank {rank}: no GPU {local} to run on (this tool has no CPU path)", file=sys.stderr, flush=True)
        return 1
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist.init_process_group(backend=os.environ.get("PFP_DIST_BACKEND", "nccl"), rank=rank, world_size=world)
    pkg = load_package()          # raises if libpfpgpu.so is missing: there is no CPU path
    d = importlib.import_module("bigbwt_amd.dist")

    n = os.path.getsize(a.text)
    lo, hi = shard_range(n, rank, world)
    t0 = time.perf_counter()
    host = np.fromfile(a.text, dtype=np.uint8, count=hi - lo, offset=lo) if hi > lo else np.empty(0, np.uint8)
    shard = torch.from_numpy(host).to(dev)
    del host
    ctx = pkg.Context(local)
    code = 0
    try:
        res = d.run(ctx, shard, a.w, a.p, a.flags, halo=a.halo or d.DEFAULT_HALO)
        t1 = time.perf_counter()
        d.write_outputs(ctx, a.base, res)
        dist.barrier()
        if rank == 0:
            st = res["stats"]
            with open(a.base + ".log", "a") as log:
                log.write(f"Ranks: {world}\nFound {st['glob']['words']} distinct words\nTotal number of words: {st['phrases_total']}\n")
            if a.v:
                c = st["collectives"]
                print(f"  {world} ranks ({c['backend']}): {c['count']} collectives {c['ms']:.1f} ms, {c['bytes_received']} bytes received per rank; "
                      f"chain {1e3 * (t1 - t0):.1f} ms, files {1e3 * (time.perf_counter() - t1):.1f} ms", flush=True)
    except pkg.PfpError as ex:
        # every rank raises at the same collective (dist._Step), so nobody is left waiting
        if rank == 0:
            print(f"bigbwt: {ex}", file=sys.stderr, flush=True)
        code = 1
    ctx.close()
    dist.destroy_process_group()
    return code


if __name__ == "__main__":
    sys.exit(main())
