#!/usr/bin/env python3
"""Builder-side run of the 64-bit index build on a dictionary of more than 4 GiB (one GPU).

    python tools/check_wide.py [--workload wide] [--flags 1]

Builds the synthetic workload in HBM (big-bwt_amd/synth.py), runs the device-resident chain with a full SA
(-S) and checks the result through size-independent properties on the GPU:
  * index_bits == 64 and dict_size > 2^32 (the wide build was the one that ran),
  * the BWT is a permutation of text + EOS, SA[0] = n, SA[1..n] is a permutation of 0..n-1,
  * BWT[i] = T[SA[i] - 1] for every i,
  * suffixes at sampled adjacent SA positions are in order (compared byte by byte on the host).
Prints one JSON line (kept under profiles/).
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="wide")
    ap.add_argument("--flags", type=int, default=0)
    ap.add_argument("--samples", type=int, default=4000)
    args = ap.parse_args()
    pkg = entry.load_package()
    synth = importlib.import_module("bigbwt_amd.synth")
    dev = torch.device("cuda", 0)
    wl = synth.WORKLOADS[args.workload]
    text = synth.workload_text_torch(dev, args.workload)
    torch.cuda.empty_cache()          # the generator's temporaries go back to the driver: the chain needs the room
    n = text.numel()
    bwt = torch.empty(n + 17, dtype=torch.uint8, device=dev)
    sa = torch.empty(n + 1, dtype=torch.int64, device=dev) if args.flags else None
    torch.cuda.synchronize()
    ctx = pkg.Context(0)
    ctx.set_profiling(True)
    ctx.set_kernel_trace(True)
    t0 = time.perf_counter()
    try:
        used = ctx.bigbwt_dev(text.data_ptr(), n, bwt.data_ptr(), sa.data_ptr() if args.flags else None, wl["w"], wl["p"], args.flags)
    except pkg.PfpError as ex:
        print(json.dumps(dict(error=str(ex), n=n, stats=ctx.stats(), mem=ctx.mem_stats())), flush=True)
        raise
    torch.cuda.synchronize()
    secs = time.perf_counter() - t0
    st = ctx.stats()
    mem = ctx.mem_stats()
    kt = sorted(ctx.kernel_trace(), key=lambda r: -r['total_ms'])[:14]
    ctx.set_kernel_trace(False)
    out = dict(workload=wl["desc"], n=n, flags=args.flags, seconds=round(secs, 3), MBps=round(n / secs / 1e6, 1),
               index_bits=st["index_bits"], dict_size=st["dict_size"], dict_over_4GiB=st["dict_size"] > (1 << 32),
               phrases=st["n_phrases"], words=st["n_words"], sa_rounds_dict=st["sa_rounds_dict"],
               phases_ms={k: round(st[k], 1) for k in st if k.startswith("ms_")}, peak_device_bytes=mem["peak"],
               top_kernels_ms={r['name']: [round(r['total_ms'], 1), r['launches']] for r in kt})
    assert used == n
    checks = {}
    # permutation of text + EOS
    def hist(t):
        h = torch.zeros(256, dtype=torch.int64, device=dev)
        for s in range(0, t.numel(), 1 << 28):
            h += torch.bincount(t[s:s + (1 << 28)].to(torch.int64), minlength=256)
        return h
    ht = hist(text); ht[0] += 1
    checks["bwt_is_permutation_of_text_plus_eos"] = bool(torch.equal(ht, hist(bwt[: n + 1])))
    if args.flags & 1:
        checks["sa0_is_n"] = int(sa[0]) == n
        seen = torch.zeros(n + 1, dtype=torch.bool, device=dev)
        ok_range = True
        for s in range(0, n + 1, 1 << 28):
            v = sa[s:s + (1 << 28)]
            ok_range &= bool((v >= 0).all()) and bool((v <= n).all())
            seen[v] = True
        checks["sa_is_permutation"] = ok_range and bool(seen.all())
        del seen
        ok = True
        for s in range(0, n + 1, 1 << 28):
            v = sa[s:s + (1 << 28)]
            prev = torch.where(v > 0, text[torch.clamp(v - 1, min=0)], torch.zeros((), dtype=torch.uint8, device=dev))
            ok &= bool(torch.equal(prev, bwt[s:s + v.numel()]))
        checks["bwt_i_is_text_sa_i_minus_1"] = ok
        # sampled order of adjacent suffixes
        rng = np.random.default_rng(1)
        idx = torch.from_numpy(rng.integers(1, n, size=args.samples)).to(dev)
        a = sa[idx].cpu().numpy(); b = sa[idx + 1].cpu().numpy()
        bad = 0
        L = 4096
        for x, y in zip(a, b):
            k = 0
            while True:
                sx = text[x + k: x + k + L].cpu().numpy(); sy = text[y + k: y + k + L].cpu().numpy()
                m = min(len(sx), len(sy))
                d = np.nonzero(sx[:m] != sy[:m])[0]
                if len(d):
                    bad += int(sx[d[0]] > sy[d[0]]); break
                if m < L:          # one suffix ended: the shorter one must come first
                    bad += int(len(sx) > len(sy)); break
                k += L
        checks["sampled_adjacent_suffixes_in_order"] = bad == 0
    # the outputs are functions of the text alone: a second run with another window / modulus -
    # another parse, another dictionary, another suffix array - must give the same BWT byte for byte
    ctx.close()
    ctx = pkg.Context(0)
    bwt2 = torch.empty(n + 17, dtype=torch.uint8, device=dev)
    used2 = ctx.bigbwt_dev(text.data_ptr(), n, bwt2.data_ptr(), None, 12, 200, 0)
    torch.cuda.synchronize()
    st2 = ctx.stats()
    out["second_parse"] = dict(w=12, p=200, dict_size=st2["dict_size"], index_bits=st2["index_bits"], phrases=st2["n_phrases"])
    # the same call again on the now warm pool: the first call above pays the driver for > 200 GB of fresh allocations
    t1 = time.perf_counter()
    ctx.bigbwt_dev(text.data_ptr(), n, bwt2.data_ptr(), None, 12, 200, 0)
    torch.cuda.synchronize()
    warm = time.perf_counter() - t1
    out["warm_pool"] = dict(seconds=round(warm, 3), MBps=round(n / warm / 1e6, 1), note="second call of the -w 12 -p 200 parse on the same context")
    checks["same_bwt_from_a_different_parse"] = used2 == n and bool(torch.equal(bwt[: n + 1], bwt2[: n + 1]))
    out["checks"] = checks
    out["all_ok"] = all(checks.values()) and out["index_bits"] == 64
    ctx.close()
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
