"""A union dictionary between 2^31 and 2^32 bytes in the multi-GPU chain (GPU box): R virtual ranks on one card, each with C copies
of a yeast-sized genome at 3 % SNPs (nearly every phrase distinct: the dictionary is as long as the text); the BWT of the R shards
against the single-GPU chain's BWT of the concatenation.

    python tools/check_range31.py [R] [C]     (defaults 2, 100: union dictionary ~3.3 GB)
"""
import hashlib
import importlib
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

pkg = entry.load_package()
D = importlib.import_module("bigbwt_amd.dist")
synth = importlib.import_module("bigbwt_amd.synth")
dev = torch.device("cuda", 0)
R = int(sys.argv[1]) if len(sys.argv) > 1 else 2
C = int(sys.argv[2]) if len(sys.argv) > 2 else 100
G, r, seed = 12_100_020, 3e-2, 3


def sha(t):
    h = hashlib.sha256()
    for s in range(0, t.numel(), 1 << 28):
        h.update(t[s:s + (1 << 28)].cpu().numpy().tobytes())
    return h.hexdigest()


texts = [synth.collection_torch(dev, G, C, r, seed, [], v) for v in range(R)]
ctxs = [pkg.Context(0) for _ in range(R)]
t0 = time.perf_counter()
res = D.simulate(ctxs, texts, 10, 100, 0, trim=True)
torch.cuda.synchronize()
t_multi = time.perf_counter() - t0
glob = res[0]["stats"]["glob"]
multi = torch.cat([x["bwt"] for x in res])
h_multi = sha(multi)
n = sum(t.numel() for t in texts)
del res, multi
for c in ctxs:
    c.close()
whole = torch.cat(texts)
del texts
torch.cuda.empty_cache()
ctx = pkg.Context(0)
bwt = torch.empty(n + 1 + 16, dtype=torch.uint8, device=dev)
t0 = time.perf_counter()
used = ctx.bigbwt_dev(whole.data_ptr(), n, bwt.data_ptr(), None, 10, 100, 0)
torch.cuda.synchronize()
t_one = time.perf_counter() - t0
h_one = sha(bwt[: n + 1])
print(json.dumps(dict(ranks=R, copies_per_rank=C, text_bytes=n, union_dict_bytes=glob["dict_bytes"], index_bits_multi=glob["index_bits"],
                      rounds=glob["rounds"], complete=glob["complete"],
                      s_multi=round(t_multi, 3), s_one_gpu=round(t_one, 3), index_bits_one_gpu=ctx.stats()["index_bits"],
                      bwt_equal=h_multi == h_one, used=used == n)))
sys.exit(0 if h_multi == h_one else 1)
