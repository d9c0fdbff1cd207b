"""one workload through pfp_bigbwt_formats_dev, K times, nothing else: what tools/pmc_kernel.sh profiles.
    python tools/one_chain.py [workload] [K]"""
import sys, importlib, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as e
pkg = e.load_package()
import torch
synth = importlib.import_module("bigbwt_amd.synth")
name = sys.argv[1] if len(sys.argv) > 1 else "c3"
wl = synth.WORKLOADS[name]
dev = torch.device("cuda", 0)
t = synth.workload_text_torch(dev, name)
n = t.numel()
bwt = torch.empty(n + 64, dtype=torch.uint8, device=dev)
ctx = pkg.Context(0)
for _ in range(int(sys.argv[2]) if len(sys.argv) > 2 else 1):
    used, outs = ctx.bigbwt_formats_dev(t.data_ptr(), n, bwt.data_ptr(), wl["w"], wl["p"], wl["flags"])
    for ptr, _ in outs.values():
        ctx.dev_free(ptr)
torch.cuda.synchronize()
del ctx
