import sys, os, importlib, hashlib, json, time
sys.path.insert(0, ".")
import __graft_entry__ as e
pkg = e.load_package()
import torch
synth = importlib.import_module("bigbwt_amd.synth")
pfpmod = importlib.import_module("bigbwt_amd.pfp")
g = json.load(open("tests/golden/golden_full.json"))["huge_w12"]
ranks = int(sys.argv[1]) if len(sys.argv) > 1 else 8
text = synth.workload_text_torch(torch.device("cuda", 0), "huge_w12").cpu().numpy()
torch.cuda.empty_cache()
os.environ["PFP_MULTI_LOOPBACK"] = "1"
base = "/dev/shm/pfp_w12_multi"
t0 = time.time()
try:
    st = pfpmod.bigbwt_files_multi(text, base, [0] * ranks, g["w"], g["p"], g["flags"])
    print("ok", st, time.time() - t0)
    for key, ext in (("bwt", ".bwt"), ("ssa", ".ssa")):
        h = hashlib.sha256()
        with open(base + ext, "rb") as fh:
            for blk in iter(lambda: fh.read(1 << 26), b""):
                h.update(blk)
        print(ext, h.hexdigest() == g[key + "_sha256"])
except Exception as ex:
    print("FAILED", type(ex).__name__, ex)
finally:
    for ext in (".bwt", ".ssa"):
        if os.path.exists(base + ext):
            os.unlink(base + ext)
