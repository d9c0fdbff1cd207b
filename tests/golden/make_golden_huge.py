#!/usr/bin/env python3
"""Reference digests for the 12.6 GB north-star workload (big-bwt_amd/synth.py "huge" / "huge_s": 1024 mutated copies,
-w 10 -p 100, BWT + -s).  Same idea as make_golden_full.py, written for a text that must not be held in memory three
times: the text is generated in pieces straight into a file, the real reference (oracle/_ref: newscanNT.x -> bwtparse
-> pfbwtNT.x -s, built from /root/reference by oracle/Makefile) runs on that file, and the outputs are hashed from
disk in chunks.  Only digests are committed (tests/golden/golden_full.json, entries "huge" and "huge_s" - one run gives
both: the .bwt does not depend on the SA flags).

    python tests/golden/make_golden_huge.py [workdir]        # build container, ~1 h of one core, ~30 GB of disk
"""
import hashlib
import importlib
import json
import os
import shutil
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "golden_full.json")
REFDIR = os.path.join(ROOT, "oracle", "_ref")


def sha_file(path, chunk=1 << 26):
    h = hashlib.sha256()
    with open(path, "rb") as fh:
        while True:
            b = fh.read(chunk)
            if not b:
                break
            h.update(b)
    return h.hexdigest(), os.path.getsize(path)


def main():
    work = sys.argv[1] if len(sys.argv) > 1 else "/tmp/pfp_golden_huge"
    os.makedirs(work, exist_ok=True)
    entry.load_package()
    synth = importlib.import_module("bigbwt_amd.synth")
    wl = synth.WORKLOADS["huge_s"]
    f = os.path.join(work, "t")
    t0 = time.time()
    text = synth.workload_text_np("huge_s")
    n = int(text.size)
    h = hashlib.sha256()
    with open(f, "wb") as fh:
        for s in range(0, n, 1 << 28):
            piece = text[s:s + (1 << 28)]
            h.update(piece)
            fh.write(piece)
    text_sha = h.hexdigest()
    del text
    print("text written", n, text_sha, f"{time.time() - t0:.0f}s", flush=True)
    secs = {}
    dn = subprocess.DEVNULL
    for name, cmd in (("parse", [os.path.join(REFDIR, "newscanNT.x"), f, "-w", str(wl["w"]), "-p", str(wl["p"]), "-s"]),
                      ("bwtparse", [os.path.join(REFDIR, "bwtparse"), f, "-s"]),
                      ("pfbwt", [os.path.join(REFDIR, "pfbwtNT.x"), "-w", str(wl["w"]), f, "-s"])):
        t1 = time.time()
        subprocess.check_call(cmd, stdout=dn, stderr=dn)
        secs[name] = round(time.time() - t1, 1)
        print(name, secs[name], "s", flush=True)
    bwt_sha, bwt_bytes = sha_file(f + ".bwt")
    ssa_sha, ssa_bytes = sha_file(f + ".ssa")
    res = json.load(open(OUT)) if os.path.exists(OUT) else {}
    base = dict(n=n, w=wl["w"], p=wl["p"], text_sha256=text_sha, ref_seconds=secs, bwt_sha256=bwt_sha, bwt_bytes=bwt_bytes)
    res["huge"] = dict(base, desc=synth.WORKLOADS["huge"]["desc"], flags=0)
    res["huge_s"] = dict(base, desc=wl["desc"], flags=wl["flags"], ssa_sha256=ssa_sha, ssa_bytes=ssa_bytes)
    with open(OUT, "w") as fh:
        json.dump(res, fh, indent=1, sort_keys=True)
    print(json.dumps(res["huge_s"]), flush=True)
    shutil.rmtree(work, ignore_errors=True)


if __name__ == "__main__":
    main()
