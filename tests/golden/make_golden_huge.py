#!/usr/bin/env python3
"""Reference digests for the 12.6 GB north-star workload (big-bwt_amd/synth.py "huge" / "huge_s": 1024 mutated copies,
-w 10 -p 100, BWT + -s).  Same idea as make_golden_full.py, written for a text that must not be held in memory three
times: the text is generated in pieces straight into a file, the real reference (oracle/_ref: newscanNT.x -> bwtparse
-> pfbwtNT.x -s, built from /root/reference by oracle/Makefile) runs on that file, and the outputs are hashed from
disk in chunks.  Only digests are committed (tests/golden/golden_full.json, entries "huge" and "huge_s" - one run gives
both: the .bwt does not depend on the SA flags).

    python tests/golden/make_golden_huge.py [workdir]        # build container, ~1 h of one core, ~30 GB of disk

Round 4: `make_golden_huge.py workdir NAME` runs the same recipe for any synth workload - "big_S" (6.3 GB, -S: a 31.5 GB .sa
whose 5-byte values pass 2^32, utils.c:112-129) and "huge_w12" (12.6 GB, -w 12 -p 200 -s) - and stores one entry NAME.
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
    name = sys.argv[2] if len(sys.argv) > 2 else "huge_s"
    wl = synth.WORKLOADS[name]
    f = os.path.join(work, "t")
    t0 = time.time()
    text = synth.workload_text_np(name)
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
    saflag = {1: "-S", 2: "-s", 4: "-e"}[wl["flags"]]          # one SA flag per workload here
    for stage, cmd in (("parse", [os.path.join(REFDIR, "newscanNT.x"), f, "-w", str(wl["w"]), "-p", str(wl["p"]), "-s"]),
                       ("bwtparse", [os.path.join(REFDIR, "bwtparse"), f, "-s"]),
                       ("pfbwt", [os.path.join(REFDIR, "pfbwtNT.x"), "-w", str(wl["w"]), f, saflag])):
        # (the 32-bit executables: bigbwt:113-151 switches on the parse's and the dictionary's sizes, both below 2^31 here)
        t1 = time.time()
        subprocess.check_call(cmd, stdout=dn, stderr=dn)
        secs[stage] = round(time.time() - t1, 1)
        print(stage, secs[stage], "s", flush=True)
    assert os.path.getsize(f + ".dict") < 2**31 - 4 and os.path.getsize(f + ".parse") // 4 < 2**31 - 1
    bwt_sha, bwt_bytes = sha_file(f + ".bwt")
    ext = {1: "sa", 2: "ssa", 4: "esa"}[wl["flags"]]
    x_sha, x_bytes = sha_file(f + "." + ext)
    base = dict(n=n, w=wl["w"], p=wl["p"], text_sha256=text_sha, ref_seconds=secs, bwt_sha256=bwt_sha, bwt_bytes=bwt_bytes)
    extra = {}
    if ext == "sa":          # the largest value and the count of values >= 2^32, read back from the 5-byte file in pieces
        import numpy as np
        big, top = 0, 0
        with open(f + ".sa", "rb") as fh:
            while True:
                b = np.frombuffer(fh.read(5 * (1 << 24)), dtype=np.uint8)
                if b.size == 0:
                    break
                hi = b[4::5]
                big += int(np.count_nonzero(hi))
                top = max(top, int(hi.max()))
        extra = dict(sa_values_over_4g=big, sa_top_byte_max=top)
    res = json.load(open(OUT)) if os.path.exists(OUT) else {}          # (re-read: another run may have added entries meanwhile)
    if name == "huge_s":
        res["huge"] = dict(base, desc=synth.WORKLOADS["huge"]["desc"], flags=0)
    res[name] = dict(base, desc=wl["desc"], flags=wl["flags"], **{ext + "_sha256": x_sha, ext + "_bytes": x_bytes}, **extra)
    with open(OUT, "w") as fh:
        json.dump(res, fh, indent=1, sort_keys=True)
    print(json.dumps(res[name]), flush=True)
    shutil.rmtree(work, ignore_errors=True)


if __name__ == "__main__":
    main()
