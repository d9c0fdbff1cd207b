#!/usr/bin/env python3
"""Generate tests/golden/golden.json by running the REAL reference binaries (oracle/_ref, built
by oracle/Makefile from /root/reference) on deterministic inputs.  Run in the build container:

    python tests/golden/make_golden.py

The JSON holds only data: input specs, and for each (input, w, p, flags) the sha256 of every
file the reference wrote (raw hex too when the file is tiny).  Nothing of the reference's source
is stored.
"""
import hashlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as entry  # noqa: E402
from textgen import make_text  # noqa: E402

O = entry.load_oracle()
DNA = b"ACGT".hex()
DNAN = b"ACGTN\n".hex()

CASES = [
    dict(name="kat1", spec=dict(kind="literal", hex=b"CCGATTACAT!GATTACAT!GATTAGATA".hex()), w=4, p=11, raw=True),
    dict(name="kat_q1", spec=dict(kind="literal", hex=b"GATTACAT!GATTACAT!GATTAGATA".hex()), w=4, p=11, raw=True,
         note="SURVEY 2.2-Q1: first window triggers, reference emits 0x02 where the EOS belongs"),
    dict(name="tiny_dna_w4", spec=dict(kind="rand", seed="tiny1", n=300, alphabet_hex=DNA), w=4, p=11, raw=True),
    dict(name="tiny_bytes_w5", spec=dict(kind="rand", seed="tiny2", n=500, alphabet_hex=bytes(range(3, 256)).hex()), w=5, p=10, raw=True),
    dict(name="repeat_w6", spec=dict(kind="repeat", seed="rep1", unit=700, copies=12, mutations=25, alphabet_hex=DNA), w=6, p=20),
    dict(name="gen_small", spec=dict(kind="gen", G=20000, C=3, r=0.002, seed=11), w=10, p=100),
    dict(name="gen_p64", spec=dict(kind="gen", G=30000, C=2, r=0.001, seed=12), w=10, p=64),
    dict(name="gen_p101_w12", spec=dict(kind="gen", G=30000, C=2, r=0.001, seed=13), w=12, p=101),
    dict(name="gen_w17", spec=dict(kind="gen", G=30000, C=2, r=0.001, seed=14), w=17, p=50),
    dict(name="gen_w20", spec=dict(kind="gen", G=30000, C=2, r=0.001, seed=15), w=20, p=50,
         note="w > 17 takes the generic (non register) scan kernel"),
    dict(name="n_run", spec=dict(kind="rand", seed="nrun", n=60000, alphabet_hex=DNA, runs=[[5000, 30000, ord("N")]]), w=10, p=100,
         note="one phrase of ~30 kB (long-phrase hashing/copy path, deep prefix doubling)"),
    dict(name="gen_nblock", spec=dict(kind="gen", G=40000, C=2, r=0.001, seed=16, nblocks=[[3000, 12000], [30000, 500]]), w=10, p=100,
         note="N blocks with newlines every 60 columns (periodic text)"),
    dict(name="special_byte", spec=dict(kind="rand", seed="sp", n=5000, alphabet_hex=DNA, runs=[[3210, 1, 2]]), w=10, p=100,
         note="SURVEY 2.2-Q8: parsing stops at the first byte <= 2"),
    dict(name="high_bytes", spec=dict(kind="rand", seed="hb", n=20000, alphabet_hex="80fffe03c1"), w=8, p=30),
    dict(name="gen_1e6x4", spec=dict(kind="gen", G=1000000, C=4, r=0.001, seed=7), w=10, p=100,
         note="SURVEY.md section 4 golden: .bwt sha256 ebf1fb17..., .sa 99407362..., .ssa 00a20762..., .esa d581e202..."),
]

INTERMEDIATES = ["dict", "occ", "parse", "last", "sai", "ilist", "bwlast", "bwsai", "parse_old"]


def main():
    out = []
    for case in CASES:
        text = make_text(case["spec"], O)
        entry_ = dict(name=case["name"], spec=case["spec"], w=case["w"], p=case["p"], note=case.get("note", ""),
                      n=int(len(text)), text_sha256=hashlib.sha256(text.tobytes()).hexdigest(), runs={})
        for flags in (0, 1, 6):
            r = O.run_ref(text.tobytes(), case["w"], case["p"], flags, check=(flags == 0))
            rec = {}
            keys = ["bwt"] + (["sa"] if flags & 1 else []) + (["ssa", "esa"] if flags & 6 else [])
            if flags == 6:
                keys += INTERMEDIATES
            if flags == 0:
                keys += ["Bwt"]
            for k in keys:
                rec[k + "_sha256"] = hashlib.sha256(r[k]).hexdigest()
                rec[k + "_len"] = len(r[k])
                if case.get("raw"):
                    rec[k + "_hex"] = r[k].hex()
            entry_["runs"][str(flags)] = rec
        entry_["bwt_equals_simplebwt"] = entry_["runs"]["0"]["bwt_sha256"] == entry_["runs"]["0"]["Bwt_sha256"]
        out.append(entry_)
        print(case["name"], "n=%d" % len(text), entry_["runs"]["0"]["bwt_sha256"][:16], "match -c:", entry_["bwt_equals_simplebwt"])
    with open(os.path.join(HERE, "golden.json"), "w") as fh:
        json.dump(out, fh, indent=1)


if __name__ == "__main__":
    main()
