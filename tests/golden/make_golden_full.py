#!/usr/bin/env python3
"""Reference digests for the FULL-SIZE benchmark workloads (BASELINE.json configs[1], configs[2], the
16-copy stand-ins for configs[3]/[4] and SURVEY.md section 4's GEN(12e6,16,1e-3,42)).

Runs the real reference (oracle/_ref, built from /root/reference by oracle/Makefile) on the texts that
big-bwt_amd/synth.py / the oracle's GEN generator define, and stores sha256 digests of its outputs in
tests/golden/golden_full.json.  Only digests are committed: the texts are re-generated from their specs.

    python tests/golden/make_golden_full.py [name ...]        # in the build container (needs oracle/_ref)
"""
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "golden_full.json")


def sha(b):
    return hashlib.sha256(b).hexdigest()


def main():
    O = entry.load_oracle()
    entry.load_package()
    import importlib
    synth = importlib.import_module("bigbwt_amd.synth")
    cases = {
        # BASELINE configs[0]: the plumbing case with the -c check (simplebwt's .Bwt must equal the .bwt): the synth-family text
        # and BASELINE.md section 5's literal GEN(12.1e6,1,0,1)
        "c1": lambda: (synth.workload_text_np("c1"), dict(synth.WORKLOADS["c1"], check=True)),
        "c1_gen": lambda: (O.gen_fasta(12_100_000, 1, 0.0, 1), dict(w=10, p=100, flags=0, check=True, desc="BASELINE.md section 5 C1-syn: GEN(12.1e6,1,0,1), -w 10 -p 100 -c")),
        "c2": lambda: (synth.workload_text_np("c2"), synth.WORKLOADS["c2"]),
        "c2r": lambda: (synth.workload_text_np("c2r"), synth.WORKLOADS["c2r"]),      # configs[1] with repeat families, satellites, microsatellites
        "c3": lambda: (synth.workload_text_np("c3"), synth.WORKLOADS["c3"]),
        "c4s": lambda: (synth.workload_text_np("c4s"), synth.WORKLOADS["c4s"]),
        "c5s": lambda: (synth.workload_text_np("c5s"), synth.WORKLOADS["c5s"]),
        "small": lambda: (synth.workload_text_np("small"), synth.WORKLOADS["small"]),
        # scalar-xorshift GEN, digests quoted there are re-derived here
        "gen16": lambda: (O.gen_fasta(12_000_000, 16, 1e-3, 42), dict(w=10, p=100, flags=6, desc="SURVEY.md section 4 GEN(12e6,16,1e-3,42)")),
    }
    want = sys.argv[1:] or list(cases)
    res = json.load(open(OUT)) if os.path.exists(OUT) else {}
    for name in want:
        text, wl = cases[name]()
        t0 = time.time()
        tmp = os.path.join("/tmp", f"pfp_golden_full_{name}")
        r = O.run_ref(text.tobytes(), wl["w"], wl["p"], wl["flags"], threads=0, keep_dir=tmp, want_intermediates=False, check=bool(wl.get("check")))
        rec = dict(desc=wl["desc"], n=int(len(text)), w=wl["w"], p=wl["p"], flags=wl["flags"], text_sha256=sha(text.tobytes()),
                   ref_seconds={k: round(v, 1) for k, v in r["seconds"].items()})
        if wl.get("check"):          # bigbwt -c (bigbwt:177-194): `cmp file.bwt file.Bwt`
            rec["check_bwts_match"] = bool(r["Bwt"] == r["bwt"])
            assert rec["check_bwts_match"]
        for ext in ("bwt", "sa", "ssa", "esa"):
            if ext in r:
                rec[ext + "_sha256"] = sha(r[ext])
                rec[ext + "_bytes"] = len(r[ext])
        res[name] = rec
        import shutil
        shutil.rmtree(tmp, ignore_errors=True)
        print(name, rec, f"{time.time() - t0:.0f}s", flush=True)
        with open(OUT, "w") as fh:
            json.dump(res, fh, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
