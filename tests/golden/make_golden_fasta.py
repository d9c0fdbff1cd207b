#!/usr/bin/env python3
"""Generate tests/golden/golden_fasta.json: what the REAL reference (oracle/_ref/newscanNT.x -f ->
bwtparse -> pfbwtNT.x) makes of small FASTA/FASTQ inputs.  For every case the JSON holds the raw
input bytes (hex), the text the reference parsed (rebuilt from the .dict/.parse files it wrote:
phrases overlap by w, the first starts with 0x02 and the last ends with w 0x02 bytes) and its .bwt.

    python tests/golden/make_golden_fasta.py
"""
import gzip
import hashlib
import json
import os
import random
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.path.join(ROOT, "oracle", "_ref")


def dna(seed, n, alphabet="ACGT"):
    r = random.Random(seed)
    return "".join(r.choice(alphabet) for _ in range(n))


def wrap(s, width, eol="\n"):
    return eol.join(s[i:i + width] for i in range(0, len(s), width)) + eol


def cases():
    a, b, c = dna(1, 150), dna(2, 97), dna(3, 61, "acgtn")
    out = []
    out.append(("two_records", (">chr1 some comment\n" + wrap(a, 60) + ">chr2\n" + wrap(b, 60)).encode()))
    out.append(("lowercase_and_crlf", (">r1 x\r\n" + wrap(c, 20, "\r\n") + ">r2\r\n" + wrap(a[:50].lower(), 25, "\r\n")).encode()))
    out.append(("junk_blank_lines_no_final_newline",
                ("leading junk\n\n>s1\n\n" + wrap(a[:70], 30) + "\n\n>empty_record\n>s2\tc\n" + wrap(b, 40)).encode()[:-1]))
    out.append(("header_chars_inside_lines", (">id>with>gt @at\n" + a[:30] + ">" + a[30:50] + "@x+y\n" + b[:40] + "\n").encode()))
    q1 = "@" + "I" * (len(a[:80]) - 1)            # a quality line that starts with '@'
    q2 = ">" + "#" * (len(b[:60]) - 1)
    out.append(("fastq_four_line", ("@read1 d\n" + a[:80] + "\n+\n" + q1 + "\n@read2\n" + b[:60] + "\n+read2\n" + q2 + "\n").encode()))
    out.append(("fastq_multi_line", ("@m1\n" + wrap(a[:100], 40) + "+\n" + wrap("F" * 100, 40) + "@m2\n" + wrap(b[:50], 40) + "+\n" + wrap("F" * 50, 40)).encode()))
    out.append(("fastq_truncated_quality", ("@t1\n" + a[:60] + "\n+\n" + "I" * 60 + "\n@t2\n" + b[:40] + "\n+\n" + "I" * 20 + "\n").encode()))
    s = a[:90]
    out.append(("stops_at_special_byte", (">x\n" + s[:50] + "\n>y\n" + s[50:70] + "\x02" + s[70:] + "\n>z\n" + b + "\n").encode()))
    big = "".join(">copy%d\n" % k + wrap(dna(9, 3000)[: 3000 - k] , 60) for k in range(3))
    out.append(("gzip_three_copies", gzip.compress(big.encode(), mtime=0)))
    return out


def rebuild_text(d, parse, w):
    words = d[:-1].split(b"\x01")[:-1] if d.endswith(b"\x00") else d.split(b"\x01")
    ids = np.frombuffer(parse, dtype="<u4")
    t = bytearray()
    for k, i in enumerate(ids):
        ph = words[i - 1]
        t += ph if k == 0 else ph[w:]
    assert t[0] == 2 and t[-w:] == b"\x02" * w
    return bytes(t[1:-w])


def main():
    res = []
    for name, raw in cases():
        for (w, p) in ((4, 11), (10, 100)):
            tmp = tempfile.mkdtemp(prefix="pfpfa_", dir="/dev/shm")
            f = os.path.join(tmp, "t")
            open(f, "wb").write(raw)
            dn = subprocess.DEVNULL
            subprocess.check_call([os.path.join(REF, "newscanNT.x"), f, "-w", str(w), "-p", str(p), "-f"], stdout=dn, stderr=dn)
            d = open(f + ".dict", "rb").read()
            parse = open(f + ".parse", "rb").read()
            if subprocess.call([os.path.join(REF, "bwtparse"), f], stdout=dn, stderr=dn) != 0:
                print(name, w, p, "skipped: the reference's bwtparse aborts on a one-phrase parse")
                continue
            subprocess.check_call([os.path.join(REF, "pfbwtNT.x"), "-w", str(w), f], stdout=dn, stderr=dn)
            bwt = open(f + ".bwt", "rb").read()
            text = rebuild_text(d, parse, w)
            assert len(bwt) == len(text) + 1
            res.append(dict(name=name, w=w, p=p, raw_hex=raw.hex(), text_hex=text.hex(), bwt_hex=bwt.hex() if len(bwt) < 2000 else None,
                            bwt_sha256=hashlib.sha256(bwt).hexdigest(), dict_sha256=hashlib.sha256(d).hexdigest()))
            print(name, w, p, "raw", len(raw), "text", len(text))
            subprocess.call(["rm", "-rf", tmp])
    # reader-only vectors: byte soups of the characters that steer the reader; text as newscanNT.x -f saw it
    rng = np.random.default_rng(2024)
    alphabet = np.frombuffer(b">@+\n\n\r \tacgtACGTN", dtype=np.uint8)
    soups = []
    tmp = tempfile.mkdtemp(prefix="pfpfa_", dir="/dev/shm")
    f = os.path.join(tmp, "t")
    for _ in range(300):
        raw = alphabet[rng.integers(0, alphabet.size, size=int(rng.integers(0, 160)))].tobytes()
        open(f, "wb").write(raw)
        subprocess.check_call([os.path.join(REF, "newscanNT.x"), f, "-w", "4", "-p", "10", "-f"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        text = rebuild_text(open(f + ".dict", "rb").read(), open(f + ".parse", "rb").read(), 4)
        soups.append([raw.hex(), text.hex()])
    subprocess.call(["rm", "-rf", tmp])
    print("soups:", len(soups), "non-empty texts:", sum(1 for s in soups if s[1]))
    # newscanNT.x -c: the compressed dictionary (.dicz) and the parse unparse rebuilds the text from
    dicz = []
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import __graft_entry__ as entry
    from textgen import make_text
    O = entry.load_oracle()
    spec = dict(kind="gen", G=20000, C=3, r=0.002, seed=11)
    inputs = [("gen_small", make_text(spec, O).tobytes(), False, spec)] + [(n, r, True, None) for n, r in cases() if n in ("gzip_three_copies", "two_records")]
    for name, raw, fasta, spec in inputs:
        tmp = tempfile.mkdtemp(prefix="pfpfa_", dir="/dev/shm")
        f = os.path.join(tmp, "t")
        open(f, "wb").write(raw)
        subprocess.check_call([os.path.join(REF, "newscanNT.x"), f, "-w", "10", "-p", "100", "-c"] + (["-f"] if fasta else []),
                              stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        subprocess.check_call([os.path.join(REF, "unparse"), f], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        z = open(f + ".dicz", "rb").read()
        out = open(f + ".out", "rb").read()
        dicz.append(dict(name=name, fasta=fasta, spec=spec, raw_hex=None if spec else raw.hex(), w=10, p=100,
                         dicz_sha256=hashlib.sha256(z).hexdigest(), dicz_len=len(z),
                         parse_sha256=hashlib.sha256(open(f + ".parse", "rb").read()).hexdigest(),
                         unparse_sha256=hashlib.sha256(out).hexdigest(), unparse_len=len(out)))
        print("dicz", name, len(z), "unparse", len(out))
        subprocess.call(["rm", "-rf", tmp])
    json.dump(dict(cases=res, soups=soups, dicz=dicz), open(os.path.join(HERE, "golden_fasta.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
