"""The reference's stage executables re-done over libpfpgpu.so (big-bwt_amd/bin/*, source
host/stages.c): same names, flags and files as newscan[NT].x / bwtparse / pfbwt[NT].x /
simplebwt / unparse.  Every file is compared with what the real
reference wrote (tests/golden/*.json); where oracle/_ref is present the stages are also mixed
with the reference's own executables, one at a time."""
import hashlib
import json
import os
import shutil
import subprocess

import numpy as np
import pytest

from textgen import make_text

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "big-bwt_amd", "bin")
REF = os.path.join(ROOT, "oracle", "_ref")
with open(os.path.join(ROOT, "tests", "golden", "golden_fasta.json")) as fh:
    DICZ = json.load(fh)["dicz"]


def sha_file(path):
    with open(path, "rb") as fh:
        return hashlib.sha256(fh.read()).hexdigest()


def run(cmd, ok=True):
    out = subprocess.run(cmd, capture_output=True, text=True)
    if ok:
        assert out.returncode == 0, " ".join(cmd) + "\n" + out.stdout + out.stderr
    return out


def test_stage_tools_built():
    names = "newscanNT.x newscan.x pscan.x bwtparse bwtparse64 pfbwtNT.x pfbwt.x pfbwtNT64.x pfbwt64.x simplebwt simplebwt64 unparse".split()
    for n in names:
        assert os.access(os.path.join(BIN, n), os.X_OK), n


def test_argument_checks_match_reference(tmp_path):
    f = tmp_path / "x"
    f.write_bytes(b"ACGT" * 100)
    assert "Windows size must be at least 4" in run([os.path.join(BIN, "newscanNT.x"), str(f), "-w", "3"], ok=False).stdout
    assert "Modulus must be at least 10" in run([os.path.join(BIN, "newscanNT.x"), str(f), "-p", "9"], ok=False).stdout
    assert "NT version cannot use threads" in run([os.path.join(BIN, "newscanNT.x"), str(f), "-t", "2"], ok=False).stdout
    assert "not both" in run([os.path.join(BIN, "pfbwtNT.x"), str(f), "-S", "-s"], ok=False).stdout
    assert run([os.path.join(BIN, "pfpstage")], ok=False).returncode == 2


def test_unparse_rebuilds_text_on_cpu(tmp_path, O):
    """unparse needs no GPU: .dicz/.parse made from the oracle's parse"""
    text = O.gen_fasta(5000, 2, 0.01, 9)
    pr = O.parse(text, 10, 100)
    words = pr["dict"].tobytes()[:-1].split(b"\x01")[:-1]
    z = b"".join((wd[1:] if wd[:1] == b"\x02" else wd)[: len(wd) - 10 - (1 if wd[:1] == b"\x02" else 0)] + b"\x01" for wd in words) + b"\x00"
    base = tmp_path / "t"
    (tmp_path / "t.dicz").write_bytes(z)
    (tmp_path / "t.parse").write_bytes(pr["parse"].astype("<u4").tobytes())
    run([os.path.join(BIN, "unparse"), str(base)])
    assert (tmp_path / "t.out").read_bytes() == text.tobytes()
    run([os.path.join(BIN, "unparse"), "-o", str(tmp_path / "again"), str(base)])
    assert (tmp_path / "again").read_bytes() == text.tobytes()
    (tmp_path / "t.parse").write_bytes(np.array([len(words) + 1], dtype="<u4").tobytes())
    assert run([os.path.join(BIN, "unparse"), str(base)], ok=False).returncode != 0


STAGE_FILES = ["dict", "occ", "parse", "last", "sai", "ilist", "bwlast", "bwsai"]


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["gen_small", "gen_w17", "n_run", "kat_q1"])
def test_three_stage_chain_matches_reference_files(golden, O, tmp_path, name):
    c = {x["name"]: x for x in golden}[name]
    f = tmp_path / "t"
    f.write_bytes(make_text(c["spec"], O).tobytes())
    w, p = str(c["w"]), str(c["p"])
    run([os.path.join(BIN, "newscanNT.x"), str(f), "-w", w, "-p", p, "-s"])
    run([os.path.join(BIN, "bwtparse"), str(f), "-s"])
    run([os.path.join(BIN, "pfbwtNT.x"), "-w", w, str(f), "-s", "-e"])
    g = c["runs"]["6"]
    for ext in STAGE_FILES + ["bwt", "ssa", "esa"]:
        assert sha_file(str(f) + "." + ext) == g[ext + "_sha256"], ext
    run([os.path.join(BIN, "pfbwt.x"), "-w", w, str(f), "-S", "-t", "4"])
    assert sha_file(str(f) + ".sa") == c["runs"]["1"]["sa_sha256"]
    run([os.path.join(BIN, "simplebwt"), str(f)])
    assert sha_file(str(f) + ".Bwt") == c["runs"]["0"]["Bwt_sha256"]


@pytest.mark.gpu
@pytest.mark.parametrize("d", DICZ, ids=[d["name"] for d in DICZ])
def test_compressed_dictionary_and_unparse(d, O, tmp_path):
    f = tmp_path / "t"
    f.write_bytes(make_text(d["spec"], O).tobytes() if d["spec"] else bytes.fromhex(d["raw_hex"]))
    run([os.path.join(BIN, "newscanNT.x"), str(f), "-w", str(d["w"]), "-p", str(d["p"]), "-c"] + (["-f"] if d["fasta"] else []))
    assert not os.path.exists(str(f) + ".dict")
    assert sha_file(str(f) + ".dicz") == d["dicz_sha256"]
    assert sha_file(str(f) + ".parse") == d["parse_sha256"]
    run([os.path.join(BIN, "unparse"), str(f)])
    assert sha_file(str(f) + ".out") == d["unparse_sha256"]


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.exists(os.path.join(REF, "bwtparse")), reason="oracle/_ref not built")
@pytest.mark.parametrize("mine", ["scan", "bwtparse", "pfbwt"])
def test_any_one_stage_swaps_into_the_reference_chain(golden, O, tmp_path, mine):
    """two stages from the reference, one from here, threaded file layout (-t 2: segmented .last/.sai)"""
    c = {x["name"]: x for x in golden}["gen_small"]
    f = tmp_path / "t"
    f.write_bytes(make_text(c["spec"], O).tobytes())
    scan = [os.path.join(BIN if mine == "scan" else REF, "pscan.x"), str(f), "-w", "10", "-p", "100", "-s", "-t", "2"]
    bwp = [os.path.join(BIN if mine == "bwtparse" else REF, "bwtparse"), str(f), "-s", "-t", "2"]
    pfb = [os.path.join(BIN if mine == "pfbwt" else REF, "pfbwtNT.x"), "-w", "10", str(f), "-S"]
    for cmd in (scan, bwp, pfb):
        run(cmd)
    assert os.path.exists(str(f) + ".1.last") and not os.path.exists(str(f) + ".last")
    assert sha_file(str(f) + ".bwt") == c["runs"]["1"]["bwt_sha256"]
    assert sha_file(str(f) + ".sa") == c["runs"]["1"]["sa_sha256"]


@pytest.mark.gpu
def test_driver_parsing_and_compress_modes(O, tmp_path):
    """bigbwt --parsing leaves .dicz + .parse; --compress packs them into .parse.txz (bigbwt:81-105)"""
    d = [x for x in DICZ if x["name"] == "gen_small"][0]
    f = tmp_path / "t"
    f.write_bytes(make_text(d["spec"], O).tobytes())
    exe = os.path.join(ROOT, "big-bwt_amd", "bigbwt")
    out = run([exe, "--parsing", str(f)])
    assert "Stopping after the parsing phase" in out.stdout
    assert sha_file(str(f) + ".dicz") == d["dicz_sha256"] and sha_file(str(f) + ".parse") == d["parse_sha256"]
    for ext in ("dict", "occ", "last", "bwt"):
        assert not os.path.exists(str(f) + "." + ext), ext
    os.remove(str(f) + ".dicz")
    os.remove(str(f) + ".parse")
    if shutil.which("tar") and shutil.which("xz"):
        out = run([exe, "--compress", str(f)])
        assert "xz-compressed as requested" in out.stdout
        assert not os.path.exists(str(f) + ".parse") and sha_file(str(f) + ".dicz") == d["dicz_sha256"]
        os.remove(str(f) + ".dicz")
        run(["tar", "-xJf", str(f) + ".parse.txz", "-C", "/"])
        assert sha_file(str(f) + ".dicz") == d["dicz_sha256"] and sha_file(str(f) + ".parse") == d["parse_sha256"]
        run([os.path.join(BIN, "unparse"), str(f)])
        assert sha_file(str(f) + ".out") == d["unparse_sha256"]
