"""bigbwt -f: FASTA/FASTQ input.  Golden vectors: tests/golden/golden_fasta.json,
made by the real reference (newscanNT.x -f -> bwtparse -> pfbwtNT.x), see make_golden_fasta.py.
CPU: the oracle's reader and the driver's reader (big-bwt_amd/host/fasta.c, loaded alone as
libpfphost.so) against the text the reference parsed.  GPU: the C driver's `-f` end to end."""
import ctypes as C
import gzip
import hashlib
import json
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
with open(os.path.join(ROOT, "tests", "golden", "golden_fasta.json")) as fh:
    _G = json.load(fh)
CASES, SOUPS = _G["cases"], _G["soups"]
IDS = ["%s-w%d-p%d" % (c["name"], c["w"], c["p"]) for c in CASES]


def plain(raw):
    return gzip.decompress(raw) if raw[:2] == b"\x1f\x8b" else raw


def host_reader():
    lib = C.CDLL(os.path.join(ROOT, "big-bwt_amd", "libpfphost.so"))
    lib.pfp_fasta_text.restype = C.c_size_t
    lib.pfp_fasta_text.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p]
    lib.pfp_read_maybe_gz.restype = C.c_void_p
    lib.pfp_read_maybe_gz.argtypes = [C.c_char_p, C.POINTER(C.c_size_t)]
    return lib


def host_fasta_text(lib, raw):
    out = C.create_string_buffer(max(len(raw), 1))
    n = lib.pfp_fasta_text(raw, len(raw), out)
    return out.raw[:n]


def test_golden_covers_the_reader_edge_cases():
    names = {c["name"] for c in CASES}
    assert {"lowercase_and_crlf", "fastq_four_line", "fastq_truncated_quality", "stops_at_special_byte", "gzip_three_copies"} <= names
    trunc = [c for c in CASES if c["name"] == "fastq_truncated_quality"][0]
    assert len(bytes.fromhex(trunc["text_hex"])) == 60          # the record with the short quality is not delivered
    stop = [c for c in CASES if c["name"] == "stops_at_special_byte"][0]
    assert len(bytes.fromhex(stop["text_hex"])) == 70


@pytest.mark.parametrize("c", CASES, ids=IDS)
def test_oracle_reader_and_chain_match_reference(c, O):
    raw = plain(bytes.fromhex(c["raw_hex"]))
    text = O.fasta_text(raw)
    assert text.tobytes() == bytes.fromhex(c["text_hex"])
    got = O.bigbwt(text, c["w"], c["p"], 0)
    assert hashlib.sha256(got["bwt"].tobytes()).hexdigest() == c["bwt_sha256"]


@pytest.mark.parametrize("c", CASES, ids=IDS)
def test_driver_reader_matches_reference(c):
    lib = host_reader()
    assert host_fasta_text(lib, plain(bytes.fromhex(c["raw_hex"]))) == bytes.fromhex(c["text_hex"])


def test_both_readers_match_reference_on_byte_soups(O):
    """300 random strings over the characters that steer the reader, text as the reference read it"""
    lib = host_reader()
    assert len(SOUPS) == 300 and sum(1 for s in SOUPS if s[1]) > 100
    for raw_hex, text_hex in SOUPS:
        raw, want = bytes.fromhex(raw_hex), bytes.fromhex(text_hex)
        assert O.fasta_text(raw).tobytes() == want, raw
        assert host_fasta_text(lib, raw) == want, raw


def test_driver_reader_agrees_with_oracle_on_random_inputs(O):
    """both readers on byte soups made of the characters that steer the reader"""
    lib = host_reader()
    rng = np.random.default_rng(5)
    alphabet = np.frombuffer(b">@+\n\n\r \tacgtACGTN", dtype=np.uint8)
    for trial in range(400):
        raw = alphabet[rng.integers(0, alphabet.size, size=int(rng.integers(0, 120)))].tobytes()
        assert host_fasta_text(lib, raw) == O.fasta_text(raw).tobytes(), raw


def test_driver_reads_gzip_and_plain_files(tmp_path):
    lib = host_reader()
    data = bytes(range(256)) * 300
    for name, blob in (("plain", data), ("gz", gzip.compress(data))):
        f = tmp_path / name
        f.write_bytes(blob)
        n = C.c_size_t(0)
        p = lib.pfp_read_maybe_gz(str(f).encode(), C.byref(n))
        assert p and n.value == len(data)
        assert C.string_at(p, n.value) == data
    n = C.c_size_t(0)
    assert not lib.pfp_read_maybe_gz(str(tmp_path / "missing").encode(), C.byref(n))


@pytest.mark.gpu
@pytest.mark.parametrize("c", CASES, ids=IDS)
def test_c_driver_fasta_mode(c, tmp_path):
    f = tmp_path / "in.fa"
    f.write_bytes(bytes.fromhex(c["raw_hex"]))
    exe = os.path.join(ROOT, "big-bwt_amd", "bigbwt")
    out = subprocess.run([exe, "-f", "-w", str(c["w"]), "-p", str(c["p"]), "-c", str(f)], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "BWTs match" in out.stdout
    bwt = np.fromfile(str(f) + ".bwt", dtype=np.uint8).tobytes()
    assert hashlib.sha256(bwt).hexdigest() == c["bwt_sha256"]
    if c["bwt_hex"]:
        assert bwt.hex() == c["bwt_hex"]
