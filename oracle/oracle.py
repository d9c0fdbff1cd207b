"""ctypes binding of oracle/liboracle.so + runner for the real reference binaries (oracle/_ref).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg; never by the product package.
"""
import ctypes as C
import os
import shutil
import subprocess
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "liboracle.so")
REFDIR = os.path.join(HERE, "_ref")

FLAG_SA, FLAG_SSA, FLAG_ESA = 1, 2, 4


class _ParseT(C.Structure):
    _fields_ = [("n_used", C.c_uint64), ("dict", C.POINTER(C.c_uint8)), ("dsize", C.c_uint64),
                ("occ", C.POINTER(C.c_uint32)), ("d", C.c_uint32),
                ("parse", C.POINTER(C.c_uint32)), ("P", C.c_uint64),
                ("last", C.POINTER(C.c_uint8)), ("sai", C.POINTER(C.c_uint64)),
                ("phash", C.POINTER(C.c_uint64))]


class _BwtT(C.Structure):
    _fields_ = [("bwt", C.POINTER(C.c_uint8)), ("nbwt", C.c_uint64),
                ("sa", C.POINTER(C.c_uint64)), ("nsa", C.c_uint64),
                ("ssa", C.POINTER(C.c_uint64)), ("nssa", C.c_uint64),
                ("esa", C.POINTER(C.c_uint64)), ("nesa", C.c_uint64),
                ("full_words", C.c_uint64), ("easy", C.c_uint64), ("hard", C.c_uint64)]


_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", HERE, "all"])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            build()
        _lib = C.CDLL(LIB)
        _lib.orc_kr_window.restype = C.c_uint64
        _lib.orc_kr_hash.restype = C.c_uint64
        _lib.orc_gen_fasta.restype = C.c_uint64
    return _lib


def _u8(a):
    a = np.ascontiguousarray(np.frombuffer(a, dtype=np.uint8) if isinstance(a, (bytes, bytearray)) else a, dtype=np.uint8)
    return a, a.ctypes.data_as(C.POINTER(C.c_uint8))


def _take(ptr, n, dtype):
    if n == 0 or not ptr:
        return np.zeros(0, dtype=dtype)
    return np.ctypeslib.as_array(ptr, shape=(int(n),)).astype(dtype, copy=True)


def kr_window(win: bytes) -> int:
    a, p = _u8(win)
    return int(lib().orc_kr_window(p, C.c_int(len(a))))


def kr_hash(s: bytes) -> int:
    a, p = _u8(s)
    return int(lib().orc_kr_hash(p, C.c_uint64(len(a))))


def scan(text, w, p):
    a, ptr = _u8(text)
    ends = C.POINTER(C.c_uint64)()
    n = C.c_uint64()
    rc = lib().orc_scan(ptr, C.c_uint64(len(a)), C.c_int(w), C.c_uint64(p), C.byref(ends), C.byref(n))
    if rc:
        raise RuntimeError(f"orc_scan rc={rc}")
    out = _take(ends, n.value, np.uint64)
    lib().orc_free(ends)
    return out


def parse(text, w, p):
    """Stage 1 -> dict of numpy arrays in the reference's file formats (sai as u64 values)."""
    a, ptr = _u8(text)
    o = _ParseT()
    rc = lib().orc_parse(ptr, C.c_uint64(len(a)), C.c_int(w), C.c_uint64(p), C.byref(o))
    if rc:
        raise RuntimeError(f"orc_parse rc={rc}")
    res = dict(n_used=int(o.n_used), dict=_take(o.dict, o.dsize, np.uint8), occ=_take(o.occ, o.d, np.uint32),
               parse=_take(o.parse, o.P, np.uint32), last=_take(o.last, o.P, np.uint8),
               sai=_take(o.sai, o.P, np.uint64), phash=_take(o.phash, o.P, np.uint64))
    lib().orc_parse_free(C.byref(o))
    return res


def sacak(s):
    a, ptr = _u8(s)
    sa = np.zeros(len(a), dtype=np.uint32)
    rc = lib().orc_sacak(ptr, sa.ctypes.data_as(C.POINTER(C.c_uint32)), C.c_uint64(len(a)))
    if rc < 0:
        raise RuntimeError("orc_sacak")
    return sa


def sacak_int(s, k=0):
    s = np.ascontiguousarray(s, dtype=np.uint32)
    sa = np.zeros(len(s), dtype=np.uint32)
    rc = lib().orc_sacak_int(s.ctypes.data_as(C.POINTER(C.c_uint32)), sa.ctypes.data_as(C.POINTER(C.c_uint32)),
                             C.c_uint64(len(s)), C.c_uint64(k))
    if rc < 0:
        raise RuntimeError("orc_sacak_int")
    return sa


def gsacak(s, want_lcp=True):
    a, ptr = _u8(s)
    sa = np.zeros(len(a), dtype=np.uint32)
    lcp = np.zeros(len(a), dtype=np.int32) if want_lcp else None
    rc = lib().orc_gsacak(ptr, sa.ctypes.data_as(C.POINTER(C.c_uint32)),
                          lcp.ctypes.data_as(C.POINTER(C.c_int32)) if want_lcp else None, C.c_uint64(len(a)))
    if rc < 0:
        raise RuntimeError("orc_gsacak")
    return sa, lcp


def bwtparse(parse_, last, sai, occ):
    parse_ = np.ascontiguousarray(parse_, dtype=np.uint32)
    last = np.ascontiguousarray(last, dtype=np.uint8)
    occ = np.ascontiguousarray(occ, dtype=np.uint32)
    P = len(parse_)
    ilist = np.zeros(P + 1, dtype=np.uint32)
    bwlast = np.zeros(P + 1, dtype=np.uint8)
    bwsai = np.zeros(P + 1, dtype=np.uint64)
    sai_p = None
    if sai is not None:
        sai = np.ascontiguousarray(sai, dtype=np.uint64)
        sai_p = sai.ctypes.data_as(C.POINTER(C.c_uint64))
    rc = lib().orc_bwtparse(parse_.ctypes.data_as(C.POINTER(C.c_uint32)), C.c_uint64(P),
                            last.ctypes.data_as(C.POINTER(C.c_uint8)), sai_p,
                            occ.ctypes.data_as(C.POINTER(C.c_uint32)), C.c_uint32(len(occ)),
                            ilist.ctypes.data_as(C.POINTER(C.c_uint32)),
                            bwlast.ctypes.data_as(C.POINTER(C.c_uint8)),
                            bwsai.ctypes.data_as(C.POINTER(C.c_uint64)))
    if rc:
        raise RuntimeError(f"orc_bwtparse rc={rc}")
    return ilist, bwlast, (bwsai if sai is not None else None)


def _bwt_result(o):
    res = dict(bwt=_take(o.bwt, o.nbwt, np.uint8), sa=_take(o.sa, o.nsa, np.uint64),
               ssa=_take(o.ssa, 2 * o.nssa, np.uint64).reshape(-1, 2),
               esa=_take(o.esa, 2 * o.nesa, np.uint64).reshape(-1, 2),
               full_words=int(o.full_words), easy=int(o.easy), hard=int(o.hard))
    lib().orc_bwt_free(C.byref(o))
    return res


def pfbwt(dict_, occ, ilist, bwlast, bwsai, w, flags=0):
    d_, dp = _u8(dict_)
    occ = np.ascontiguousarray(occ, dtype=np.uint32)
    ilist = np.ascontiguousarray(ilist, dtype=np.uint32)
    bwlast = np.ascontiguousarray(bwlast, dtype=np.uint8)
    bp = None
    if bwsai is not None:
        bwsai = np.ascontiguousarray(bwsai, dtype=np.uint64)
        bp = bwsai.ctypes.data_as(C.POINTER(C.c_uint64))
    o = _BwtT()
    rc = lib().orc_pfbwt(dp, C.c_uint64(len(d_)), occ.ctypes.data_as(C.POINTER(C.c_uint32)), C.c_uint32(len(occ)),
                         ilist.ctypes.data_as(C.POINTER(C.c_uint32)), bwlast.ctypes.data_as(C.POINTER(C.c_uint8)),
                         bp, C.c_uint64(len(ilist)), C.c_int(w), C.c_int(flags), C.byref(o))
    if rc:
        raise RuntimeError(f"orc_pfbwt rc={rc}")
    return _bwt_result(o)


def bigbwt(text, w=10, p=100, flags=0):
    a, ptr = _u8(text)
    o = _BwtT()
    rc = lib().orc_bigbwt(ptr, C.c_uint64(len(a)), C.c_int(w), C.c_uint64(p), C.c_int(flags), C.byref(o))
    if rc:
        raise RuntimeError(f"orc_bigbwt rc={rc}")
    return _bwt_result(o)


def simplebwt(text):
    a, ptr = _u8(text)
    out = np.zeros(len(a) + 1, dtype=np.uint8)
    rc = lib().orc_simplebwt(ptr, C.c_uint64(len(a)), out.ctypes.data_as(C.POINTER(C.c_uint8)))
    if rc:
        raise RuntimeError("orc_simplebwt")
    return out


def pack5(v):
    v = np.ascontiguousarray(v, dtype=np.uint64).reshape(-1)
    return v.view(np.uint8).reshape(-1, 8)[:, :5].copy().reshape(-1)


def unpack5(b):
    b = np.frombuffer(bytes(b), dtype=np.uint8).reshape(-1, 5)
    out = np.zeros((len(b), 8), dtype=np.uint8)
    out[:, :5] = b
    return out.view(np.uint64).reshape(-1)


# ---------------------------------------------------------------------------------------
# the real reference (oracle/_ref), run exactly as bigbwt:69-156 chains the three stages
# ---------------------------------------------------------------------------------------

def have_ref():
    return os.path.exists(os.path.join(REFDIR, "newscanNT.x"))


def run_ref(text: bytes, w=10, p=100, flags=0, threads=0, keep_dir=None, want_intermediates=True, check=False):
    """Run newscanNT.x|pscan.x -> bwtparse -> pfbwtNT.x|pfbwt.x on `text` in a temp dir.

    threads>0 uses pscan.x/pfbwt.x -t N (newscan.x -t N is broken for plain files);
    with -s/-e the last stage is single threaded as in bigbwt:132,141.
    Returns dict of raw file bytes keyed by extension, plus 'seconds' per stage.
    """
    import time
    if not have_ref():
        raise RuntimeError("oracle/_ref not built")
    tmp = keep_dir or tempfile.mkdtemp(prefix="pfpref_", dir="/dev/shm" if os.path.isdir("/dev/shm") and len(text) < (1 << 28) else None)
    os.makedirs(tmp, exist_ok=True)
    f = os.path.join(tmp, "t")
    with open(f, "wb") as fh:
        fh.write(bytes(text))
    sa_any = bool(flags)
    secs = {}
    dn = subprocess.DEVNULL
    try:
        t0 = time.time()
        if threads > 0:
            cmd = [os.path.join(REFDIR, "pscan.x"), f, "-w", str(w), "-p", str(p), "-t", str(threads)]
        else:
            cmd = [os.path.join(REFDIR, "newscanNT.x"), f, "-w", str(w), "-p", str(p)]
        if sa_any:
            cmd.append("-s")
        subprocess.check_call(cmd, stdout=dn, stderr=dn)
        secs["parse"] = time.time() - t0
        t0 = time.time()
        cmd = [os.path.join(REFDIR, "bwtparse"), f] + (["-s"] if sa_any else []) + (["-t", str(threads)] if threads > 0 else [])
        subprocess.check_call(cmd, stdout=dn, stderr=dn)
        secs["bwtparse"] = time.time() - t0
        t0 = time.time()
        sampled = bool(flags & (FLAG_SSA | FLAG_ESA))
        if threads > 0 and not sampled:
            cmd = [os.path.join(REFDIR, "pfbwt.x"), "-w", str(w), f, "-t", str(threads)]
        else:
            cmd = [os.path.join(REFDIR, "pfbwtNT.x"), "-w", str(w), f]
        if flags & FLAG_SSA:
            cmd.append("-s")
        if flags & FLAG_ESA:
            cmd.append("-e")
        if flags & FLAG_SA:
            cmd.append("-S")
        subprocess.check_call(cmd, stdout=dn, stderr=dn)
        secs["pfbwt"] = time.time() - t0
        if check:
            t0 = time.time()
            subprocess.check_call([os.path.join(REFDIR, "simplebwt"), f], stdout=dn, stderr=dn)
            secs["simplebwt"] = time.time() - t0
        out = {"seconds": secs}
        exts = ["bwt", "sa", "ssa", "esa", "Bwt"]
        if want_intermediates:
            exts += ["dict", "occ", "parse", "parse_old", "last", "sai", "ilist", "bwlast", "bwsai"]
        for ext in exts:
            fn = f + "." + ext
            if os.path.exists(fn):
                with open(fn, "rb") as fh:
                    out[ext] = fh.read()
        return out
    finally:
        if keep_dir is None:
            shutil.rmtree(tmp, ignore_errors=True)


# ---------------------------------------------------------------------------------------
# synthetic repetitive FASTA generator, SURVEY.md section 4 "GEN"
# ---------------------------------------------------------------------------------------

_M64 = (1 << 64) - 1


def fasta_text(raw):
    """the text `bigbwt -f` parses: sequences of a FASTA/FASTQ buffer, upper-cased (orc_fasta_text)"""
    raw = np.ascontiguousarray(np.frombuffer(bytes(raw), dtype=np.uint8))
    out = np.zeros(max(raw.size, 1), dtype=np.uint8)
    lib().orc_fasta_text.restype = C.c_uint64
    got = lib().orc_fasta_text(raw.ctypes.data_as(C.POINTER(C.c_uint8)), C.c_uint64(raw.size),
                               out.ctypes.data_as(C.POINTER(C.c_uint8)))
    return out[:got].copy()


def gen_fasta(G, C_, r, seed, n_blocks=()):
    """GEN(G,C,r,seed) of SURVEY.md section 4, scalar xorshift64 spec (run in C: orc_gen_fasta).
    n_blocks: optional [(start,len),...] ranges of the base genome overwritten by 'N' (config 2)."""
    nl = (G + 59) // 60
    hdr = sum(len(b">copy%d\n" % c) for c in range(C_))
    total = hdr + C_ * (G + nl)
    out = np.zeros(total, dtype=np.uint8)
    nb = np.ascontiguousarray(np.array(n_blocks, dtype=np.uint64).reshape(-1))
    got = lib().orc_gen_fasta(C.c_uint64(G), C.c_uint32(C_), C.c_double(r), C.c_uint64(seed),
                              nb.ctypes.data_as(C.POINTER(C.c_uint64)), C.c_uint32(len(nb) // 2),
                              out.ctypes.data_as(C.POINTER(C.c_uint8)), C.c_uint64(total))
    assert got == total, (got, total)
    return out


def gen_fasta_fast(G, C_, r, seed, n_blocks=()):
    """Same text family as gen_fasta but drawn with numpy's PCG64 (vectorised).  Deterministic
    in (G,C,r,seed); a different stream from the scalar spec."""
    rng = np.random.Generator(np.random.PCG64(seed))
    lut = np.frombuffer(b"ACGT", dtype=np.uint8)
    base = lut[rng.integers(0, 4, size=G, dtype=np.uint8)]
    for (st, ln) in n_blocks:
        base[st:st + ln] = ord("N")
    nl = (G + 59) // 60
    where = np.arange(G, dtype=np.int64)
    where += where // 60
    parts = []
    for c in range(C_):
        seq = base.copy()
        if r > 0:
            k = rng.binomial(G, r)
            pos = rng.integers(0, G, size=k)
            seq[pos] = lut[rng.integers(0, 4, size=k, dtype=np.uint8)]
        body = np.full(G + nl, ord("\n"), dtype=np.uint8)
        body[where] = seq
        parts.append(np.frombuffer(b">copy%d\n" % c, dtype=np.uint8))
        parts.append(body)
    return np.concatenate(parts)
