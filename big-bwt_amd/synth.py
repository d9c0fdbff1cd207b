"""Synthetic repetitive FASTA for benchmarks and full-size parity tests (host-side utility).

One definition, two evaluators that give the SAME bytes: numpy (CPU, used where the reference is run
to make golden digests) and torch (any device, used by bench.py to build the text directly in HBM).
The text family is SURVEY.md section 4's GEN - a base genome over ACGT, C copies with point mutations
at rate r, `>copy<c>` headers, 60-column lines - but every random draw is a pure function of
(seed, copy, position) through the splitmix64 finaliser, so it can be evaluated in parallel and on
any device (GEN's scalar xorshift stream cannot).

    base[i]        = "ACGT"[mix(seed*K0 + i) & 3]                  overwritten by 'N' inside nblocks
    copy c, pos i  : h = mix((seed*K1 + c + 1 + variant*4096)*K2 + i)
                     mutated iff (h >> 11) < r * 2^53 and base[i] != 'N';  then "ACGT"[(h >> 3) & 3]
"""
import numpy as np

K0, K1, K2 = 0x9E3779B97F4A7C15, 0xD1B54A32D192ED03, 0x8CB92BA72F3D8DD7
_M64 = (1 << 64) - 1
KR_PRIME = 1999999973            # newscan.cpp:172


def _s64(x):                     # python int -> two's complement int64 value
    x &= _M64
    return x - (1 << 64) if x >> 63 else x


def mix64_np(x):
    """splitmix64 finaliser on a uint64 array"""
    with np.errstate(over="ignore"):
        z = x + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def mix64_torch(x):
    """the same on an int64 tensor (wrapping multiplies, logical shifts spelled out)"""
    def lsr(v, k):
        return (v >> k) & ((1 << (64 - k)) - 1)
    z = x + _s64(0x9E3779B97F4A7C15)
    z = (z ^ lsr(z, 30)) * _s64(0xBF58476D1CE4E5B9)
    z = (z ^ lsr(z, 27)) * _s64(0x94D049BB133111EB)
    return z ^ lsr(z, 31)


def header(c):
    return b">copy%d\n" % c


# Repeat structure of a chromosome (round 4, the "c2r" workload): every choice a pure function of (seed, position), so both
# evaluators give the same bytes.  In order of precedence (after the N blocks):
#   satellite arrays   [(start, length), ...]: a 171-base monomer repeated, every base replaced by a random one with probability 1/64
#   microsatellites    in every 10 000-base window, at offset 5 000, 20-60 bases of a unit of 1-6 bases repeated exactly
#   interspersed       the first 300 bases of every 3 000-base window are a copy of one of `families` consensus sequences,
#                      every base replaced by a random one with probability 1/8 (~10 % of the genome, ~9 % divergence)
_KR = (0xA0761D6478BD642F, 0xE7037ED1A0B428DB, 0x8EBC6AF09C88C6E3, 0x589965CC75374CC3, 0x1D8E4E27C47D124F, 0xEB44ACCAB455D165,
       0x2D358DCCAA6C78A5, 0x8BB84B93962EACC9)


def _repeats_np(base, i0, seed, rep):
    """overwrite base (the positions i0 .. i0 + len(base)) with the repeat structure `rep` describes"""
    lut = np.frombuffer(b"ACGT", dtype=np.uint8)
    n = len(base)
    i = np.arange(i0, i0 + n, dtype=np.uint64)
    U = np.uint64

    def k(q):
        return U(((seed + 1) * _KR[q]) & _M64)

    def pick(h):
        return lut[(h & U(3)).astype(np.int64)]
    with np.errstate(over="ignore"):
        # interspersed repeat copies
        j, o = i // U(3000), i % U(3000)
        fam = mix64_np(k(0) + j) % U(rep["families"])
        cons = pick(mix64_np(k(1) + fam * U(300) + o))
        hd = mix64_np(k(2) + i)
        cons = np.where((hd & U(7)) == U(0), pick(hd >> U(3)), cons)
        base[:] = np.where(o < U(300), cons, base)
        # microsatellites
        j2, o2 = i // U(10000), i % U(10000)
        ln = U(20) + mix64_np(k(3) + j2) % U(41)
        ul = U(1) + mix64_np(k(4) + j2) % U(6)
        inside = (o2 >= U(5000)) & (o2 < U(5000) + ln)
        unit = pick(mix64_np(k(5) + j2 * U(8) + (o2 - U(5000)) % ul))
        base[:] = np.where(inside, unit, base)
        # satellite arrays
        for st, sl in rep["sat"]:
            lo, hi = max(st, i0), min(st + sl, i0 + n)
            if lo < hi:
                ii = i[lo - i0:hi - i0]
                mono = pick(mix64_np(k(6) + (ii - U(st)) % U(171)))
                hm = mix64_np(k(7) + ii)
                base[lo - i0:hi - i0] = np.where((hm & U(63)) == U(0), pick(hm >> U(6)), mono)


def _repeats_torch(base, i0, seed, rep):
    """the same on a uint8 device tensor (positions i0 .. i0 + base.numel())"""
    import torch
    dev = base.device
    lut = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)
    n = base.numel()
    i = torch.arange(i0, i0 + n, dtype=torch.int64, device=dev)

    def k(q):
        return _s64((seed + 1) * _KR[q])

    def umod(h, m):          # h is an unsigned 64-bit value held in an int64: h mod m for 0 < m < 2^31
        hi = (h >> 32) & 0xFFFFFFFF
        lo = h & 0xFFFFFFFF
        return ((hi % m) * ((1 << 32) % m) + lo % m) % m

    def lsr(v, s):
        return (v >> s) & ((1 << (64 - s)) - 1)

    def pick(h):
        return lut[h & 3]
    j, o = i // 3000, i % 3000
    fam = umod(mix64_torch(j + k(0)), rep["families"])
    cons = pick(mix64_torch(fam * 300 + o + k(1)))
    hd = mix64_torch(i + k(2))
    cons = torch.where((hd & 7) == 0, pick(lsr(hd, 3)), cons)
    base.copy_(torch.where(o < 300, cons, base))
    del cons, fam, hd
    j2, o2 = i // 10000, i % 10000
    ln = 20 + umod(mix64_torch(j2 + k(3)), 41)
    ul = 1 + umod(mix64_torch(j2 + k(4)), 6)
    inside = (o2 >= 5000) & (o2 < 5000 + ln)
    unit = pick(mix64_torch(j2 * 8 + (o2 - 5000).clamp(min=0) % ul + k(5)))
    base.copy_(torch.where(inside, unit, base))
    del unit, inside, ln, ul, j2, o2
    for st, sl in rep["sat"]:
        lo, hi = max(st, i0), min(st + sl, i0 + n)
        if lo < hi:
            ii = i[lo - i0:hi - i0]
            mono = pick(mix64_torch((ii - st) % 171 + k(6)))
            hm = mix64_torch(ii + k(7))
            base[lo - i0:hi - i0] = torch.where((hm & 63) == 0, pick(lsr(hm, 6)), mono)


def collection_np(G, C, r, seed, nblocks=(), variant=0, first_copy=0, header_base=None, repeats=None):
    """numpy evaluator -> uint8 array"""
    assert G % 60 == 0
    lut = np.frombuffer(b"ACGT", dtype=np.uint8)
    idx = np.arange(G, dtype=np.uint64)
    with np.errstate(over="ignore"):
        base = lut[(mix64_np(np.uint64((seed * K0) & _M64) + idx) & np.uint64(3)).astype(np.int64)]
    if repeats:
        for s0 in range(0, G, 1 << 26):
            _repeats_np(base[s0:s0 + (1 << 26)], s0, seed, repeats)
    for st, ln in nblocks:
        base[st:st + ln] = ord("N")
    thr = np.uint64(int(r * (1 << 53)))
    nl = np.full((G // 60, 1), ord("\n"), dtype=np.uint8)
    parts = []
    for c in range(first_copy, first_copy + C):
        seq = base
        if r > 0:
            with np.errstate(over="ignore"):
                k = np.uint64((((seed * K1 + c + 1 + variant * 4096) & _M64) * K2) & _M64)
                h = mix64_np(k + idx)
            mut = ((h >> np.uint64(11)) < thr) & (base != ord("N"))
            seq = np.where(mut, lut[((h >> np.uint64(3)) & np.uint64(3)).astype(np.int64)], base)
        parts.append(np.frombuffer(header(c + (variant * C if header_base is None else header_base)), dtype=np.uint8))
        parts.append(np.concatenate([seq.reshape(-1, 60), nl], axis=1).reshape(-1))
    return np.concatenate(parts)


def collection_torch(dev, G, C, r, seed, nblocks=(), variant=0, first_copy=0, header_base=None, repeats=None):
    """torch evaluator (any device) -> uint8 tensor with the same bytes as collection_np"""
    import torch
    assert G % 60 == 0
    lut = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)
    CH = 1 << 28          # positions per piece (the int64 temporaries of a piece stay at a few GB)
    base = torch.empty(G, dtype=torch.uint8, device=dev)
    for s0 in range(0, G, CH):
        i = torch.arange(s0, min(G, s0 + CH), dtype=torch.int64, device=dev)
        base[s0:s0 + i.numel()] = lut[mix64_torch(i + _s64(seed * K0)) & 3]
    del i
    if repeats:
        for s0 in range(0, G, 1 << 26):
            _repeats_torch(base[s0:s0 + (1 << 26)], s0, seed, repeats)
    for st, ln in nblocks:
        base[st:st + ln] = ord("N")
    thr = int(r * (1 << 53))
    nl = torch.full((G // 60, 1), ord("\n"), dtype=torch.uint8, device=dev)
    parts = []
    for c in range(first_copy, first_copy + C):
        seq = base
        if r > 0:
            k = _s64((((seed * K1 + c + 1 + variant * 4096) & _M64) * K2) & _M64)
            seq = torch.empty_like(base)
            for s0 in range(0, G, CH):
                i = torch.arange(s0, min(G, s0 + CH), dtype=torch.int64, device=dev)
                b = base[s0:s0 + i.numel()]
                h = mix64_torch(i + k)
                mut = (((h >> 11) & ((1 << 53) - 1)) < thr) & (b != ord("N"))
                seq[s0:s0 + i.numel()] = torch.where(mut, lut[(h >> 3) & 3], b)
            del h, mut, i, b
        parts.append(torch.tensor(list(header(c + (variant * C if header_base is None else header_base))), dtype=torch.uint8, device=dev))
        parts.append(torch.cat([seq.view(-1, 60), nl], dim=1).reshape(-1))
    out = torch.cat(parts).contiguous()
    return out


def kr_window_hash(win):
    """hash of one window as KR_window leaves it after w characters (newscan.cpp:168-202)"""
    h = 0
    for b in bytes(win):
        h = (h * 256 + b) % KR_PRIME
    return h


def first_window_triggers(first_bytes, w, p):
    """the reference writes 0x02 where the EOS belongs when the first window triggers"""
    return kr_window_hash(bytes(first_bytes[:w])) % p == 0


# BASELINE.json configs as synthetic workloads, shared by bench.py and the parity tests
WORKLOADS = {
    "c1": dict(G=12_100_020, C=1, r=0.0, nblocks=[], w=10, p=100, flags=0, seed=1,
               desc="BASELINE configs[0]: 1x yeast-shaped FASTA (~12.3 MB), -w 10 -p 100, BWT checked against the whole-text SACA-K BWT (-c)"),
    "c2": dict(G=249_000_000, C=1, r=0.0, nblocks=[(120_000_000, 18_000_000), (30_000_000, 10_000), (200_000_000, 10_000)],
               w=10, p=100, flags=0, seed=2,
               desc="BASELINE configs[1]: 1x human-chr1-shaped FASTA (~253 MB), -w 10 -p 100, BWT only"),
    "c2r": dict(G=249_000_000, C=1, r=0.0, nblocks=[(120_000_000, 18_000_000), (30_000_000, 10_000), (200_000_000, 10_000)],
                repeats=dict(families=64, sat=[(60_000_000, 3_000_000), (180_000_000, 1_500_000)]),
                w=10, p=100, flags=0, seed=2,
                desc="configs[1] with a chromosome's repeat structure: ~10 % interspersed repeats (64 families of 300 bases, ~9 % divergence), "
                     "3 Mb + 1.5 Mb satellite arrays of a 171-base monomer at 1.6 % divergence, a microsatellite every 10 kb, the same N blocks"),
    "c3": dict(G=12_100_020, C=64, r=1e-3, nblocks=[], w=10, p=100, flags=6, seed=3,
               desc="BASELINE configs[2]: 64x mutated yeast-shaped FASTA (~0.79 GB), -w 10 -p 100, BWT + -s -e sampled SA"),
    "big": dict(G=12_100_020, C=512, r=1e-3, nblocks=[], w=10, p=100, flags=0, seed=3,
                desc="512x mutated yeast-shaped FASTA (~6.3 GB > 2^32 bytes), -w 10 -p 100, BWT only (robustness / scaling probe)"),
    "huge": dict(G=12_100_020, C=1024, r=1e-3, nblocks=[], w=10, p=100, flags=0, seed=3,
                 desc="1024x mutated yeast-shaped FASTA (~12.6 GB; the north star's >= 10 GB repetitive input on one GPU), BWT only"),
    "huge_s": dict(G=12_100_020, C=1024, r=1e-3, nblocks=[], w=10, p=100, flags=2, seed=3,
                   desc="1024x mutated yeast-shaped FASTA (~12.6 GB; the north star's >= 10 GB repetitive input on one GPU), BWT + -s sampled SA"),
    "big_S": dict(G=12_100_020, C=512, r=1e-3, nblocks=[], w=10, p=100, flags=1, seed=3,
                  desc="BASELINE configs[3] flag set (-w 10 -p 100 -S, full SA in 5-byte integers) on 512 copies (~6.3 GB > 2^32 bytes: SA values above 4 G)"),
    "big_w12": dict(G=12_100_020, C=512, r=1e-3, nblocks=[], w=12, p=200, flags=2, seed=3,
                    desc="BASELINE configs[4] flag set (-w 12 -p 200 -s) on 512 copies (~6.3 GB; 1.5 GB dictionary when parsed as -p 200 says)"),
    "huge_w12": dict(G=12_100_020, C=1024, r=1e-3, nblocks=[], w=12, p=200, flags=2, seed=3,
                     desc="BASELINE configs[4] flag set (-w 12 -p 200 -s) on the 12.6 GB, 1024-copy text"),
    "wide": dict(G=4_260_000_000, C=1, r=0.0, nblocks=[], w=10, p=100, flags=0, seed=3,
                 desc="one 4.26 G-base random genome as FASTA (~4.33 GB, non-repetitive): dictionary > 4 GiB, exercises the 64-bit index build"),
    "wide31": dict(G=12_100_020, C=200, r=3e-2, nblocks=[], w=10, p=100, flags=0, seed=3,
                   desc="200x yeast-shaped FASTA at 3 % SNPs (~2.5 GB): dictionary between 2^31 and 2^32 bytes (32-bit indices without a spare bit)"),
    "c4s": dict(G=12_100_020, C=16, r=1e-3, nblocks=[], w=10, p=100, flags=1, seed=3,
                desc="BASELINE configs[3] parameters (-w 10 -p 100 -S, full SA) on a 16-copy, 0.2 GB stand-in (parity probe, not a reportable number)"),
    "c5s": dict(G=12_100_020, C=16, r=1e-3, nblocks=[], w=12, p=200, flags=2, seed=3,
                desc="BASELINE configs[4] parameters (-w 12 -p 200 -s) on a 16-copy, 0.2 GB stand-in (parity probe, not a reportable number)"),
    "small": dict(G=6_000_000, C=4, r=1e-3, nblocks=[(1_000_000, 300_000)], w=10, p=100, flags=0, seed=3,
                  desc="reduced smoke workload (not a reportable number)"),
}


def workload_seed(name, evaluator=collection_np, **kw):
    """first seed >= the workload's nominal one whose text does not start with a trigger window (Q1)"""
    wl = WORKLOADS[name]
    seed = wl["seed"]
    while True:
        # the first w bytes are ">copy0\\n" + the first bases: enough to evaluate 60 bases of copy 0
        v = kw.get("variant", 0)
        head = collection_np(60, 1, wl["r"], seed, [], v, header_base=v * wl["C"])[: wl["w"]]
        if not first_window_triggers(head, wl["w"], wl["p"]):
            return seed
        seed += 1


def workload_text_np(name, variant=0):
    wl = WORKLOADS[name]
    return collection_np(wl["G"], wl["C"], wl["r"], workload_seed(name, variant=variant), wl["nblocks"], variant, repeats=wl.get("repeats"))


def workload_text_torch(dev, name, variant=0):
    wl = WORKLOADS[name]
    return collection_torch(dev, wl["G"], wl["C"], wl["r"], workload_seed(name, variant=variant), wl["nblocks"], variant, repeats=wl.get("repeats"))
