"""Deterministic test inputs, described by small JSON-able specs (stored in tests/golden/golden.json)."""
import hashlib

import numpy as np


def _stream(seed: str, n: int) -> np.ndarray:
    return np.frombuffer(hashlib.shake_256(seed.encode()).digest(n), dtype=np.uint8)


def make_text(spec, O=None) -> np.ndarray:
    kind = spec["kind"]
    if kind == "literal":
        return np.frombuffer(bytes.fromhex(spec["hex"]), dtype=np.uint8).copy()
    if kind == "gen":           # SURVEY.md section 4 GEN, scalar xorshift spec (oracle C generator)
        return O.gen_fasta(spec["G"], spec["C"], spec["r"], spec["seed"], [tuple(b) for b in spec.get("nblocks", [])])
    if kind == "rand":          # i.i.d. symbols from an alphabet, optional overwritten runs / single bytes
        alpha = np.frombuffer(bytes.fromhex(spec["alphabet_hex"]), dtype=np.uint8)
        t = alpha[_stream(spec["seed"], spec["n"]) % len(alpha)].copy()
        for pos, ln, byte in spec.get("runs", []):
            t[pos:pos + ln] = byte
        return t
    if kind == "repeat":        # a random unit repeated with point mutations
        alpha = np.frombuffer(bytes.fromhex(spec["alphabet_hex"]), dtype=np.uint8)
        unit = alpha[_stream(spec["seed"], spec["unit"]) % len(alpha)]
        t = np.tile(unit, spec["copies"]).copy()
        mut = _stream(spec["seed"] + "/mut", 8 * spec["mutations"]).view(np.uint64)
        for k in range(spec["mutations"]):
            t[int(mut[k] % len(t))] = alpha[int((mut[k] >> 40) % len(alpha))]
        return t
    raise ValueError(kind)
