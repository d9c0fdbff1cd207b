"""The synthetic-workload generator (big-bwt_amd/synth.py): its numpy and torch evaluators give the same
bytes, and the committed full-size reference digests (tests/golden/golden_full.json) belong to the texts
the generator makes today."""
import hashlib

import numpy as np
import pytest


def test_numpy_and_torch_evaluators_agree(synth):
    import torch
    for (G, C, r, seed, nb, variant) in [(600, 1, 0.0, 2, [], 0), (6000, 3, 0.01, 5, [(100, 50)], 2), (1200, 5, 0.2, 9, [(0, 70)], 1)]:
        a = synth.collection_np(G, C, r, seed, nb, variant)
        b = synth.collection_torch("cpu", G, C, r, seed, nb, variant).numpy()
        assert np.array_equal(a, b)
        assert len(a) == sum(len(synth.header(c + variant * C)) for c in range(C)) + C * (G + G // 60)
    x = np.arange(1000, dtype=np.uint64) * np.uint64(0x123456789ABCDEF1)
    assert np.array_equal(synth.mix64_np(x).view(np.int64), synth.mix64_torch(torch.from_numpy(x.view(np.int64))).numpy())


def test_repeat_structure_evaluators_agree(synth):
    """the chromosome-like repeat structure of the c2r workload (interspersed families, microsatellites, satellite arrays): both
    evaluators give the same bytes, the features are where the definition puts them"""
    rep = dict(families=64, sat=[(600_000, 300_000), (1_800_000, 150_000)])
    a = synth.collection_np(2_400_000, 2, 0.001, 2, [(1_200_000, 180_000)], repeats=rep)
    b = synth.collection_torch("cpu", 2_400_000, 2, 0.001, 2, [(1_200_000, 180_000)], repeats=rep).numpy()
    assert np.array_equal(a, b)
    seq = a[len(synth.header(0)):][: 2_400_000 + 40_000]
    seq = seq[seq != 10][: 2_400_000]          # the bases of copy 0 without the line ends
    mono = seq[600_000:600_000 + 171 * 100].reshape(100, 171)
    assert (mono != mono[0]).mean() < 0.05          # a satellite array: copies of one monomer, a few per cent apart
    micro = seq[15_000:15_020]
    assert any(np.array_equal(micro[u:], micro[:-u]) for u in range(1, 7))          # a microsatellite: a unit of 1-6 bases repeated
    assert (seq[1_200_000:1_380_000] == ord("N")).all()
    fam = {}
    for j in range(0, 200):          # copies of the same family are ~9 % apart, different families ~75 %
        fam.setdefault(j, seq[3000 * j:3000 * j + 300])
    d = np.array([[(fam[x] != fam[y]).mean() for y in range(40)] for x in range(40)])
    assert ((d > 0.01) & (d < 0.3)).sum() >= 10 and (d > 0.6).sum() > 1000


def test_text_shape_and_mutation_rate(synth):
    t = synth.collection_np(60000, 4, 0.01, 3)
    lines = bytes(t).split(b"\n")
    assert lines[0] == b">copy0" and all(len(l) == 60 for l in lines[1:1001])
    base = synth.collection_np(60000, 4, 0.0, 3)
    rate = float((t != base).mean())
    assert 0.005 < rate < 0.009          # r * 3/4 of the draws change the base
    assert set(np.unique(t)) <= set(b">copy0123\nACGT")


def test_first_window_rule(synth):
    # workloads never start with a trigger window
    for name, wl in synth.WORKLOADS.items():
        seed = synth.workload_seed(name)
        head = synth.collection_np(60, 1, wl["r"], seed)[: wl["w"]]
        assert not synth.first_window_triggers(head, wl["w"], wl["p"])
    assert synth.kr_window_hash(b"GATT") == 1195463764
    assert synth.kr_window_hash(b"CCGA") % 11 == 6


def test_golden_full_texts_are_reproducible(synth, golden_full):
    g = golden_full["small"]
    t = synth.workload_text_np("small")
    assert len(t) == g["n"] and hashlib.sha256(t.tobytes()).hexdigest() == g["text_sha256"]


def test_oracle_matches_full_size_reference_digest(synth, golden_full, O):
    """the CPU oracle on the 24 MB `small` workload against the real reference's digest"""
    g = golden_full["small"]
    t = synth.workload_text_np("small")
    got = O.bigbwt(t, g["w"], g["p"], 0)
    assert hashlib.sha256(got["bwt"].tobytes()).hexdigest() == g["bwt_sha256"]
