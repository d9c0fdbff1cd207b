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
