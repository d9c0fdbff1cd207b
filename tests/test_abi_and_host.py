"""CPU: the C-ABI library loads and exports every symbol include/pfpgpu.h declares; host-side
helpers; loud failure without a GPU (no CPU fallback in the product path)."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "pfpgpu.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(pfp_[a-z0-9_]+)\s*\(", hdr)))


def test_header_symbols_exported(pkg):
    lib = pkg.load_library()
    names = declared_symbols()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"libpfpgpu.so does not export {n}"
    assert set(pkg.SYMBOLS) == set(names)


def test_version_and_strerror(pkg):
    lib = pkg.load_library()
    assert b"gfx950" in lib.pfp_version()
    assert lib.pfp_strerror(0) == b"ok" and lib.pfp_strerror(-4) == b"phrase hash collision"


def test_library_contains_gfx950_code_objects():
    so = os.path.join(ROOT, "big-bwt_amd", "libpfpgpu.so")
    data = open(so, "rb").read()
    assert b"gfx950" in data and b"kr_flag_kernel" in data and b"hard_groups_kernel" in data


def test_no_cpu_fallback_without_gpu(pkg):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(pkg.PfpError) as ei:
        pkg.Context(0)
    assert ei.value.code == -2   # PFP_ENODEV


def test_product_does_not_import_oracle():
    """the product path must never route through the oracle"""
    for dp, _, fs in os.walk(os.path.join(ROOT, "big-bwt_amd")):
        for f in fs:
            if f.endswith((".py", ".hip", ".hpp", ".c", ".h", ".cpp")) or f == "Makefile":
                src = open(os.path.join(dp, f), errors="replace").read()
                assert "oracle" not in src.lower().replace("no cpu fallback", ""), f"{f} mentions the oracle"


def test_pack5_roundtrip(pkg, O):
    v = np.array([0, 1, 255, 256, (1 << 40) - 1, 123456789012], dtype=np.uint64)
    b = pkg.pack5(v)
    assert len(b) == 30 and np.array_equal(pkg.unpack5(b), v)
    assert np.array_equal(O.pack5(v), b)


def test_c_driver_built_and_usage():
    exe = os.path.join(ROOT, "big-bwt_amd", "bigbwt")
    assert os.path.exists(exe)
    out = subprocess.run([exe, "-h"], capture_output=True, text=True)
    assert out.returncode == 0 and "--parsing" in out.stdout and "-S" in out.stdout and "--gpus N" in out.stdout and "--halo" in out.stdout
    both = subprocess.run([exe, "-S", "-s", "/dev/null"], capture_output=True, text=True)
    assert "not both" in both.stdout       # bigbwt:59-61


def test_textgen_is_deterministic(O):
    from textgen import make_text
    a = make_text(dict(kind="rand", seed="x", n=1000, alphabet_hex=b"ACGT".hex()))
    b = make_text(dict(kind="rand", seed="x", n=1000, alphabet_hex=b"ACGT".hex()))
    assert np.array_equal(a, b) and set(a.tolist()) <= set(b"ACGT")
    g = O.gen_fasta(1000, 2, 0.01, 3)
    assert bytes(g[:7]) == b">copy0\n" and g[-1] == 10


def test_bench_starts_its_own_ranks(monkeypatch):
    """`python bench.py --gpus N` with no RANK in the environment must start N ranks itself - as a child process of a
    parent that has not touched the GPU - instead of benchmarking one GPU under the name of N."""
    import importlib
    import subprocess
    import sys
    sys.path.insert(0, ROOT)
    bench = importlib.import_module("bench")
    seen = {}

    class Done:
        returncode = 7

    def fake_run(cmd, env=None, **kw):
        seen["cmd"], seen["env"] = cmd, env
        return Done()
    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3"])
    monkeypatch.delenv("RANK", raising=False)
    try:
        bench.main()
        assert False, "bench.main() went on in the parent"
    except SystemExit as e:
        assert e.code == 7          # the child's exit code is the parent's
    cmd = seen["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=4" in cmd and "--master-addr" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-4:] == ["--gpus", "4", "--steps", "3"] and cmd[-5].endswith("bench.py")
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_rank_script_splits_the_input_like_the_reference_threads():
    """dist_main.py (the Python ranks of `bigbwt -G N` under PFP_MULTI_PYTHON=1): byte ranges [n r / N, n (r + 1) / N) - consecutive,
    covering, never empty for n >= N (pscan.hpp:114-165 splits the input the same way)"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("pfp_dist_main", os.path.join(ROOT, "big-bwt_amd", "dist_main.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    for n in (0, 1, 7, 1000, 12_596_936_618):
        for size in (1, 2, 3, 8):
            rs = [m.shard_range(n, r, size) for r in range(size)]
            assert rs[0][0] == 0 and rs[-1][1] == n and all(rs[k][1] == rs[k + 1][0] for k in range(size - 1))
            if n >= size:
                assert all(hi > lo for lo, hi in rs)

