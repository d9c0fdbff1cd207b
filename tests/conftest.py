import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import __graft_entry__ as entry  # noqa: E402


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def O():
    """CPU oracle (checker)."""
    return entry.load_oracle()


@pytest.fixture(scope="session")
def pkg():
    return entry.load_package()


@pytest.fixture(scope="session")
def ctx(pkg):
    """One GPU context for the whole session; fails loudly when the HIP path is unavailable."""
    c = pkg.Context(0)
    yield c
    c.close()


@pytest.fixture(params=[32, 64], ids=["idx32", "idx64"])
def wctx(request, ctx):
    """the session context in both index widths: by size (32-bit positions for these inputs) and the forced
    64-bit build (pfp_set_index_bits; what a dictionary of 4 GiB or more selects, bigbwt:109-151)"""
    ctx.set_index_bits(64 if request.param == 64 else 0)
    yield ctx
    ctx.set_index_bits(0)


@pytest.fixture(autouse=True)
def _pool_debug_check(request):
    """PFP_POOL_DEBUG=1 (exact-size device blocks with canary bands): fail the test that damaged a band"""
    yield
    if os.environ.get("PFP_POOL_DEBUG") and request.node.get_closest_marker("gpu") and "ctx" in request.fixturenames:
        request.getfixturevalue("ctx").debug_check()


@pytest.fixture(scope="session")
def synth(pkg):
    import importlib
    return importlib.import_module("bigbwt_amd.synth")


@pytest.fixture(scope="session")
def golden_full():
    import json
    with open(os.path.join(ROOT, "tests", "golden", "golden_full.json")) as fh:
        return json.load(fh)


@pytest.fixture(scope="session")
def golden():
    import json
    with open(os.path.join(ROOT, "tests", "golden", "golden.json")) as fh:
        return json.load(fh)
