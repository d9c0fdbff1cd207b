"""big-bwt_amd: MI355X-native prefix-free-parsing BWT builder (HIP kernels behind a C ABI).

The directory name carries a hyphen (it mirrors the reference's name), so import it through
`__graft_entry__.load_package()` / `tests/conftest.py`, which register it as `bigbwt_amd`.
"""
from .pfp import (Context, PfpError, FLAG_SA, FLAG_SSA, FLAG_ESA, SYMBOLS, LIB_PATH, load_library,  # noqa: F401
                  pack5, unpack5)
