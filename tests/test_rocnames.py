"""tools/rocnames.py: key / value types out of rocPRIM's radix sort kernel names, whatever config type they carry (round 4: prims.hip
passes its own onesweep configuration, so the names hold a nested config instead of `default_config`)."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
from rocnames import sort_types, top_level_args  # noqa: E402

NS = "rocprim::ROCPRIM_400200_NS"


def name(cfg, key, value):
    return (f"void {NS}::detail::trampoline_kernel<{NS}::detail::wrapped_radix_sort_onesweep_config<{cfg}, {key}, {value}>, "
            f"({NS}::detail::target_arch)950, {NS}::detail::radix_sort_onesweep_iteration<{cfg}, false, {key} const*, {key}*>>")


def test_default_and_tuned_configs():
    tuned = (f"{NS}::radix_sort_onesweep_config<{NS}::kernel_config<1024u, 7u, 4294967295u>, {NS}::kernel_config<1024u, 7u, 4294967295u>, "
             f"8u, ({NS}::block_radix_rank_algorithm)2>")
    for cfg in (f"{NS}::default_config", tuned):
        assert sort_types(name(cfg, "unsigned long", "unsigned int")) == ("u64", "u32")
        assert sort_types(name(cfg, "unsigned int", "unsigned int")) == ("u32", "u32")
        assert sort_types(name(cfg, "unsigned long", "unsigned long")) == ("u64", "u64")
        assert sort_types(name(cfg, "unsigned long", f"{NS}::empty_type")) == ("u64", None)


def test_other_kernels_are_not_sorts():
    assert sort_types("pfp::slot_records_kernel<unsigned int>(unsigned long, unsigned int const*)") is None
    assert sort_types("void rocprim::detail::lookback_scan_kernel<false, rocprim::default_config>") is None


def test_argument_split_counts_brackets():
    s = "f<a<b, c>, (x)1, d<e<f, g>, h>>"
    assert top_level_args(s, 1) == ["a<b, c>", "(x)1", "d<e<f, g>, h>"]
