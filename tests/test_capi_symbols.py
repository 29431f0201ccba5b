"""CPU: the HIP shared library loads and exports every entry point include/dau_conv.h declares,
and argument validation that needs no device works.  No compute is launched here."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "dau_conv.h")
LIB = os.path.join(ROOT, "dau-convnet_amd", "dau_conv", "libdau_conv_hip.so")


def declared_symbols():
    src = open(HEADER).read()
    return sorted(set(re.findall(r"DAU_API\s+[\w\s\*]+?\b(dau_conv_\w+)\s*\(", src)))


def test_header_declares_expected_entry_points():
    names = declared_symbols()
    for must in ("dau_conv_plan_create", "dau_conv_plan_destroy", "dau_conv_workspace_bytes", "dau_conv_forward",
                 "dau_conv_backward", "dau_conv_check_status", "dau_conv_last_error", "dau_conv_abi_version",
                 "dau_conv_backward_param_sums", "dau_conv_finalize_param_grads", "dau_conv_last_status",
                 "dau_conv_filters", "dau_conv_unit_table", "dau_conv_plan_get_info", "dau_conv_profile_begin",
                 "dau_conv_profile_end", "dau_conv_build_id"):
        assert must in names


def test_library_exports_every_declared_symbol():
    assert os.path.exists(LIB), "build the library first: make -C dau-convnet_amd/csrc"
    lib = ctypes.CDLL(LIB)
    for name in declared_symbols():
        assert hasattr(lib, name), "missing export %s" % name
    assert lib.dau_conv_abi_version() == 2


def test_build_id_is_the_fingerprint_of_the_sources():
    """The library names the sources it was built from (bench.py matches committed counter files by it); the Makefile and
    bench.source_fingerprint() must compute the same thing, wherever the tree lies."""
    import bench
    from dau_conv import _capi
    assert len(_capi.build_id()) == 16 and _capi.build_id() == bench.source_fingerprint()


def test_plan_validation_without_device():
    from dau_conv import _capi
    p = _capi.Plan(2, 3, 4, 2, 8, 9, max_kernel_size=9, sigma_hint=0.5)
    assert p.info["offset_bucket"] == 4 and p.info["blur_support"] == 7
    assert p.workspace_bytes(_capi.PASS_FORWARD) > 0 and p.workspace_bytes(_capi.PASS_BACKWARD) > p.workspace_bytes(_capi.PASS_FORWARD)
    for k, bucket in ((17, 8), (33, 16), (37, 18), (41, 20), (49, 24), (65, 32), (11, 8), (35, 18)):
        assert _capi.Plan(1, 1, 1, 2, 8, 8, max_kernel_size=k).info["offset_bucket"] == bucket
    with pytest.raises(_capi.InvalidArgumentError):
        _capi.Plan(1, 1, 1, 2, 8, 8, max_kernel_size=67)          # offsets beyond 32 px (dau_conv_op.cpp:245-248)
    with pytest.raises(_capi.FailedPreconditionError):
        _capi.Plan(1, 1, 1, 2, 8, 8, sigma_hint=0.0)               # DAU_CHECK(sigma > 0)
    with pytest.raises(_capi.InvalidArgumentError):
        _capi.Plan(1, 1, 1, 2, 8, 8, sigma_hint=2.0)               # prefilter larger than 17x17 (convolve.cu:40)
    with pytest.raises(_capi.InvalidArgumentError):
        _capi.Plan(1, 1, 1, 2, 8, 8, number_units_ignore=2)
    # kernel sets a call can choose from: every bucket up to the static one, unless pinned
    assert _capi.Plan(2, 4, 8, 2, 32, 32, max_kernel_size=65).info["bucket_sets"] == 7
    assert _capi.Plan(2, 4, 8, 2, 32, 32, max_kernel_size=9).info["bucket_sets"] == 1
    assert _capi.Plan(2, 4, 8, 2, 32, 32, max_kernel_size=65,
                      flags=_capi.FLAG_USE_INTERPOLATION | _capi.FLAG_STATIC_BUCKET).info["bucket_sets"] == 1
    assert p.last_status() is None                                  # no call has run
    ut = _capi.Plan(1, 1, 1, 2, 16, 65, flags=_capi.FLAG_USE_INTERPOLATION | _capi.FLAG_UNIT_TESTING)
    assert ut.info["drop_last_row"] == 1 and ut.info["drop_last_col"] == 0
