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
                 "dau_conv_profile_end", "dau_conv_build_id", "dau_conv_filter_support"):
        assert must in names


def test_library_exports_every_declared_symbol():
    assert os.path.exists(LIB), "build the library first: make -C dau-convnet_amd/csrc"
    lib = ctypes.CDLL(LIB)
    for name in declared_symbols():
        assert hasattr(lib, name), "missing export %s" % name
    assert lib.dau_conv_abi_version() == 4


def test_release_library_reads_no_tuning_knobs():
    """A plan's behaviour is fully determined by dau_conv_desc (+ DAU_WORKSPACE_BUDGET_GB): the environment variables that pin
    kernel variants, chunkings and staging tiles for A/B runs exist only in the tuning build (make tuning, -DDAU_TUNING).  The
    shipped library must not even contain their names."""
    blob = open(LIB, "rb").read()
    for name in (b"DAU_DOT_", b"DAU_GATHER_", b"DAU_BLUR", b"DAU_DENSE_FT", b"DAU_DENSE_STAGE", b"DAU_DENSE_WGRAD",
                 b"DAU_DENSE_SCATTER", b"DAU_DENSE_R3", b"DAU_DENSE_SPLIT", b"DAU_SPLIT_", b"DAU_WGRAD_", b"DAU_DYNAMIC_BUCKET", b"DAU_DIAG"):
        assert name not in blob, "release library contains the tuning knob %s*" % name.decode()
    assert b"DAU_WORKSPACE_BUDGET_GB" in blob
    tuning = LIB.replace("libdau_conv_hip.so", "libdau_conv_hip_tuning.so")
    assert os.path.exists(tuning), "build the tuning library too: make -C dau-convnet_amd/csrc tuning"
    tblob = open(tuning, "rb").read()
    assert b"DAU_GATHER_VARIANT" in tblob and b"DAU_DOT_RW" in tblob
    tlib = ctypes.CDLL(tuning)
    lib = ctypes.CDLL(LIB)
    tlib.dau_conv_build_id.restype = lib.dau_conv_build_id.restype = ctypes.c_char_p
    assert tlib.dau_conv_build_id() == lib.dau_conv_build_id()       # same sources


def test_filter_support_rule():
    from dau_conv import _capi
    # 2*ceil(5*sigma)+1 in float32 (base_dau_conv_layer.cpp:146); 0.8f * 5 rounds to exactly 4 in float32
    assert [_capi.filter_support(s) for s in (0.3, 0.5, 0.6, 0.8, 1.0, 1.2, 1.6)] == [5, 7, 7, 9, 11, 13, 17]


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


def test_which_dense_members_a_default_plan_holds():
    """Plan creation needs no device: the radii of the two-limb f16 dense gather-sum a plan holds by default are those that pay
    for its unit count on its tiling (split_pays, dau_conv_api.hip); the flags force all / none; the bf16-product dense flag and a
    static bucket exclude them.  (The same table runs on the GPU in tests/test_gpu_dense_split.py.)"""
    from dau_conv import _capi
    for (S, F, G, H, W), want in (((256, 256, 4, 56, 56), 0b11100), ((256, 256, 6, 56, 56), 0b11100), ((256, 256, 2, 56, 56), 0b00100),
                                  ((256, 256, 1, 56, 56), 0), ((96, 256, 4, 27, 27), 0b01100), ((512, 512, 4, 28, 28), 0b11100),
                                  ((7, 5, 4, 16, 16), 0), ((64, 64, 2, 56, 56), 0), ((128, 128, 4, 16, 16), 0b11100)):
        assert _capi.Plan(2, S, F, G, H, W, max_kernel_size=9, sigma_hint=0.5).info["gather_dense_split"] == want, (S, F, G, H, W)
    I = _capi.FLAG_USE_INTERPOLATION
    assert _capi.Plan(2, 7, 5, 1, 16, 16, flags=I | _capi.FLAG_DENSE_SPLIT_F16).info["gather_dense_split"] == 0b11100
    assert _capi.Plan(2, 256, 256, 4, 56, 56, flags=I | _capi.FLAG_NO_DENSE_SPLIT).info["gather_dense_split"] == 0
    assert _capi.Plan(2, 256, 256, 4, 56, 56, flags=I | _capi.FLAG_STATIC_BUCKET).info["gather_dense_split"] == 0
    assert _capi.Plan(2, 256, 256, 4, 56, 56, flags=I | _capi.FLAG_IO_BF16 | _capi.FLAG_DENSE_BF16).info["gather_dense_split"] == 0
    assert _capi.Plan(2, 256, 256, 4, 56, 56, max_kernel_size=65).info["gather_dense_split"] == 0b11100      # any kernel size: the members serve calls within +-4
    assert _capi.Plan(2, 256, 256, 4, 56, 56, sigma_hint=1.2).info["gather_dense_split"] == 0                 # 13-tap prefilter: no staging instantiation
    with pytest.raises(_capi.InvalidArgumentError):
        _capi.Plan(2, 8, 8, 2, 8, 8, flags=I | _capi.FLAG_NO_DENSE_SPLIT | _capi.FLAG_DENSE_SPLIT_F16)


def test_plan_cache_keys_on_the_prefilter_support_and_is_bounded():
    """A trainable sigma moves every optimizer step; a plan depends on it only through the prefilter support, so the cache
    must keep returning the same plan (with its kernel sets, hint and pending status) -- and it must not grow without bound."""
    import torch
    import importlib
    dc = importlib.import_module("dau_conv.dau_conv")        # (the package attribute of that name is the op function)
    dc._PLANS.clear()
    x = torch.zeros(2, 3, 8, 9)
    w = torch.zeros(1, 3, 2, 4)
    st = lambda sigma: dc._settings(torch.full((1,), sigma), num_output=4, kernel_size=9)
    first = dc._get_plan(x, w, st(0.5))
    for i in range(1, 40):
        assert dc._get_plan(x, w, st(0.5 + 1e-3 * i)) is first        # 0.501 ... 0.539: still the 7 x 7 prefilter
    assert len(dc._PLANS) == 1
    other = dc._get_plan(x, w, st(0.61))                               # 2*ceil(3.05)+1 = 9: another plan
    assert other is not first and other.info["blur_support"] == 9 and len(dc._PLANS) == 2
    for h in range(8, 8 + dc._PLAN_CACHE_MAX + 10):                    # many shapes: least recently used plans are dropped
        dc._get_plan(torch.zeros(2, 3, h, 9), w, st(0.5))
    assert len(dc._PLANS) == dc._PLAN_CACHE_MAX
    dc._PLANS.clear()


def test_check_offsets_is_normalised():
    import numpy as np
    import torch
    import importlib
    dc = importlib.import_module("dau_conv.dau_conv")        # (the package attribute of that name is the op function)
    sg = torch.full((1,), 0.5)
    for given, want in (("async", "async"), (True, True), (False, False), (1, True), (1.0, True), (np.True_, True), (0, False),
                        (np.bool_(False), False), ("yes", True)):
        got = dc._settings(sg, check_offsets=given)["check_offsets"]
        assert got is want or got == want and type(got) is type(want), (given, got)
