"""CPU: the C oracle (oracle/dau_oracle.c) against the golden vectors produced by the
reference's own numpy oracle (tests/golden/make_golden.py).  This pins the oracle."""
import os

import numpy as np
import pytest

from oracle import dau_oracle as orc
from util import GOLDEN, assert_parity, golden_cases, load_case


@pytest.mark.parametrize("tag", ["sigma05_sd0_fp0", "sigma05_sd1_fp0", "sigma05_sd1_fp1", "sigma08_sd0_fp0",
                                 "sigma08_sd1_fp0", "sigma08_sd1_fp1"])
def test_filters_match_reference_oracle(tag):
    z = np.load(os.path.join(GOLDEN, "filters_%s.npz" % tag))
    sd, fp = int(tag[-5]), int(tag[-1])
    # same 9x9 support as the numpy oracle (dau_conv_test.py:178-180)
    f = orc.filters(float(z["sigma"]), k=9, single_dim_kernel=sd, forbid_positive_dim1=fp)
    for name in ("Gn", "Dw", "Dmu1", "Dmu2", "Dsigma", "Gerr"):
        assert_parity(f[name], z[name], "%s/%s" % (tag, name), rel=1e-6, floor=1e-7)


def test_support_rule_and_truncation():
    # C++ support rule 2*ceil(5*sigma)+1 (base_dau_conv_layer.cpp:146): 7x7 at 0.5, 9x9 at 0.8
    assert orc.filter_support(0.5) == 7 and orc.filter_support(0.8) == 9 and orc.filter_support(1.6) == 17
    f7, f9 = orc.filters(0.5, k=7), orc.filters(0.5, k=9)
    for name in f7:
        assert np.abs(f9[name][1:-1, 1:-1] - f7[name]).max() < 1e-12  # outer ring of the 9x9 is ~1e-14
        assert np.abs(f9[name][0]).max() < 1e-12


@pytest.mark.parametrize("name", golden_cases())
def test_forward_backward_match_reference_oracle(name):
    c = load_case(name)
    kw = dict(sigma=float(c["sigma"]), ignore=int(c["ignore"]), use_interpolation=bool(c["use_interpolation"]),
              single_dim_kernel=bool(c["single_dim_kernel"]), forbid_positive_dim1=bool(c["forbid_positive_dim1"]))
    # k=0 -> the product's support rule (7x7 at sigma 0.5); the numpy oracle used 9x9, identical to ~1e-14
    y = orc.forward(c["x"], c["w"], c["mu1"], c["mu2"], k=0, **kw)
    assert_parity(y, c["y"], name + "/y", rel=1e-5, floor=1e-6)
    g = orc.backward(c["x"], c["dy"], c["w"], c["mu1"], c["mu2"], k=0, unit_testing=bool(c["unit_testing"]),
                     mu_learning_rate_factor=1.0, **kw)
    for key in ("dx", "dw", "dmu1", "dmu2", "dsigma"):
        # the numpy oracle accumulates in float32 (dau_conv_test.py:59,173), the C oracle in double
        assert_parity(g[key], c[key], name + "/" + key, rel=1e-5, floor=2e-6)


def test_unit_table_bit_exact_definition():
    rs = np.random.RandomState(1)
    mu1 = rs.uniform(-8, 8, 500).astype(np.float32)
    mu2 = rs.uniform(-8, 8, 500).astype(np.float32)
    mu1[:4] = [-4.0, 4.0, 3.99, -0.0]
    off, b = orc.unit_table(mu1, mu2)
    fx = mu1 - np.floor(mu1)
    fy = mu2 - np.floor(mu2)
    assert np.array_equal(off[:, 0], np.floor(mu1).astype(np.int32))
    assert np.array_equal(off[:, 1], np.floor(mu2).astype(np.int32))
    one = np.float32(1)
    assert np.array_equal(b[:, 0], (one - fx) * (one - fy)) and np.array_equal(b[:, 1], fx * (one - fy))
    assert np.array_equal(b[:, 2], (one - fx) * fy) and np.array_equal(b[:, 3], fx * fy)
    off0, b0 = orc.unit_table(mu1, mu2, use_interpolation=False)
    assert np.array_equal(off0, off) and np.array_equal(b0[:, 0], np.ones(500, np.float32)) and not b0[:, 1:].any()


def test_edge_rule_matches_oracle_rule():
    e = np.ones((2, 3, 16, 65), np.float32)
    o = orc.apply_edge_rule(e)
    assert o[..., -1, :].sum() == 0 and o[..., :-1, -1].min() == 1  # H=16 dropped, W=65 kept
    o = orc.apply_edge_rule(np.ones((1, 1, 6, 8), np.float32))
    assert o[..., :, -1].sum() == 0 and o[..., -1, :-1].min() == 1  # W=8 dropped, H=6 kept


def test_linearity_and_shift_properties():
    # size-independent properties: forward is linear in x and in w; a unit with integer offset is a pure shift
    rs = np.random.RandomState(2)
    N, S, F, G, H, W = 1, 2, 3, 2, 11, 13
    x1, x2 = rs.rand(N, S, H, W).astype(np.float32), rs.rand(N, S, H, W).astype(np.float32)
    w = rs.randn(1, S, G, F).astype(np.float32)
    mu1 = rs.uniform(-3, 3, (1, S, G, F)).astype(np.float32)
    mu2 = rs.uniform(-3, 3, (1, S, G, F)).astype(np.float32)
    ya, yb = orc.forward(x1, w, mu1, mu2, 0.5), orc.forward(x2, w, mu1, mu2, 0.5)
    yab = orc.forward(x1 + x2, w, mu1, mu2, 0.5)
    assert_parity(yab, ya + yb, "linearity", rel=1e-5, floor=1e-6)
    xb = orc.blur(x1, orc.filters(0.5)["Gn"])
    wi = np.zeros((1, S, G, F), np.float32); wi[0, 1, 0, 2] = 1
    m1 = np.zeros_like(wi); m2 = np.zeros_like(wi); m1[0, 1, 0, 2] = 2; m2[0, 1, 0, 2] = -1
    y = orc.offset_and_sum(xb, wi, m1, m2)
    want = np.zeros((H, W), np.float32); want[1:, :W - 2] = xb[0, 1, :H - 1, 2:]
    assert np.array_equal(y[0, 2], want) and not y[0, :2].any()
