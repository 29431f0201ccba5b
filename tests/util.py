"""Shared helpers of the parity tests."""
import glob
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden_cases():
    return sorted(os.path.basename(p)[5:-4] for p in glob.glob(os.path.join(GOLDEN, "case_*.npz")))


def load_case(name):
    with np.load(os.path.join(GOLDEN, "case_%s.npz" % name)) as z:
        return {k: z[k] for k in z.files}


def parity_error(got, want, rel=1e-4, floor=1e-6):
    """Largest violation of |got-want| <= rel*|want| + floor*max|want| (<= 0 means pass).

    `rel` is the north-star tolerance (1e-4 relative, fp32); `floor` is the absolute floor of
    SURVEY.md section 8(d): 1e-6 of the tensor's max-norm.
    """
    got = np.asarray(got, np.float64)
    want = np.asarray(want, np.float64)
    assert got.shape == want.shape, (got.shape, want.shape)
    scale = np.abs(want).max() if want.size else 0.0
    # 1e-12: a tensor that is zero by symmetry (e.g. the vertical-offset gradient of a one-row image) holds only
    # rounding noise of the order 1e-15 on both sides
    tol = rel * np.abs(want) + floor * scale + 1e-12
    return float((np.abs(got - want) - tol).max()) if want.size else 0.0


def assert_parity(got, want, name, rel=1e-4, floor=1e-6):
    assert np.all(np.isfinite(np.asarray(got))), "%s has non-finite values" % name
    viol = parity_error(got, want, rel, floor)
    if viol > 0:
        g, w = np.asarray(got, np.float64), np.asarray(want, np.float64)
        idx = np.unravel_index(np.argmax(np.abs(g - w)), g.shape)
        raise AssertionError("%s: parity violated by %.3e (max |diff| %.3e at %s: got %.7g want %.7g, max|want| %.3g)"
                             % (name, viol, np.abs(g - w).max(), idx, g[idx], w[idx], np.abs(w).max()))


def case_kernel_size(c):
    m = float(c["max_offset"])
    for k in (9, 17, 33, 65):
        if m <= k // 2:
            return k
    raise ValueError(m)
