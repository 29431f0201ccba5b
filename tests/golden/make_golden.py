#!/usr/bin/env python3
"""Generate golden vectors from the reference's own numpy oracle.

Runs ONLY in the build container (needs /root/reference).  It reads the text of
plugins/tensorflow/tests/dau_conv_test.py, takes the `class DAUConvPython` block
(lines 13-295; the rest of the file needs TensorFlow/pylab and is Python 2), and
executes that class in memory.  Nothing of the reference is written to disk: the
.npz files hold inputs and the arrays the class returned, i.e. data.

Two Python-2 integer divisions inside `_get_filters` (only reached with
single_dim_kernel / aggr_forbid_positive) are rewritten to `//` in memory.

Usage:  python tests/golden/make_golden.py   (writes tests/golden/*.npz)
"""
import os
import sys

import numpy as np
from scipy.ndimage import correlate

REF = "/root/reference/plugins/tensorflow/tests/dau_conv_test.py"
OUT = os.path.dirname(os.path.abspath(__file__))


def load_reference_oracle():
    with open(REF) as fh:
        lines = fh.read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith("class DAUConvPython"))
    end = next(i for i, l in enumerate(lines) if l.startswith("class DAUConvTest"))
    src = "\n".join(lines[start:end])
    src = src.replace("valid_filter.shape[0]/2", "valid_filter.shape[0]//2")
    src = src.replace("valid_filter.shape[1]/2+1", "valid_filter.shape[1]//2+1")
    ns = {"np": np, "correlate": correlate}
    exec(compile(src, "<reference DAUConvPython>", "exec"), ns)
    return ns["DAUConvPython"]


# name, N, S, F, H, W, G_stored, ignore, max_kernel_size, mu_init_range, sigma, extras
CASES = [
    dict(name="split_w65_h8", N=2, S=3, F=4, H=8, W=65, G=2, m=3.0),
    dict(name="small_8x8", N=1, S=4, F=8, H=8, W=8, G=2, m=3.0),
    dict(name="k9_16x16_g4", N=2, S=4, F=8, H=16, W=16, G=4, m=3.0),
    dict(name="k17_16x16_g4", N=2, S=4, F=8, H=16, W=16, G=4, m=6.0),
    dict(name="odd_s3_32x32", N=2, S=3, F=8, H=32, W=32, G=4, m=3.0),
    dict(name="tiny_6x6_m8", N=2, S=2, F=4, H=6, W=6, G=2, m=8.0),
    dict(name="k33_40x40_m10", N=1, S=2, F=4, H=40, W=40, G=4, m=10.0),
    dict(name="k65_64x64_m20", N=1, S=2, F=2, H=64, W=64, G=2, m=20.0),
    dict(name="single_unit_ignore1", N=2, S=3, F=4, H=12, W=12, G=2, ignore=1, m=3.0),
    dict(name="sigma08", N=1, S=2, F=4, H=14, W=18, G=2, m=3.0, sigma=0.8),
    dict(name="no_unit_testing", N=2, S=3, F=4, H=16, W=24, G=2, m=3.0, unit_testing=False),
    dict(name="no_interpolation", N=2, S=3, F=4, H=10, W=13, G=2, m=3.0, use_interpolation=False),
    dict(name="single_dim_1d", N=2, S=3, F=4, H=9, W=20, G=2, m=3.0, single_dim_kernel=True, mu2_zero=True),
    dict(name="forbid_positive", N=1, S=2, F=4, H=9, W=20, G=2, m=3.0, single_dim_kernel=True,
         forbid_positive=True, mu2_zero=True),
    dict(name="config0_quick", N=2, S=16, F=32, H=32, W=32, G=2, m=3.0),
    # added later (appended, so that the random stream of the cases above is unchanged): unit counts and map sizes that
    # exercise the unit passes, stacked planes and 7-row regions of the tiled kernels
    dict(name="g6_28x28", N=3, S=4, F=8, H=28, W=28, G=6, m=3.0),
    dict(name="g8_14x14", N=5, S=3, F=8, H=14, W=14, G=8, m=3.0),
    dict(name="g3_7x7_k17", N=4, S=4, F=8, H=7, W=7, G=3, m=6.0),
    dict(name="g5_20x100", N=1, S=2, F=4, H=20, W=100, G=5, m=3.0),
]


def main():
    Oracle = load_reference_oracle()
    rs = np.random.RandomState(0)
    # filter tables (the oracle's fixed 9x9 support)
    for sig in (0.5, 0.8):
        for sd, fp in ((False, False), (True, False), (True, True)):
            f = Oracle()._get_filters(sig, single_dim_kernel=sd, aggr_forbid_positive=fp)
            np.savez_compressed(os.path.join(OUT, "filters_sigma%02d_sd%d_fp%d.npz" % (int(sig * 10), sd, fp)),
                                sigma=np.float32(sig), Gn=f[0], Dw=f[1], Dmu1=f[2], Dmu2=f[3], Dsigma=f[4], Gerr=f[5])
    for c in CASES:
        N, S, F, H, W, G = c["N"], c["S"], c["F"], c["H"], c["W"], c["G"]
        m = c["m"]
        sigma = c.get("sigma", 0.5)
        ignore = c.get("ignore", 0)
        ut = c.get("unit_testing", True)
        interp = c.get("use_interpolation", True)
        sd = c.get("single_dim_kernel", False)
        fp = c.get("forbid_positive", False)
        # distributions of dau_conv_test.py:342-368, rounded to float32 first
        x = rs.rand(N, S, H, W).astype(np.float32)
        w = (rs.randn(1, S, G, F) * 0.1).astype(np.float32)
        if ignore:
            w[:, :, G - ignore:, :] = 0
        mu1 = rs.uniform(-m, m, (1, S, G, F)).astype(np.float32)
        mu2 = rs.uniform(-m, m, (1, S, G, F)).astype(np.float32)
        if c.get("mu2_zero"):
            mu2[:] = 0
        dy = rs.randn(N, F, H, W).astype(np.float32)
        o = Oracle()
        y = o.forward_cpu(x=x.astype(np.float64), w=w, mu1=mu1, mu2=mu2, sigma=[sigma],
                          num_dau_units_ignore=ignore, single_dim_kernel=sd, aggr_forbid_positive=fp,
                          use_interpolation=interp)
        dx, dw, dmu1, dmu2, dsigma = o.backward_cpu(x=x.astype(np.float64), error=dy.copy(), w=w, mu1=mu1, mu2=mu2,
                                                   sigma=[sigma], num_dau_units_ignore=ignore, unit_testing=ut,
                                                   single_dim_kernel=sd, aggr_forbid_positive=fp,
                                                   use_interpolation=interp)
        np.savez_compressed(os.path.join(OUT, "case_%s.npz" % c["name"]),
                            x=x, w=w, mu1=mu1, mu2=mu2, dy=dy, sigma=np.float32(sigma), ignore=np.int32(ignore),
                            unit_testing=np.int32(ut), use_interpolation=np.int32(interp),
                            single_dim_kernel=np.int32(sd), forbid_positive_dim1=np.int32(fp),
                            max_offset=np.float32(m),
                            y=y.astype(np.float32), dx=dx.astype(np.float32), dw=dw.astype(np.float32),
                            dmu1=dmu1.astype(np.float32), dmu2=dmu2.astype(np.float32),
                            dsigma=dsigma.astype(np.float32))
        print("wrote case_%s" % c["name"], y.shape, float(np.abs(y).max()))


if __name__ == "__main__":
    sys.exit(main())
