"""ctypes front-end of the CPU oracle (oracle/dau_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py -- never by the product package.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libdau_oracle.so")
_lib = None

_f32p = ctypes.POINTER(ctypes.c_float)
_i32p = ctypes.POINTER(ctypes.c_int32)


def build(force=False):
    src = os.path.join(_HERE, "dau_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE] + (["-B"] if force else []))
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_SO)
        _lib.dau_oracle_num_threads.restype = ctypes.c_int
        _lib.dau_oracle_filter_support.restype = ctypes.c_int
        _lib.dau_oracle_filter_support.argtypes = [ctypes.c_float]
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(_f32p)


def _c(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def num_threads():
    return lib().dau_oracle_num_threads()


def filter_support(sigma):
    return lib().dau_oracle_filter_support(ctypes.c_float(sigma))


def filters(sigma, k=None, single_dim_kernel=False, forbid_positive_dim1=False):
    """-> dict of six (k,k) float32 filters: Gn, Dw, Dmu1, Dmu2, Dsigma, Gerr."""
    k = filter_support(sigma) if k is None else int(k)
    out = {n: np.zeros((k, k), np.float32) for n in ("Gn", "Dw", "Dmu1", "Dmu2", "Dsigma", "Gerr")}
    lib().dau_oracle_filters(ctypes.c_float(sigma), k, int(single_dim_kernel), int(forbid_positive_dim1),
                             *[_p(out[n]) for n in ("Gn", "Dw", "Dmu1", "Dmu2", "Dsigma", "Gerr")])
    return out


def blur(x, filt):
    x = _c(x)
    filt = _c(filt)
    H, W = x.shape[-2:]
    out = np.empty_like(x)
    lib().dau_oracle_blur(_p(x), ctypes.c_long(x.size // (H * W)), H, W, _p(filt), filt.shape[0], _p(out))
    return out


def unit_table(mu1, mu2, use_interpolation=True):
    mu1, mu2 = _c(mu1).ravel(), _c(mu2).ravel()
    off = np.zeros((mu1.size, 2), np.int32)
    b = np.zeros((mu1.size, 4), np.float32)
    lib().dau_oracle_unit_table(_p(mu1), _p(mu2), ctypes.c_long(mu1.size), int(use_interpolation),
                                off.ctypes.data_as(_i32p), _p(b))
    return off, b


def _dims(x, w):
    N, S, H, W = x.shape
    _, S2, G, F = w.shape
    assert S2 == S, (x.shape, w.shape)
    return N, S, F, G, H, W


def offset_and_sum(xb, w, mu1, mu2, ignore=0, use_interpolation=True):
    xb, w, mu1, mu2 = _c(xb), _c(w), _c(mu1), _c(mu2)
    N, S, F, G, H, W = _dims(xb, w)
    y = np.empty((N, F, H, W), np.float32)
    lib().dau_oracle_offset_and_sum(_p(xb), N, S, F, G, H, W, _p(w), _p(mu1), _p(mu2), int(ignore),
                                    int(use_interpolation), _p(y))
    return y


def offset_and_dot(xk, err, mu1, mu2, ignore=0, use_interpolation=True):
    xk, err, mu1, mu2 = _c(xk), _c(err), _c(mu1), _c(mu2)
    N, S, H, W = xk.shape
    _, _, G, F = mu1.shape
    out = np.empty((1, S, G, F), np.float32)
    lib().dau_oracle_offset_and_dot(_p(xk), _p(err), N, S, F, G, H, W, _p(mu1), _p(mu2), int(ignore),
                                    int(use_interpolation), _p(out))
    return out


def apply_edge_rule(err):
    err = _c(err)
    H, W = err.shape[-2:]
    out = np.empty_like(err)
    lib().dau_oracle_apply_edge_rule(_p(err), ctypes.c_long(err.size // (H * W)), H, W, _p(out))
    return out


def forward(x, w, mu1, mu2, sigma, k=0, ignore=0, use_interpolation=True, single_dim_kernel=False,
            forbid_positive_dim1=False):
    x, w, mu1, mu2 = _c(x), _c(w), _c(mu1), _c(mu2)
    N, S, F, G, H, W = _dims(x, w)
    y = np.empty((N, F, H, W), np.float32)
    lib().dau_oracle_forward(_p(x), N, S, F, G, H, W, _p(w), _p(mu1), _p(mu2), ctypes.c_float(sigma),
                             int(k), int(ignore), int(use_interpolation), int(single_dim_kernel),
                             int(forbid_positive_dim1), _p(y))
    return y


def backward(x, dy, w, mu1, mu2, sigma, k=0, ignore=0, use_interpolation=True, single_dim_kernel=False,
             forbid_positive_dim1=False, unit_testing=False, mu_learning_rate_factor=1.0,
             need=("dx", "dw", "dmu1", "dmu2", "dsigma")):
    x, dy, w, mu1, mu2 = _c(x), _c(dy), _c(w), _c(mu1), _c(mu2)
    N, S, F, G, H, W = _dims(x, w)
    out = {
        "dx": np.empty((N, S, H, W), np.float32) if "dx" in need else None,
        "dw": np.empty((1, S, G, F), np.float32) if "dw" in need else None,
        "dmu1": np.empty((1, S, G, F), np.float32) if "dmu1" in need else None,
        "dmu2": np.empty((1, S, G, F), np.float32) if "dmu2" in need else None,
        "dsigma": np.empty((1, S, G, F), np.float32) if "dsigma" in need else None,
    }
    lib().dau_oracle_backward(_p(x), _p(dy), N, S, F, G, H, W, _p(w), _p(mu1), _p(mu2),
                              ctypes.c_float(sigma), int(k), int(ignore), int(use_interpolation),
                              int(single_dim_kernel), int(forbid_positive_dim1), int(unit_testing),
                              ctypes.c_float(mu_learning_rate_factor),
                              *[_p(out[n]) for n in ("dx", "dw", "dmu1", "dmu2", "dsigma")])
    return out
