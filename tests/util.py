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


def make_inputs(seed, N, S, F, G, H, W, k, m, ignore=0):
    """Synthetic tensors of SURVEY.md 8(d): x ~ U[0,1), dy ~ N(0,1), w ~ N(0, 0.1^2), mu ~ U(-m, m) clipped to the kernel."""
    rs = np.random.RandomState(seed)
    x = rs.rand(N, S, H, W).astype(np.float32)
    dy = rs.randn(N, F, H, W).astype(np.float32)
    w = (rs.randn(1, S, G, F) * 0.1).astype(np.float32)
    if ignore:
        w[:, :, G - ignore:, :] = 0.0
    lim = k // 2 - 0.01
    mu1 = np.clip(rs.uniform(-m, m, (1, S, G, F)), -lim, lim).astype(np.float32)
    mu2 = np.clip(rs.uniform(-m, m, (1, S, G, F)), -lim, lim).astype(np.float32)
    return x, dy, w, mu1, mu2


def run_plan(plan, x, dy, w, mu1, mu2, dtype=None, calls=1, sigma=0.5):
    """forward + backward through the C ABI (status checked after every call) -> dict of the six tensors as numpy fp32."""
    import torch
    dtype = dtype or torch.float32
    dev = lambda a: torch.from_numpy(a).cuda()
    S, G, F = w.shape[1:]
    sg = torch.full((1, S, G, F), float(sigma), device="cuda")
    xd, dyd = dev(x).to(dtype), dev(dy).to(dtype)
    wd, m1, m2 = dev(w), dev(mu1), dev(mu2)
    for _ in range(calls):
        y = plan.forward(xd, wd, m1, m2, sg)
        plan.check_status()
        g = plan.backward(xd, dyd, wd, m1, m2, sg)
        plan.check_status()
    torch.cuda.synchronize()
    return dict(y=y.float().cpu().numpy(), dx=g[0].float().cpu().numpy(), dw=g[1].cpu().numpy(), dmu1=g[2].cpu().numpy(),
                dmu2=g[3].cpu().numpy(), dsigma=g[4].cpu().numpy())


def margins(got, want):
    """{tensor: max|got - want| / max|want|} -- what a parity test prints so that the distance to its bar is on record."""
    out = {}
    for key in want:
        w_ = np.asarray(want[key], np.float64)
        out[key] = float(np.abs(np.asarray(got[key], np.float64) - w_).max() / max(np.abs(w_).max(), 1e-300))
    return out


def record_margins(name, got, want, bar):
    """Append {test, bar, max|got-want|/max|want| per tensor} to gpurun_out/parity_margins.jsonl (where the GPU box's scratch
    directory exists): the distance of the big parity tests to their bars, kept under profiles/ per round."""
    import json
    m = margins(got, want)
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(out):
        with open(os.path.join(out, "parity_margins.jsonl"), "a") as fh:
            fh.write(json.dumps(dict(test=name, bar=bar, max_err_over_max_norm={k: float("%.3e" % v) for k, v in m.items()})) + "\n")
    return m


_TUNING = []


def tuning_capi():
    """A second instance of the ctypes binding, over libdau_conv_hip_tuning.so (`make tuning`: the same sources with
    -DDAU_TUNING).  Only that build reads the variant-pinning environment variables (DAU_GATHER_VARIANT, DAU_DOT_RW, ...), which the
    variant tests use to run the kernels production picks for large batches on shapes the oracle finishes in seconds."""
    if not _TUNING:
        import importlib.util
        pkg = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "dau-convnet_amd", "dau_conv")
        so = os.path.join(pkg, "libdau_conv_hip_tuning.so")
        assert os.path.exists(so), "%s missing: run `make -C dau-convnet_amd/csrc tuning` (or __graft_entry__.build())" % so
        spec = importlib.util.spec_from_file_location("dau_conv_capi_tuning", os.path.join(pkg, "_capi.py"))
        mod = importlib.util.module_from_spec(spec)
        old = os.environ.get("DAU_CONV_LIB")
        os.environ["DAU_CONV_LIB"] = so
        try:
            spec.loader.exec_module(mod)
        finally:
            if old is None:
                del os.environ["DAU_CONV_LIB"]
            else:
                os.environ["DAU_CONV_LIB"] = old
        _TUNING.append(mod)
    return _TUNING[0]
