"""GPU: the densified bf16 gather-sum (DAU_FLAG_DENSE_BF16 + DAU_FLAG_IO_BF16, k_dense_bf16.hip): for calls whose offsets lie
within +-4 the two gather-sum passes (y and dx) run as an implicit GEMM on the bf16 matrix cores over a dense 9 x 9
kernel per channel pair.  Bar (SURVEY.md 8d, bf16 configuration): 2e-2 relative + 4e-3 of the max-norm against the fp32
oracle fed the bf16-rounded inputs; the parameter gradients do not use the dense form and keep the fp32 bar."""
import numpy as np
import pytest
import torch

from oracle import dau_oracle as orc
from util import assert_parity

pytestmark = pytest.mark.gpu


def _case(seed, N, S, F, G, H, W, m):
    rs = np.random.RandomState(seed)
    xb = torch.from_numpy(rs.rand(N, S, H, W).astype(np.float32)).to(torch.bfloat16)
    dyb = torch.from_numpy(rs.randn(N, F, H, W).astype(np.float32)).to(torch.bfloat16)
    w = (rs.randn(1, S, G, F) * 0.1).astype(np.float32)
    mu1 = rs.uniform(-m, m, (1, S, G, F)).astype(np.float32)
    mu2 = rs.uniform(-m, m, (1, S, G, F)).astype(np.float32)
    return xb, dyb, w, mu1, mu2


def _run(plan, xb, dyb, w, mu1, mu2, calls=1):
    dev = lambda a: torch.from_numpy(a).cuda()
    S, G, F = w.shape[1:]
    sigma = torch.full((1, S, G, F), 0.5, device="cuda")
    for _ in range(calls):
        y = plan.forward(xb.cuda(), dev(w), dev(mu1), dev(mu2), sigma)
        plan.check_status()
        g = plan.backward(xb.cuda(), dyb.cuda(), dev(w), dev(mu1), dev(mu2), sigma)
        plan.check_status()
    torch.cuda.synchronize()
    return y, g


def _check(y, g, xb, dyb, w, mu1, mu2, name, dense_params=False):
    x32, dy32 = xb.float().numpy(), dyb.float().numpy()
    want_y = orc.forward(x32, w, mu1, mu2, 0.5)
    want = orc.backward(x32, dy32, w, mu1, mu2, 0.5)
    assert y.dtype == torch.bfloat16 and g[0].dtype == torch.bfloat16
    assert_parity(y.float().cpu().numpy(), want_y, name + "/y", rel=2e-2, floor=4e-3)
    assert_parity(g[0].float().cpu().numpy(), want["dx"], name + "/dx", rel=2e-2, floor=4e-3)
    for t, key in zip(g[1:], ("dw", "dmu1", "dmu2", "dsigma")):
        if dense_params:      # the dense correlations round Xk and the error to bfloat16: the bf16 bar
            assert_parity(t.cpu().numpy(), want[key], name + "/" + key, rel=2e-2, floor=4e-3)
        else:                 # exact fp32 gather-dot on the bf16 inputs: the fp32 bar
            assert_parity(t.cpu().numpy(), want[key], name + "/" + key)


@pytest.mark.parametrize("shape", [
    dict(N=2, S=16, F=128, G=4, H=56, W=56),      # one column block of 7 subtiles, whole chunks and channel blocks
    dict(N=3, S=20, F=40, G=3, H=30, W=45),       # ragged everything: channels, rows, columns, odd batch
    dict(N=2, S=7, F=5, G=2, H=9, W=6),           # tiny
    dict(N=2, S=33, F=130, G=6, H=28, W=28),      # two channel blocks, the second almost empty
    dict(N=1, S=16, F=16, G=2, H=20, W=130),      # three column blocks
    dict(N=2, S=32, F=64, G=1, H=17, W=64),       # eight subtiles per block
    dict(N=4, S=24, F=32, G=5, H=14, W=14),       # two subtiles
])
def test_dense_bf16_gather_against_oracle(shape):
    from dau_conv import _capi
    N, S, F, G, H, W = (shape[q] for q in ("N", "S", "F", "G", "H", "W"))
    xb, dyb, w, mu1, mu2 = _case(41, N, S, F, G, H, W, 3.99)
    mu1.flat[0] = 3.99; mu2.flat[0] = -3.99; mu1.flat[1] = -4.0; mu2.flat[1] = 4.0       # the corners of the 9 x 9 kernel (and the +5 tap of weight 0 it leaves out)
    flags = _capi.FLAG_USE_INTERPOLATION | _capi.FLAG_IO_BF16 | _capi.FLAG_DENSE_BF16
    plan = _capi.Plan(N, S, F, G, H, W, max_kernel_size=9, sigma_hint=0.5, flags=flags)
    # three or more units: the parameter gradients take the dense form too
    assert plan.info["gather_dense_bf16"] == (2 if G >= 3 else 1)
    y, g = _run(plan, xb, dyb, w, mu1, mu2)
    _check(y, g, xb, dyb, w, mu1, mu2, "dense", dense_params=plan.info["gather_dense_bf16"] == 2)


@pytest.mark.parametrize("shape", [
    dict(N=2, S=16, F=32, G=6, H=56, W=56),       # two blocks of 30 columns, whole channel blocks
    dict(N=3, S=20, F=40, G=5, H=30, W=45),       # ragged channels, odd batch (one image pair without partner, 13 empty images)
    dict(N=17, S=7, F=5, G=8, H=9, W=6),          # two image chunks, the second almost empty; one block of 30 columns
    dict(N=2, S=33, F=290, G=2, H=28, W=28),      # ten 32-channel blocks of F = two workgroup groups; forced for two units
    dict(N=36, S=8, F=8, G=1, H=12, W=60),        # three image chunks (split), the widest row the form takes
    dict(N=4, S=40, F=24, G=9, H=14, W=31),
    # rows of more than 60 pixels are walked in segments of an instantiated length
    dict(N=2, S=8, F=8, G=3, H=6, W=112),         # 2 x 56
    dict(N=2, S=8, F=40, G=4, H=5, W=64),         # 5 x 14 (70 columns)
    dict(N=3, S=5, F=8, G=2, H=4, W=130),         # 5 x 28 (140)
])
@pytest.mark.parametrize("unit_testing", [False, True])
def test_dense_parameter_gradients_against_oracle(shape, unit_testing):
    """k_dense_wgrad.hip on its own: DAU_FLAG_DENSE_WGRAD_ALWAYS takes the dense correlations from one unit on."""
    from dau_conv import _capi
    N, S, F, G, H, W = (shape[q] for q in ("N", "S", "F", "G", "H", "W"))
    xb, dyb, w, mu1, mu2 = _case(47, N, S, F, G, H, W, 3.99)
    mu1.flat[0] = 3.99; mu2.flat[0] = -3.99; mu1.flat[1] = -4.0; mu2.flat[1] = 4.0       # the corners of the displacement range
    flags = _capi.FLAG_USE_INTERPOLATION | _capi.FLAG_IO_BF16 | _capi.FLAG_DENSE_BF16 | (_capi.FLAG_UNIT_TESTING if unit_testing else 0)
    plan = _capi.Plan(N, S, F, G, H, W, max_kernel_size=9, sigma_hint=0.5, flags=flags | _capi.FLAG_DENSE_WGRAD_ALWAYS)
    assert plan.info["gather_dense_bf16"] == 2
    dev = lambda a: torch.from_numpy(a).cuda()
    sigma = torch.full((1, S, G, F), 0.5, device="cuda")
    got = plan.backward(xb.cuda(), dyb.cuda(), dev(w), dev(mu1), dev(mu2), sigma)
    plan.check_status()
    want = orc.backward(xb.float().numpy(), dyb.float().numpy(), w, mu1, mu2, 0.5, unit_testing=unit_testing)
    for t, key in zip(got[1:], ("dw", "dmu1", "dmu2", "dsigma")):
        assert_parity(t.cpu().numpy(), want[key], key, rel=2e-2, floor=4e-3)
    # and against the exact gather-dot of the same plan shape (fp32 arithmetic on the same bf16 inputs)
    exact = _capi.Plan(N, S, F, G, H, W, max_kernel_size=9, sigma_hint=0.5, flags=flags | _capi.FLAG_DENSE_WGRAD_NEVER)
    assert exact.info["gather_dense_bf16"] == 1
    ref = exact.backward(xb.cuda(), dyb.cuda(), dev(w), dev(mu1), dev(mu2), sigma)
    for a, b, key in zip(got[1:], ref[1:], ("dw", "dmu1", "dmu2", "dsigma")):
        assert_parity(a.cpu().numpy(), b.cpu().numpy(), key + " (dense vs exact)", rel=2e-2, floor=4e-3)


@pytest.mark.parametrize("sigma", [0.2, 0.35, 0.7, 1.0])
@pytest.mark.parametrize("shape", [dict(N=3, S=20, F=40, G=4, H=30, W=45), dict(N=18, S=9, F=33, G=3, H=19, W=72)])
def test_dense_forms_under_other_prefilter_supports(shape, sigma):
    """The staging kernels of the dense forms are instantiated per prefilter support (3, 5, 7, 9 taps: dense_stage_rows_kernel,
    wg_filter_kernel); wider prefilters (sigma 1.0: 11 taps) take the general staging kernels.  y, dx and the dense parameter
    gradients at sigma 0.2 / 0.35 / 0.7 / 1.0 against the oracle at the bf16 bar; W = 45 takes the element-wise loads, W = 72 the
    16-byte ones with two column segments."""
    from dau_conv import _capi
    N, S, F, G, H, W = (shape[q] for q in ("N", "S", "F", "G", "H", "W"))
    xb, dyb, w, mu1, mu2 = _case(53, N, S, F, G, H, W, 3.99)
    flags = _capi.FLAG_USE_INTERPOLATION | _capi.FLAG_IO_BF16 | _capi.FLAG_DENSE_BF16
    plan = _capi.Plan(N, S, F, G, H, W, max_kernel_size=9, sigma_hint=sigma, flags=flags)
    assert plan.info["gather_dense_bf16"] == 2
    dev = lambda a: torch.from_numpy(a).cuda()
    sig = torch.full((1, S, G, F), sigma, device="cuda")
    y = plan.forward(xb.cuda(), dev(w), dev(mu1), dev(mu2), sig)
    g = plan.backward(xb.cuda(), dyb.cuda(), dev(w), dev(mu1), dev(mu2), sig)
    plan.check_status()
    x32, dy32 = xb.float().numpy(), dyb.float().numpy()
    assert_parity(y.float().cpu().numpy(), orc.forward(x32, w, mu1, mu2, sigma), "y sigma %g" % sigma, rel=2e-2, floor=4e-3)
    want = orc.backward(x32, dy32, w, mu1, mu2, sigma)
    assert_parity(g[0].float().cpu().numpy(), want["dx"], "dx sigma %g" % sigma, rel=2e-2, floor=4e-3)
    for t, key in zip(g[1:], ("dw", "dmu1", "dmu2", "dsigma")):
        assert_parity(t.cpu().numpy(), want[key], "%s sigma %g" % (key, sigma), rel=2e-2, floor=1e-2)


@pytest.mark.parametrize("shape", [
    dict(N=2, S=16, F=128, G=4, H=56, W=56),
    dict(N=3, S=20, F=40, G=3, H=30, W=45),
    dict(N=17, S=7, F=5, G=8, H=9, W=6),
    dict(N=2, S=33, F=130, G=6, H=28, W=28),
    dict(N=2, S=8, F=8, G=3, H=6, W=112),
    dict(N=18, S=9, F=33, G=3, H=19, W=72),
])
@pytest.mark.parametrize("unit_testing", [False, True])
def test_dense_forms_of_radius_three(shape, unit_testing):
    """Calls whose offsets lie within +-3 take the 7 x 7 members of the dense forms (49 taps / displacements instead of 81; the
    device guard of the call decides between (-1, 3] and (3, 4]).  Offsets up to exactly +-3.0 -- the corners of the 7 x 7 kernel
    and the +4 tap of weight 0 it leaves out -- against the oracle, y, dx and the dense parameter gradients; then the same plan
    with ONE offset moved to 3.5 (the 9 x 9 members) and back."""
    from dau_conv import _capi
    N, S, F, G, H, W = (shape[q] for q in ("N", "S", "F", "G", "H", "W"))
    xb, dyb, w, mu1, mu2 = _case(61, N, S, F, G, H, W, 3.0)
    mu1.flat[0] = 3.0; mu2.flat[0] = -3.0; mu1.flat[1] = -3.0; mu2.flat[1] = 3.0
    flags = _capi.FLAG_USE_INTERPOLATION | _capi.FLAG_IO_BF16 | _capi.FLAG_DENSE_BF16 | (_capi.FLAG_UNIT_TESTING if unit_testing else 0)
    plan = _capi.Plan(N, S, F, G, H, W, max_kernel_size=9, sigma_hint=0.5, flags=flags | _capi.FLAG_DENSE_WGRAD_ALWAYS)
    assert plan.info["gather_dense_bf16"] == 2
    dev = lambda a: torch.from_numpy(a).cuda()
    sigma = torch.full((1, S, G, F), 0.5, device="cuda")
    x32, dy32 = xb.float().numpy(), dyb.float().numpy()

    def run_and_check(m1, m2, tag):
        y = plan.forward(xb.cuda(), dev(w), dev(m1), dev(m2), sigma)
        g = plan.backward(xb.cuda(), dyb.cuda(), dev(w), dev(m1), dev(m2), sigma)
        plan.check_status()
        assert_parity(y.float().cpu().numpy(), orc.forward(x32, w, m1, m2, 0.5), tag + "/y", rel=2e-2, floor=4e-3)
        want = orc.backward(x32, dy32, w, m1, m2, 0.5, unit_testing=unit_testing)
        assert_parity(g[0].float().cpu().numpy(), want["dx"], tag + "/dx", rel=2e-2, floor=4e-3)
        for t, key in zip(g[1:], ("dw", "dmu1", "dmu2", "dsigma")):
            assert_parity(t.cpu().numpy(), want[key], tag + "/" + key, rel=2e-2, floor=4e-3)
        return y, g

    y3, g3 = run_and_check(mu1, mu2, "r3")
    far = mu1.copy(); far.flat[2] = 3.5
    run_and_check(far, mu2, "r4")
    y3b, g3b = run_and_check(mu1, mu2, "r3 again")
    assert torch.equal(y3, y3b)
    for a, b in zip(g3, g3b):
        assert torch.equal(a, b)


def test_dense_parameter_gradients_without_the_sigma_kind():
    """A call that does not want dsigma (the layer's default: dau_sigma_trainable=False; the reference's last_k_optional,
    dau_conv_backward.cpp:219) runs three of the four kinds of GEMMs: dw, dmu1, dmu2 are bit-identical to the call that wants
    all five, dsigma is not produced, and the next full call is unaffected."""
    from dau_conv import _capi
    N, S, F, G, H, W = 18, 33, 40, 4, 30, 45
    xb, dyb, w, mu1, mu2 = _case(59, N, S, F, G, H, W, 3.99)
    flags = _capi.FLAG_USE_INTERPOLATION | _capi.FLAG_IO_BF16 | _capi.FLAG_DENSE_BF16
    plan = _capi.Plan(N, S, F, G, H, W, max_kernel_size=9, sigma_hint=0.5, flags=flags)
    assert plan.info["gather_dense_bf16"] == 2
    dev = lambda a: torch.from_numpy(a).cuda()
    sigma = torch.full((1, S, G, F), 0.5, device="cuda")
    args = (xb.cuda(), dyb.cuda(), dev(w), dev(mu1), dev(mu2), sigma)
    full = plan.backward(*args)
    part = plan.backward(*args, need_mask=_capi.NEED_ALL & ~_capi.NEED_DSIGMA)
    again = plan.backward(*args)
    plan.check_status()
    assert part[4] is None
    for a, b, c, key in zip(full[:4], part[:4], again[:4], ("dx", "dw", "dmu1", "dmu2")):
        assert torch.equal(a, b) and torch.equal(a, c), key
    assert torch.equal(full[4], again[4])
    want = orc.backward(xb.float().numpy(), dyb.float().numpy(), w, mu1, mu2, 0.5)
    for t, key in zip(part[1:4], ("dw", "dmu1", "dmu2")):
        assert_parity(t.cpu().numpy(), want[key], key, rel=2e-2, floor=4e-3)


def test_dense_bf16_under_a_larger_kernel_follows_the_offsets():
    """max_kernel_size 17 with the dense flag: calls with |mu| <= 4 take the dense GEMM (after the first call has left its
    hint), calls with larger offsets the exact gather of bucket 8; both match the oracle."""
    from dau_conv import _capi
    N, S, F, G, H, W = 3, 18, 36, 4, 33, 40
    flags = _capi.FLAG_USE_INTERPOLATION | _capi.FLAG_IO_BF16 | _capi.FLAG_DENSE_BF16
    plan = _capi.Plan(N, S, F, G, H, W, max_kernel_size=17, sigma_hint=0.5, flags=flags)
    assert plan.info["gather_dense_bf16"] == 2 and plan.info["bucket_sets"] == 2
    small = _case(42, N, S, F, G, H, W, 3.5)
    y, g = _run(plan, *small, calls=2)
    _check(y, g, *small, "small offsets (dense)", dense_params=True)
    big = _case(43, N, S, F, G, H, W, 7.5)
    y, g = _run(plan, *big)                        # stale hint (3.5): the guard sends the call to the bucket-8 gather
    _check(y, g, *big, "large offsets (gather)")
    y, g = _run(plan, *big)
    _check(y, g, *big, "large offsets again")


def test_dense_arithmetic_depends_on_the_offsets_only_not_on_the_hint():
    """With DAU_FLAG_DENSE_BF16 the bucket-4 member is the one kernel set whose arithmetic differs (bf16 products).  It is
    enqueued as a guarded candidate on EVERY call, so the first call of a plan (no hint), a hinted call and a call after a
    stale hint (larger offsets in between) give the same bits for the same inputs."""
    from dau_conv import _capi
    N, S, F, G, H, W = 3, 10, 24, 4, 20, 26
    flags = _capi.FLAG_USE_INTERPOLATION | _capi.FLAG_IO_BF16 | _capi.FLAG_DENSE_BF16
    plan = _capi.Plan(N, S, F, G, H, W, max_kernel_size=33, sigma_hint=0.5, flags=flags)
    assert plan.info["gather_dense_bf16"] == 2 and plan.info["bucket_sets"] == 3
    small = _case(52, N, S, F, G, H, W, 3.5)
    y1, g1 = _run(plan, *small)                            # no hint yet
    y2, g2 = _run(plan, *small)                            # hinted: bucket 4
    mid = _case(53, N, S, F, G, H, W, 7.0)
    ym, gm = _run(plan, *mid)                              # bucket 8 (exact fp32 gather), leaves a hint of 7
    _check(ym, gm, *mid, "bucket 8 between two dense calls")
    y3, g3 = _run(plan, *small)                            # stale hint of 7: dense, bucket-8 and static candidates, dense runs
    _check(y1, g1, *small, "first call (dense)", dense_params=True)
    for a, b, c, name in zip((y1,) + tuple(g1), (y2,) + tuple(g2), (y3,) + tuple(g3), ("y", "dx", "dw", "dmu1", "dmu2", "dsigma")):
        assert torch.equal(a, b) and torch.equal(a, c), name
    static = _capi.Plan(N, S, F, G, H, W, max_kernel_size=33, sigma_hint=0.5, flags=flags | _capi.FLAG_STATIC_BUCKET)
    assert static.info["gather_dense_bf16"] == 0           # the dense member is unreachable: not reported


def test_dense_flag_needs_bf16_io():
    from dau_conv import _capi
    with pytest.raises(_capi.InvalidArgumentError):
        _capi.Plan(2, 4, 8, 2, 16, 16, flags=_capi.FLAG_USE_INTERPOLATION | _capi.FLAG_DENSE_BF16)


def test_dense_bf16_through_the_layer():
    import dau_conv
    torch.manual_seed(1)
    kw = dict(filters=24, dau_units=(2, 2), max_kernel_size=9, in_channels=10, use_bias=False)
    exact = dau_conv.DAUConv2d(**kw).cuda()
    dense = dau_conv.DAUConv2d(dense_bf16=True, **kw).cuda()
    dense.load_state_dict(exact.state_dict())
    x = torch.rand(3, 10, 21, 37, device="cuda").to(torch.bfloat16)
    xa, xb = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
    ya, yb = exact(xa), dense(xb)
    assert yb.dtype == torch.bfloat16
    dy = torch.randn_like(ya)
    ya.backward(dy); yb.backward(dy)
    scale = float(ya.float().abs().max())
    assert float((ya.float() - yb.float()).abs().max()) <= 2e-2 * scale
    gs = float(xa.grad.float().abs().max())
    assert float((xa.grad.float() - xb.grad.float()).abs().max()) <= 2e-2 * gs
    # four units: the parameter gradients take the dense correlations as well (bf16 bar against the exact gather-dot)
    for pa, pb in ((exact.weights, dense.weights), (exact.mu1, dense.mu1), (exact.mu2, dense.mu2), (exact.sigma, dense.sigma)):
        if pa.grad is None:
            continue
        ps = float(pa.grad.abs().max())
        assert float((pa.grad - pb.grad).abs().max()) <= 2e-2 * ps
