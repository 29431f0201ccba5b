"""GPU parity tests: the HIP path (through the C ABI) against the golden vectors of the
reference's numpy oracle and against the C oracle on seeded inputs.

Tolerance (north star): 1e-4 relative, fp32, with an absolute floor of 1e-6 of the tensor's
max-norm; per-unit offset/fraction bookkeeping is bit-exact.
The reference's own tests drop the last output column before comparing
(dau_conv_test.py:398-404); this implementation is exact there, so nothing is dropped.
"""
import numpy as np
import pytest
import torch

from oracle import dau_oracle as orc
from util import assert_parity, case_kernel_size, golden_cases, load_case, tuning_capi

pytestmark = pytest.mark.gpu

ALGOS = {"direct": 1, "auto": 0}


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _flags(c, capi):
    f = 0
    if int(c["use_interpolation"]): f |= capi.FLAG_USE_INTERPOLATION
    if int(c["unit_testing"]): f |= capi.FLAG_UNIT_TESTING
    if int(c["single_dim_kernel"]): f |= capi.FLAG_SINGLE_DIM_KERNEL
    if int(c["forbid_positive_dim1"]): f |= capi.FLAG_FORBID_POSITIVE_DIM1
    return f


def _run_case(c, algo, lr=1.0):
    from dau_conv import _capi
    N, S, H, W = c["x"].shape
    _, _, G, F = c["w"].shape
    plan = _capi.Plan(N, S, F, G, H, W, max_kernel_size=case_kernel_size(c), number_units_ignore=int(c["ignore"]),
                      flags=_flags(c, _capi), algo=algo, sigma_hint=float(c["sigma"]), mu_learning_rate_factor=lr)
    x, w, mu1, mu2, dy = (_dev(c[k]) for k in ("x", "w", "mu1", "mu2", "dy"))
    sigma = torch.full((1, S, G, F), float(c["sigma"]), device="cuda")
    y = plan.forward(x, w, mu1, mu2, sigma)
    plan.check_status()
    dx, dw, dmu1, dmu2, dsigma = plan.backward(x, dy, w, mu1, mu2, sigma)
    plan.check_status()
    torch.cuda.synchronize()
    out = dict(y=y, dx=dx, dw=dw, dmu1=dmu1, dmu2=dmu2, dsigma=dsigma)
    return plan, {k: v.cpu().numpy() for k, v in out.items()}


@pytest.mark.parametrize("algo", sorted(ALGOS))
@pytest.mark.parametrize("name", golden_cases())
def test_golden_vectors(name, algo):
    """Every golden case of the reference oracle, forward + all five gradients."""
    c = load_case(name)
    _, got = _run_case(c, ALGOS[algo])
    for key in ("y", "dx", "dw", "dmu1", "dmu2", "dsigma"):
        # golden values carry the numpy oracle's own float32 accumulation error (~1e-6 of max-norm)
        assert_parity(got[key], c[key], "%s/%s/%s" % (name, algo, key), rel=1e-4, floor=3e-6)


def test_tiled_kernels_are_selected_for_benchmark_shapes():
    """AUTO must pick the LDS-tiled MFMA gather for the shapes the benchmark runs (no silent slow path)."""
    from dau_conv import _capi
    for (H, W, k) in ((56, 56, 9), (32, 32, 9), (32, 32, 17), (16, 16, 9), (8, 8, 9), (27, 27, 9), (28, 28, 9), (24, 24, 9),
                      # patch decomposition: feature maps of ResNet / CIFAR stages, odd sizes, large images, kernels 17 and 33
                      (14, 14, 9), (7, 7, 9), (64, 64, 9), (112, 112, 9), (8, 65, 9), (90, 100, 9), (224, 224, 9), (56, 56, 17),
                      (64, 64, 33), (128, 96, 17), (512, 512, 33),
                      # kernel 65: four offset-window passes of the kernel-33 gather
                      (64, 64, 65), (40, 100, 65)):
        info = _capi.Plan(2, 4, 8, 2, H, W, max_kernel_size=k).info
        assert info["algo_forward"] == _capi.ALGO_TILED, (H, W, k, info)
    # gather-dot: any image size, any unit count, kernels 9 and 17
    for (H, W, k, G) in ((56, 56, 9, 4), (27, 27, 9, 4), (65, 8, 9, 2), (32, 32, 17, 6), (56, 56, 17, 8), (100, 90, 9, 1),
                         (64, 64, 33, 4), (64, 64, 65, 9), (512, 512, 33, 9), (16, 2000, 9, 2)):
        info = _capi.Plan(2, 4, 8, G, H, W, max_kernel_size=k).info
        assert info["algo_backward"] == _capi.ALGO_TILED, (H, W, k, G, info)


def test_mu_learning_rate_factor_and_need_mask():
    """dmu1/dmu2 are scaled inside the op (dau_conv_grad_op.cpp:297-303); skipped outputs stay None."""
    from dau_conv import _capi
    c = load_case("k9_16x16_g4")
    _, base = _run_case(c, 0, lr=1.0)
    _, got = _run_case(c, 0, lr=1000.0)
    assert_parity(got["dmu1"], base["dmu1"] * 1000.0, "dmu1*lr", rel=1e-6, floor=1e-7)
    assert_parity(got["dmu2"], base["dmu2"] * 1000.0, "dmu2*lr", rel=1e-6, floor=1e-7)
    assert_parity(got["dw"], base["dw"], "dw", rel=0, floor=0)
    assert_parity(got["dsigma"], base["dsigma"], "dsigma", rel=0, floor=0)
    N, S, H, W = c["x"].shape
    _, _, G, F = c["w"].shape
    plan = _capi.Plan(N, S, F, G, H, W, flags=_capi.FLAG_USE_INTERPOLATION | _capi.FLAG_UNIT_TESTING)
    sigma = torch.full((1, S, G, F), 0.5, device="cuda")
    out = plan.backward(_dev(c["x"]), _dev(c["dy"]), _dev(c["w"]), _dev(c["mu1"]), _dev(c["mu2"]), sigma,
                        need_mask=_capi.NEED_DX | _capi.NEED_DW)
    assert out[2] is None and out[3] is None and out[4] is None
    assert_parity(out[0].cpu().numpy(), base["dx"], "dx only", rel=0, floor=0)
    assert_parity(out[1].cpu().numpy(), base["dw"], "dw only", rel=1e-6, floor=1e-7)


def test_unit_bookkeeping_bit_exact():
    """floor(mu), fractions and the four bilinear factors are bit-identical to the oracle -- in THE table the gather kernels
    consume (dau_conv_unit_table runs prepare_units_kernel, the first kernel of every call, into the caller's buffer), in
    all three forms a step uses: [S][G][F] with bare factors (parameter gradients, dau_conv_backward_core.hpp:2078-2081),
    [S][G][F] premultiplied by w (forward, dau_conv_forward_core.hpp:2025-2028,2135-2213) and [F][G][S] with negated offsets
    premultiplied by w (input gradient, base_dau_conv_layer.cu:299-325); an ignored unit has zero factors."""
    from dau_conv import _capi
    rs = np.random.RandomState(3)
    S, G, F, ignore = 5, 4, 24, 1
    mu1 = rs.uniform(-8, 8, (1, S, G, F)).astype(np.float32)
    mu2 = rs.uniform(-8, 8, (1, S, G, F)).astype(np.float32)
    mu1.flat[:6] = [-8.0, 8.0, 7.99, -7.99, -0.0, 3.0]
    mu2.flat[:6] = [8.0, -8.0, -7.99, 7.99, 1e-8, -3.0]
    w = (rs.randn(1, S, G, F) * 0.1).astype(np.float32)
    live = (np.arange(G) < G - ignore).astype(np.float32).reshape(1, 1, G, 1)
    bits = lambda a: np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)
    for interp in (True, False):
        plan = _capi.Plan(1, S, F, G, 8, 8, max_kernel_size=17, number_units_ignore=ignore,
                          flags=_capi.FLAG_USE_INTERPOLATION if interp else 0)
        eoff, efac = orc.unit_table(mu1, mu2, use_interpolation=interp)          # [S*G*F] in SGF order
        # parameter-gradient pass: bare factors (ignored units: zero)
        off, fac = plan.unit_table(_dev(mu1), _dev(mu2))
        assert np.array_equal(off.cpu().numpy(), eoff)
        want = efac.reshape(S, G, F, 4) * live.reshape(1, G, 1, 1)
        assert np.array_equal(bits(fac.cpu().numpy()), bits(want.reshape(-1, 4)))
        # forward pass: premultiplied by w
        off, fac = plan.unit_table(_dev(mu1), _dev(mu2), w=_dev(w))
        assert np.array_equal(off.cpu().numpy(), eoff)
        wl = np.where(live > 0, w, np.float32(0.0)).reshape(S, G, F, 1).astype(np.float32)   # an ignored unit's weight is +0
        want = wl * efac.reshape(S, G, F, 4)                                      # one fp32 multiply, as the kernel does
        assert np.array_equal(bits(fac.cpu().numpy()), bits(want.reshape(-1, 4)))
        # input-gradient pass: [F][G][S] order, offsets negated (floor(-mu), its own fractions), premultiplied by w; every
        # unit is live here (the reference transposes the zero weights of ignored units along, :299-325)
        noff, nfac = orc.unit_table(-mu1, -mu2, use_interpolation=interp)
        off, fac = plan.unit_table(_dev(mu1), _dev(mu2), w=_dev(w), form=1)
        t = lambda a, c: np.ascontiguousarray(a.reshape(S, G, F, c).transpose(2, 1, 0, 3)).reshape(-1, c)
        assert np.array_equal(off.cpu().numpy(), t(noff, 2))
        want = w.reshape(S, G, F, 1).astype(np.float32) * nfac.reshape(S, G, F, 4)
        assert np.array_equal(bits(fac.cpu().numpy()), bits(t(want, 4)))


@pytest.mark.parametrize("sigma,sd,fp", [(0.5, 0, 0), (0.8, 0, 0), (1.2, 0, 0), (0.5, 1, 0), (0.5, 1, 1)])
def test_filter_synthesis(sigma, sd, fp):
    from dau_conv import _capi
    flags = _capi.FLAG_USE_INTERPOLATION | (_capi.FLAG_SINGLE_DIM_KERNEL if sd else 0) | (_capi.FLAG_FORBID_POSITIVE_DIM1 if fp else 0)
    plan = _capi.Plan(1, 1, 1, 2, 8, 8, flags=flags, sigma_hint=sigma)
    k = plan.info["blur_support"]
    assert k == orc.filter_support(sigma)
    got = plan.filters(torch.full((1, 1, 2, 1), sigma, device="cuda")).cpu().numpy()
    want = orc.filters(sigma, k=k, single_dim_kernel=sd, forbid_positive_dim1=fp)
    for i, name in enumerate(("Gn", "Dw", "Dmu1", "Dmu2", "Dsigma", "Gerr")):
        assert_parity(got[i], want[name], name, rel=1e-6, floor=1e-7)


def test_zero_sized_dimensions_are_rejected():
    from dau_conv import _capi
    for bad in (dict(N=0), dict(S=0), dict(F=0), dict(G=0), dict(H=0), dict(W=0)):
        kw = dict(N=2, S=3, F=4, G=2, H=8, W=8); kw.update(bad)
        with pytest.raises(_capi.InvalidArgumentError):
            _capi.Plan(kw["N"], kw["S"], kw["F"], kw["G"], kw["H"], kw["W"])


def test_error_convention():
    """NaN offsets -> FAILED_PRECONDITION, offsets beyond the kernel -> INVALID_ARGUMENT
    (dau_conv_op.cpp:250-262); wrong shapes are rejected before any launch."""
    from dau_conv import _capi
    N, S, F, G, H, W = 1, 2, 4, 2, 8, 8
    plan = _capi.Plan(N, S, F, G, H, W, max_kernel_size=9)
    x = torch.rand(N, S, H, W, device="cuda")
    w = torch.randn(1, S, G, F, device="cuda")
    mu = torch.zeros(1, S, G, F, device="cuda")
    sigma = torch.full((1, S, G, F), 0.5, device="cuda")
    plan.forward(x, w, mu, mu, sigma)
    assert plan.check_status() == 0.0
    bad = mu.clone(); bad[0, 1, 1, 2] = float("nan")
    plan.forward(x, w, bad, mu, sigma)
    with pytest.raises(_capi.FailedPreconditionError):
        plan.check_status()
    far = mu.clone(); far[0, 0, 0, 0] = 6.5
    y = plan.forward(x, w, mu, far, sigma)
    with pytest.raises(_capi.InvalidArgumentError):
        plan.check_status()
    assert torch.isfinite(y).all()
    with pytest.raises(_capi.InvalidArgumentError):
        plan.forward(x[:, :1], w, mu, mu, sigma)
    with pytest.raises(_capi.InvalidArgumentError):
        plan.forward(x, w[..., :2], mu, mu, sigma)


@pytest.mark.parametrize("shape", [
    # shape matrix of the reference tests (dau_conv_test.py:418-465), channel counts reduced so the
    # CPU oracle finishes in seconds: patch splitting W=65/H=8, small batches, 9/17 kernels, odd S,
    # 6x6 images, large kernels
    dict(N=2, W=65, H=8, S=33, F=32, G=2, k=9, m=3),
    dict(N=1, W=8, H=8, S=32, F=32, G=2, k=9, m=3),
    dict(N=4, W=32, H=32, S=8, F=32, G=4, k=9, m=3),
    dict(N=4, W=32, H=32, S=8, F=32, G=4, k=17, m=6),
    dict(N=4, W=32, H=32, S=3, F=32, G=4, k=17, m=3),
    dict(N=4, W=6, H=6, S=16, F=64, G=2, k=17, m=8),
    dict(N=2, W=64, H=64, S=3, F=32, G=4, k=33, m=10),
    dict(N=2, W=64, H=64, S=4, F=16, G=4, k=65, m=20),
    # shapes of the benchmark configs at reduced N / channels
    dict(N=2, W=27, H=27, S=12, F=32, G=4, k=9, m=3),
    dict(N=3, W=56, H=56, S=8, F=32, G=4, k=9, m=3),
    dict(N=2, W=28, H=28, S=16, F=16, G=4, k=9, m=3),
    # patch decomposition of the tiled gather: 2x2 patches of 32 and of 56, odd sizes, one ragged patch row
    dict(N=3, W=64, H=64, S=5, F=12, G=4, k=9, m=3),
    dict(N=2, W=112, H=112, S=3, F=8, G=2, k=9, m=3),
    dict(N=2, W=50, H=40, S=4, F=8, G=4, k=9, m=3),
    dict(N=3, W=14, H=14, S=8, F=16, G=4, k=9, m=3),
    dict(N=5, W=7, H=7, S=8, F=16, G=2, k=9, m=3),
    dict(N=2, W=75, H=33, S=3, F=8, G=2, k=17, m=7),
    # offset windows of the gather-dot for kernels 33 and 65 with offsets over the whole range
    dict(N=2, W=40, H=40, S=5, F=33, G=3, k=33, m=16),
    dict(N=1, W=48, H=40, S=3, F=8, G=2, k=65, m=32),
    # very wide maps: the error rows are transposed in chunks of 512 padded columns
    dict(N=1, W=700, H=9, S=2, F=3, G=2, k=9, m=3),
    dict(N=2, W=1100, H=8, S=1, F=2, G=1, k=17, m=7),
    # degenerate sizes: one pixel, one channel, one unit; a 2x3 image
    dict(N=1, W=1, H=1, S=1, F=1, G=1, k=9, m=3),
    dict(N=3, W=3, H=2, S=2, F=3, G=2, k=9, m=3),
    dict(N=1, W=9, H=1, S=3, F=2, G=4, k=17, m=7),
    # gather-dot passes: 5 = 4 + 1, 7 = 4 + 3 (two four-unit blocks), 9 = 8 + 1, 10 = 8 + 2 units; odd channel counts
    dict(N=2, W=16, H=24, S=35, F=33, G=5, k=9, m=3),
    dict(N=2, W=16, H=16, S=9, F=40, G=7, k=9, m=3),
    dict(N=1, W=24, H=16, S=66, F=20, G=9, k=9, m=3),
    dict(N=3, W=16, H=16, S=5, F=32, G=10, k=17, m=7),
    # unit counts of the reference's dau_units (1x1 .. 4x2): one, odd, six and eight units per channel
    dict(N=2, W=16, H=16, S=5, F=40, G=1, k=9, m=3),
    dict(N=2, W=24, H=24, S=6, F=32, G=3, k=9, m=3),
    dict(N=3, W=32, H=32, S=9, F=32, G=6, k=9, m=3),
    dict(N=2, W=16, H=16, S=33, F=48, G=8, k=9, m=3),
    # kernel 17 on a 56x56 map (single-tile gather-dot, R = 8 gather-sum variant) and with six units
    dict(N=3, W=56, H=56, S=6, F=32, G=4, k=17, m=7),
    dict(N=2, W=32, H=32, S=17, F=32, G=6, k=17, m=7),
])
@pytest.mark.parametrize("algo", sorted(ALGOS))
def test_seeded_shapes_against_c_oracle(shape, algo):
    from dau_conv import _capi
    rs = np.random.RandomState(7)
    N, S, F, G, H, W, k, m = (shape[q] for q in ("N", "S", "F", "G", "H", "W", "k", "m"))
    # distributions of dau_conv_test.py:342-368
    x = rs.rand(N, S, H, W).astype(np.float32)
    w = (rs.randn(1, S, G, F) * 0.1).astype(np.float32)
    lim = k // 2 - 0.01  # the layer clips mu to +-(floor(k/2) - border) (dau_conv.py:183,190-191)
    mu1 = np.clip(rs.uniform(-m, m, (1, S, G, F)), -lim, lim).astype(np.float32)
    mu2 = np.clip(rs.uniform(-m, m, (1, S, G, F)), -lim, lim).astype(np.float32)
    dy = rs.randn(N, F, H, W).astype(np.float32)
    lr = 1000.0
    plan = _capi.Plan(N, S, F, G, H, W, max_kernel_size=k, flags=_capi.FLAG_USE_INTERPOLATION | _capi.FLAG_UNIT_TESTING,
                      algo=ALGOS[algo], sigma_hint=0.5, mu_learning_rate_factor=lr)
    sigma = torch.full((1, S, G, F), 0.5, device="cuda")
    y = plan.forward(_dev(x), _dev(w), _dev(mu1), _dev(mu2), sigma)
    got = plan.backward(_dev(x), _dev(dy), _dev(w), _dev(mu1), _dev(mu2), sigma)
    plan.check_status()
    want_y = orc.forward(x, w, mu1, mu2, 0.5)
    want = orc.backward(x, dy, w, mu1, mu2, 0.5, unit_testing=True, mu_learning_rate_factor=lr)
    assert_parity(y.cpu().numpy(), want_y, "y")
    for t, key in zip(got, ("dx", "dw", "dmu1", "dmu2", "dsigma")):
        assert_parity(t.cpu().numpy(), want[key], key)


@pytest.mark.parametrize("shape", [
    # small feature maps with several (image pair, patch) planes stacked per workgroup; N chosen so that the last
    # group is short (planes % stack != 0) and N is odd (zero image in the last pair)
    dict(N=5, W=32, H=32, S=5, F=12, G=4, k=9, m=3, variant=8, stack=2, patch=32),
    dict(N=7, W=24, H=24, S=4, F=8, G=2, k=9, m=3, variant=9, stack=4, patch=24),
    dict(N=9, W=16, H=16, S=6, F=10, G=3, k=9, m=3, variant=10, stack=4, patch=16),    # 8 channels per workgroup
    dict(N=21, W=8, H=8, S=8, F=20, G=4, k=9, m=3, variant=11, stack=4, patch=8),      # 16 channels per workgroup
    dict(N=7, W=28, H=28, S=5, F=8, G=4, k=9, m=3, variant=12, stack=3, patch=32),     # whole 28x28 images
    dict(N=3, W=27, H=27, S=3, F=8, G=6, k=9, m=3, variant=12, stack=3, patch=32),
    dict(N=3, W=40, H=20, S=3, F=8, G=4, k=9, m=3, variant=9, stack=4, patch=24),      # two patch columns, stacked across patches and pairs
    dict(N=5, W=21, H=19, S=4, F=12, G=4, k=9, m=3, variant=13, stack=2, patch=24),    # whole images, no edge tiles
    dict(N=9, W=14, H=14, S=6, F=10, G=3, k=9, m=3, variant=14, stack=4, patch=16),    # 14x14 maps
    dict(N=11, W=9, H=15, S=3, F=9, G=2, k=9, m=3, variant=14, stack=4, patch=16),
    dict(N=21, W=7, H=7, S=8, F=20, G=4, k=9, m=3, variant=15, stack=8, patch=8),      # 7x7 maps
    dict(N=3, W=5, H=6, S=8, F=33, G=5, k=9, m=3, variant=15, stack=8, patch=8),
    dict(N=4, W=100, H=60, S=3, F=8, G=4, k=9, m=3, variant=8, stack=2, patch=32),     # a large image cut into stacked patches
    # edge-free 31 pixel patches for large offsets (rows 16, 17): buckets 16 / 20 / 24 / 32 without offset windows, and
    # bucket 32 in windows of radius 16; images that are not a multiple of the patch, odd channel counts
    dict(N=2, W=33, H=31, S=3, F=8, G=2, k=33, m=15, variant=16, stack=1, patch=32),
    dict(N=3, W=70, H=45, S=5, F=12, G=4, k=41, m=19.5, variant=16, stack=1, patch=32),
    dict(N=2, W=62, H=64, S=4, F=16, G=5, k=65, m=31, variant=16, stack=1, patch=32),       # windows of radius 16
    dict(N=2, W=64, H=64, S=4, F=16, G=5, k=49, m=23.5, variant=17, stack=1, patch=32),
    dict(N=3, W=40, H=90, S=3, F=9, G=3, k=65, m=31.5, variant=17, stack=1, patch=32),      # bucket 32 in one pass
    dict(N=2, W=100, H=37, S=2, F=20, G=9, k=65, m=31.5, variant=17, stack=1, patch=32),    # nine units: windows
    # three plane buffers, partner waves one unit behind (row 20): every unit-count class of its group logic
    dict(N=4, W=56, H=56, S=9, F=12, G=4, k=9, m=3, variant=20, stack=1, patch=56),
    dict(N=3, W=50, H=41, S=5, F=7, G=1, k=9, m=3, variant=20, stack=1, patch=56),
    dict(N=2, W=56, H=56, S=4, F=8, G=2, k=9, m=3, variant=20, stack=1, patch=56),
    dict(N=2, W=56, H=56, S=3, F=9, G=3, k=9, m=3, variant=20, stack=1, patch=56),
    dict(N=2, W=80, H=56, S=1, F=8, G=6, k=9, m=3, variant=20, stack=1, patch=56),
    dict(N=2, W=56, H=56, S=2, F=4, G=9, k=9, m=3, variant=20, stack=1, patch=56),
    # twelve output channels per workgroup on one 25..31 pixel image (row 21): whole and partial channel blocks
    dict(N=3, W=27, H=27, S=5, F=24, G=4, k=9, m=3, variant=21, stack=1, patch=32),
    dict(N=4, W=28, H=30, S=3, F=17, G=3, k=9, m=3, variant=21, stack=1, patch=32),
    dict(N=2, W=64, H=40, S=4, F=26, G=5, k=49, m=23.5, variant=22, stack=1, patch=32),  # twelve channels, 31 pixel patches
    dict(N=2, W=70, H=40, S=4, F=26, G=5, k=37, m=17.9, variant=23, stack=1, patch=32),  # twelve channels on a pitch-72 plane (buckets <= 20)
    dict(N=3, W=33, H=64, S=3, F=12, G=9, k=41, m=19.9, variant=23, stack=1, patch=32),
    # tiles of 32 x 2 positions: whole 28 / 27 pixel images in 15 / 14 tiles; larger images in patches of 31 x 29 / 31 x 27
    dict(N=3, W=28, H=28, S=5, F=24, G=4, k=9, m=3, variant=24, stack=1, patch=32),
    dict(N=4, W=29, H=26, S=3, F=17, G=3, k=9, m=3, variant=24, stack=1, patch=32),
    dict(N=3, W=27, H=27, S=5, F=13, G=6, k=9, m=3, variant=25, stack=1, patch=32),
    dict(N=2, W=70, H=61, S=3, F=12, G=2, k=9, m=3, variant=24, stack=1, patch=32),
    dict(N=2, W=31, H=30, S=2, F=25, G=4, k=9, m=3, variant=25, stack=1, patch=32),
    # the twelve-channel rows with lagged partner waves (three plane buffers)
    dict(N=3, W=28, H=28, S=5, F=24, G=4, k=9, m=3, variant=26, stack=1, patch=32),
    dict(N=3, W=27, H=27, S=4, F=13, G=1, k=9, m=3, variant=26, stack=1, patch=32),
    dict(N=2, W=70, H=40, S=4, F=26, G=5, k=37, m=17.9, variant=27, stack=1, patch=32),
    dict(N=3, W=33, H=64, S=3, F=12, G=9, k=41, m=19.9, variant=27, stack=1, patch=32),
])
def test_stacked_gather_variants(shape, monkeypatch):
    # the tuning build of the same sources: only it reads DAU_GATHER_VARIANT (at plan creation); the release library picks these
    # rows by itself for the large batches they were made for
    _capi = tuning_capi()
    monkeypatch.setenv("DAU_GATHER_VARIANT", str(shape["variant"]))
    rs = np.random.RandomState(11)
    N, S, F, G, H, W, k, m = (shape[q] for q in ("N", "S", "F", "G", "H", "W", "k", "m"))
    x = rs.rand(N, S, H, W).astype(np.float32)
    w = (rs.randn(1, S, G, F) * 0.1).astype(np.float32)
    lim = k // 2 - 0.01
    mu1 = np.clip(rs.uniform(-m, m, (1, S, G, F)), -lim, lim).astype(np.float32)
    mu2 = np.clip(rs.uniform(-m, m, (1, S, G, F)), -lim, lim).astype(np.float32)
    dy = rs.randn(N, F, H, W).astype(np.float32)
    plan = _capi.Plan(N, S, F, G, H, W, max_kernel_size=k, sigma_hint=0.5)
    assert plan.info["algo_forward"] == _capi.ALGO_TILED and plan.info["gather_variant"] == shape["variant"]
    assert (plan.info["gather_stack"], plan.info["gather_patch"]) == (shape["stack"], shape["patch"]), plan.info
    sigma = torch.full((1, S, G, F), 0.5, device="cuda")
    y = plan.forward(_dev(x), _dev(w), _dev(mu1), _dev(mu2), sigma)
    got = plan.backward(_dev(x), _dev(dy), _dev(w), _dev(mu1), _dev(mu2), sigma)
    plan.check_status()
    assert_parity(y.cpu().numpy(), orc.forward(x, w, mu1, mu2, 0.5), "y")
    want = orc.backward(x, dy, w, mu1, mu2, 0.5)
    for t, key in zip(got, ("dx", "dw", "dmu1", "dmu2", "dsigma")):
        assert_parity(t.cpu().numpy(), want[key], key)


@pytest.mark.parametrize("shape", [
    # gather-dot over 14 x 4 regions (maps whose width pads less in steps of 14 than of 8; unit counts that give every wave
    # two unit pairs): whole and partial regions, odd batches, partial channel blocks, both passes' layouts (AS 2 and 1)
    dict(N=3, W=28, H=28, S=5, F=40, G=4, region=1404),
    dict(N=4, W=27, H=27, S=33, F=20, G=4, region=1404),
    dict(N=2, W=42, H=30, S=3, F=8, G=8, region=1404),
    dict(N=5, W=13, H=9, S=4, F=33, G=3, region=1404),
    dict(N=2, W=28, H=26, S=2, F=12, G=7, region=1404),
    # the same maps where the form does not apply: one unit pair per wave somewhere (G % 4 in {1, 2}), or no gain
    dict(N=3, W=28, H=28, S=3, F=16, G=6, region=807),
    dict(N=2, W=56, H=56, S=2, F=8, G=4, region=808),
])
def test_gather_dot_region_forms(shape, monkeypatch):
    from dau_conv import _capi
    rs = np.random.RandomState(17)
    N, S, F, G, H, W = (shape[q] for q in ("N", "S", "F", "G", "H", "W"))
    x = rs.rand(N, S, H, W).astype(np.float32)
    dy = rs.randn(N, F, H, W).astype(np.float32)
    w = (rs.randn(1, S, G, F) * 0.1).astype(np.float32)
    mu1 = rs.uniform(-3.9, 3.9, (1, S, G, F)).astype(np.float32)
    mu2 = rs.uniform(-3.9, 3.9, (1, S, G, F)).astype(np.float32)
    plan = _capi.Plan(N, S, F, G, H, W, max_kernel_size=9, sigma_hint=0.5, flags=_capi.FLAG_USE_INTERPOLATION | _capi.FLAG_UNIT_TESTING)
    assert plan.info["dot_region"] == shape["region"], plan.info
    sigma = torch.full((1, S, G, F), 0.5, device="cuda")
    got = plan.backward(_dev(x), _dev(dy), _dev(w), _dev(mu1), _dev(mu2), sigma)
    plan.check_status()
    want = orc.backward(x, dy, w, mu1, mu2, 0.5, unit_testing=True)
    for t, key in zip(got, ("dx", "dw", "dmu1", "dmu2", "dsigma")):
        assert_parity(t.cpu().numpy(), want[key], key)
    if shape["region"] == 1404:
        # the 8-column form of the same plan (tuning build, DAU_DOT_RW=8 at plan creation) agrees to rounding
        monkeypatch.setenv("DAU_DOT_RW", "8")
        plan8 = tuning_capi().Plan(N, S, F, G, H, W, max_kernel_size=9, sigma_hint=0.5, flags=_capi.FLAG_USE_INTERPOLATION | _capi.FLAG_UNIT_TESTING)
        assert plan8.info["dot_region"] in (807, 808), plan8.info
        got8 = plan8.backward(_dev(x), _dev(dy), _dev(w), _dev(mu1), _dev(mu2), sigma)
        for a8, a14, key in zip(got8[1:], got[1:], ("dw", "dmu1", "dmu2", "dsigma")):
            assert_parity(a14.cpu().numpy(), a8.cpu().numpy(), key + " (14 x 4 vs 8-column regions)")


def test_calls_on_a_side_stream():
    """Every kernel is launched on the stream handed to the ABI (no default-stream work, no device-wide sync)."""
    from dau_conv import _capi
    rs = np.random.RandomState(21)
    N, S, F, G, H, W = 4, 6, 32, 4, 28, 28
    x = rs.rand(N, S, H, W).astype(np.float32); dy = rs.randn(N, F, H, W).astype(np.float32)
    w = (rs.randn(1, S, G, F) * 0.1).astype(np.float32)
    mu1 = rs.uniform(-3, 3, (1, S, G, F)).astype(np.float32); mu2 = rs.uniform(-3, 3, (1, S, G, F)).astype(np.float32)
    plan = _capi.Plan(N, S, F, G, H, W, max_kernel_size=9, sigma_hint=0.5)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        xd, dyd, wd, m1d, m2d = (_dev(a) for a in (x, dy, w, mu1, mu2))
        sigma = torch.full((1, S, G, F), 0.5, device="cuda")
        y = plan.forward(xd, wd, m1d, m2d, sigma)
        got = plan.backward(xd, dyd, wd, m1d, m2d, sigma)
        plan.check_status()
    side.synchronize()
    assert_parity(y.cpu().numpy(), orc.forward(x, w, mu1, mu2, 0.5), "y")
    want = orc.backward(x, dy, w, mu1, mu2, 0.5)
    for t, key in zip(got, ("dx", "dw", "dmu1", "dmu2", "dsigma")):
        assert_parity(t.cpu().numpy(), want[key], key)


def test_widest_prefilter_stays_on_the_tiled_kernels():
    """sigma = 1.6 gives the 17-tap prefilter (the largest supported, convolve.cu:40,245); staging bands keep it in LDS."""
    from dau_conv import _capi
    rs = np.random.RandomState(9)
    N, S, F, G, H, W, k, sg = 2, 3, 8, 2, 60, 70, 17, 1.6
    x = rs.rand(N, S, H, W).astype(np.float32); dy = rs.randn(N, F, H, W).astype(np.float32)
    w = (rs.randn(1, S, G, F) * 0.1).astype(np.float32)
    mu1 = rs.uniform(-7, 7, (1, S, G, F)).astype(np.float32); mu2 = rs.uniform(-7, 7, (1, S, G, F)).astype(np.float32)
    plan = _capi.Plan(N, S, F, G, H, W, max_kernel_size=k, sigma_hint=sg)
    assert plan.info["blur_support"] == 17
    assert plan.info["algo_forward"] == _capi.ALGO_TILED and plan.info["algo_backward"] == _capi.ALGO_TILED
    sigma = torch.full((1, S, G, F), sg, device="cuda")
    y = plan.forward(_dev(x), _dev(w), _dev(mu1), _dev(mu2), sigma)
    got = plan.backward(_dev(x), _dev(dy), _dev(w), _dev(mu1), _dev(mu2), sigma)
    plan.check_status()
    assert_parity(y.cpu().numpy(), orc.forward(x, w, mu1, mu2, sg), "y")
    want = orc.backward(x, dy, w, mu1, mu2, sg)
    for t, key in zip(got, ("dx", "dw", "dmu1", "dmu2", "dsigma")):
        assert_parity(t.cpu().numpy(), want[key], key)


def test_direct_kernels_take_more_than_65535_planes():
    """The any-shape kernels fold (image, output channel) into grid.x: N*F beyond the 65535 limit of grid.y still runs."""
    from dau_conv import _capi
    rs = np.random.RandomState(2)
    N, S, F, G, H, W = 300, 1, 256, 1, 4, 5
    x = rs.rand(N, S, H, W).astype(np.float32)
    w = (rs.randn(1, S, G, F) * 0.1).astype(np.float32)
    mu1 = rs.uniform(-2, 2, (1, S, G, F)).astype(np.float32)
    mu2 = rs.uniform(-2, 2, (1, S, G, F)).astype(np.float32)
    plan = _capi.Plan(N, S, F, G, H, W, max_kernel_size=9, sigma_hint=0.5, algo=_capi.ALGO_DIRECT)
    assert plan.info["algo_forward"] == _capi.ALGO_DIRECT
    y = plan.forward(_dev(x), _dev(w), _dev(mu1), _dev(mu2), torch.full((1, S, G, F), 0.5, device="cuda"))
    plan.check_status()
    assert_parity(y.cpu().numpy(), orc.forward(x, w, mu1, mu2, 0.5), "y (N*F = 76800 planes)")


def test_library_gets_the_stream_of_the_tensors_device(monkeypatch):
    """The stream handed to the C ABI is torch's current stream of the TENSORS' device, and the call runs with that device
    current (a model on another device than the current one must not be enqueued on the wrong stream)."""
    from dau_conv import _capi
    seen = {}
    real = _capi.lib.dau_conv_forward

    def spy(plan, stream, *rest):
        seen["stream"] = stream.value if hasattr(stream, "value") else stream
        seen["device"] = torch.cuda.current_device()
        return real(plan, stream, *rest)

    monkeypatch.setattr(_capi.lib, "dau_conv_forward", spy)
    N, S, F, G, H, W = 1, 2, 4, 2, 8, 8
    plan = _capi.Plan(N, S, F, G, H, W)
    x = torch.rand(N, S, H, W, device="cuda:0")
    p = lambda: torch.zeros(1, S, G, F, device="cuda:0")
    side = torch.cuda.Stream(device=0)
    with torch.cuda.stream(side):
        plan.forward(x, p(), p(), p(), torch.full((1, S, G, F), 0.5, device="cuda:0"))
    assert seen["stream"] == side.cuda_stream and seen["device"] == 0
    with pytest.raises(_capi.InvalidArgumentError):
        plan.forward(x, p().cpu(), p(), p(), p())


def test_forward_backward_capture_into_a_hip_graph():
    """The calls only enqueue work on the caller's stream (kernels and one 16-byte memset; the launch attributes were set
    by the first call), so a warmed-up forward + backward can be captured into a HIP graph and replayed: same results as
    the eager calls, also after the inputs changed in place.  The offset-bucket candidates are frozen at capture time;
    the device-side guards still pick the set the replay's offsets need."""
    from dau_conv import _capi
    rs = np.random.RandomState(8)
    N, S, F, G, H, W, k = 4, 12, 24, 4, 28, 28, 17
    plan = _capi.Plan(N, S, F, G, H, W, max_kernel_size=k, sigma_hint=0.5)
    x = torch.rand(N, S, H, W, device="cuda")
    dy = torch.randn(N, F, H, W, device="cuda")
    w = torch.randn(1, S, G, F, device="cuda") * 0.1
    mu1 = (torch.rand(1, S, G, F, device="cuda") * 2 - 1) * 3.0
    mu2 = (torch.rand(1, S, G, F, device="cuda") * 2 - 1) * 3.0
    sigma = torch.full((1, S, G, F), 0.5, device="cuda")
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        for _ in range(2):                                   # warm-up on the capture stream: workspace, attributes, hint
            plan.forward(x, w, mu1, mu2, sigma); plan.backward(x, dy, w, mu1, mu2, sigma)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        y = plan.forward(x, w, mu1, mu2, sigma)
        grads = plan.backward(x, dy, w, mu1, mu2, sigma)
    for trial in range(2):
        if trial == 1:                                       # new inputs and LARGER offsets in place (beyond the hinted bucket)
            x.copy_(torch.rand_like(x)); mu1.mul_(2.4); mu2.mul_(-2.4)
        graph.replay()
        torch.cuda.synchronize()
        want_y = orc.forward(x.cpu().numpy(), w.cpu().numpy(), mu1.cpu().numpy(), mu2.cpu().numpy(), 0.5)
        want = orc.backward(x.cpu().numpy(), dy.cpu().numpy(), w.cpu().numpy(), mu1.cpu().numpy(), mu2.cpu().numpy(), 0.5)
        assert_parity(y.cpu().numpy(), want_y, "graph replay %d: y" % trial)
        for t, key in zip(grads, ("dx", "dw", "dmu1", "dmu2", "dsigma")):
            assert_parity(t.cpu().numpy(), want[key], "graph replay %d: %s" % (trial, key))
