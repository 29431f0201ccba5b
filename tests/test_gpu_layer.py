"""GPU: the DAUConv2d / DAUConv1d layer surface, written the way the reference's own test drives it
(plugins/tensorflow/tests/dau_conv_test.py:335-416, `_run_DAUConv_forward_and_backward`): build the layer with
random initialisers, run forward + gradients w.r.t. [x, weights, mu1, mu2, sigma], compare with the oracle, with
the oracle's dmu multiplied by mu_learning_rate_factor = 1000 as the reference test does (:406-407)."""
import numpy as np
import pytest
import torch

from oracle import dau_oracle as orc
from util import assert_parity

pytestmark = pytest.mark.gpu


def _run_DAUConv_forward_and_backward(N, W, H, S, F, dau_uints, max_kernel_size, max_offset_init, use_interpolation=True,
                                      layer_cls=None, **layer_kw):
    import dau_conv
    torch.manual_seed(1234)
    mu_learning_rate_factor = 1000
    sigma = 0.5
    layer_cls = layer_cls or dau_conv.DAUConv2d
    kw = dict(filters=F, dau_units=dau_uints, max_kernel_size=max_kernel_size, use_bias=False,
              weight_initializer=dau_conv.random_normal_initializer(stddev=0.1),
              mu1_initializer=dau_conv.random_uniform_initializer(-max_offset_init, max_offset_init),
              sigma_initializer=dau_conv.constant_initializer(sigma), mu_learning_rate_factor=mu_learning_rate_factor,
              dau_mu_interpolation=use_interpolation, dau_sigma_trainable=True, unit_testing=True, in_channels=S)
    if layer_cls is dau_conv.DAUConv2d:
        kw["mu2_initializer"] = dau_conv.random_uniform_initializer(-max_offset_init, max_offset_init)
    kw.update(layer_kw)
    op = layer_cls(**kw).cuda()
    x = torch.rand(N, S, H, W, device="cuda", requires_grad=True)
    result = op(x)
    result_error = torch.randn_like(result)
    result.backward(result_error)
    torch.cuda.synchronize()

    w, mu1, mu2 = (t.detach().cpu().numpy() for t in (op.dau_weights, op.dau_mu1, op.dau_mu2))
    lim = np.floor(max_kernel_size / 2.0) - 0.01          # the layer clips mu in the graph (dau_conv.py:190-191)
    mu1c, mu2c = np.clip(mu1, -lim, lim), np.clip(mu2, -lim, lim)
    kwo = dict(ignore=op.num_dau_units_ignore, use_interpolation=use_interpolation,
               single_dim_kernel=op.dau_unit_single_dim, forbid_positive_dim1=op.dau_aggregation_forbid_positive_dim1)
    xn, en = x.detach().cpu().numpy(), result_error.cpu().numpy()
    gt_fwd = orc.forward(xn, w, mu1c, mu2c, sigma, **kwo)
    gt = orc.backward(xn, en, w, mu1c, mu2c, sigma, unit_testing=True, mu_learning_rate_factor=mu_learning_rate_factor, **kwo)
    assert_parity(result.detach().cpu().numpy(), gt_fwd, "fwd_output")
    assert_parity(x.grad.cpu().numpy(), gt["dx"], "bwd_error")
    assert_parity(op.weights.grad.cpu().numpy(), gt["dw"], "bwd_w_grad")
    # clipped units receive zero mu gradient through the clamp
    m1 = (np.abs(mu1) <= lim).astype(np.float32)
    m2 = (np.abs(mu2) <= lim).astype(np.float32)
    assert_parity(op.mu1.grad.cpu().numpy(), gt["dmu1"] * m1, "bwd_mu1_grad")
    if not op.dau_unit_single_dim:
        assert_parity(op.mu2.grad.cpu().numpy(), gt["dmu2"] * m2, "bwd_mu2_grad")
    # sigma is one scalar variable tiled to the parameter shape: its gradient is the sum (dau_conv.py:417-430)
    want = float(gt["dsigma"].astype(np.float64).sum())
    got = float(op.sigma.grad.item())
    # a sum of n = S*G*F signed per-unit gradients, each held to 1e-4 relative + 1e-6 of the max-norm: the errors of the terms add
    # up like a random walk, so the sum gets the relative bar + 4 sqrt(n) floors (n = 4096: 2.6e-4 of max|dsigma|)
    n_terms = gt["dsigma"].size
    assert abs(got - want) <= 1e-4 * abs(want) + 4e-6 * np.sqrt(n_terms) * float(np.abs(gt["dsigma"]).max()), (got, want, n_terms)
    return op


# The reference's own shape matrix, verbatim (batch, channel counts, image sizes, units, kernels, offset ranges): the C oracle
# takes well under a second for the largest of them on the GPU box's host cores, so nothing is reduced.
_QUICK = [  # dau_conv_test.py:418-437 test_DAUConvQuick
    dict(N=2, W=65, H=8, S=33, F=32, dau_uints=(1, 2), max_kernel_size=9, max_offset_init=3),
    dict(N=1, W=65, H=8, S=32, F=32, dau_uints=(1, 2), max_kernel_size=9, max_offset_init=3),
    dict(N=1, W=8, H=8, S=32, F=32, dau_uints=(1, 2), max_kernel_size=9, max_offset_init=3),
    dict(N=4, W=8, H=8, S=32, F=32, dau_uints=(1, 2), max_kernel_size=9, max_offset_init=3),
    dict(N=16, W=32, H=32, S=32, F=32, dau_uints=(2, 2), max_kernel_size=9, max_offset_init=3),
    dict(N=16, W=32, H=32, S=32, F=32, dau_uints=(2, 2), max_kernel_size=17, max_offset_init=6),
    dict(N=16, W=32, H=32, S=32, F=32, dau_uints=(2, 2), max_kernel_size=17, max_offset_init=3),
    dict(N=16, W=32, H=32, S=3, F=32, dau_uints=(2, 2), max_kernel_size=17, max_offset_init=3),
    dict(N=16, W=64, H=64, S=3, F=32, dau_uints=(2, 2), max_kernel_size=33, max_offset_init=10),
]
_FULL = [  # dau_conv_test.py:439-465 test_DAUConv: the cases Quick does not already hold
    dict(N=2, W=65, H=8, S=32, F=32, dau_uints=(1, 2), max_kernel_size=9, max_offset_init=3),
    dict(N=2, W=8, H=8, S=32, F=32, dau_uints=(1, 2), max_kernel_size=9, max_offset_init=3),
    dict(N=16, W=6, H=6, S=64, F=256, dau_uints=(2, 1), max_kernel_size=17, max_offset_init=8),
    dict(N=16, W=64, H=64, S=32, F=32, dau_uints=(2, 2), max_kernel_size=33, max_offset_init=10),
    dict(N=16, W=64, H=64, S=32, F=32, dau_uints=(2, 2), max_kernel_size=65, max_offset_init=20),
]
_INTERPOLATION = [  # dau_conv_test.py:467-501 test_DAUConvInterpolation (use_interpolation=False)
    dict(N=2, W=65, H=8, S=32, F=32, dau_uints=(1, 2), max_kernel_size=9, max_offset_init=3),
    dict(N=1, W=65, H=8, S=32, F=32, dau_uints=(1, 2), max_kernel_size=9, max_offset_init=3),
    dict(N=1, W=8, H=8, S=32, F=32, dau_uints=(1, 2), max_kernel_size=9, max_offset_init=3),
    dict(N=2, W=8, H=8, S=32, F=32, dau_uints=(1, 2), max_kernel_size=9, max_offset_init=3),
    dict(N=4, W=8, H=8, S=32, F=32, dau_uints=(1, 2), max_kernel_size=9, max_offset_init=3),
    dict(N=16, W=32, H=32, S=32, F=32, dau_uints=(2, 2), max_kernel_size=9, max_offset_init=3),
    dict(N=16, W=32, H=32, S=32, F=32, dau_uints=(2, 2), max_kernel_size=17, max_offset_init=6),
    dict(N=16, W=6, H=6, S=64, F=256, dau_uints=(2, 1), max_kernel_size=17, max_offset_init=8),
    dict(N=16, W=32, H=32, S=32, F=32, dau_uints=(2, 2), max_kernel_size=17, max_offset_init=3),
    dict(N=16, W=64, H=64, S=16, F=32, dau_uints=(2, 2), max_kernel_size=33, max_offset_init=10),
    dict(N=16, W=64, H=64, S=32, F=32, dau_uints=(2, 2), max_kernel_size=33, max_offset_init=10),
    dict(N=16, W=64, H=64, S=32, F=32, dau_uints=(2, 2), max_kernel_size=65, max_offset_init=20),
]


def _id(c):
    return "N%d_%dx%d_S%d_F%d_u%dx%d_k%d_m%d" % (c["N"], c["H"], c["W"], c["S"], c["F"], c["dau_uints"][0], c["dau_uints"][1],
                                                 c["max_kernel_size"], c["max_offset_init"])


@pytest.mark.parametrize("cfg", _QUICK + [
    # :631-633 test_DAUConvSingleUnit -- one unit becomes two with one ignored
    dict(N=4, W=32, H=32, S=8, F=16, dau_uints=(1, 1), max_kernel_size=9, max_offset_init=3),
    # mu beyond the clip range: clipped units must get zero mu gradient
    dict(N=2, W=16, H=16, S=4, F=8, dau_uints=(2, 2), max_kernel_size=9, max_offset_init=6),
], ids=_id)
def test_DAUConvQuick(cfg):
    _run_DAUConv_forward_and_backward(**cfg)


@pytest.mark.parametrize("cfg", _FULL, ids=_id)
def test_DAUConv(cfg):
    _run_DAUConv_forward_and_backward(**cfg)


@pytest.mark.parametrize("cfg", _INTERPOLATION, ids=_id)
def test_DAUConvInterpolation(cfg):
    _run_DAUConv_forward_and_backward(use_interpolation=False, **cfg)


def test_DAUConv1d():
    import dau_conv
    op = _run_DAUConv_forward_and_backward(N=2, W=32, H=8, S=8, F=16, dau_uints=(1, 2), max_kernel_size=9, max_offset_init=3,
                                           layer_cls=dau_conv.DAUConv1d)
    assert op.dau_unit_single_dim and float(op.dau_mu2.abs().max()) == 0.0
    assert op.mu2.grad is None or float(op.mu2.grad.abs().max()) == 0.0


def test_layer_semantics_stride_bias_activation_and_names():
    import dau_conv
    torch.manual_seed(0)
    layer = dau_conv.DAUConv2d(filters=8, dau_units=(2, 2), max_kernel_size=9, strides=2, use_bias=True,
                               activation=torch.relu, in_channels=4,
                               bias_initializer=dau_conv.constant_initializer(0.25)).cuda()
    assert sorted(k for k, _ in layer.named_parameters()) == ["bias", "mu1", "mu2", "sigma", "weights"]
    assert tuple(layer.weights.shape) == (1, 4, 4, 8) and tuple(layer.sigma.shape) == (1,) and not layer.sigma.requires_grad
    x = torch.rand(2, 4, 17, 20, device="cuda")
    y = layer(x)
    ref = dau_conv.DAUConv2d(filters=8, dau_units=(2, 2), max_kernel_size=9, strides=1, use_bias=False, in_channels=4).cuda()
    ref.load_state_dict({k: v for k, v in layer.state_dict().items() if k != "bias"})
    want = torch.relu(ref(x)[:, :, ::2, ::2] + 0.25)       # stride emulated by slicing (dau_conv.py:497-498)
    assert y.shape == want.shape == (2, 8, 9, 10)
    assert torch.allclose(y, want, rtol=1e-6, atol=1e-6)
    # functional slim-style form keeps variables per scope
    out1 = dau_conv.dau_conv2d(x, 8, (2, 2), 9, data_format="NCHW", scope="t_scope")
    out2 = dau_conv.dau_conv2d(x, 8, (2, 2), 9, data_format="NCHW", scope="t_scope", reuse=True)
    assert torch.equal(out1, out2) and (out1 >= 0).all()
    with pytest.raises(ValueError):
        dau_conv.dau_conv2d(x, 8, (2, 2), 9, data_format="NHWC")


def test_layer_raises_like_the_reference_ops():
    import dau_conv
    x = torch.rand(1, 2, 8, 8, device="cuda")
    w = torch.randn(1, 2, 2, 4, device="cuda")
    mu = torch.zeros(1, 2, 2, 4, device="cuda")
    sigma = torch.full((1, 2, 2, 4), 0.5, device="cuda")
    bad = mu.clone(); bad[0, 0, 0, 0] = float("nan")
    # check_offsets=True: wait for the call and raise at once, as the reference ops do
    with pytest.raises(dau_conv.FailedPreconditionError):      # dau_conv_op.cpp:256-261
        dau_conv.dau_conv(x, w, bad, mu, sigma, num_output=4, kernel_size=9, check_offsets=True)
    far = mu.clone(); far[0, 0, 0, 0] = 40.0
    with pytest.raises(dau_conv.InvalidArgumentError):         # dau_conv_op.cpp:245-248
        dau_conv.dau_conv(x, w, far, mu, sigma, num_output=4, kernel_size=65, check_offsets=True)
    with pytest.raises(dau_conv.InvalidArgumentError):         # shape function: last dim == num_output
        dau_conv.dau_conv(x, w, mu, mu, sigma, num_output=8, kernel_size=9)


def test_async_offset_check_raises_one_call_late_without_a_sync():
    """check_offsets="async" (the default): the call with the bad offsets returns, the NEXT call of the same plan (or
    check_pending_offsets) raises the reference's error from the status the device left in pinned host memory."""
    import dau_conv
    x = torch.rand(1, 2, 9, 11, device="cuda")
    w = torch.randn(1, 2, 2, 4, device="cuda")
    mu = torch.zeros(1, 2, 2, 4, device="cuda")
    sigma = torch.full((1, 2, 2, 4), 0.5, device="cuda")
    bad = mu.clone(); bad[0, 1, 0, 3] = float("nan")
    y = dau_conv.dau_conv(x, w, bad, mu, sigma, num_output=4, kernel_size=9)        # enqueued, not checked yet
    torch.cuda.synchronize()
    assert torch.isfinite(y).all()                                                   # a NaN offset reads as offset 0
    with pytest.raises(dau_conv.FailedPreconditionError):
        dau_conv.dau_conv(x, w, mu, mu, sigma, num_output=4, kernel_size=9)         # same plan: sees the previous status
    # a clean call clears it again
    dau_conv.dau_conv(x, w, mu, mu, sigma, num_output=4, kernel_size=9, check_offsets=False)
    dau_conv.check_pending_offsets()
    far = mu.clone(); far[0, 0, 0, 0] = 6.0
    dau_conv.dau_conv(x, w, far, mu, sigma, num_output=4, kernel_size=9)
    with pytest.raises(dau_conv.InvalidArgumentError):
        dau_conv.check_pending_offsets()
    dau_conv.dau_conv(x, w, mu, mu, sigma, num_output=4, kernel_size=9, check_offsets=False)
    dau_conv.check_pending_offsets()


def test_bad_offsets_of_a_middle_layer_are_not_overwritten_by_later_calls():
    """Plans are shared by every layer of one shape.  Three calls A1, A2, A3 of one plan where only A2 has bad offsets: A3
    overwrites the plan's 'most recent call' status, so the bad one has to be recorded STICKY in the pinned mirror until the
    host has reported it (the kernels keep going with a NaN offset read as 0 and a far one clamped: training would
    continue silently wrong)."""
    import dau_conv
    import importlib
    dc = importlib.import_module("dau_conv.dau_conv")        # (the package attribute of that name is the op function)
    x = torch.rand(2, 3, 10, 12, device="cuda")
    w = torch.randn(1, 3, 2, 4, device="cuda")
    mu = torch.zeros(1, 3, 2, 4, device="cuda")
    sigma = torch.full((1, 3, 2, 4), 0.5, device="cuda")
    call = lambda m1: dau_conv.dau_conv(x, w, m1, mu, sigma, num_output=4, kernel_size=9, check_offsets=False)
    for bad_value, error in ((float("nan"), dau_conv.FailedPreconditionError), (7.5, dau_conv.InvalidArgumentError)):
        bad = mu.clone(); bad[0, 2, 1, 3] = bad_value
        call(mu); call(bad); call(mu); call(mu)
        torch.cuda.synchronize()
        plan = dc._get_plan(x, w, dc._settings(sigma, num_output=4, kernel_size=9))
        with pytest.raises(error):
            plan.last_status()
        assert plan.last_status() is None                  # reported once; the mirror is back to "nothing to report"
        call(mu)
        dau_conv.check_pending_offsets()
    # the same through the default mode: the error surfaces on a later call although good calls ran in between
    bad = mu.clone(); bad[0, 0, 0, 0] = float("nan")
    dau_conv.dau_conv(x, w, bad, mu, sigma, num_output=4, kernel_size=9)
    call(mu); call(mu)
    torch.cuda.synchronize()
    with pytest.raises(dau_conv.FailedPreconditionError):
        dau_conv.dau_conv(x, w, mu, mu, sigma, num_output=4, kernel_size=9)
    dau_conv.check_pending_offsets()


def test_sigma_host_copy_follows_the_tensor():
    """The prefilter support comes from a host copy of sigma; it must follow load_state_dict / in-place updates
    (the reference re-reads sigma whenever it builds its layer, base_dau_conv_layer.cpp:140-146)."""
    import dau_conv
    from oracle import dau_oracle as orc
    torch.manual_seed(0)
    layer = dau_conv.DAUConv2d(filters=4, dau_units=(1, 2), max_kernel_size=9, use_bias=False, in_channels=3).cuda()
    x = torch.rand(2, 3, 12, 12, device="cuda")
    layer(x)
    assert layer._sigma_host == 0.5
    sd = layer.state_dict(); sd["sigma"] = torch.tensor([0.8], device="cuda")
    layer.load_state_dict(sd)
    y = layer(x)
    assert abs(layer._sigma_host - 0.8) < 1e-6
    m = layer._dau_convolution_op.mean_max_allowed_offset
    want = orc.forward(x.cpu().numpy(), layer.weights.detach().cpu().numpy(),
                       layer.mu1.detach().clamp(-m, m).cpu().numpy(), layer.mu2.detach().clamp(-m, m).cpu().numpy(), 0.8)
    from util import assert_parity
    assert_parity(y.detach().cpu().numpy(), want, "y after sigma 0.5 -> 0.8")      # 9x9 prefilter, not a truncated 7x7


def test_toy_training_loop_reduces_the_loss():
    """examples/train_toy.py: two stacked DAUConv2d layers fitted with Adam to a teacher DAU layer; weights, offsets and
    biases all move through the operator's gradients, so the loss has to fall."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location(
        "train_toy", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples", "train_toy.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    losses = mod.run(steps=60, verbose=False)
    assert all(l == l for l in losses)                      # no NaN
    assert sum(losses[-5:]) / 5 < 0.6 * sum(losses[:5]) / 5, (losses[:5], losses[-5:])
