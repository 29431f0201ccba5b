"""GPU, full BASELINE size (north star: N=128, C=256->256, 56x56, four units, kernel 9): the oracle needs minutes for
this shape, so parity is established through size-independent properties plus oracle spot checks:

  * per-image outputs (y, dx) of images taken out of the full-size run equal the oracle run on just those images;
  * y is linear in x;
  * sum(y * dy) == sum(w * dw)  (y is linear in w and dw is its gradient: ties the gather-dot to the gather-sum);
  * the batch gradients equal the sum of the gradients of the two half batches (chunked partial sums, image pairing);
  * the LDS-tiled kernels agree with the independent direct kernels (one thread per output, double accumulation).
"""
import numpy as np
import pytest
import torch

from oracle import dau_oracle as orc
from util import assert_parity

pytestmark = pytest.mark.gpu

N, S, F, G, H, W, K = 128, 256, 256, 4, 56, 56, 9


@pytest.fixture(scope="module")
def ns():
    from dau_conv import _capi
    g = torch.Generator(device="cuda"); g.manual_seed(2024)
    t = dict(x=torch.rand((N, S, H, W), device="cuda", generator=g),
             dy=torch.randn((N, F, H, W), device="cuda", generator=g),
             w=torch.randn((1, S, G, F), device="cuda", generator=g) * 0.1,
             mu1=(torch.rand((1, S, G, F), device="cuda", generator=g) * 2 - 1) * 3.0,
             mu2=(torch.rand((1, S, G, F), device="cuda", generator=g) * 2 - 1) * 3.0,
             sigma=torch.full((1, S, G, F), 0.5, device="cuda"))
    plan = _capi.Plan(N, S, F, G, H, W, max_kernel_size=K, sigma_hint=0.5)
    assert plan.info["algo_forward"] == _capi.ALGO_TILED and plan.info["algo_backward"] == _capi.ALGO_TILED
    t["plan"] = plan
    t["y"] = plan.forward(t["x"], t["w"], t["mu1"], t["mu2"], t["sigma"])
    t["grads"] = plan.backward(t["x"], t["dy"], t["w"], t["mu1"], t["mu2"], t["sigma"])
    plan.check_status()
    yield t
    t.clear()
    torch.cuda.empty_cache()


def _rel_to_max(got, want):
    return float((got.double() - want.double()).abs().max() / want.double().abs().max())


def test_images_of_the_full_run_match_the_oracle(ns):
    idx = [0, 77, 127]          # first, an odd one from the middle, the last image of the last pair
    x = ns["x"][idx].cpu().numpy(); dy = ns["dy"][idx].cpu().numpy()
    w, mu1, mu2 = (ns[k].cpu().numpy() for k in ("w", "mu1", "mu2"))
    want_y = orc.forward(x, w, mu1, mu2, 0.5)
    want = orc.backward(x, dy, w, mu1, mu2, 0.5, need=("dx",))
    assert_parity(ns["y"][idx].cpu().numpy(), want_y, "y of images %s" % idx)
    assert_parity(ns["grads"][0][idx].cpu().numpy(), want["dx"], "dx of images %s" % idx)


def test_forward_is_linear_in_x(ns):
    plan = ns["plan"]
    x2 = torch.rand_like(ns["x"])
    y2 = plan.forward(x2, ns["w"], ns["mu1"], ns["mu2"], ns["sigma"])
    y12 = plan.forward(ns["x"] * 0.75 + x2, ns["w"], ns["mu1"], ns["mu2"], ns["sigma"])
    assert _rel_to_max(y12, ns["y"] * 0.75 + y2) < 1e-5


def test_weight_gradient_identity(ns):
    lhs = float((ns["y"].double() * ns["dy"].double()).sum())
    rhs = float((ns["w"].double() * ns["grads"][1].double()).sum())
    scale = float((ns["w"].double().abs() * ns["grads"][1].double().abs()).sum())
    assert abs(lhs - rhs) <= 1e-4 * scale, (lhs, rhs, scale)


def test_batch_gradients_are_the_sum_of_half_batches(ns):
    from dau_conv import _capi
    half = _capi.Plan(N // 2, S, F, G, H, W, max_kernel_size=K, sigma_hint=0.5)
    need = _capi.NEED_DW | _capi.NEED_DMU1 | _capi.NEED_DMU2 | _capi.NEED_DSIGMA
    a = half.backward(ns["x"][:N // 2], ns["dy"][:N // 2], ns["w"], ns["mu1"], ns["mu2"], ns["sigma"], need_mask=need)
    b = half.backward(ns["x"][N // 2:], ns["dy"][N // 2:], ns["w"], ns["mu1"], ns["mu2"], ns["sigma"], need_mask=need)
    for i, name in ((1, "dw"), (2, "dmu1"), (3, "dmu2"), (4, "dsigma")):
        assert _rel_to_max(a[i] + b[i], ns["grads"][i]) < 2e-5, name


def test_tiled_kernels_agree_with_direct_kernels(ns):
    from dau_conv import _capi
    direct = _capi.Plan(N, S, F, G, H, W, max_kernel_size=K, sigma_hint=0.5, algo=_capi.ALGO_DIRECT)
    y = direct.forward(ns["x"], ns["w"], ns["mu1"], ns["mu2"], ns["sigma"])
    # both sides accumulate 4096 products per output in fp32, in different orders: the floor is 1e-5 of the max-norm here
    # (against the double-accumulating oracle, above, it is 1e-6)
    assert_parity(ns["y"].cpu().numpy(), y.cpu().numpy(), "y tiled vs direct", floor=1e-5)
    del y
    grads = direct.backward(ns["x"], ns["dy"], ns["w"], ns["mu1"], ns["mu2"], ns["sigma"])
    assert_parity(ns["grads"][0].cpu().numpy(), grads[0].cpu().numpy(), "dx tiled vs direct", floor=1e-5)
    # parameter gradients: sums of 4e5 products each; the direct kernel accumulates in double, the tiled one in fp32
    # per lane and chunk: bound the difference against the gradient's max-norm
    for i, name in ((1, "dw"), (2, "dmu1"), (3, "dmu2"), (4, "dsigma")):
        assert _rel_to_max(ns["grads"][i], grads[i]) < 1e-4, name
