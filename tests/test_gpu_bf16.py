"""GPU: bfloat16 activations (DAU_FLAG_IO_BF16; BASELINE config 2 names bf16 I/O with fp32 accumulation).  x, dy go in
as bfloat16, y and dx come back as bfloat16, parameters and their gradients stay fp32.  Bar: 2e-2 relative
against the fp32 oracle fed the bf16-rounded inputs; in practice the only loss is the final rounding of y / dx (2^-9), and
the parameter gradients, which are not rounded, match to the fp32 tolerance."""
import numpy as np
import pytest
import torch

from oracle import dau_oracle as orc
from util import assert_parity

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("shape", [
    dict(N=3, S=6, F=40, G=4, H=56, W=56, k=9, m=3),
    dict(N=4, S=8, F=16, G=6, H=28, W=28, k=9, m=3),          # stacked planes, two gather-dot passes
    dict(N=2, S=5, F=8, G=2, H=40, W=72, k=17, m=7),
    dict(N=2, S=3, F=8, G=3, H=33, W=20, k=65, m=20),         # window passes accumulate into a bf16 output
])
def test_bf16_io_against_oracle(shape):
    from dau_conv import _capi
    rs = np.random.RandomState(5)
    N, S, F, G, H, W, k, m = (shape[q] for q in ("N", "S", "F", "G", "H", "W", "k", "m"))
    xb = torch.from_numpy(rs.rand(N, S, H, W).astype(np.float32)).to(torch.bfloat16)
    dyb = torch.from_numpy(rs.randn(N, F, H, W).astype(np.float32)).to(torch.bfloat16)
    w = (rs.randn(1, S, G, F) * 0.1).astype(np.float32)
    lim = k // 2 - 0.01
    mu1 = np.clip(rs.uniform(-m, m, (1, S, G, F)), -lim, lim).astype(np.float32)
    mu2 = np.clip(rs.uniform(-m, m, (1, S, G, F)), -lim, lim).astype(np.float32)
    plan = _capi.Plan(N, S, F, G, H, W, max_kernel_size=k, sigma_hint=0.5,
                      flags=_capi.FLAG_USE_INTERPOLATION | _capi.FLAG_IO_BF16)
    dev = lambda a: torch.from_numpy(a).cuda()
    sigma = torch.full((1, S, G, F), 0.5, device="cuda")
    y = plan.forward(xb.cuda(), dev(w), dev(mu1), dev(mu2), sigma)
    dx, dw, dmu1, dmu2, dsigma = plan.backward(xb.cuda(), dyb.cuda(), dev(w), dev(mu1), dev(mu2), sigma)
    plan.check_status()
    assert y.dtype == torch.bfloat16 and dx.dtype == torch.bfloat16 and dw.dtype == torch.float32
    x32, dy32 = xb.float().numpy(), dyb.float().numpy()          # what the kernels actually read
    want_y = orc.forward(x32, w, mu1, mu2, 0.5)
    want = orc.backward(x32, dy32, w, mu1, mu2, 0.5)
    # bf16 outputs: half an ulp of bfloat16 (2^-9 relative) per rounding, one rounding per window pass
    assert_parity(y.float().cpu().numpy(), want_y, "y", rel=2e-2, floor=4e-3)
    assert_parity(dx.float().cpu().numpy(), want["dx"], "dx", rel=2e-2, floor=4e-3)
    # fp32 outputs keep the fp32 bar
    for got, key in ((dw, "dw"), (dmu1, "dmu1"), (dmu2, "dmu2"), (dsigma, "dsigma")):
        assert_parity(got.cpu().numpy(), want[key], key)


def test_bf16_layer_forward_backward():
    import dau_conv
    torch.manual_seed(0)
    layer = dau_conv.DAUConv2d(filters=16, dau_units=(2, 2), max_kernel_size=9, in_channels=8, use_bias=False).cuda()
    x = torch.rand(4, 8, 32, 32, device="cuda").to(torch.bfloat16).requires_grad_(True)
    y = layer(x)
    assert y.dtype == torch.bfloat16
    y.float().sum().backward()
    assert x.grad.dtype == torch.bfloat16 and layer.weights.grad.dtype == torch.float32
    ref = layer(x.detach().float())                               # the same layer on the fp32 copy of the bf16 input
    assert torch.allclose(y.float(), ref, rtol=2e-2, atol=2e-2)


def test_bf16_needs_the_tiled_kernels():
    from dau_conv import _capi
    with pytest.raises(_capi.InvalidArgumentError):
        _capi.Plan(2, 4, 8, 2, 16, 16, flags=_capi.FLAG_USE_INTERPOLATION | _capi.FLAG_IO_BF16, algo=_capi.ALGO_DIRECT)


def test_bf16_store_keeps_a_nan_a_nan():
    """A NaN in the input must come out as a NaN in the bfloat16 output (the rounding add on the bits would turn some NaN
    payloads into zero or infinity)."""
    from dau_conv import _capi
    N, S, F, G, H, W = 1, 1, 2, 1, 8, 8
    plan = _capi.Plan(N, S, F, G, H, W, flags=_capi.FLAG_USE_INTERPOLATION | _capi.FLAG_IO_BF16)
    x = torch.ones(N, S, H, W, device="cuda")
    x[0, 0, 3, 3] = float("nan")
    w = torch.ones(1, S, G, F, device="cuda")
    z = torch.zeros(1, S, G, F, device="cuda")
    y = plan.forward(x.to(torch.bfloat16), w, z, z.clone(), torch.full((1, S, G, F), 0.5, device="cuda"))
    assert torch.isnan(y[0, 0, 3, 3]) and torch.isnan(y[0, 1, 3, 3])
    assert torch.isfinite(y[0, 0, 7, 7])                      # beyond the 7 x 7 prefilter around the NaN
