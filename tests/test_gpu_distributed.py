"""GPU: the exchange path bench.py uses for N>1 (dau_conv_backward_param_sums -> asynchronous RCCL all-reduce of the flat
[4,S,G,F] buffer of raw sums, hidden under the dx pass -> dau_conv_finalize_param_grads AFTER the exchange), rehearsed
with one rank on the one GPU of the test box; the world-size-2 logic is covered on CPU with gloo
(test_distributed_cpu.py)."""
import os

import pytest
import torch
import torch.distributed as dist

pytestmark = pytest.mark.gpu


def test_overlapped_backward_single_rank_rccl():
    from dau_conv import _capi
    from dau_conv.distributed import OverlappedBackward
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29541")
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        N, S, F, G, H, W = 6, 8, 32, 4, 24, 24
        g = torch.Generator(device=dev); g.manual_seed(3)
        x = torch.rand((N, S, H, W), device=dev, generator=g)
        dy = torch.randn((N, F, H, W), device=dev, generator=g)
        w = torch.randn((1, S, G, F), device=dev, generator=g) * 0.1
        mu1 = (torch.rand((1, S, G, F), device=dev, generator=g) * 2 - 1) * 3
        mu2 = (torch.rand((1, S, G, F), device=dev, generator=g) * 2 - 1) * 3
        sigma = torch.full((1, S, G, F), 0.5, device=dev)
        plan = _capi.Plan(N, S, F, G, H, W, max_kernel_size=9, sigma_hint=0.5)
        want = plan.backward(x, dy, w, mu1, mu2, sigma)
        ex = OverlappedBackward((1, S, G, F), dev)
        dx = ex.run(lambda out: plan.backward_param_sums(x, dy, mu1, mu2, sigma, out=out),
                    lambda: plan.backward(x, dy, w, mu1, mu2, sigma, need_mask=_capi.NEED_DX)[0],
                    lambda sums: plan.finalize_param_grads(sums, w))
        got = ex.wait()
        torch.cuda.synchronize()
        assert torch.equal(dx, want[0])
        for a, b in zip(got, want[1:]):          # one rank: the reduced bucket equals the local gradients bit for bit
            assert torch.equal(a, b)
    finally:
        dist.destroy_process_group()


def test_param_sums_then_finalize_equals_backward():
    """The two-step parameter-gradient path (sums, then the elementwise tail) is the one-call path, bit for bit, incl. an
    ignored unit, the lr factor and the need mask."""
    from dau_conv import _capi
    dev = torch.device("cuda", 0)
    N, S, F, G, H, W = 5, 6, 40, 6, 20, 31
    g = torch.Generator(device=dev); g.manual_seed(11)
    x = torch.rand((N, S, H, W), device=dev, generator=g)
    dy = torch.randn((N, F, H, W), device=dev, generator=g)
    w = torch.randn((1, S, G, F), device=dev, generator=g) * 0.1
    mu1 = (torch.rand((1, S, G, F), device=dev, generator=g) * 2 - 1) * 7
    mu2 = (torch.rand((1, S, G, F), device=dev, generator=g) * 2 - 1) * 7
    sigma = torch.full((1, S, G, F), 0.5, device=dev)
    plan = _capi.Plan(N, S, F, G, H, W, max_kernel_size=17, number_units_ignore=1, sigma_hint=0.5, mu_learning_rate_factor=250.0)
    want = plan.backward(x, dy, w, mu1, mu2, sigma)
    sums = plan.backward_param_sums(x, dy, mu1, mu2, sigma)
    got = plan.finalize_param_grads(sums, w)
    for a, b in zip(got, want[1:]):
        assert torch.equal(a, b)
    assert float(got[0][:, :, G - 1].abs().max()) == 0.0            # the ignored unit
    only = plan.finalize_param_grads(sums, w, need_mask=_capi.NEED_DMU1)
    assert only[0] is None and only[2] is None and only[3] is None and torch.equal(only[1], want[2])
