"""GPU: the exchange path bench.py uses for N>1 (need-mask split of dau_conv_backward + asynchronous RCCL all-reduce of
the flat [dw|dmu1|dmu2|dsigma] bucket), rehearsed with one rank on the one GPU of the test box; the world-size-2 logic
is covered on CPU with gloo (test_distributed_cpu.py)."""
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
        dx = ex.run(lambda need: plan.backward(x, dy, w, mu1, mu2, sigma, need_mask=need))
        got = ex.wait()
        torch.cuda.synchronize()
        assert torch.equal(dx, want[0])
        for a, b in zip(got, want[1:]):          # one rank: the reduced bucket equals the local gradients bit for bit
            assert torch.equal(a, b)
    finally:
        dist.destroy_process_group()
