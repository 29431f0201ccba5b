"""GPU: the exchange path bench.py uses for N>1 (dau_conv_backward_param_sums -> asynchronous all-reduce of the flat
[4,S,G,F] buffer of raw sums, hidden under the dx pass -> dau_conv_finalize_param_grads AFTER the exchange): with one rank over
RCCL, and with TWO ranks -- fresh child processes that share the one GPU of the test box, backend gloo -- on the HIP path,
through the C ABI and through DAUConv2d(process_group=...), against the single-process full-batch result.  (The same logic on
CPU with the oracle: test_distributed_cpu.py.)"""
import os
import subprocess
import sys

import numpy as np

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


@pytest.fixture(scope="module")
def two_ranks(tmp_path_factory):
    """Two child processes (rank 0 and 1 of a gloo group, both on GPU 0) run tests/dist_worker.py on the halves of one batch."""
    out = str(tmp_path_factory.mktemp("dist2"))
    here = os.path.dirname(os.path.abspath(__file__))
    port = 29600 + os.getpid() % 2000
    procs = [subprocess.Popen([sys.executable, os.path.join(here, "dist_worker.py"), str(r), "2", str(port), out],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(2)]
    logs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=300)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        logs.append(o.decode("utf-8", "replace"))
    for r, p in enumerate(procs):
        assert p.returncode == 0, "rank %d failed:\n%s" % (r, logs[r][-3000:])
    return [dict(np.load(os.path.join(out, "rank%d.npz" % r))) for r in range(2)]


def _close(got, want, name, floor=1e-6):
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    err = np.abs(got - want).max()
    assert err <= floor * np.abs(want).max(), "%s: %.3e of the max-norm" % (name, err / np.abs(want).max())


def test_two_ranks_c_abi_sums_allreduce_finalize_equals_full_batch(two_ranks):
    """HIP backward_param_sums on the two shard_bounds halves -> all_reduce -> finalize_param_grads == the single-process
    full-batch dau_conv_backward (<= 1e-6 of the max-norm: the two orders of summation), ranks bit-identical."""
    import dist_worker as dw
    from dau_conv import _capi
    c = dw.SHAPE
    x, dy, w, mu1, mu2 = dw.inputs()
    dev = lambda a: torch.from_numpy(a).cuda()
    sigma = torch.full((1, c["S"], c["G"], c["F"]), 0.5, device="cuda")
    plan = _capi.Plan(c["N"], c["S"], c["F"], c["G"], c["H"], c["W"], max_kernel_size=c["k"], sigma_hint=0.5,
                      mu_learning_rate_factor=c["lr"])
    full = plan.backward(dev(x), dev(dy), dev(w), dev(mu1), dev(mu2), sigma)
    full_sums = plan.backward_param_sums(dev(x), dev(dy), dev(mu1), dev(mu2), sigma)
    r0, r1 = two_ranks
    assert tuple(r0["lo_hi"]) == (0, 5) and tuple(r1["lo_hi"]) == (5, 10)
    for key in ("abi_sums", "abi_dw", "abi_dmu1", "abi_dmu2", "abi_dsigma"):
        assert np.array_equal(r0[key], r1[key]), key + ": the ranks differ"
    _close(r0["abi_sums"], full_sums.cpu().numpy(), "reduced raw sums")
    for i, key in enumerate(("dw", "dmu1", "dmu2", "dsigma")):
        _close(r0["abi_" + key], full[i + 1].cpu().numpy(), key)
    # dx is per image: the shards' dx are the rows of the full-batch dx, bit for bit (same kernels, other batch size only
    # changes which workgroup takes an image pair) -- compared at the fp32 bar
    dx = np.concatenate([r0["abi_dx"], r1["abi_dx"]])
    _close(dx, full[0].cpu().numpy(), "dx", floor=1e-6)


def test_two_ranks_layer_process_group_equals_full_batch(two_ranks):
    """DAUConv2d(process_group=True): y and dx of the shards are the rows of the full-batch result; the weights / mu1 / mu2 /
    sigma gradients every rank ends up with equal the single-process full-batch gradients (grad_reduce="sum") or their mean
    over the ranks ("mean")."""
    import dist_worker as dw
    c = dw.SHAPE
    x, dy, *_ = dw.inputs()
    layer = dw.make_layer(process_group=None)
    xin = torch.from_numpy(x).cuda().requires_grad_(True)
    y = layer(xin)
    y.backward(torch.from_numpy(dy).cuda())
    torch.cuda.synchronize()
    r0, r1 = two_ranks
    _close(np.concatenate([r0["layer_y"], r1["layer_y"]]), y.detach().cpu().numpy(), "y")
    _close(np.concatenate([r0["layer_dx"], r1["layer_dx"]]), xin.grad.cpu().numpy(), "dx")
    for name in ("weights", "mu1", "mu2", "sigma"):
        want = getattr(layer, name).grad.cpu().numpy()
        for mode, scale in (("sum", 1.0), ("mean", 0.5)):
            key = "layer_%s_%s" % (mode, name)
            assert np.array_equal(r0[key], r1[key]), key + ": the ranks differ"
            _close(r0[key], want * scale, key, floor=2e-6 if name == "sigma" else 1e-6)
