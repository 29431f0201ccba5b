"""CPU, world_size 2, gloo: the batch-sharded data-parallel path.  Each rank computes the parameter
gradient SUMS of its batch shard (here with the CPU oracle's building blocks, since there is no GPU in this suite), the
flat [4,S,G,F] buffer of raw sums is all-reduced, the elementwise tail (dmu *= w*lr, ...) runs AFTER the exchange
(SURVEY.md 8e), and the result must equal the full-batch gradients."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, out_dir):
    for p in (ROOT, os.path.join(ROOT, "dau-convnet_amd"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from dau_conv import distributed as ddp
    from oracle import dau_oracle as orc
    rs = np.random.RandomState(5)
    N, S, F, G, H, W = 5, 3, 4, 2, 9, 10
    x = rs.rand(N, S, H, W).astype(np.float32)
    dy = rs.randn(N, F, H, W).astype(np.float32)
    w = (rs.randn(1, S, G, F) * 0.1).astype(np.float32)
    mu1 = rs.uniform(-3, 3, (1, S, G, F)).astype(np.float32)
    mu2 = rs.uniform(-3, 3, (1, S, G, F)).astype(np.float32)
    lo, hi = ddp.shard_bounds(N, rank, world)
    lr = 3.0
    filt = orc.filters(0.5)

    # the three callables of OverlappedBackward, here from the oracle's building blocks (on the GPU they are
    # Plan.backward_param_sums / Plan.backward(NEED_DX) / Plan.finalize_param_grads)
    def sums_fn(out):
        for k, name in enumerate(("Dw", "Dmu1", "Dmu2", "Dsigma")):
            out[k] = torch.from_numpy(orc.offset_and_dot(orc.blur(x[lo:hi], filt[name]), dy[lo:hi], mu1, mu2)[0])

    def dx_fn():
        return torch.from_numpy(orc.backward(x[lo:hi], dy[lo:hi], w, mu1, mu2, 0.5, need=("dx",))["dx"])

    order = []

    def finalize_fn(sums):
        order.append("finalize")
        wt = torch.from_numpy(w)
        return sums[0:1].clone(), sums[1:2] * wt * lr, sums[2:3] * wt * lr, sums[3:4] * wt

    ex = ddp.OverlappedBackward(w.shape, torch.device("cpu"))
    local = torch.empty_like(ex.sums)
    sums_fn(local)
    dx_shard = ex.run(sums_fn, dx_fn, finalize_fn)
    assert order == [], "finalize must not run before the exchange has been joined"
    red = ex.wait()
    assert order == ["finalize"]
    assert dx_shard.shape == (hi - lo, S, H, W)
    # the buffer that travelled holds the SUM over ranks of the raw sums (not of finalized gradients)
    gathered = [torch.empty_like(local) for _ in range(world)]
    dist.all_gather(gathered, local)
    assert torch.allclose(ex.sums, sum(gathered), rtol=1e-6, atol=1e-7)
    # same result through the plain helper
    again = local.clone()
    ddp.all_reduce_param_sums(again)
    assert torch.equal(again, ex.sums)
    if rank == 0:
        full = orc.backward(x, dy, w, mu1, mu2, 0.5, mu_learning_rate_factor=lr, need=("dw", "dmu1", "dmu2", "dsigma"))
        np.savez(os.path.join(out_dir, "r.npz"), **{k: v.numpy() for k, v in zip(("dw", "dmu1", "dmu2", "dsigma"), red)},
                 **{"full_" + k: full[k] for k in ("dw", "dmu1", "dmu2", "dsigma")})
    dist.barrier()
    dist.destroy_process_group()


def test_shard_bounds_cover_batch():
    from dau_conv.distributed import shard_bounds
    for n in (1, 5, 128, 1024):
        for world in (1, 2, 3, 8):
            spans = [shard_bounds(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(hi - lo for lo, hi in spans) - min(hi - lo for lo, hi in spans) <= 1


@pytest.mark.timeout(120)
def test_two_rank_allreduce_matches_full_batch(tmp_path):
    from util import assert_parity
    port = 29500 + os.getpid() % 2000
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    z = np.load(os.path.join(str(tmp_path), "r.npz"))
    for k in ("dw", "dmu1", "dmu2", "dsigma"):
        assert_parity(z[k], z["full_" + k], k, rel=1e-5, floor=1e-6)
