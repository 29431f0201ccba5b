"""One rank of the world-size-2 HIP-path tests (tests/test_gpu_distributed.py starts two of these as child processes that
share GPU 0, backend gloo -- the exchange logic is the one RCCL runs at --gpus N; a 1-GPU box cannot host two RCCL ranks).

    python tests/dist_worker.py <rank> <world> <port> <out_dir>

Part 1 (C ABI): dau_conv_backward_param_sums on the rank's shard_bounds slice -> all_reduce of the raw [4,S,G,F] sums ->
dau_conv_finalize_param_grads, through OverlappedBackward with the dx pass in between.
Part 2 (layer): DAUConv2d(process_group=True) forward + autograd backward on the slice.
Everything is written to <out_dir>/rank<r>.npz; the parent compares with its own single-process full-batch run."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "dau-convnet_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

SHAPE = dict(N=10, S=12, F=40, G=4, H=28, W=28, k=9, m=3.0, lr=25.0, seed=77)


def inputs():
    from util import make_inputs
    c = SHAPE
    return make_inputs(c["seed"], c["N"], c["S"], c["F"], c["G"], c["H"], c["W"], c["k"], c["m"])


def make_layer(process_group, grad_reduce="sum"):
    import dau_conv
    import torch
    c = SHAPE
    _, _, w, mu1, mu2 = inputs()
    layer = dau_conv.DAUConv2d(filters=c["F"], dau_units=(2, 2), max_kernel_size=c["k"], use_bias=False, in_channels=c["S"],
                               mu_learning_rate_factor=c["lr"], dau_sigma_trainable=True, process_group=process_group,
                               grad_reduce=grad_reduce)
    with torch.no_grad():
        layer.weights.copy_(torch.from_numpy(w)); layer.mu1.copy_(torch.from_numpy(mu1)); layer.mu2.copy_(torch.from_numpy(mu2))
    return layer.cuda()


def main():
    rank, world, port, out_dir = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from dau_conv import _capi
    from dau_conv.distributed import OverlappedBackward, shard_bounds
    c = SHAPE
    x, dy, w, mu1, mu2 = inputs()
    lo, hi = shard_bounds(c["N"], rank, world)
    dev = torch.device("cuda", 0)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    xs, dys, wd, m1, m2 = t(x[lo:hi]), t(dy[lo:hi]), t(w), t(mu1), t(mu2)
    sigma = torch.full((1, c["S"], c["G"], c["F"]), 0.5, device=dev)
    out = {}

    # part 1: the C ABI's two-step path, exchange hidden under the dx pass
    plan = _capi.Plan(hi - lo, c["S"], c["F"], c["G"], c["H"], c["W"], max_kernel_size=c["k"], sigma_hint=0.5,
                      mu_learning_rate_factor=c["lr"])
    ex = OverlappedBackward(wd.shape, dev)
    dx = ex.run(lambda buf: plan.backward_param_sums(xs, dys, m1, m2, sigma, out=buf),
                lambda: plan.backward(xs, dys, wd, m1, m2, sigma, need_mask=_capi.NEED_DX)[0],
                lambda sums: plan.finalize_param_grads(sums, wd))
    grads = ex.wait()
    torch.cuda.synchronize()
    out["abi_dx"] = dx.cpu().numpy()
    out["abi_sums"] = ex.sums.cpu().numpy()
    for name, g in zip(("dw", "dmu1", "dmu2", "dsigma"), grads):
        out["abi_" + name] = g.cpu().numpy()

    # part 2: the layer, exchange inside its autograd backward
    for reduce_ in ("sum", "mean"):
        layer = make_layer(process_group=True, grad_reduce=reduce_)
        xin = xs.clone().requires_grad_(True)
        y = layer(xin)
        y.backward(dys)
        torch.cuda.synchronize()
        if reduce_ == "sum":
            out["layer_y"] = y.detach().cpu().numpy()
            out["layer_dx"] = xin.grad.cpu().numpy()
        for name in ("weights", "mu1", "mu2", "sigma"):
            out["layer_%s_%s" % (reduce_, name)] = getattr(layer, name).grad.cpu().numpy()
    out["lo_hi"] = np.array([lo, hi])
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), **out)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
