"""Batch-sharded data parallelism for the DAU operator (one process per GPU, torch.distributed).

The path shards over the batch: forward and dx are per-image, the parameter gradients are sums
over n (dau_conv_test.py:173-174).  So the only exchange is one all-reduce(sum) of the flat
[dw, dmu1, dmu2, dsigma] buffer per step (backend "nccl" = RCCL over xGMI on MI355X, "gloo" in the
CPU tests).  The reference has no multi-GPU support; this is new.
"""
import torch
import torch.distributed as dist


def shard_bounds(n_total, rank, world):
    """Contiguous, balanced [lo, hi) slice of the batch owned by `rank`."""
    base, rem = divmod(n_total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class GradBucket(object):
    """One flat fp32 buffer holding [dw | dmu1 | dmu2 | dsigma]; reduced with a single collective."""

    def __init__(self, param_shape, device):
        self.numel = 1
        for d in param_shape:
            self.numel *= int(d)
        self.shape = tuple(param_shape)
        self.flat = torch.empty(4 * self.numel, dtype=torch.float32, device=device)

    def pack(self, dw, dmu1, dmu2, dsigma):
        torch.cat([dw.reshape(-1), dmu1.reshape(-1), dmu2.reshape(-1), dsigma.reshape(-1)], out=self.flat)
        return self.flat

    def views(self):
        n = self.numel
        return tuple(self.flat[i * n:(i + 1) * n].view(self.shape) for i in range(4))

    def all_reduce(self, group=None, async_op=False):
        return dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group, async_op=async_op)


def all_reduce_param_grads(dw, dmu1, dmu2, dsigma, bucket=None, group=None):
    """Sum the four parameter-gradient tensors over all ranks; returns the reduced views."""
    if bucket is None:
        bucket = GradBucket(dw.shape, dw.device)
    bucket.pack(dw, dmu1, dmu2, dsigma)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        bucket.all_reduce(group=group)
    return bucket.views()


class OverlappedBackward(object):
    """Backward of one batch shard with the gradient exchange hidden under the dx pass.

    dau_conv_backward is called twice through its need mask: first for [dw, dmu1, dmu2, dsigma] (the gather-dot
    pass), whose flat bucket goes out on the collective's own stream (async all-reduce), then for dx (the gather-sum
    pass over the mirrored error), which runs while the 4*S*G*F floats travel over xGMI.  `wait()` joins the two.

    `backward_fn(need_mask) -> (dx, dw, dmu1, dmu2, dsigma)` is `Plan.backward` bound to its tensors; the CPU tests
    drive the same class with the oracle through the same signature.
    """

    def __init__(self, param_shape, device, group=None):
        self.bucket = GradBucket(param_shape, device)
        self.group = group
        self._work = None

    def run(self, backward_fn, need_dx=True):
        from ._capi import NEED_DX, NEED_DW, NEED_DMU1, NEED_DMU2, NEED_DSIGMA
        _, dw, dmu1, dmu2, dsigma = backward_fn(NEED_DW | NEED_DMU1 | NEED_DMU2 | NEED_DSIGMA)
        self.bucket.pack(dw, dmu1, dmu2, dsigma)
        if dist.is_available() and dist.is_initialized():
            self._work = self.bucket.all_reduce(group=self.group, async_op=True)
        dx = backward_fn(NEED_DX)[0] if need_dx else None
        return dx

    def wait(self):
        """Make the current stream wait for the exchange; returns the reduced (dw, dmu1, dmu2, dsigma) views."""
        if self._work is not None:
            self._work.wait()
            self._work = None
        return self.bucket.views()
