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
