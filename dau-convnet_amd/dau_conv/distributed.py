"""Batch-sharded data parallelism for the DAU operator (one process per GPU, torch.distributed).

The path shards over the batch: forward and dx are per-image, the parameter gradients are sums
over n (dau_conv_test.py:173-174).  So the only exchange is one all-reduce(sum) of the flat
buffer of raw parameter-gradient sums [4, S, G, F] per step (backend "nccl" = RCCL over xGMI on MI355X, "gloo" in the
CPU tests).  The reference has no multi-GPU support; this is new.
"""
import torch
import torch.distributed as dist


def shard_bounds(n_total, rank, world):
    """Contiguous, balanced [lo, hi) slice of the batch owned by `rank`."""
    base, rem = divmod(n_total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def all_reduce_param_sums(sums, group=None, async_op=False):
    """Sum the flat raw-sum buffer [4, S, G, F] over all ranks in place (one collective); returns the work handle when
    async_op.  Call Plan.finalize_param_grads on the result, never before."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        return dist.all_reduce(sums.view(-1), op=dist.ReduceOp.SUM, group=group, async_op=async_op)
    return None


class OverlappedBackward(object):
    """Backward of one batch shard with the gradient exchange hidden under the dx pass.

    Order: raw parameter-gradient sums of the shard (gather-dot pass, `sums_fn`) -> ONE asynchronous
    all-reduce of that flat [4, S, G, F] buffer on the collective's own stream -> the dx pass (gather-sum over the
    mirrored error, `dx_fn`) runs while the floats travel over xGMI -> `wait()` joins and only THEN applies the
    elementwise tail (`finalize_fn`: dmu *= w * lr, dsigma *= w, ignored units -> 0, NaN -> 0) to the reduced sums.
    Finalizing after the exchange keeps all ranks bit-identical and lets a NaN on one rank surface on all of them
    instead of being zeroed locally.

      sums_fn(out) -> fills `out` ([4, S, G, F] float32) with the shard's raw sums   (Plan.backward_param_sums)
      dx_fn()      -> dx of the shard                                                 (Plan.backward, NEED_DX)
      finalize_fn(sums) -> (dw, dmu1, dmu2, dsigma)                                   (Plan.finalize_param_grads)

    The CPU tests drive the same class with the oracle through the same three callables.
    """

    def __init__(self, param_shape, device, group=None):
        shape = tuple(int(d) for d in param_shape)
        self.sums = torch.empty((4,) + shape[1:], dtype=torch.float32, device=device)
        self.group = group
        self._work = None
        self._finalize = None
        # measure_exposed(True): event pairs around the join in wait() -- the time the compute stream stands still for the
        # exchange after the dx pass has been enqueued (what of the all-reduce is NOT hidden); read with exposed_ms()
        self._events = None

    def run(self, sums_fn, dx_fn, finalize_fn):
        sums_fn(self.sums)
        if dist.is_available() and dist.is_initialized():
            self._work = dist.all_reduce(self.sums.view(-1), op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self._finalize = finalize_fn
        return dx_fn() if dx_fn is not None else None

    def measure_exposed(self, on=True):
        self._events = [] if on else None

    def exposed_ms(self):
        """(total ms, joins) the compute stream waited in wait() since measure_exposed(True); synchronizes the recorded events."""
        if not self._events:
            return 0.0, 0
        total = 0.0
        for a, b in self._events:
            b.synchronize()
            total += a.elapsed_time(b)
        n = len(self._events)
        self._events = []
        return total, n

    def wait(self):
        """Make the current stream wait for the exchange, then finalize: returns (dw, dmu1, dmu2, dsigma)."""
        if self._work is not None:
            if self._events is not None and self.sums.is_cuda:
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                self._work.wait()
                b.record()
                self._events.append((a, b))
            else:
                self._work.wait()
            self._work = None
        fin, self._finalize = self._finalize, None
        return fin(self.sums)
