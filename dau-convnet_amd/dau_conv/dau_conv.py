"""Python surface of the DAU convolution operator, PyTorch-ROCm binding.

Mirrors plugins/tensorflow/dau_conv/dau_conv.py of the reference (same class / function /
argument / variable names and defaults) on top of the C ABI in include/dau_conv.h:

    reference                                   here
    ------------------------------------------  -------------------------------------------
    dau_conv_op_module.dau_conv  (:212-219)     dau_conv(...)        raw op, TF attr names
    dau_conv_grad_module.dau_conv_grad          dau_conv_grad(...)   raw grad op
      (_dau_conv_grad_op.py:40-60)
    @ops.RegisterGradient("DAUConv") (:14)      _DAUConvFunction (torch.autograd.Function)
    DAUGridMean (:24-74), ZeroNLast (:76-110)   same names (initializers)
    _DAUConvolution2d (:113-219)                same name
    DAUConv2d (:221-555), DAUConv1d (:557-570)  same names (torch.nn.Module)
    dau_conv2d (:580-690), dau_conv1d (:692-795) same names (scope-keyed functional form)

Parameters keep the reference's names and shapes: `weights`, `mu1`, `mu2` [1,S,G,F],
`sigma` (1,), `bias` (F,)  (dau_conv.py:389-440), so checkpoints map one to one.
There is no CPU / eager fallback: every call goes through libdau_conv_hip.so.
"""
import collections
import math
import warnings

import numpy as np
import torch
import torch.nn as nn

from . import _capi

__all__ = ["DAUGridMean", "ZeroNLast", "DAUConv2d", "DAUConv1d", "dau_conv2d", "dau_conv1d", "dau_conv",
           "dau_conv_grad", "check_pending_offsets"]


# ----------------------------------------------------------------------------------------------
# raw ops (attr names and defaults of REGISTER_OP("DAUConv") / ("DAUConvGrad"),
# plugins/tensorflow/src/dau_conv_op.cpp:22-48, dau_conv_grad_op.cpp:18-49)
# ----------------------------------------------------------------------------------------------
# Plans are cached per (shape, attrs, prefilter support, device) in least-recently-used order.  sigma enters a plan only
# through the prefilter support 2*ceil(5*sigma)+1 (the taps come from the device tensor on every call), so the key holds that
# number, not sigma: a trainable sigma changes every optimizer step and must keep hitting the same plan -- its kernel sets, its
# offset-bucket hint and its pending status.  The cache is bounded; a dropped plan is asked for its status first.
_PLANS = collections.OrderedDict()
_PLAN_CACHE_MAX = 64


def _get_plan(x, w, settings):
    N, S, H, W = x.shape
    _, S2, G, F = w.shape
    if S2 != S:
        raise _capi.InvalidArgumentError("weights dim 1 (%d) must equal the input channels (%d)" % (S2, S))
    if F != settings["num_output"]:
        # shape function of the op: last dim of every parameter == num_output (dau_conv_op.cpp:62-76)
        raise _capi.InvalidArgumentError("last dim of weights (%d) must equal num_output (%d)" % (F, settings["num_output"]))
    if settings["stride"] != 1:
        raise _capi.InvalidArgumentError("DAUConv: only stride=1 is supported")  # base_dau_conv_layer.cpp:68-69
    flags = 0
    if settings["use_interpolation"]:
        flags |= _capi.FLAG_USE_INTERPOLATION
    if settings["unit_testing"]:
        flags |= _capi.FLAG_UNIT_TESTING
    if settings["single_dim_kernel"]:
        flags |= _capi.FLAG_SINGLE_DIM_KERNEL
    if settings["forbid_positive_dim1"]:
        flags |= _capi.FLAG_FORBID_POSITIVE_DIM1
    if x.dtype == torch.bfloat16:
        flags |= _capi.FLAG_IO_BF16      # bfloat16 activations (input, output and their gradients); fp32 parameters
        if settings["dense_bf16"]:
            # calls whose offsets lie within +-4: the gather-sum passes (and, from three units on, the parameter gradients)
            # as densified bf16 MFMA GEMMs
            flags |= _capi.FLAG_DENSE_BF16
    # the two-limb f16 dense gather-sum (fp32 accuracy): None = the library's choice (the radii that pay for this unit count)
    if settings["dense_split"] is True and not (flags & _capi.FLAG_DENSE_BF16):
        flags |= _capi.FLAG_DENSE_SPLIT_F16
    elif settings["dense_split"] is False:
        flags |= _capi.FLAG_NO_DENSE_SPLIT
    key = (N, S, F, G, H, W, settings["kernel_size"], settings["number_units_ignore"], flags, settings["algo"],
           _capi.filter_support(settings["sigma_hint"]), float(settings["mu_learning_rate_factor"]), x.device.index)
    plan = _PLANS.get(key)
    if plan is not None:
        _PLANS.move_to_end(key)
        return plan
    plan = _capi.Plan(N, S, F, G, H, W, max_kernel_size=settings["kernel_size"],
                      number_units_ignore=settings["number_units_ignore"], flags=flags, algo=settings["algo"],
                      sigma_hint=settings["sigma_hint"], mu_learning_rate_factor=settings["mu_learning_rate_factor"],
                      device=x.device)
    if settings["algo"] == _capi.ALGO_AUTO and _capi.ALGO_DIRECT in (plan.info["algo_forward"], plan.info["algo_backward"]):
        # the LDS-tiled kernels refused this shape (e.g. more than 16 units per channel pair under a kernel larger than 17, a
        # prefilter window or error row larger than the LDS): correct, but orders of magnitude slower -- say so once per plan
        which = [n for n, a in (("forward / input gradient", plan.info["algo_forward"]), ("parameter gradients", plan.info["algo_backward"]))
                 if a == _capi.ALGO_DIRECT]
        warnings.warn("DAUConv: N=%d C=%d->%d %dx%d, %d units, max_kernel_size=%d: the %s run on the plain one-thread-per-output "
                      "kernels (the tiled MFMA kernels do not support this shape); expect them to be orders of magnitude slower"
                      % (N, S, F, H, W, G, settings["kernel_size"], " and the ".join(which)), RuntimeWarning, stacklevel=3)
    _PLANS[key] = plan
    while len(_PLANS) > _PLAN_CACHE_MAX:
        _, old = _PLANS.popitem(last=False)
        old.last_status()        # a NaN / out-of-range offset it has seen and not yet reported is raised here, not lost
    return plan


def _check_before(plan, mode):
    """check_offsets="async": before enqueueing a call, look at what the plan's most recent COMPLETED call reported
    (pinned host memory, no sync) and raise the reference's errors for it (NaN -> FailedPrecondition, offsets beyond the
    kernel -> InvalidArgument, dau_conv_op.cpp:250-262) -- one call late, without stalling the stream."""
    if mode == "async":
        plan.last_status()


def _check_after(plan, mode):
    """check_offsets=True: the reference's behaviour -- wait for the call and raise at once (the reference blocks on a
    D2H copy of the amax in every Compute, dau_conv_op.cpp:229-235)."""
    if mode is True:                                  # _settings normalises: "async", True or False, nothing else
        plan.check_status()


def check_pending_offsets(device=None):
    """Wait for the device and raise if any plan's latest call saw a NaN / out-of-range offset (for check_offsets="async"
    users: call at the end of a step or before reading results)."""
    torch.cuda.synchronize(device)
    for plan in list(_PLANS.values()):
        plan.last_status()


def _settings(sigma, number_units_x=2, number_units_y=2, number_units_ignore=0, num_output=64, kernel_size=9, pad=4,
              stride=1, unit_normalization=True, square_unit_normalization=False, mean_iteration_step=1,
              sigma_iteration_step=1, component_border_bound=1.0, sigma_lower_bound=0.3, merge_iteration_step=0,
              merge_threshold=1, unit_testing=False, mu_learning_rate_factor=1.0, single_dim_kernel=False,
              forbid_positive_dim1=False, use_interpolation=True, sigma_hint=None, check_offsets="async",
              algo=_capi.ALGO_AUTO, dense_bf16=False, dense_split=None, process_group=None, grad_reduce="mean", name=None):
    if not unit_normalization or square_unit_normalization:
        raise _capi.InvalidArgumentError("only unit_normalization=True, square_unit_normalization=False is implemented")
    if sigma_hint is None:
        # same host read of sigma[0] the reference performs in LayerSetUp (base_dau_conv_layer.cpp:140-143)
        sigma_hint = float(sigma[(0,) * sigma.dim()].item())
    if grad_reduce not in ("mean", "sum"):
        raise _capi.InvalidArgumentError("grad_reduce must be \"mean\" or \"sum\"")
    # "async" or a truth value (1, 1.0, numpy.bool_ ... count as True): nothing else reaches _check_before / _check_after
    mode = "async" if (isinstance(check_offsets, str) and check_offsets == "async") else bool(check_offsets)
    return dict(number_units_ignore=int(number_units_ignore), num_output=int(num_output), kernel_size=int(kernel_size),
                stride=int(stride), unit_testing=bool(unit_testing), mu_learning_rate_factor=float(mu_learning_rate_factor),
                single_dim_kernel=bool(single_dim_kernel), forbid_positive_dim1=bool(forbid_positive_dim1),
                use_interpolation=bool(use_interpolation), sigma_hint=float(sigma_hint),
                check_offsets=mode, algo=int(algo), dense_bf16=bool(dense_bf16),
                dense_split=(None if dense_split is None else bool(dense_split)), process_group=process_group,
                grad_reduce=grad_reduce)


def _c(t):
    return t.contiguous() if not t.is_contiguous() else t


def dau_conv_grad(grad, input, weights, mu1, mu2, sigma, need_mask=_capi.NEED_ALL, **attrs):
    """DAUConvGrad op -> (grad_input, grad_weights, grad_mu1, grad_mu2, grad_sigma)."""
    attrs.setdefault("component_border_bound", 0.0)  # the only default that differs (dau_conv_grad_op.cpp:37)
    st = _settings(sigma, **attrs)
    plan = _get_plan(input, weights, st)
    _check_before(plan, st["check_offsets"])
    out = plan.backward(_c(input), _c(grad), _c(weights), _c(mu1), _c(mu2), _c(sigma), need_mask)
    _check_after(plan, st["check_offsets"])
    return out


class _DAUConvFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, input, weights, mu1, mu2, sigma, st):
        input, weights, mu1, mu2, sigma = _c(input), _c(weights), _c(mu1), _c(mu2), _c(sigma)
        plan = _get_plan(input, weights, st)
        _check_before(plan, st["check_offsets"])
        y = plan.forward(input, weights, mu1, mu2, sigma)
        _check_after(plan, st["check_offsets"])
        ctx.save_for_backward(input, weights, mu1, mu2, sigma)
        ctx.plan, ctx.st = plan, st
        return y

    @staticmethod
    def backward(ctx, grad):
        input, weights, mu1, mu2, sigma = ctx.saved_tensors
        need = 0
        for i, bit in enumerate((_capi.NEED_DX, _capi.NEED_DW, _capi.NEED_DMU1, _capi.NEED_DMU2, _capi.NEED_DSIGMA)):
            if ctx.needs_input_grad[i]:
                need |= bit
        if need == 0:
            return (None,) * 6
        _check_before(ctx.plan, ctx.st["check_offsets"])
        group = ctx.st["process_group"]
        param_need = need & ~_capi.NEED_DX
        if group is not None and param_need and _group_size(group) > 1:
            out = _data_parallel_backward(ctx.plan, input, _c(grad), weights, mu1, mu2, sigma, need, group,
                                          ctx.st["grad_reduce"])
        else:
            out = ctx.plan.backward(input, _c(grad), weights, mu1, mu2, sigma, need)
        _check_after(ctx.plan, ctx.st["check_offsets"])
        return out + (None,)


def _group_size(group):
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return 1
    return dist.get_world_size(None if group is True else group)


def _data_parallel_backward(plan, input, grad, weights, mu1, mu2, sigma, need, group, grad_reduce):
    """Backward of one batch shard in the order of SURVEY.md 8(e): raw parameter-gradient sums of the shard -> ONE asynchronous
    all-reduce of the flat [4, S, G, F] buffer -> the dx pass runs while the floats travel -> the elementwise tail
    (dmu *= w * lr, dsigma *= w, ignored units, NaN -> 0) on the REDUCED sums, so every rank ends bit-identical and a NaN on one
    rank reaches all of them.  grad_reduce="mean" divides the reduced sums by the group size (what DistributedDataParallel
    does to finalized gradients), "sum" leaves the sum over the global batch."""
    from .distributed import OverlappedBackward
    pg = None if group is True else group
    world = _group_size(group)
    ex = OverlappedBackward(weights.shape, input.device, group=pg)
    dx_fn = (lambda: plan.backward(input, grad, weights, mu1, mu2, sigma, _capi.NEED_DX)[0]) if need & _capi.NEED_DX else None

    def finalize(sums):
        if grad_reduce == "mean":
            sums.mul_(1.0 / world)
        return plan.finalize_param_grads(sums, weights, need_mask=need & ~_capi.NEED_DX)

    dx = ex.run(lambda out: plan.backward_param_sums(input, grad, mu1, mu2, sigma, out=out), dx_fn, finalize)
    return (dx,) + tuple(ex.wait())


def dau_conv(input, weights, mu1, mu2, sigma, **attrs):
    """DAUConv op: output[n,f] = sum_{s,g} w * bilinear(blur_sigma(input[n,s]), . + (mu2, mu1)); differentiable."""
    st = _settings(sigma, **attrs)
    return _DAUConvFunction.apply(input, weights, mu1, mu2, sigma, st)


# ----------------------------------------------------------------------------------------------
# initializers
# ----------------------------------------------------------------------------------------------
class DAUGridMean(object):
    """Initializer placing the DAU offsets on a regular grid, equally spaced in each dimension.

    Args follow the reference (dau_conv.py:24-37): dau_units = units in each direction,
    max_value = limit of the unit positions from the centre, dau_unit_axis 2 => mu1, 1 => mu2.
    The value formula is the reference's (:50).  The reference then tiles with a list indexed
    by the unit COUNT (:61-62), which only produces a grid for some unit shapes; this
    implementation broadcasts along the requested axis, i.e. the documented intent
    (README.md:194-202).  `legacy_values()` returns the bare per-axis value vector.
    """

    def __init__(self, dau_units, max_value, dau_unit_axis=2):
        self.dau_units = tuple(int(u) for u in dau_units)
        self.dau_unit_axis = dau_unit_axis
        self.max_value = max_value

    def legacy_values(self, num_units):
        mv = self.max_value
        return np.arange(num_units) * (2 * mv + 1) / float(num_units) + (-0.5 + (2 * mv + 1) / float(2 * num_units)) - mv

    def __call__(self, shape, dtype=None, partition_info=None):
        assert len(shape) == 4, "DAUGridMean requires input of rank 4 with dims=[1, input_channels, mu1*mu2, out_filters]"
        shape = list(shape)
        separated = shape[2] != self.dau_units[0] * self.dau_units[1]
        if not separated:
            shape = [shape[1], self.dau_units[0], self.dau_units[1], shape[-1]]
        num_units = shape[self.dau_unit_axis]
        vals = self.legacy_values(num_units)
        view = [1, 1, 1, 1]
        view[self.dau_unit_axis] = num_units
        out = np.broadcast_to(vals.reshape(view), shape).astype(np.float32)
        if not separated:
            out = out.reshape(1, shape[0], shape[1] * shape[2], shape[3])
        return torch.from_numpy(np.ascontiguousarray(out))

    def get_config(self):
        return {"dau_units": self.dau_units, "dau_unit_axis": self.dau_unit_axis, "max_value": self.max_value}


class ZeroNLast(object):
    """Wrapper initializer that zeros the last N values along `axis` (dau_conv.py:76-110)."""

    def __init__(self, base_init, last_num_to_zero, axis):
        self.base_init = base_init
        self.last_num_to_zero = last_num_to_zero
        self.axis = axis

    def __call__(self, shape, dtype=None, partition_info=None):
        vals = _run_init(self.base_init, shape)
        idx = [slice(None)] * vals.dim()
        idx[self.axis] = slice(vals.shape[self.axis] - self.last_num_to_zero, None)
        vals[tuple(idx)] = 0
        return vals

    def get_config(self):
        return {"last_num_to_zero": self.last_num_to_zero, "axis": self.axis}


def random_normal_initializer(mean=0.0, stddev=1.0):
    return lambda shape, dtype=None, partition_info=None: torch.randn(*shape) * stddev + mean


def random_uniform_initializer(minval=0.0, maxval=1.0):
    return lambda shape, dtype=None, partition_info=None: torch.rand(*shape) * (maxval - minval) + minval


def constant_initializer(value=0.0):
    return lambda shape, dtype=None, partition_info=None: torch.full(tuple(shape), float(value))


def zeros_initializer():
    return constant_initializer(0.0)


def _run_init(init, shape):
    if torch.is_tensor(init):
        t = init.detach().clone().float()
    elif isinstance(init, (int, float)):
        t = torch.full(tuple(shape), float(init))
    else:
        t = init(tuple(shape))
        if isinstance(t, np.ndarray):
            t = torch.from_numpy(t)
        t = t.detach().clone().float()
    if tuple(t.shape) != tuple(shape):
        raise ValueError("initializer returned shape %s, expected %s" % (tuple(t.shape), tuple(shape)))
    return t


# ----------------------------------------------------------------------------------------------
# layer
# ----------------------------------------------------------------------------------------------
class _DAUConvolution2d(object):
    """Helper that clips the offsets and calls the op (dau_conv.py:113-219)."""

    def __init__(self, input_shape, num_output, dau_units, max_kernel_size, padding, data_format=None, strides=None,
                 num_dau_units_ignore=0, mu_learning_rate_factor=500, dau_unit_border_bound=0.01,
                 dau_unit_sigma_bound=0.01, dau_unit_single_dim=False, dau_aggregation_forbid_positive_dim1=False,
                 dau_mu_interpolation=True, unit_testing=False, name=None, check_offsets="async", algo=_capi.ALGO_AUTO,
                 dense_bf16=False, process_group=None, grad_reduce="mean", dense_split=None):
        if len(input_shape) != 4:
            raise ValueError("Only two dimensional DAUConv supported (rank-4 NCHW input).")
        if data_format is None or data_format == "NHWC":
            raise ValueError("data_format \"NHWC\" not supported - TODO: manually convert to NHWC.")
        if data_format != "NCHW":
            raise ValueError("data_format must be \"NHWC\" or \"NCHW\".")
        if strides is None:
            strides = 1
        if strides > 1:
            raise ValueError("Only strides=1 supported.")
        self.num_output = num_output
        self.padding = padding
        self.name = name
        self.dau_units = dau_units
        self.num_dau_units_ignore = num_dau_units_ignore
        self.max_kernel_size = max_kernel_size
        self.mu_learning_rate_factor = mu_learning_rate_factor
        self.dau_unit_border_bound = dau_unit_border_bound
        self.dau_unit_sigma_bound = dau_unit_sigma_bound
        self.dau_unit_single_dim = dau_unit_single_dim
        self.dau_aggregation_forbid_positive_dim1 = dau_aggregation_forbid_positive_dim1
        self.dau_mu_interpolation = dau_mu_interpolation
        self.unit_testing = unit_testing
        self.check_offsets = check_offsets
        self.algo = algo
        self.dense_bf16 = dense_bf16
        self.dense_split = dense_split
        self.process_group = process_group
        self.grad_reduce = grad_reduce
        self.mean_max_allowed_offset = float(np.floor(self.max_kernel_size / 2.0) - self.dau_unit_border_bound)

    def __call__(self, inp, w, mu1, mu2, sigma, sigma_hint=None):
        # clip in the graph, so clipped units receive zero mu-gradient (dau_conv.py:190-191)
        m = self.mean_max_allowed_offset
        mu1 = torch.clamp(mu1, min=-m, max=m)
        mu2 = torch.clamp(mu2, min=-m, max=m)
        return dau_conv(inp, w, mu1, mu2, sigma,
                        num_output=self.num_output, number_units_x=self.dau_units[0], number_units_y=self.dau_units[1],
                        number_units_ignore=self.num_dau_units_ignore, kernel_size=self.max_kernel_size,
                        pad=int(self.padding), component_border_bound=self.dau_unit_border_bound,
                        sigma_lower_bound=self.dau_unit_sigma_bound, mu_learning_rate_factor=self.mu_learning_rate_factor,
                        single_dim_kernel=self.dau_unit_single_dim,
                        forbid_positive_dim1=self.dau_aggregation_forbid_positive_dim1,
                        use_interpolation=self.dau_mu_interpolation, unit_testing=self.unit_testing,
                        sigma_hint=sigma_hint, check_offsets=self.check_offsets, algo=self.algo,
                        dense_bf16=self.dense_bf16, dense_split=self.dense_split, process_group=self.process_group,
                        grad_reduce=self.grad_reduce, name=self.name)


class DAUConv2d(nn.Module):
    """DAU convolution layer; constructor arguments as in the reference (dau_conv.py:226-258).

    Additions: `in_channels` (build the variables at construction), `algo`, and `check_offsets`: how the op's offset
    preconditions (NaN, |mu| beyond the kernel; dau_conv_op.cpp:250-262) are enforced -- "async" (default): read the
    previous call's on-device result from pinned host memory before each call, no stall, errors surface one call late
    (`dau_conv.check_pending_offsets()` flushes); True: wait for every call and raise at once, as the reference does;
    False: never read the result back.  `dense_split` (None: the library's choice, True: always, False: never): calls whose
    offsets lie within +-2 / +-3 / +-4 run their two gather-sum passes as a densified GEMM on the f16 matrix cores with both
    operands split into two binary16 limbs -- fp32 accuracy (the exact kernels' parity bar), about 1.7x faster at four units per
    channel pair; by default a plan holds the radii that pay for its unit count.  `dense_bf16=True` (bfloat16 inputs only): calls whose offsets lie within +-4 run
    their forward and input-gradient passes as a densified bf16 matrix-core GEMM (DAU_FLAG_DENSE_BF16: taps and blurred
    activations rounded to bf16, fp32 sums) and, from three units per channel on, their
    parameter gradients as dense cross-correlations on the same cores -- the whole step about 2x faster than the exact
    path at six units, at the bf16 tolerance.
    `process_group` (a torch.distributed group, or True for the default one): batch-sharded data parallelism INSIDE the
    layer's backward -- the raw parameter-gradient sums of the local shard are all-reduced (one flat [4,S,G,F] buffer, RCCL
    over xGMI with backend "nccl") while the input gradient is computed, and the elementwise tail runs on the reduced sums;
    weights / mu1 / mu2 / sigma gradients come out identical on every rank, averaged over the ranks (`grad_reduce="mean"`,
    the DistributedDataParallel convention) or summed (`"sum"`).  The bias gradient is ordinary autograd and is not exchanged
    here.  Wrapping such a layer in DistributedDataParallel as well is harmless but redundant (it averages identical values).
    """

    # the reference kernels process units in pairs; odd unit counts get one zero-weight ignored unit
    DAU_UNITS_GROUP = 2

    def __init__(self, filters, dau_units, max_kernel_size, strides=1, data_format='channels_first', activation=None,
                 use_bias=True, weight_initializer=None, mu1_initializer=None, mu2_initializer=None,
                 sigma_initializer=None, bias_initializer=None, weight_regularizer=None, mu1_regularizer=None,
                 mu2_regularizer=None, sigma_regularizer=None, bias_regularizer=None, activity_regularizer=None,
                 weight_constraint=None, mu1_constraint=None, mu2_constraint=None, sigma_constraint=None,
                 bias_constraint=None, trainable=True, mu_learning_rate_factor=500, dau_unit_border_bound=0.01,
                 dau_unit_single_dim=False, dau_aggregation_forbid_positive_dim1=False, dau_sigma_trainable=False,
                 dau_mu_interpolation=True, unit_testing=False, name=None, in_channels=None, check_offsets="async",
                 algo=_capi.ALGO_AUTO, dense_bf16=False, process_group=None, grad_reduce="mean", dense_split=None, **kwargs):
        super(DAUConv2d, self).__init__()
        self.rank = 2
        self.filters = int(filters)
        du = (dau_units, dau_units) if isinstance(dau_units, int) else tuple(dau_units)
        if len(du) != self.rank:
            raise ValueError("dau_components must be an int or a tuple of %d ints" % self.rank)
        self.dau_units = tuple(int(u) for u in du)
        self.max_kernel_size = max_kernel_size
        self.padding = np.floor(self.max_kernel_size / 2.0)
        self.strides = strides
        if data_format in ("channels_first", "NCHW"):
            self.data_format = "channels_first"
        elif data_format in ("channels_last", "NHWC"):
            self.data_format = "channels_last"
        else:
            raise ValueError("The `data_format` argument must be one of \"channels_first\", \"channels_last\".")
        self.activation = activation
        self.use_bias = bool(use_bias)
        self.trainable = trainable
        self.name = name
        self.weight_initializer = weight_initializer if weight_initializer is not None else random_normal_initializer(stddev=0.1)
        self.bias_initializer = bias_initializer if bias_initializer is not None else zeros_initializer()
        self.mu1_initializer = mu1_initializer
        self.mu2_initializer = mu2_initializer
        self.sigma_initializer = sigma_initializer
        self.regularizers = dict(weights=weight_regularizer, mu1=mu1_regularizer, mu2=mu2_regularizer,
                                 sigma=sigma_regularizer, bias=bias_regularizer)
        self.activity_regularizer = activity_regularizer
        self.constraints = dict(weights=weight_constraint, mu1=mu1_constraint, mu2=mu2_constraint,
                                sigma=sigma_constraint, bias=bias_constraint)
        if self.mu1_initializer is None:
            self.mu1_initializer = DAUGridMean(dau_units=self.dau_units, max_value=np.floor(self.max_kernel_size / 2.0) - 1,
                                               dau_unit_axis=2)
        if self.mu2_initializer is None:
            self.mu2_initializer = DAUGridMean(dau_units=self.dau_units, max_value=np.floor(self.max_kernel_size / 2.0) - 1,
                                               dau_unit_axis=1)
        if self.sigma_initializer is None:
            self.sigma_initializer = constant_initializer(0.5)
        self.mu_learning_rate_factor = mu_learning_rate_factor
        self.unit_testing = unit_testing
        self.dau_unit_border_bound = dau_unit_border_bound
        self.num_dau_units_all = int(np.prod(self.dau_units))
        self.num_dau_units_ignore = 0
        self.dau_mu_interpolation = dau_mu_interpolation
        self.dau_unit_single_dim = dau_unit_single_dim
        self.dau_aggregation_forbid_positive_dim1 = dau_aggregation_forbid_positive_dim1
        self.check_offsets = check_offsets
        self.algo = algo
        self.dense_bf16 = dense_bf16
        self.dense_split = dense_split
        self.process_group = process_group
        self.grad_reduce = grad_reduce
        # odd number of units: add one dummy (zero weight, ignored) unit (dau_conv.py:317-329)
        if self.num_dau_units_all % self.DAU_UNITS_GROUP != 0:
            new_num_units = int(np.ceil(self.num_dau_units_all / float(self.DAU_UNITS_GROUP)) * self.DAU_UNITS_GROUP)
            self.num_dau_units_ignore = new_num_units - self.num_dau_units_all
            if self.dau_units[0] < self.dau_units[1]:
                self.dau_units = (self.dau_units[0] + self.num_dau_units_ignore, self.dau_units[1])
            else:
                self.dau_units = (self.dau_units[0], self.dau_units[1] + self.num_dau_units_ignore)
            self.num_dau_units_all = new_num_units
            self.weight_initializer = ZeroNLast(self.weight_initializer, last_num_to_zero=self.num_dau_units_ignore, axis=2)
        self.dau_sigma_trainable = dau_sigma_trainable
        self._manual = {}
        self._sigma_host = None
        self._sigma_tag = None
        self.built = False
        if self.strides > 1:
            warnings.warn('NOTICE: using stride>=2 in DAU convolution uses the same computational resources as with '
                          'stride=1 (current implementation only emulates stride>=2 using tensor slicing).')
        if in_channels is not None:
            self.build((None, int(in_channels), None, None))

    # -- variables --------------------------------------------------------------------------------
    def set_dau_variables_manually(self, w=None, mu1=None, mu2=None, sigma=None):
        """Use caller-provided tensors for w/mu1/mu2/sigma. Call before build() / the first forward."""
        for k, v in (("weights", w), ("mu1", mu1), ("mu2", mu2), ("sigma", sigma)):
            if v is not None:
                self._manual[k] = v

    def _get_input_channel_axis(self):
        if self.data_format == 'channels_first':
            return 1
        raise ValueError('Only `channels_first` supported, i.e., NCHW format.')

    def _get_input_channels(self, input_shape):
        c = input_shape[self._get_input_channel_axis()]
        if c is None:
            raise ValueError('The channel dimension of the inputs should be defined. Found `None`.')
        return int(c)

    def get_dau_variable_shape(self, input_shape):
        return (1, self._get_input_channels(input_shape), self.num_dau_units_all, self.filters)

    def _add(self, name, shape, initializer, trainable=True):
        p = nn.Parameter(_run_init(initializer, shape), requires_grad=bool(trainable and self.trainable))
        self.register_parameter(name, p)
        return p

    def add_dau_weights_var(self, input_shape):
        return self._add('weights', self.get_dau_variable_shape(input_shape), self.weight_initializer)

    def add_dau_mu1_var(self, input_shape):
        return self._add('mu1', self.get_dau_variable_shape(input_shape), self.mu1_initializer)

    def add_dau_mu2_var(self, input_shape):
        return self._add('mu2', self.get_dau_variable_shape(input_shape), self.mu2_initializer)

    def add_dau_sigma_var(self, input_shape, trainable=False):
        # one scalar variable shared by the whole layer (dau_conv.py:417-430)
        return self._add('sigma', (1,), self.sigma_initializer, trainable=trainable)

    def add_bias_var(self):
        return self._add('bias', (self.filters,), self.bias_initializer)

    def build(self, input_shape):
        shape = self.get_dau_variable_shape(input_shape)
        for key, adder in (("weights", self.add_dau_weights_var), ("mu1", self.add_dau_mu1_var),
                           ("mu2", self.add_dau_mu2_var), ("sigma", None)):
            if key in self._manual:
                if tuple(self._manual[key].shape) != tuple(shape):
                    raise ValueError('Shape mismatch for variable `dau_%s`' % key)
            elif adder is not None:
                adder(input_shape)
            else:
                self.add_dau_sigma_var(input_shape, trainable=self.dau_sigma_trainable)
        if self.use_bias:
            self.add_bias_var()
        self._param_shape = shape
        self._dau_convolution_op = _DAUConvolution2d(
            (None, shape[1], None, None), num_output=self.filters, dau_units=self.dau_units,
            max_kernel_size=self.max_kernel_size, padding=self.padding, strides=1,
            num_dau_units_ignore=self.num_dau_units_ignore, mu_learning_rate_factor=self.mu_learning_rate_factor,
            dau_unit_border_bound=self.dau_unit_border_bound, dau_unit_single_dim=self.dau_unit_single_dim,
            dau_aggregation_forbid_positive_dim1=self.dau_aggregation_forbid_positive_dim1,
            dau_mu_interpolation=self.dau_mu_interpolation, unit_testing=self.unit_testing, data_format="NCHW",
            name=self.name, check_offsets=self.check_offsets, algo=self.algo, dense_bf16=self.dense_bf16,
            process_group=self.process_group, grad_reduce=self.grad_reduce, dense_split=self.dense_split)
        self.built = True

    def _var(self, key):
        if key in self._manual:
            return self._manual[key]
        return self._parameters.get(key)

    # the reference exposes the variables as op.dau_weights / dau_mu1 / dau_mu2 / dau_sigma
    dau_weights = property(lambda self: self._var("weights"))
    dau_mu1 = property(lambda self: self._var("mu1"))
    dau_mu2 = property(lambda self: self._var("mu2"))
    dau_sigma = property(lambda self: self._var("sigma"))

    def _sigma_tensor_and_hint(self):
        s = self.dau_sigma
        if s.numel() == 1:
            # tile the scalar to the parameter shape, as the reference graph does (dau_conv.py:429-430)
            # the host copy sizes the prefilter support (2*ceil(5*sigma)+1, base_dau_conv_layer.cpp:146) and keys the plan; it
            # is re-read whenever the tensor has been written since (optimizer step, load_state_dict, constraints,
            # set_dau_variables_manually): the reference re-reads sigma every time it builds its layer (:140-146)
            tag = (s._version, s.data_ptr())
            if self._sigma_host is None or self._sigma_tag != tag:
                self._sigma_host = float(s.detach().reshape(-1)[0].item())
                self._sigma_tag = tag
            return s.reshape(1, 1, 1, 1).expand(self._param_shape), self._sigma_host
        return s, None

    # -- call ---------------------------------------------------------------------------------------
    def forward(self, inputs):
        if not self.built:
            self.build(tuple(inputs.shape))
            self.to(inputs.device)
        if inputs.dim() != self.rank + 2:
            raise ValueError('DAU convolution not supported for input with rank %d' % inputs.dim())
        sigma_t, hint = self._sigma_tensor_and_hint()
        outputs = self._dau_convolution_op(inputs, self.dau_weights, self.dau_mu1, self.dau_mu2, sigma_t, sigma_hint=hint)
        # strides > 1 are emulated by sampling the stride-1 output (dau_conv.py:497-498)
        if self.strides > 1:
            outputs = outputs[:, :, ::self.strides, ::self.strides]
        if self.use_bias:
            outputs = outputs + self._parameters["bias"].reshape(1, self.filters, 1, 1)
        if self.activation is not None:
            return self.activation(outputs)
        return outputs

    call = forward

    def compute_output_shape(self, input_shape):
        n, _, h, w = input_shape
        return (n, self.filters, int(math.ceil(h / float(self.strides))), int(math.ceil(w / float(self.strides))))

    def regularization_loss(self):
        """Sum of the per-variable regularizers passed to the constructor (TF collected these implicitly)."""
        total = 0.0
        for name, reg in self.regularizers.items():
            p = self._var(name)
            if reg is not None and p is not None:
                total = total + reg(p)
        return total

    @torch.no_grad()
    def apply_constraints(self):
        """Project variables with the constraint callables (TF applied them after each optimizer update)."""
        for name, con in self.constraints.items():
            p = self._var(name)
            if con is not None and p is not None:
                p.copy_(con(p))


class DAUConv1d(DAUConv2d):
    """1-D variant: mu2 fixed at zero, blur only along x (dau_conv.py:557-570)."""

    def __init__(self, filters, dau_units, max_kernel_size, **kwargs):
        def mu_zero_constraint(w):
            return torch.zeros_like(w)
        kwargs.pop("mu2_initializer", None)
        kwargs.pop("mu2_regularizer", None)
        kwargs.pop("mu2_constraint", None)
        kwargs.pop("dau_unit_single_dim", None)
        super(DAUConv1d, self).__init__(filters, dau_units, max_kernel_size, mu2_initializer=zeros_initializer(),
                                        mu2_regularizer=None, mu2_constraint=mu_zero_constraint,
                                        dau_unit_single_dim=True, **kwargs)


# ----------------------------------------------------------------------------------------------
# slim-style functional forms; variables live in a scope-keyed registry (TF: variable_scope + reuse)
# ----------------------------------------------------------------------------------------------
_SCOPES = {}
_SCOPE_COUNTER = [0]


def _scoped_layer(scope, reuse, default_name, factory):
    if scope is None:
        _SCOPE_COUNTER[0] += 1
        scope = default_name if _SCOPE_COUNTER[0] == 1 else "%s_%d" % (default_name, _SCOPE_COUNTER[0] - 1)
    if scope in _SCOPES:
        if reuse is False:
            raise ValueError("Variable scope %s already exists, disallowed. Did you mean to set reuse=True?" % scope)
        return _SCOPES[scope]
    if reuse is True:
        raise ValueError("Variable scope %s does not exist, cannot reuse" % scope)
    layer = factory(scope)
    _SCOPES[scope] = layer
    return layer


def get_scope_layer(scope):
    """The DAUConv2d/1d module that dau_conv2d/dau_conv1d created under `scope` (for optimizers / checkpoints)."""
    return _SCOPES[scope]


def dau_conv2d(inputs, filters, dau_units, max_kernel_size, stride=1, mu_learning_rate_factor=500, data_format=None,
               activation_fn=torch.relu, normalizer_fn=None, normalizer_params=None, weights_initializer=None,
               weights_regularizer=None, weights_constraint=None, mu1_initializer=None, mu1_regularizer=None,
               mu1_constraint=None, mu2_initializer=None, mu2_regularizer=None, mu2_constraint=None,
               sigma_initializer=None, sigma_regularizer=None, sigma_constraint=None, biases_initializer=zeros_initializer(),
               biases_regularizer=None, biases_constraint=None, dau_unit_border_bound=0.01, dau_sigma_trainable=False,
               dau_mu_interpolation=True, reuse=None, variables_collections=None, outputs_collections=None,
               trainable=True, scope=None):
    if data_format not in [None, 'NCHW']:
        raise ValueError('Invalid data_format: %r' % (data_format,))
    if inputs.dim() != 4:
        raise ValueError('DAU convolution not supported for input with rank', inputs.dim())
    df = 'channels_first' if data_format and data_format.startswith('NC') else 'channels_last'
    layer = _scoped_layer(scope, reuse, 'DAUConv', lambda name: DAUConv2d(
        filters, dau_units, max_kernel_size, strides=stride, data_format=df, activation=None,
        use_bias=bool(not normalizer_fn and biases_initializer), mu_learning_rate_factor=mu_learning_rate_factor,
        weight_initializer=weights_initializer, mu1_initializer=mu1_initializer, mu2_initializer=mu2_initializer,
        sigma_initializer=sigma_initializer, bias_initializer=biases_initializer, weight_regularizer=weights_regularizer,
        mu1_regularizer=mu1_regularizer, mu2_regularizer=mu2_regularizer, sigma_regularizer=sigma_regularizer,
        bias_regularizer=biases_regularizer, weight_constraint=weights_constraint, mu1_constraint=mu1_constraint,
        mu2_constraint=mu2_constraint, sigma_constraint=sigma_constraint, bias_constraint=biases_constraint,
        dau_unit_border_bound=dau_unit_border_bound, dau_sigma_trainable=dau_sigma_trainable,
        dau_mu_interpolation=dau_mu_interpolation, trainable=trainable, unit_testing=False, name=name))
    outputs = layer(inputs)
    if normalizer_fn is not None:
        outputs = normalizer_fn(outputs, **(normalizer_params or {}))
    if activation_fn is not None:
        outputs = activation_fn(outputs)
    if outputs_collections is not None:
        outputs_collections.append(outputs)
    return outputs


def dau_conv1d(inputs, filters, dau_units, max_kernel_size, stride=1, mu_learning_rate_factor=500, data_format=None,
               activation_fn=torch.relu, normalizer_fn=None, normalizer_params=None, weights_initializer=None,
               weights_regularizer=None, weights_constraint=None, mu1_initializer=None, mu1_regularizer=None,
               mu1_constraint=None, sigma_initializer=None, sigma_regularizer=None, sigma_constraint=None,
               biases_initializer=zeros_initializer(), biases_regularizer=None, dau_unit_border_bound=0.01,
               dau_sigma_trainable=False, dau_aggregation_forbid_positive_dim1=False, dau_mu_interpolation=True,
               reuse=None, variables_collections=None, outputs_collections=None, trainable=True, scope=None):
    if data_format not in [None, 'NCHW']:
        raise ValueError('Invalid data_format: %r' % (data_format,))
    if inputs.dim() != 4:
        raise ValueError('DAU convolution not supported for input with rank', inputs.dim())
    df = 'channels_first' if data_format and data_format.startswith('NC') else 'channels_last'
    layer = _scoped_layer(scope, reuse, 'DAUConv', lambda name: DAUConv1d(
        filters, dau_units, max_kernel_size, strides=stride, data_format=df, activation=None,
        use_bias=bool(not normalizer_fn and biases_initializer), mu_learning_rate_factor=mu_learning_rate_factor,
        weight_initializer=weights_initializer, mu1_initializer=mu1_initializer, sigma_initializer=sigma_initializer,
        bias_initializer=biases_initializer, weight_regularizer=weights_regularizer, mu1_regularizer=mu1_regularizer,
        sigma_regularizer=sigma_regularizer, bias_regularizer=biases_regularizer, weight_constraint=weights_constraint,
        mu1_constraint=mu1_constraint, sigma_constraint=sigma_constraint, dau_unit_border_bound=dau_unit_border_bound,
        dau_sigma_trainable=dau_sigma_trainable,
        dau_aggregation_forbid_positive_dim1=dau_aggregation_forbid_positive_dim1,
        dau_mu_interpolation=dau_mu_interpolation, trainable=trainable, unit_testing=False, name=name))
    outputs = layer(inputs)
    if normalizer_fn is not None:
        outputs = normalizer_fn(outputs, **(normalizer_params or {}))
    if activation_fn is not None:
        outputs = activation_fn(outputs)
    if outputs_collections is not None:
        outputs_collections.append(outputs)
    return outputs
