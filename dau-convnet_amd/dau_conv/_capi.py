"""ctypes binding of the C ABI in include/dau_conv.h (libdau_conv_hip.so).

This is the only route from Python into the operator: there is no CPU or eager-PyTorch
fallback.  If the HIP library is missing the import fails loudly.
PyTorch is used for device memory and the current HIP stream only.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# DAU_CONV_LIB selects another build of the same ABI (A/B timing of kernel variants on one device)
_LIB_PATH = os.environ.get("DAU_CONV_LIB") or os.path.join(_HERE, "libdau_conv_hip.so")

DAU_OK, DAU_INVALID_ARGUMENT, DAU_FAILED_PRECONDITION, DAU_INTERNAL = 0, 1, 2, 3

FLAG_USE_INTERPOLATION = 1 << 0
FLAG_UNIT_TESTING = 1 << 1
FLAG_SINGLE_DIM_KERNEL = 1 << 2
FLAG_FORBID_POSITIVE_DIM1 = 1 << 3
FLAG_IO_BF16 = 1 << 4   # x, y, dy, dx are torch.bfloat16; parameters and their gradients stay float32
FLAG_STATIC_BUCKET = 1 << 5   # always the kernels of the bucket max_kernel_size allows (no per-call selection)
FLAG_DENSE_BF16 = 1 << 6      # with FLAG_IO_BF16: gather-sum passes of calls with |mu| <= 4 as a densified bf16 MFMA GEMM
FLAG_DENSE_WGRAD_NEVER = 1 << 7    # with FLAG_DENSE_BF16: parameter gradients always through the exact fp32 gather-dot
FLAG_DENSE_WGRAD_ALWAYS = 1 << 8   # with FLAG_DENSE_BF16: dense parameter gradients from one unit per channel on (default: three)
FLAG_DENSE_SPLIT_F16 = 1 << 9      # gather-sum passes of calls with |mu| <= 2 / 3 / 4 as a densified two-limb f16 MFMA GEMM at fp32 accuracy, whatever G
FLAG_NO_DENSE_SPLIT = 1 << 10      # never (default: the radii that pay for the plan's unit count)

ALGO_AUTO, ALGO_DIRECT, ALGO_TILED = 0, 1, 2
PASS_FORWARD, PASS_BACKWARD = 1, 2
NEED_DX, NEED_DW, NEED_DMU1, NEED_DMU2, NEED_DSIGMA, NEED_ALL = 1, 2, 4, 8, 16, 31


class DAUConvError(RuntimeError):
    """Base class; .code holds the DAU_* status."""
    code = DAU_INTERNAL


class InvalidArgumentError(DAUConvError, ValueError):      # TF: errors.InvalidArgumentError
    code = DAU_INVALID_ARGUMENT


class FailedPreconditionError(DAUConvError):                # TF: errors.FailedPreconditionError
    code = DAU_FAILED_PRECONDITION


class InternalError(DAUConvError):                          # TF: errors.InternalError
    code = DAU_INTERNAL


class _Desc(ctypes.Structure):
    _fields_ = [
        ("struct_size", ctypes.c_int32), ("batch", ctypes.c_int32), ("in_channels", ctypes.c_int32),
        ("out_channels", ctypes.c_int32), ("units_per_channel", ctypes.c_int32), ("height", ctypes.c_int32),
        ("width", ctypes.c_int32), ("max_kernel_size", ctypes.c_int32), ("number_units_ignore", ctypes.c_int32),
        ("flags", ctypes.c_int32), ("algo", ctypes.c_int32), ("sigma_hint", ctypes.c_float),
        ("mu_learning_rate_factor", ctypes.c_float),
    ]


class _Info(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in
                ("offset_bucket", "blur_support", "algo_forward", "algo_backward", "drop_last_col", "drop_last_row",
                 "gather_patch", "gather_stack", "dot_windows", "gather_windows", "bucket_sets", "gather_dense_bf16", "batch_slab_gather", "batch_slab_dot", "dot_region", "gather_fblock", "gather_variant", "dense_bf16_radius3", "gather_dense_split")]


def _load():
    if not os.path.exists(_LIB_PATH):
        raise ImportError(
            "dau_conv: %s not found. Build it with `make -C dau-convnet_amd/csrc` (hipcc, gfx950) or "
            "`python -c 'import __graft_entry__ as g; g.build()'`. There is no CPU fallback." % _LIB_PATH)
    lib = ctypes.CDLL(_LIB_PATH)
    vp, fp, ip = ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p
    lib.dau_conv_abi_version.restype = ctypes.c_int
    lib.dau_conv_last_error.restype = ctypes.c_char_p
    lib.dau_conv_build_id.restype = ctypes.c_char_p
    lib.dau_conv_plan_create.argtypes = [ctypes.POINTER(_Desc), ctypes.POINTER(vp)]
    lib.dau_conv_plan_destroy.argtypes = [vp]
    lib.dau_conv_plan_get_info.argtypes = [vp, ctypes.POINTER(_Info)]
    lib.dau_conv_workspace_bytes.argtypes = [vp, ctypes.c_int, ctypes.POINTER(ctypes.c_size_t)]
    lib.dau_conv_forward.argtypes = [vp, vp, fp, fp, fp, fp, fp, fp, vp, ctypes.c_size_t]
    lib.dau_conv_backward.argtypes = [vp, vp] + [fp] * 11 + [vp, ctypes.c_size_t, ctypes.c_int]
    lib.dau_conv_backward_param_sums.argtypes = [vp, vp] + [fp] * 6 + [vp, ctypes.c_size_t]
    lib.dau_conv_finalize_param_grads.argtypes = [vp, vp] + [fp] * 6 + [ctypes.c_int]
    lib.dau_conv_check_status.argtypes = [vp, vp, vp, ctypes.POINTER(ctypes.c_float)]
    lib.dau_conv_last_status.argtypes = [vp, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_int32)]
    lib.dau_conv_filters.argtypes = [vp, vp, fp, fp]
    lib.dau_conv_unit_table.argtypes = [vp, vp, fp, fp, fp, ctypes.c_int, vp]
    lib.dau_conv_filter_support.argtypes = [ctypes.c_float]
    lib.dau_conv_filter_support.restype = ctypes.c_int
    lib.dau_conv_profile_begin.argtypes = [vp]
    lib.dau_conv_profile_end.argtypes = [vp, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int32)]
    if lib.dau_conv_abi_version() != 4:
        raise ImportError("dau_conv: ABI version mismatch in %s" % _LIB_PATH)
    return lib


lib = _load()


def filter_support(sigma):
    """k of the k x k prefilter for this sigma (2*ceil(5*sigma)+1 in float32): all a plan keeps of sigma_hint."""
    return int(lib.dau_conv_filter_support(ctypes.c_float(float(sigma))))


def build_id():
    """Fingerprint of the kernel sources the loaded library was built from (dau_conv_build_id)."""
    return lib.dau_conv_build_id().decode()


def _check(code):
    if code == DAU_OK:
        return
    msg = lib.dau_conv_last_error().decode("utf-8", "replace")
    raise {DAU_INVALID_ARGUMENT: InvalidArgumentError, DAU_FAILED_PRECONDITION: FailedPreconditionError}.get(
        code, InternalError)(msg)


def _ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _stream(device):
    """The stream the library launches on: torch's current stream of the tensors' device (not of the current device)."""
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _same_device(*tensors):
    """All tensors of a call live on one GPU; returns it."""
    dev = tensors[0].device
    for t in tensors[1:]:
        if t is not None and t.device != dev:
            raise InvalidArgumentError("all tensors of a call must be on the same device (%s vs %s)" % (dev, t.device))
    return dev


def _req(t, name, shape=None, dtype=torch.float32):
    if not (t.is_cuda and t.dtype == dtype and t.is_contiguous()):
        raise InvalidArgumentError("%s must be a contiguous %s tensor on the GPU" % (name, str(dtype).replace("torch.", "")))
    if shape is not None and tuple(t.shape) != tuple(shape):
        raise InvalidArgumentError("%s has shape %s, expected %s" % (name, tuple(t.shape), tuple(shape)))
    return t


# One grow-only workspace per (device, stream), shared by every plan: the layers of a network run one after the other on
# a stream, so they can reuse the same scratch memory (a north-star sized layer needs 3 GB for its backward pass).  The
# status word at the head of the workspace is read back by check_status() right after the call that wrote it.
_SHARED_WS = {}


class Plan(object):
    """Owns one dau_conv_plan; scratch memory comes from the per-stream shared workspace."""

    def __init__(self, N, S, F, G, H, W, max_kernel_size=9, number_units_ignore=0, flags=FLAG_USE_INTERPOLATION,
                 algo=ALGO_AUTO, sigma_hint=0.5, mu_learning_rate_factor=1.0, device=None):
        d = _Desc(ctypes.sizeof(_Desc), N, S, F, G, H, W, int(max_kernel_size), int(number_units_ignore), int(flags),
                  int(algo), float(sigma_hint), float(mu_learning_rate_factor))
        self._h = ctypes.c_void_p()
        # a plan belongs to the device that is current when it is created (its kernels' launch attributes and its pinned
        # status mirror are set up there); `device` makes that explicit for callers whose current device is another one
        if device is not None and torch.cuda.is_available():
            with torch.cuda.device(device):
                _check(lib.dau_conv_plan_create(ctypes.byref(d), ctypes.byref(self._h)))
        else:
            _check(lib.dau_conv_plan_create(ctypes.byref(d), ctypes.byref(self._h)))
        self.N, self.S, self.F, self.G, self.H, self.W = N, S, F, G, H, W
        self.io_dtype = torch.bfloat16 if int(flags) & FLAG_IO_BF16 else torch.float32
        info = _Info()
        _check(lib.dau_conv_plan_get_info(self._h, ctypes.byref(info)))
        self.info = {n: getattr(info, n) for n, _ in _Info._fields_}
        self._ws = {}

    def __del__(self):
        h = getattr(self, "_h", None)
        if h and lib is not None:        # at interpreter shutdown the module globals may already be gone
            lib.dau_conv_plan_destroy(h)
            self._h = None

    def workspace_bytes(self, which):
        n = ctypes.c_size_t()
        _check(lib.dau_conv_workspace_bytes(self._h, which, ctypes.byref(n)))
        return n.value

    def _workspace(self, which, device):
        need = self._ws.get(which)
        if need is None:
            need = self._ws[which] = self.workspace_bytes(which)
        key = (device, torch.cuda.current_stream(device).cuda_stream)
        ws = _SHARED_WS.get(key)
        if ws is None or ws.numel() < need:
            ws = _SHARED_WS[key] = torch.empty(need, dtype=torch.uint8, device=device)
        return ws

    def forward(self, x, w, mu1, mu2, sigma):
        pshape = (1, self.S, self.G, self.F)
        _req(x, "input", (self.N, self.S, self.H, self.W), self.io_dtype)
        for t, n in ((w, "weights"), (mu1, "mu1"), (mu2, "mu2"), (sigma, "sigma")):
            _req(t, n, pshape)
        dev = _same_device(x, w, mu1, mu2, sigma)
        # the library launches on the CURRENT device: make that the tensors' device for the duration of the call
        with torch.cuda.device(dev):
            y = torch.empty((self.N, self.F, self.H, self.W), dtype=self.io_dtype, device=dev)
            ws = self._workspace(PASS_FORWARD, dev)
            _check(lib.dau_conv_forward(self._h, _stream(dev), _ptr(x), _ptr(w), _ptr(mu1), _ptr(mu2), _ptr(sigma), _ptr(y),
                                        _ptr(ws), ws.numel()))
        self._last_ws = ws
        return y

    def backward(self, x, dy, w, mu1, mu2, sigma, need_mask=NEED_ALL):
        pshape = (1, self.S, self.G, self.F)
        _req(x, "input", (self.N, self.S, self.H, self.W), self.io_dtype)
        _req(dy, "grad", (self.N, self.F, self.H, self.W), self.io_dtype)
        for t, n in ((w, "weights"), (mu1, "mu1"), (mu2, "mu2"), (sigma, "sigma")):
            _req(t, n, pshape)
        dev = _same_device(x, dy, w, mu1, mu2, sigma)
        new = lambda shape: torch.empty(shape, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            dx = torch.empty(x.shape, dtype=self.io_dtype, device=dev) if need_mask & NEED_DX else None
            dw = new(pshape) if need_mask & NEED_DW else None
            dmu1 = new(pshape) if need_mask & NEED_DMU1 else None
            dmu2 = new(pshape) if need_mask & NEED_DMU2 else None
            dsigma = new(pshape) if need_mask & NEED_DSIGMA else None
            ws = self._workspace(PASS_BACKWARD, dev)
            _check(lib.dau_conv_backward(self._h, _stream(dev), _ptr(x), _ptr(dy), _ptr(w), _ptr(mu1), _ptr(mu2), _ptr(sigma),
                                         _ptr(dx), _ptr(dw), _ptr(dmu1), _ptr(dmu2), _ptr(dsigma), _ptr(ws), ws.numel(),
                                         int(need_mask)))
        self._last_ws = ws
        return dx, dw, dmu1, dmu2, dsigma

    def backward_param_sums(self, x, dy, mu1, mu2, sigma, out=None):
        """Raw parameter-gradient sums [4, S, G, F] (kinds w, mu1, mu2, sigma) of this batch: linear in the batch, so a
        data-parallel caller all-reduces THIS buffer and only then calls finalize_param_grads()."""
        pshape = (1, self.S, self.G, self.F)
        _req(x, "input", (self.N, self.S, self.H, self.W), self.io_dtype)
        _req(dy, "grad", (self.N, self.F, self.H, self.W), self.io_dtype)
        for t, n in ((mu1, "mu1"), (mu2, "mu2"), (sigma, "sigma")):
            _req(t, n, pshape)
        dev = _same_device(x, dy, mu1, mu2, sigma, out)
        with torch.cuda.device(dev):
            if out is None:
                out = torch.empty((4, self.S, self.G, self.F), dtype=torch.float32, device=dev)
            _req(out, "sums", (4, self.S, self.G, self.F))
            ws = self._workspace(PASS_BACKWARD, dev)
            _check(lib.dau_conv_backward_param_sums(self._h, _stream(dev), _ptr(x), _ptr(dy), _ptr(mu1), _ptr(mu2), _ptr(sigma),
                                                    _ptr(out), _ptr(ws), ws.numel()))
        self._last_ws = ws
        return out

    def finalize_param_grads(self, sums, w, need_mask=NEED_DW | NEED_DMU1 | NEED_DMU2 | NEED_DSIGMA):
        """(dw, dmu1, dmu2, dsigma) from (all-reduced) sums: dmu *= w * lr, dsigma *= w, ignored units -> 0, NaN dmu -> 0."""
        pshape = (1, self.S, self.G, self.F)
        _req(sums, "sums", (4, self.S, self.G, self.F))
        _req(w, "weights", pshape)
        dev = _same_device(sums, w)
        new = lambda: torch.empty(pshape, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            dw = new() if need_mask & NEED_DW else None
            dmu1 = new() if need_mask & NEED_DMU1 else None
            dmu2 = new() if need_mask & NEED_DMU2 else None
            dsigma = new() if need_mask & NEED_DSIGMA else None
            _check(lib.dau_conv_finalize_param_grads(self._h, _stream(dev), _ptr(sums), _ptr(w), _ptr(dw), _ptr(dmu1),
                                                     _ptr(dmu2), _ptr(dsigma), int(need_mask)))
        return dw, dmu1, dmu2, dsigma

    def check_status(self):
        """Sync + raise if the last call saw NaN or out-of-bucket offsets; returns max(|mu|)."""
        mx = ctypes.c_float()
        dev = self._last_ws.device
        with torch.cuda.device(dev):
            _check(lib.dau_conv_check_status(self._h, _stream(dev), _ptr(self._last_ws), ctypes.byref(mx)))
        return mx.value

    def last_status(self):
        """No sync: max(|mu|) seen by the most recent COMPLETED call of this plan (None before any has completed);
        raises if that call saw NaN or out-of-bucket offsets.  Calling it before every op call surfaces a bad offset
        one call late without ever stalling the stream."""
        mx, valid = ctypes.c_float(), ctypes.c_int32()
        _check(lib.dau_conv_last_status(self._h, ctypes.byref(mx), ctypes.byref(valid)))
        return mx.value if valid.value else None

    def profile_begin(self):
        _check(lib.dau_conv_profile_begin(self._h))

    def profile_end(self):
        """-> {slot name: (summed ms of the dominant kernel's launches, passes they made up)} since profile_begin()."""
        ms = (ctypes.c_double * 3)()
        n = (ctypes.c_int32 * 3)()
        _check(lib.dau_conv_profile_end(self._h, ms, n))
        return {name: (ms[i], n[i]) for i, name in enumerate(("gather_sum_fwd", "gather_sum_dx", "gather_dot"))}

    def filters(self, sigma):
        k = self.info["blur_support"]
        out = torch.empty((6, k, k), dtype=torch.float32, device=sigma.device)
        with torch.cuda.device(sigma.device):
            _check(lib.dau_conv_filters(self._h, _stream(sigma.device), _ptr(sigma), _ptr(out)))
        return out

    def unit_table(self, mu1, mu2, w=None, form=0):
        """The table the gather kernels consume, from the kernel every call runs first: -> (offsets int32 [units, 2],
        factors float32 [units, 4]).  form 0: [S][G][F] order (w=None: bare factors, the parameter-gradient pass);
        form 1: [F][G][S] order with negated offsets (the input-gradient pass)."""
        units = self.S * self.G * self.F
        raw = torch.empty((units, 6), dtype=torch.int32, device=mu1.device)
        with torch.cuda.device(mu1.device):
            _check(lib.dau_conv_unit_table(self._h, _stream(mu1.device), _ptr(w), _ptr(mu1), _ptr(mu2), int(form), _ptr(raw)))
        return raw[:, :2].contiguous(), raw[:, 2:].contiguous().view(torch.float32)
