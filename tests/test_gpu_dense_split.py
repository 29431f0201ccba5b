"""GPU: the densified gather-sum with two-limb f16 operands (DAU_FLAG_DENSE_SPLIT_F16, k_dense_split.hip): for calls whose
offsets lie within +-3 the two gather-sum passes (y and dx) run as an implicit GEMM on the f16 matrix cores over a dense 7 x 7
kernel per channel pair, every operand split into hi + lo binary16 limbs.  Bar: the FP32 one (north star: 1e-4 relative + the
1e-6 floor of SURVEY.md 8d against the oracle) -- the form claims fp32 accuracy, so it gets no bar of its own; the measured
distances go to gpurun_out/parity_margins.jsonl (kept under profiles/ per round).  Replaces the same reference code as the
exact gather (dau_conv_forward_core.hpp:804-1605; tolerance of the reference's own test: dau_conv_test.py:300-333)."""
import numpy as np
import pytest
import torch

from oracle import dau_oracle as orc
from util import assert_parity, make_inputs, record_margins, run_plan

pytestmark = pytest.mark.gpu


def _plan(N, S, F, G, H, W, k=9, extra=0, **kw):
    from dau_conv import _capi
    flags = _capi.FLAG_USE_INTERPOLATION | _capi.FLAG_DENSE_SPLIT_F16 | extra
    return _capi.Plan(N, S, F, G, H, W, max_kernel_size=k, sigma_hint=kw.pop("sigma_hint", 0.5), flags=flags, **kw)


def _oracle(x, dy, w, mu1, mu2, sigma=0.5, **kw):
    want = orc.backward(x, dy, w, mu1, mu2, sigma, **kw)
    want["y"] = orc.forward(x, w, mu1, mu2, sigma, **{k: v for k, v in kw.items() if k == "ignore"})
    return want


def _check(got, want, name):
    for key in ("y", "dx", "dw", "dmu1", "dmu2", "dsigma"):
        assert_parity(got[key], want[key], name + "/" + key)
    return record_margins(name, got, want, "1e-4 rel + 1e-6 max-norm (fp32 bar; split-f16 dense gather-sum)")


@pytest.mark.parametrize("shape", [
    dict(N=2, S=16, F=128, G=4, H=56, W=56),      # one column block of 7 subtiles, whole chunks and channel blocks
    dict(N=3, S=20, F=40, G=3, H=30, W=45),       # ragged everything: channels, rows, columns, odd batch
    dict(N=2, S=7, F=5, G=2, H=9, W=6),           # tiny
    dict(N=2, S=33, F=130, G=6, H=28, W=28),      # two channel blocks, the second almost empty
    dict(N=1, S=16, F=16, G=2, H=20, W=130),      # three column blocks
    dict(N=2, S=32, F=64, G=1, H=17, W=64),       # eight subtiles per block
    dict(N=4, S=24, F=32, G=5, H=14, W=14),       # two subtiles
    dict(N=2, S=8, F=8, G=4, H=27, W=27),         # odd width: the staging kernel's scalar loads
])
def test_split_gather_against_oracle(shape):
    N, S, F, G, H, W = (shape[q] for q in ("N", "S", "F", "G", "H", "W"))
    x, dy, w, mu1, mu2 = make_inputs(43, N, S, F, G, H, W, 9, 3.0)
    mu1.flat[0] = 3.0; mu2.flat[0] = -3.0; mu1.flat[1] = -3.0; mu2.flat[1] = 3.0     # the corners of the 7 x 7 kernel (and the +4 tap of weight 0 it leaves out)
    plan = _plan(N, S, F, G, H, W)
    assert plan.info["gather_dense_split"] == 1
    got = run_plan(plan, x, dy, w, mu1, mu2)
    _check(got, _oracle(x, dy, w, mu1, mu2), "split/%dx%d" % (H, W))


def test_split_gather_hands_over_to_the_exact_kernels_beyond_radius_three():
    """One offset at 3.5: the call's device guard sends it to the exact bucket-4 gather (bit-identical to a plan without the
    flag); back within +-3 the split form runs again.  Which arithmetic a call gets depends on its own offsets only."""
    from dau_conv import _capi
    N, S, F, G, H, W = 2, 16, 32, 4, 24, 24
    x, dy, w, mu1, mu2 = make_inputs(7, N, S, F, G, H, W, 9, 3.0)
    plan = _plan(N, S, F, G, H, W)
    ref = _capi.Plan(N, S, F, G, H, W, max_kernel_size=9, sigma_hint=0.5, flags=_capi.FLAG_USE_INTERPOLATION)
    inside = run_plan(plan, x, dy, w, mu1, mu2)
    exact_inside = run_plan(ref, x, dy, w, mu1, mu2)
    assert not np.array_equal(inside["y"], exact_inside["y"])          # another arithmetic ...
    _check(inside, _oracle(x, dy, w, mu1, mu2), "split/inside")        # ... inside the same bar
    mu1b = mu1.copy(); mu1b.flat[5] = 3.5
    outside = run_plan(plan, x, dy, w, mu1b, mu2)
    exact = run_plan(ref, x, dy, w, mu1b, mu2)
    for key in ("y", "dx"):
        assert np.array_equal(outside[key], exact[key]), key
    again = run_plan(plan, x, dy, w, mu1, mu2)
    for key in ("y", "dx"):
        assert np.array_equal(again[key], inside[key]), key


@pytest.mark.parametrize("scale", [1e-6, 1.0, 3e4])
def test_split_gather_is_scale_free(scale):
    """The limbs are taken after a power-of-two scaling to the tensor's maximum: tiny gradients (the fp16 underflow of
    mixed-precision training) and large activations keep the fp32 bar; scaling the inputs by a power of two scales the outputs
    exactly."""
    N, S, F, G, H, W = 2, 16, 32, 4, 24, 24
    x, dy, w, mu1, mu2 = make_inputs(11, N, S, F, G, H, W, 9, 3.0)
    plan = _plan(N, S, F, G, H, W)
    xs, dys = (x * np.float32(scale)).astype(np.float32), (dy * np.float32(scale)).astype(np.float32)
    got = run_plan(plan, xs, dys, w, mu1, mu2)
    _check(got, _oracle(xs, dys, w, mu1, mu2), "split/scale%g" % scale)
    base = run_plan(plan, x, dy, w, mu1, mu2)
    p2 = run_plan(plan, x * np.float32(1024.0), dy * np.float32(2.0 ** -20), w, mu1, mu2)
    assert np.array_equal(p2["y"], base["y"] * np.float32(1024.0))
    assert np.array_equal(p2["dx"], base["dx"] * np.float32(2.0 ** -20))


def test_split_gather_zero_and_nonfinite_inputs():
    N, S, F, G, H, W = 2, 16, 16, 2, 16, 16
    x, dy, w, mu1, mu2 = make_inputs(3, N, S, F, G, H, W, 9, 3.0)
    plan = _plan(N, S, F, G, H, W)
    got = run_plan(plan, np.zeros_like(x), np.zeros_like(dy), w, mu1, mu2)
    assert not got["y"].any() and not got["dx"].any()
    got = run_plan(plan, x, dy, np.zeros_like(w), mu1, mu2)
    assert not got["y"].any() and not got["dx"].any()
    xn = x.copy(); xn[1, 3, 5, 5] = np.inf
    dev = lambda a: torch.from_numpy(a).cuda()
    sg = torch.full((1, S, G, F), 0.5, device="cuda")
    y = plan.forward(dev(xn), dev(w), dev(mu1), dev(mu2), sg).cpu().numpy()
    # a non-finite input makes the call's scale 1 (one scale per call): the finite image keeps finite results
    assert np.isfinite(y[0]).all() and not np.isfinite(y[1]).all()


@pytest.mark.parametrize("io", ["f32", "bf16"])
def test_split_gather_flags_and_dtypes(io):
    """bf16 activations (exact fp32 arithmetic on them), unit_testing edge rule, one ignored unit, sigma 0.8 (9-tap prefilter)."""
    from dau_conv import _capi
    N, S, F, G, H, W = 2, 12, 20, 4, 32, 32
    x, dy, w, mu1, mu2 = make_inputs(5, N, S, F, G, H, W, 9, 3.0, ignore=1)
    bf = io == "bf16"
    if bf:
        x = torch.from_numpy(x).to(torch.bfloat16).float().numpy()
        dy = torch.from_numpy(dy).to(torch.bfloat16).float().numpy()
    extra = (_capi.FLAG_IO_BF16 if bf else 0) | _capi.FLAG_UNIT_TESTING
    plan = _plan(N, S, F, G, H, W, extra=extra, number_units_ignore=1, sigma_hint=0.8)
    assert plan.info["gather_dense_split"] == 1
    got = run_plan(plan, x, dy, w, mu1, mu2, dtype=torch.bfloat16 if bf else None, sigma=0.8)
    want = _oracle(x, dy, w, mu1, mu2, sigma=0.8, ignore=1, unit_testing=True)
    if bf:     # y, dx are stored as bfloat16: the bf16 storage bar for those two, fp32 for the parameter gradients
        assert_parity(got["y"], want["y"], "split-bf16/y", rel=2e-2, floor=4e-3)
        assert_parity(got["dx"], want["dx"], "split-bf16/dx", rel=2e-2, floor=4e-3)
        for key in ("dw", "dmu1", "dmu2", "dsigma"):
            assert_parity(got[key], want[key], "split-bf16/" + key)
    else:
        _check(got, want, "split/flags")


def test_split_gather_at_north_star_depth():
    """S = F = 256, 56 x 56, G = 4 -- the depth (12 544 dense products x 3 limb pairs per output) of the headline workload, on 8
    images: y and dx against the oracle at the fp32 bar, the margin recorded."""
    N, S, F, G, H, W = 8, 256, 256, 4, 56, 56
    x, dy, w, mu1, mu2 = make_inputs(2024, N, S, F, G, H, W, 9, 3.0)
    plan = _plan(N, S, F, G, H, W)
    dev = lambda a: torch.from_numpy(a).cuda()
    sg = torch.full((1, S, G, F), 0.5, device="cuda")
    y = plan.forward(dev(x), dev(w), dev(mu1), dev(mu2), sg)
    plan.check_status()
    from dau_conv import _capi
    dx = plan.backward(dev(x), dev(dy), dev(w), dev(mu1), dev(mu2), sg, need_mask=_capi.NEED_DX)[0]
    plan.check_status()
    got = dict(y=y.cpu().numpy(), dx=dx.cpu().numpy())
    want = dict(y=orc.forward(x, w, mu1, mu2, 0.5), dx=orc.backward(x, dy, w, mu1, mu2, 0.5, need=("dx",))["dx"])
    for key in ("y", "dx"):
        assert_parity(got[key], want[key], "split/ns-depth/" + key)
    m = record_margins("split/ns-depth N=8 S=F=256 56x56 G=4", got, want, "1e-4 rel + 1e-6 max-norm (fp32 bar)")
    print("margins", m)
