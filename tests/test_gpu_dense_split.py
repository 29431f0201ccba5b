"""GPU: the densified gather-sum with two-limb f16 operands (k_dense_split.hip; DAU_FLAG_DENSE_SPLIT_F16 forces it, by default a
plan holds the radii that pay for its unit count): for calls whose offsets lie within +-2 / +-3 / +-4 the two gather-sum passes
(y and dx) run as an implicit GEMM on the f16 matrix cores over a dense 5 x 5 / 7 x 7 / 9 x 9 kernel per channel pair, every
operand split into hi + lo binary16 limbs.  Bar: the FP32 one (north star: 1e-4 relative + the
1e-6 floor of SURVEY.md 8d against the oracle) -- the form claims fp32 accuracy, so it gets no bar of its own; the measured
distances go to gpurun_out/parity_margins.jsonl (kept under profiles/ per round).  Replaces the same reference code as the
exact gather (dau_conv_forward_core.hpp:804-1605; tolerance of the reference's own test: dau_conv_test.py:300-333)."""
import numpy as np
import pytest
import torch

from oracle import dau_oracle as orc
from util import assert_parity, case_kernel_size, golden_cases, load_case, make_inputs, record_margins, run_plan, tuning_capi

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
    dict(N=2, S=16, F=128, G=4, H=56, W=56),      # column blocks of four and of three tiles, whole chunks and channel blocks
    dict(N=3, S=20, F=40, G=3, H=30, W=45),       # ragged everything: channels, rows, columns, odd batch; blocks of 3 + 3
    dict(N=2, S=7, F=5, G=2, H=9, W=6),           # tiny: one block of one tile
    dict(N=2, S=33, F=130, G=6, H=28, W=28),      # two channel blocks, the second almost empty
    dict(N=1, S=16, F=16, G=2, H=20, W=130),      # seventeen tiles: one block of five... of 4, 4, 3, 3, 3
    dict(N=2, S=32, F=64, G=1, H=17, W=64),       # two blocks of four
    dict(N=4, S=24, F=32, G=5, H=14, W=14),       # one block of two
    dict(N=2, S=8, F=8, G=4, H=27, W=27),         # odd width: the staging kernel's scalar loads
])
@pytest.mark.parametrize("radius", [2, 3, 4])
def test_split_gather_against_oracle(shape, radius):
    N, S, F, G, H, W = (shape[q] for q in ("N", "S", "F", "G", "H", "W"))
    r = float(radius)
    x, dy, w, mu1, mu2 = make_inputs(43 + radius, N, S, F, G, H, W, 9, r)
    # the corners of the (2r+1)^2 kernel (and the +r+1 tap of weight 0 it leaves out); radius 4: the layer's clip, 3.99
    c = min(r, 3.99)
    mu1.flat[0] = c; mu2.flat[0] = -c; mu1.flat[1] = -c; mu2.flat[1] = c
    assert max(np.abs(mu1).max(), np.abs(mu2).max()) > radius - 1       # this call belongs to the member of THIS radius
    plan = _plan(N, S, F, G, H, W)
    assert plan.info["gather_dense_split"] == 0b11100
    got = run_plan(plan, x, dy, w, mu1, mu2)
    _check(got, _oracle(x, dy, w, mu1, mu2), "split/r%d/%dx%d" % (radius, H, W))


@pytest.mark.parametrize("shape", [
    dict(N=2, S=16, F=130, G=4, H=28, W=28),      # 24 + 4 rows; two column halves of two tiles
    dict(N=3, S=20, F=40, G=3, H=27, W=45),       # 24 + 3 rows; blocks of three tiles: the second column half owns one
    dict(N=2, S=7, F=5, G=2, H=3, W=6),           # the four-row block alone, one tile: the second column half is idle
    dict(N=2, S=16, F=16, G=2, H=12, W=14),       # 8 + 4 rows, two tiles
    dict(N=1, S=16, F=16, G=2, H=20, W=130),      # 16 + 4 rows, blocks of 4 / 3 tiles
    dict(N=2, S=8, F=8, G=4, H=9, W=27),          # 8 + 1 rows
])
@pytest.mark.parametrize("radius", [2, 3, 4])
def test_split_gather_block_of_four_rows(shape, radius, monkeypatch):
    """The last 1 .. 4 rows of a map as a block of FOUR rows (split_gather_kernel<NSUB, RG = 1>: waves split the columns instead of
    the rows).  Production takes that form only where a pass is many rounds of workgroups long (BASELINE config 3); the tuning build
    forces it (DAU_SPLIT_ROWS4=2) on shapes the oracle finishes in no time."""
    capi = tuning_capi()
    monkeypatch.setenv("DAU_SPLIT_ROWS4", "2")
    N, S, F, G, H, W = (shape[q] for q in ("N", "S", "F", "G", "H", "W"))
    r = float(radius)
    x, dy, w, mu1, mu2 = make_inputs(143 + radius, N, S, F, G, H, W, 9, r)
    c = min(r, 3.99)
    mu1.flat[0] = c; mu2.flat[0] = -c; mu1.flat[1] = -c; mu2.flat[1] = c
    plan = capi.Plan(N, S, F, G, H, W, max_kernel_size=9, sigma_hint=0.5, flags=capi.FLAG_USE_INTERPOLATION | capi.FLAG_DENSE_SPLIT_F16)
    got = run_plan(plan, x, dy, w, mu1, mu2)
    want = _oracle(x, dy, w, mu1, mu2)
    for key in ("y", "dx", "dw", "dmu1", "dmu2", "dsigma"):
        assert_parity(got[key], want[key], "split/rows4/r%d/%dx%d/%s" % (radius, H, W, key))
    monkeypatch.setenv("DAU_SPLIT_ROWS4", "0")      # the same call through eight-row blocks only: the same sums in the same order
    plan8 = capi.Plan(N, S, F, G, H, W, max_kernel_size=9, sigma_hint=0.5, flags=capi.FLAG_USE_INTERPOLATION | capi.FLAG_DENSE_SPLIT_F16)
    ref = run_plan(plan8, x, dy, w, mu1, mu2)
    for key in ("y", "dx"):
        assert np.array_equal(got[key], ref[key]), key


@pytest.mark.parametrize("shape", [
    dict(N=2, S=16, F=16, G=4, H=28, W=28),       # config 3's map: 3 blocks of 8 rows x 7 tall tiles + the block of four rows
    dict(N=2, S=7, F=5, G=2, H=13, W=27),         # 7 tiles, the last one three columns wide; 8 + 5 rows
    dict(N=1, S=16, F=130, G=2, H=8, W=25),       # 7 tiles, one live column in the last; two channel blocks
    dict(N=2, S=8, F=8, G=4, H=16, W=20),         # 5 tiles (3 + 2)
    dict(N=2, S=24, F=8, G=3, H=11, W=17),        # 5 tiles, one live column in the last; 8 + 3 rows
])
@pytest.mark.parametrize("rows4", ["0", "2"])
@pytest.mark.parametrize("radius", [2, 3, 4])
def test_split_gather_tall_tiles(shape, radius, rows4, monkeypatch):
    """Tiles of 8 rows x 4 columns (split_gather_kernel<NSUB, 2, TT = true>) for widths that are 1 .. 4 columns more than a
    multiple of eight: a 28-pixel row is seven tiles instead of four padded ones.  Production takes them where they compute at
    least 5 % fewer tile positions (BASELINE config 3); the tuning build forces them (DAU_SPLIT_TALL=2), with and without the
    block of four rows beside them."""
    capi = tuning_capi()
    monkeypatch.setenv("DAU_SPLIT_TALL", "2")
    monkeypatch.setenv("DAU_SPLIT_ROWS4", rows4)
    N, S, F, G, H, W = (shape[q] for q in ("N", "S", "F", "G", "H", "W"))
    r = float(radius)
    x, dy, w, mu1, mu2 = make_inputs(151 + radius, N, S, F, G, H, W, 9, r)
    c = min(r, 3.99)
    mu1.flat[0] = c; mu2.flat[0] = -c; mu1.flat[1] = -c; mu2.flat[1] = c
    flags = capi.FLAG_USE_INTERPOLATION | capi.FLAG_DENSE_SPLIT_F16
    got = run_plan(capi.Plan(N, S, F, G, H, W, max_kernel_size=9, sigma_hint=0.5, flags=flags), x, dy, w, mu1, mu2)
    want = _oracle(x, dy, w, mu1, mu2)
    for key in ("y", "dx", "dw", "dmu1", "dmu2", "dsigma"):
        assert_parity(got[key], want[key], "split/tall/r%d/%dx%d/%s" % (radius, H, W, key))
    monkeypatch.setenv("DAU_SPLIT_TALL", "0")       # the same call through 4 x 8 tiles: the same sums in the same order
    ref = run_plan(capi.Plan(N, S, F, G, H, W, max_kernel_size=9, sigma_hint=0.5, flags=flags), x, dy, w, mu1, mu2)
    for key in ("y", "dx"):
        assert np.array_equal(got[key], ref[key]), key


@pytest.mark.parametrize("case", [
    # (S, F, G, H, W) -> the radii a default plan holds: those that pay for the unit count on this tiling (split_pays, dau_conv_api.hip)
    ((256, 256, 4, 56, 56), 0b11100), ((256, 256, 6, 56, 56), 0b11100), ((256, 256, 2, 56, 56), 0b00100),
    ((256, 256, 1, 56, 56), 0), ((96, 256, 4, 27, 27), 0b01100), ((512, 512, 4, 28, 28), 0b11100), ((7, 5, 4, 16, 16), 0),
])
def test_default_plans_hold_the_radii_that_pay(case):
    from dau_conv import _capi
    (S, F, G, H, W), want = case
    plan = _capi.Plan(2, S, F, G, H, W, max_kernel_size=9, sigma_hint=0.5)
    assert plan.info["gather_dense_split"] == want, bin(plan.info["gather_dense_split"])
    never = _capi.Plan(2, S, F, G, H, W, max_kernel_size=9, sigma_hint=0.5, flags=_capi.FLAG_USE_INTERPOLATION | _capi.FLAG_NO_DENSE_SPLIT)
    assert never.info["gather_dense_split"] == 0
    with pytest.raises(_capi.InvalidArgumentError):
        _capi.Plan(2, S, F, G, H, W, flags=_capi.FLAG_USE_INTERPOLATION | _capi.FLAG_NO_DENSE_SPLIT | _capi.FLAG_DENSE_SPLIT_F16)


def test_split_members_and_the_exact_kernels_share_a_plan():
    """A kernel-17 plan: offsets within +-2, +-3, +-4 take the dense members of those radii, an offset of 5 the exact bucket-8
    gather (bit-identical to a plan that has no dense member) -- decided per call on the device from the call's own offsets, so a
    result never depends on what the plan saw before."""
    from dau_conv import _capi
    N, S, F, G, H, W = 2, 16, 32, 4, 24, 24
    plan = _plan(N, S, F, G, H, W, k=17)
    ref = _capi.Plan(N, S, F, G, H, W, max_kernel_size=17, sigma_hint=0.5, flags=_capi.FLAG_USE_INTERPOLATION | _capi.FLAG_NO_DENSE_SPLIT)
    assert plan.info["gather_dense_split"] == 0b11100 and ref.info["gather_dense_split"] == 0
    first = {}
    for rnd in range(2):
        for m in (2.0, 5.0, 3.0, 3.99):
            x, dy, w, mu1, mu2 = make_inputs(7, N, S, F, G, H, W, 17, m)
            mu1.flat[3] = m
            got = run_plan(plan, x, dy, w, mu1, mu2)
            exact = run_plan(ref, x, dy, w, mu1, mu2)
            if m > 4:
                for key in ("y", "dx"):
                    assert np.array_equal(got[key], exact[key]), (m, key)
            else:
                assert not np.array_equal(got["y"], exact["y"])                 # another arithmetic ...
                _check(got, _oracle(x, dy, w, mu1, mu2), "split/k17/m%g" % m)    # ... inside the same bar
            if rnd == 0:
                first[m] = got
            else:
                for key in ("y", "dx"):
                    assert np.array_equal(got[key], first[m][key]), (m, key)


@pytest.mark.parametrize("name", golden_cases())
def test_golden_vectors_through_the_dense_members(name):
    """The golden cases of the reference's own numpy oracle (tests/golden: arrays the reference class returned) whose offsets lie
    within +-4, with the dense members forced: forward + all five gradients at the bar of the golden test of test_gpu_parity.py
    (the files carry the numpy oracle's float32 accumulation error: floor 3e-6)."""
    from dau_conv import _capi
    c = load_case(name)
    if float(np.abs(c["mu1"]).max()) > 4.0 or float(np.abs(c["mu2"]).max()) > 4.0 or float(c["sigma"]) > 1.0:
        pytest.skip("offsets beyond +-4 (or a prefilter wider than 11 taps): the exact kernels' case")
    N, S, H, W = c["x"].shape
    _, _, G, F = c["w"].shape
    fl = _capi.FLAG_DENSE_SPLIT_F16
    if int(c["use_interpolation"]): fl |= _capi.FLAG_USE_INTERPOLATION
    if int(c["unit_testing"]): fl |= _capi.FLAG_UNIT_TESTING
    if int(c["single_dim_kernel"]): fl |= _capi.FLAG_SINGLE_DIM_KERNEL
    if int(c["forbid_positive_dim1"]): fl |= _capi.FLAG_FORBID_POSITIVE_DIM1
    plan = _capi.Plan(N, S, F, G, H, W, max_kernel_size=case_kernel_size(c), number_units_ignore=int(c["ignore"]), flags=fl,
                      sigma_hint=float(c["sigma"]))
    assert plan.info["gather_dense_split"] == 0b11100
    got = run_plan(plan, c["x"], c["dy"], c["w"], c["mu1"], c["mu2"], sigma=float(c["sigma"]))
    for key in ("y", "dx", "dw", "dmu1", "dmu2", "dsigma"):
        assert_parity(got[key], c[key], "golden/%s/split/%s" % (name, key), rel=1e-4, floor=3e-6)


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


def test_split_gather_heavy_tailed_inputs():
    """Activations after a ReLU (half of them exactly zero, a few three orders of magnitude above the rest) and gradients with the
    same kind of outliers: the one scale per call is set by the outliers, the bulk sits 2^-10 below it and must keep its bits (the
    second limb is then a binary16 subnormal for part of the data)."""
    N, S, F, G, H, W = 2, 32, 32, 4, 24, 24
    x, dy, w, mu1, mu2 = make_inputs(17, N, S, F, G, H, W, 9, 3.0)
    rs = np.random.RandomState(99)
    x = np.maximum(x - 0.5, 0.0).astype(np.float32)
    x[rs.rand(*x.shape) < 1e-3] *= np.float32(1000.0)
    dy = (dy * 1e-4).astype(np.float32)
    dy[rs.rand(*dy.shape) < 1e-3] *= np.float32(1000.0)
    plan = _plan(N, S, F, G, H, W)
    got = run_plan(plan, x, dy, w, mu1, mu2)
    want = _oracle(x, dy, w, mu1, mu2)
    _check(got, want, "split/heavy-tailed")
    # the bulk of the outputs (away from the outliers' footprints) must be as accurate as without them: compare with the run
    # whose outliers are removed, on the outputs that do not change by more than rounding
    calm = want["y"][np.abs(want["y"]) < np.percentile(np.abs(want["y"]), 90)]
    err = (got["y"].astype(np.float64) - want["y"])[np.abs(want["y"]) < np.percentile(np.abs(want["y"]), 90)]
    assert np.abs(err).max() <= 1e-4 * np.abs(calm).max() + 1e-6 * np.abs(want["y"]).max()


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
    assert plan.info["gather_dense_split"] == 0b11100
    got = run_plan(plan, x, dy, w, mu1, mu2, dtype=torch.bfloat16 if bf else None, sigma=0.8)
    want = _oracle(x, dy, w, mu1, mu2, sigma=0.8, ignore=1, unit_testing=True)
    if bf:     # y, dx are stored as bfloat16: the bf16 storage bar for those two, fp32 for the parameter gradients
        assert_parity(got["y"], want["y"], "split-bf16/y", rel=2e-2, floor=4e-3)
        assert_parity(got["dx"], want["dx"], "split-bf16/dx", rel=2e-2, floor=4e-3)
        for key in ("dw", "dmu1", "dmu2", "dsigma"):
            assert_parity(got[key], want[key], "split-bf16/" + key)
    else:
        _check(got, want, "split/flags")


def _gather_passes(plan, x, dy, w, mu1, mu2):
    from dau_conv import _capi
    dev = lambda a: torch.from_numpy(a).cuda()
    S, G, F = w.shape[1:]
    sg = torch.full((1, S, G, F), 0.5, device="cuda")
    y = plan.forward(dev(x), dev(w), dev(mu1), dev(mu2), sg)
    plan.check_status()
    dx = plan.backward(dev(x), dev(dy), dev(w), dev(mu1), dev(mu2), sg, need_mask=_capi.NEED_DX)[0]
    plan.check_status()
    return dict(y=y.cpu().numpy(), dx=dx.cpu().numpy())


@pytest.mark.parametrize("cfg", [
    # name, (N, S, F, G, H, W), offsets within: the depth of the BASELINE workloads on a few images, DEFAULT plans (no flag)
    ("ns-depth N=8 S=F=256 56x56 G=4 r3", (8, 256, 256, 4, 56, 56), 3.0),
    ("ns-depth N=4 S=F=256 56x56 G=4 r2", (4, 256, 256, 4, 56, 56), 2.0),
    ("c2-depth N=4 S=F=256 56x56 G=6 r4", (4, 256, 256, 6, 56, 56), 3.99),
    ("c3-depth N=8 S=F=512 28x28 G=4 r3", (8, 512, 512, 4, 28, 28), 3.0),
    ("c1 N=64 96->256 27x27 G=4 r3", (64, 96, 256, 4, 27, 27), 3.0),
])
def test_split_gather_at_baseline_depth(cfg):
    """y and dx at the depth (input channels x taps x 3 limb pairs per output) of the BASELINE workloads against the oracle at the
    fp32 bar, through the plan a caller gets by default; the margins go on record."""
    from dau_conv import _capi
    name, (N, S, F, G, H, W), m = cfg
    x, dy, w, mu1, mu2 = make_inputs(2024, N, S, F, G, H, W, 9, m)
    plan = _capi.Plan(N, S, F, G, H, W, max_kernel_size=9, sigma_hint=0.5)
    radius = 2 if m <= 2 else 3 if m <= 3 else 4
    assert plan.info["gather_dense_split"] & (1 << radius), "the default plan of this workload holds the radius-%d member" % radius
    got = _gather_passes(plan, x, dy, w, mu1, mu2)
    want = dict(y=orc.forward(x, w, mu1, mu2, 0.5), dx=orc.backward(x, dy, w, mu1, mu2, 0.5, need=("dx",))["dx"])
    mg = record_margins("split/" + name, got, want, "1e-4 rel + 1e-6 max-norm (fp32 bar)")
    print("margins", name, mg)
    for key in ("y", "dx"):
        assert_parity(got[key], want[key], "split/%s/%s" % (name, key))
