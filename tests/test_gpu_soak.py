"""GPU: soak / leak check of the drop-in layer -- the counterpart of the reference's test_DAUConvMemtest
(plugins/tensorflow/tests/dau_conv_test.py:635-682: 10 000 forward + backward runs of one layer, watched for memory growth by
hand).  Here: 2 000 training steps of one DAUConv2d with a TRAINABLE sigma that moves every step, three input shapes taking
turns (three plans), check_offsets="async" (the default: pinned status mirror, no sync); device memory (allocated and reserved),
the host's resident set and the plan cache must be flat after the first hundred steps."""
import importlib
import os
import warnings

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _rss_kb():
    with open("/proc/self/status") as fh:
        for line in fh:
            if line.startswith("VmRSS:"):
                return int(line.split()[1])
    return 0


def test_two_thousand_layer_steps_leave_memory_and_the_plan_cache_flat():
    import dau_conv
    dc = importlib.import_module("dau_conv.dau_conv")
    dc._PLANS.clear()
    torch.manual_seed(5)
    # the reference's memtest layer (32 x 128 x 6 x 6 -> 256 channels, 2 x 1 units, kernel 9) plus two more shapes of the same layer
    shapes = [(32, 128, 6, 6), (8, 128, 14, 14), (4, 128, 28, 20)]
    layer = dau_conv.DAUConv2d(filters=256, dau_units=(2, 1), max_kernel_size=9, use_bias=False, in_channels=128,
                               dau_sigma_trainable=True, dau_unit_border_bound=0.1, mu_learning_rate_factor=10.0).cuda()
    # weights and offsets follow SGD; sigma is trainable (its gradient is computed every step) and is moved by hand, so that it
    # stays inside the prefilter support it started in
    opt = torch.optim.SGD([layer.weights, layer.mu1, layer.mu2], lr=1e-5)
    xs = [torch.rand(s, device="cuda") for s in shapes]
    dys = [torch.randn((s[0], 256, s[2], s[3]), device="cuda") for s in shapes]
    marks = []
    for step in range(2000):
        i = step % 3
        opt.zero_grad(set_to_none=True)
        y = layer(xs[i])
        y.backward(dys[i])
        assert layer.sigma.grad is not None
        layer.sigma.grad = None
        with torch.no_grad():
            # sigma moves every step, inside one prefilter support (2 * ceil(5 sigma) + 1 stays 7): the plan must be kept
            layer.sigma.add_(1e-5 * (1 if (step // 50) % 2 == 0 else -1))
        opt.step()
        if step in (100, 1999):
            torch.cuda.synchronize()
            marks.append(dict(step=step, allocated=torch.cuda.memory_allocated(), reserved=torch.cuda.memory_reserved(),
                              rss_kb=_rss_kb(), plans=len(dc._PLANS)))
    dau_conv.check_pending_offsets()          # nothing bad is pending (raises otherwise)
    a, b = marks
    assert b["plans"] == a["plans"] <= 3, marks
    assert b["allocated"] <= a["allocated"], marks                       # nothing accumulates on the device ...
    assert b["reserved"] <= a["reserved"], marks                         # ... and the caching allocator did not have to grow
    assert b["rss_kb"] - a["rss_kb"] < 32 * 1024, marks                  # host: < 32 MiB over 1900 steps (allocator noise, no trend)
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(out):
        import json
        with open(os.path.join(out, "soak_memory.json"), "w") as fh:
            json.dump(dict(test="2000 DAUConv2d fwd+bwd steps, sigma trainable, three alternating shapes, check_offsets=async", marks=marks), fh)


def test_fallback_to_the_direct_kernels_warns_once_per_plan():
    """A shape the tiled kernels refuse (here: 18 units per channel pair under kernel 33: the window passes of the gather-dot
    place at most 16) runs on the one-thread-per-output kernels; the layer says so when the plan is made, once."""
    import dau_conv
    dc = importlib.import_module("dau_conv.dau_conv")
    dc._PLANS.clear()
    layer = dau_conv.DAUConv2d(filters=8, dau_units=(6, 3), max_kernel_size=33, use_bias=False, in_channels=4).cuda()
    x = torch.rand((2, 4, 24, 24), device="cuda")
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        layer(x).sum().backward()
        layer(x).sum().backward()
        torch.cuda.synchronize()
    msgs = [str(w.message) for w in rec if issubclass(w.category, RuntimeWarning) and "one-thread-per-output" in str(w.message)]
    assert len(msgs) == 1, msgs
    assert "parameter gradients" in msgs[0]
