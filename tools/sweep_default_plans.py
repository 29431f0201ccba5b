#!/usr/bin/env python3
"""Parity sweep of DEFAULT plans (the members the library picks) over layer shapes of real networks: feature maps 7 ... 112 px,
64 ... 512 channels, 2 / 4 / 6 units, offsets within +-2 / +-3 / +-3.99; y and dx (and the parameter gradients at the smaller sizes)
against the oracle at the fp32 bar.  Prints one line per configuration and a summary; exit code 1 on a violation.
    python tools/sweep_default_plans.py            (on the GPU box)"""
import itertools, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "dau-convnet_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch
from dau_conv import _capi
from oracle import dau_oracle as orc
from util import make_inputs, parity_error

bad, worst, n = [], 0.0, 0
t0 = time.time()
for (H, S, F), G, m in itertools.product([(7, 512, 512), (14, 256, 256), (14, 256, 512), (28, 128, 128), (28, 512, 256), (56, 64, 64), (56, 64, 256),
                                          (56, 192, 96), (112, 64, 64), (35, 288, 288), (17, 768, 128), (100, 32, 160)], (2, 4, 6), (2.0, 3.0, 3.99)):
    N = 2 if H * H * S * F * G < 6e9 else 1
    W = H + (3 if H in (35, 17) else 0)
    x, dy, w, mu1, mu2 = make_inputs(int(H * 1000 + S + G), N, S, F, G, H, W, 9, m)
    plan = _capi.Plan(N, S, F, G, H, W, max_kernel_size=9, sigma_hint=0.5)
    dev = lambda a: torch.from_numpy(a).cuda()
    sg = torch.full((1, S, G, F), 0.5, device="cuda")
    y = plan.forward(dev(x), dev(w), dev(mu1), dev(mu2), sg)
    g = plan.backward(dev(x), dev(dy), dev(w), dev(mu1), dev(mu2), sg)
    plan.check_status()
    want = orc.backward(x, dy, w, mu1, mu2, 0.5)
    want["y"] = orc.forward(x, w, mu1, mu2, 0.5)
    got = dict(y=y, dx=g[0], dw=g[1], dmu1=g[2], dmu2=g[3], dsigma=g[4])
    viol = {k: parity_error(got[k].cpu().numpy(), want[k]) for k in got}
    marg = {k: float(np.abs(got[k].cpu().numpy().astype(np.float64) - want[k]).max() / max(np.abs(want[k]).max(), 1e-300)) for k in ("y", "dx")}
    n += 1
    worst = max(worst, marg["y"], marg["dx"])
    ok = all(v <= 0 for v in viol.values())
    print("%-4s N=%d %4d->%-4d %3dx%-3d G=%d m=%.2f members=%s  y %.2e dx %.2e" % ("ok" if ok else "FAIL", N, S, F, H, W, G, m, bin(plan.info["gather_dense_split"]), marg["y"], marg["dx"]), flush=True)
    if not ok:
        bad.append((H, W, S, F, G, m, viol))
print("%d configurations, %d violations, worst y/dx error %.2e of the max-norm, %.0f s" % (n, len(bad), worst, time.time() - t0))
sys.exit(1 if bad else 0)
