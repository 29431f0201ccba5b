#!/usr/bin/env python3
"""max|err| / max|want| of y and dx of the gather-sum passes against the oracle at a given depth (default plan):
    DAU_CONV_LIB=<lib> python tools/split_margin.py N S F G H W m"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "dau-convnet_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch
from dau_conv import _capi
from oracle import dau_oracle as orc
from util import make_inputs
N, S, F, G, H, W = (int(v) for v in sys.argv[1:7]); m = float(sys.argv[7])
x, dy, w, mu1, mu2 = make_inputs(2024, N, S, F, G, H, W, 9, m)
plan = _capi.Plan(N, S, F, G, H, W, max_kernel_size=9, sigma_hint=0.5)
dev = lambda a: torch.from_numpy(a).cuda()
sg = torch.full((1, S, G, F), 0.5, device="cuda")
y = plan.forward(dev(x), dev(w), dev(mu1), dev(mu2), sg).cpu().numpy()
dx = plan.backward(dev(x), dev(dy), dev(w), dev(mu1), dev(mu2), sg, need_mask=_capi.NEED_DX)[0].cpu().numpy()
wy = orc.forward(x, w, mu1, mu2, 0.5); wdx = orc.backward(x, dy, w, mu1, mu2, 0.5, need=("dx",))["dx"]
e = lambda g, t: float(np.abs(g.astype(np.float64) - t).max() / np.abs(t).max())
print("%s split_radii=%s  y %.3e  dx %.3e" % (os.environ.get("DAU_CONV_LIB", "default").split("/")[-2:], bin(plan.info["gather_dense_split"]), e(y, wy), e(dx, wdx)))
