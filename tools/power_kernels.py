#!/usr/bin/env python3
"""Socket power and shader clock while ONE kind of kernel runs back to back on the north-star shape:
phase 1 the gather-sum (forward calls), phase 2 the gather-dot (backward with only the parameter gradients).
Run on the GPU box from the repo root: python tools/power_kernels.py"""
import os, re, subprocess, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dau-convnet_amd"))
import torch
from dau_conv import _capi

N, S, F, G, H, W = 128, 256, 256, 4, 56, 56
dev = torch.device("cuda", 0)
x = torch.rand(N, S, H, W, device=dev); dy = torch.randn(N, F, H, W, device=dev)
w = torch.randn(1, S, G, F, device=dev) * 0.1
mu1 = (torch.rand(1, S, G, F, device=dev) * 2 - 1) * 3; mu2 = (torch.rand(1, S, G, F, device=dev) * 2 - 1) * 3
sigma = torch.full((1, S, G, F), 0.5, device=dev)
plan = _capi.Plan(N, S, F, G, H, W, max_kernel_size=9, sigma_hint=0.5)
phase, samples, stop = ["idle"], [], [False]

def sampler():
    while not stop[0]:
        out = subprocess.run(["rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True).stdout
        p = re.search(r"Package Power \(W\): ([0-9.]+)", out); c = re.search(r"sclk clock level: \S+ \((\d+)Mhz\)", out)
        if p and c: samples.append((phase[0], float(p.group(1)), int(c.group(1))))
        time.sleep(0.3)

t = threading.Thread(target=sampler); t.start()
need = _capi.NEED_DW | _capi.NEED_DMU1 | _capi.NEED_DMU2 | _capi.NEED_DSIGMA
for name, fn, reps in (("gather_sum", lambda: plan.forward(x, w, mu1, mu2, sigma), 700),
                       ("gather_dot", lambda: plan.backward(x, dy, w, mu1, mu2, sigma, need_mask=need), 350)):
    fn(); torch.cuda.synchronize(); phase[0] = name
    t0 = time.time()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); phase[0] = "idle"
    print("%s: %.2f ms per call" % (name, (time.time() - t0) / reps * 1e3))
    time.sleep(1.0)
stop[0] = True; t.join()
for name in ("gather_sum", "gather_dot"):
    s = [v for v in samples if v[0] == name][2:]      # drop the ramp-up
    if s: print("%s: %d samples, power %.0f W (max %.0f), sclk %.0f MHz" % (name, len(s), sum(v[1] for v in s) / len(s), max(v[1] for v in s), sum(v[2] for v in s) / len(s)))
