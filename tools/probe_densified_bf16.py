#!/usr/bin/env python3
"""One measured data point for SURVEY.md section 7 hard part (A): the DENSIFIED bf16 formulation of the forward pass.

The G units of every (s, f) pair are scattered into a dense (2R+2) x (2R+2) kernel (R = 4: 10 x 10 taps: integer offsets
-4..4 plus the second bilinear tap) and the forward pass becomes an ordinary convolution of the Gaussian-blurred input,
M = F, K = S * 100, N = pixels -- 100 / (4 G) times the gather's FLOPs, but on the bf16 matrix cores (16 x the fp32
rate).  This probe runs that convolution through the library (MIOpen via torch.nn.functional.conv2d, bf16 in, fp32
accumulate), i.e. the cheapest way to get a real number for the alternative; it is a measurement tool, not part of the
product path.  It prints one JSON line: times of the densified conv (and of its preparation) next to the gather-sum of
this repository on the same inputs, and the parity of both against the CPU oracle on bf16-rounded inputs.

    python tools/probe_densified_bf16.py [--workload c2|ns] [--images N]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, "dau-convnet_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np
import torch
import torch.nn.functional as Fn


def densify(w, mu1, mu2, R):
    """[1,S,G,F] units -> dense kernel [F, S, 2R+2, 2R+2] (fp32), tap (oy+dy+R, ox+dx+R) += w * b_dydx."""
    _, S, G, F = w.shape
    K = 2 * R + 2
    ox, oy = torch.floor(mu1), torch.floor(mu2)
    fx, fy = mu1 - ox, mu2 - oy
    dense = torch.zeros((F, S, K * K), device=w.device, dtype=torch.float32)
    f_idx = torch.arange(F, device=w.device).view(1, 1, 1, F).expand(1, S, G, F)
    s_idx = torch.arange(S, device=w.device).view(1, S, 1, 1).expand(1, S, G, F)
    for dy in (0, 1):
        for dx in (0, 1):
            b = (fx if dx else 1 - fx) * (fy if dy else 1 - fy)
            tap = ((oy + dy + R) * K + (ox + dx + R)).long()
            dense.index_put_((f_idx.reshape(-1), s_idx.reshape(-1), tap.reshape(-1)), (w * b).reshape(-1), accumulate=True)
    return dense.view(F, S, K, K)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="c2", choices=["c2", "ns"])
    ap.add_argument("--images", type=int, default=128)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--radius", type=int, default=4, help="offset bucket R: 4 (kernel 9, 10 x 10 dense taps) or 8 (kernel 17, 18 x 18)")
    ap.add_argument("--units", type=int, default=0, help="units per channel (default: the workload's)")
    args = ap.parse_args()
    from dau_conv import _capi
    from oracle import dau_oracle as orc
    dev = torch.device("cuda", 0)
    N, S, F, H, W, R = args.images, 256, 256, 56, 56, args.radius
    G = args.units or (6 if args.workload == "c2" else 4)
    g = torch.Generator(device=dev); g.manual_seed(7)
    x = torch.rand((N, S, H, W), device=dev, generator=g).to(torch.bfloat16)
    w = torch.randn((1, S, G, F), device=dev, generator=g) * 0.1
    mu1 = ((torch.rand((1, S, G, F), device=dev, generator=g) * 2 - 1) * (R - 1.0))
    mu2 = ((torch.rand((1, S, G, F), device=dev, generator=g) * 2 - 1) * (R - 1.0))
    sigma = torch.full((1, S, G, F), 0.5, device=dev)

    def timed(fn, steps):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(steps):
            out = fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / steps, out

    # --- this repository: gather-sum on the fp32 matrix cores, bf16 activations in HBM -------------------------------
    plan = _capi.Plan(N, S, F, G, H, W, max_kernel_size=2 * R + 1, sigma_hint=0.5,
                      flags=_capi.FLAG_USE_INTERPOLATION | _capi.FLAG_IO_BF16)
    t_gather, y_gather = timed(lambda: plan.forward(x, w, mu1, mu2, sigma), args.steps)
    plan.profile_begin()
    for _ in range(args.steps):
        plan.forward(x, w, mu1, mu2, sigma)
    prof = plan.profile_end()
    t_gather_kernel = prof["gather_sum_fwd"][0] / max(prof["gather_sum_fwd"][1], 1)

    # --- densified alternative ---------------------------------------------------------------------------------------
    filt = plan.filters(sigma)[0]                                    # Gn, k x k, from the same synthesis kernel
    k = filt.shape[0]
    gk = filt.view(1, 1, k, k).expand(S, 1, k, k).contiguous()

    def prepare():
        dense = densify(w, mu1, mu2, R).to(torch.bfloat16)
        xb = Fn.conv2d(x.float(), gk, padding=k // 2, groups=S)      # depthwise Gaussian, zero padded
        xp = Fn.pad(xb, (R, R + 1, R, R + 1)).to(torch.bfloat16)      # offsets -R .. R+1
        return dense, xp

    t_prep, (dense, xp) = timed(prepare, 3)
    res = {}
    for fmt_name, fmt in (("nchw", torch.contiguous_format), ("nhwc", torch.channels_last)):
        try:
            torch.backends.cudnn.benchmark = True
            xpf, df = xp.contiguous(memory_format=fmt), dense.contiguous(memory_format=fmt)
            t_conv, y_dense = timed(lambda: Fn.conv2d(xpf, df), args.steps)
            res[fmt_name] = (t_conv, y_dense)
        except Exception as e:                                       # the library may have no solver for a layout
            res[fmt_name] = (None, repr(e)[:200])
    best = min((v for v in res.values() if v[0] is not None), key=lambda v: v[0], default=(None, None))

    # --- parity of both against the oracle on the bf16-rounded inputs (first two images) -------------------------------
    nchk = min(2, N)
    want = orc.forward(x[:nchk].float().cpu().numpy(), w.cpu().numpy(), mu1.cpu().numpy(), mu2.cpu().numpy(), 0.5)

    def viol(got):
        got = got[:nchk].float().cpu().numpy().astype(np.float64)
        return float((np.abs(got - want) - (2e-2 * np.abs(want) + 4e-3 * np.abs(want).max())).max()), float(np.abs(got - want).max() / np.abs(want).max())

    gather_flops = 8.0 * G * N * H * W * S * F
    dense_flops = 2.0 * (2 * R + 2) ** 2 * N * H * W * S * F
    out = dict(workload="%s: N=%d C=%d->%d HW=%d G=%d k=%d, bf16 activations" % (args.workload, N, S, F, H, G, 2 * R + 1),
               gather_sum_fp32_mfma=dict(ms_call=round(t_gather, 3), ms_kernel=round(t_gather_kernel, 3),
                                         tflops_algorithmic=round(gather_flops / (t_gather_kernel * 1e-3) / 1e12, 1),
                                         parity_violation=viol(y_gather)[0], err_rel_to_max=viol(y_gather)[1]),
               densified_bf16_conv=dict(
                   ms_conv={k_: (round(v[0], 3) if v[0] is not None else v[1]) for k_, v in res.items()},
                   ms_prepare_densify_blur_pad=round(t_prep, 3),
                   tflops_dense=(round(dense_flops / (best[0] * 1e-3) / 1e12, 1) if best[0] else None),
                   flop_inflation=round(dense_flops / gather_flops, 2),
                   parity_violation=(viol(best[1])[0] if best[0] else None),
                   err_rel_to_max=(viol(best[1])[1] if best[0] else None),
                   engine="torch.nn.functional.conv2d (MIOpen), bf16 in / bf16 out, cudnn.benchmark=True"),
               note="forward pass only; the densified form needs the dense kernel rebuilt every step (weights and offsets "
                    "train) and a (2R+2)^2 dense weight-gradient convolution per gradient kind in the backward pass")
    print(json.dumps(out))


if __name__ == "__main__":
    main()
