#!/usr/bin/env python3
"""Benchmark of the DAU forward+backward hot path (BASELINE.json metric: GSamples/s = N*H*W / (t_fwd+t_bwd)).

    python bench.py --gpus N --steps K --warmup W          (N>1: launched by torch.distributed.run, one rank per GPU)

One "step" = dau_conv_forward + dau_conv_backward (dx, dw, dmu1, dmu2, dsigma) over one synthetic batch that is
already resident in HBM, plus -- for N>1 -- the RCCL all-reduce of the four parameter-gradient tensors.  The batch
is sharded over ranks (weak scaling: 128 images per GPU).  Rank 0 prints ONE JSON line.

Extra objects in the line:
  roofline      dominant kernel's ALGORITHMIC FLOPs per launch / its average launch time, measured live with HIP
                events on the launch stream (dau_conv_profile_begin/_end).  The operator is compute bound (1640 FLOP/B
                at this shape), so the roof is the fp32 matrix/vector peak of 157.3 TFLOP/s.
  cpu_baseline  the CPU oracle (oracle/dau_oracle.c, OpenMP) timed on this box's host cores on a small slice of the
                same workload.  A reported baseline, not a target.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "dau-convnet_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np
import torch
import torch.distributed as dist

WORKLOADS = {
    # north star of BASELINE.json: N=128 C=256->256 HW=56, 4-unit DAU, fp32, max_kernel_size 9, mu ~ U(-3,3)
    "ns": dict(N=128, S=256, F=256, H=56, W=56, G=4, k=9, m=3.0,
               label="north-star N=128/GPU C=256->256 HW=56 G=4(2x2) k=9 mu~U(-3,3) sigma=0.5 fp32 fwd+bwd(dx,dw,dmu1,dmu2,dsigma)"),
    # AlexNet-DAU conv2 shape (configs[1])
    "c1": dict(N=64, S=96, F=256, H=27, W=27, G=4, k=9, m=3.0,
               label="AlexNet-DAU conv2 N=64/GPU C=96->256 HW=27 G=4 k=9 fp32 fwd+bwd"),
    # configs[2] at fp32 (bf16 is not implemented): six units per channel
    "c2": dict(N=128, S=256, F=256, H=56, W=56, G=6, k=9, m=3.0,
               label="ResNet-50-DAU layer N=128/GPU C=256->256 HW=56 G=6 k=9 fp32 fwd+bwd"),
    # configs[3]: per-GPU share of the N=1024 batch-sharded step
    "c3": dict(N=128, S=512, F=512, H=28, W=28, G=4, k=9, m=3.0,
               label="batch-sharded step N=128/GPU C=512->512 HW=28 G=4 k=9 fp32 fwd+bwd"),
    # configs[4]: segmentation-scale maps, nine units, offsets up to +-17 => offset bucket 32 (max_kernel_size 65)
    "c4": dict(N=16, S=256, F=256, H=512, W=512, G=9, k=65, m=17.0,
               label="seg-scale N=16/GPU C=256->256 HW=512 G=9 k=65 mu~U(-17,17) fp32 fwd+bwd"),
    # the same maps with offsets within +-16 (bucket 16, max_kernel_size 33)
    "c4k33": dict(N=16, S=256, F=256, H=512, W=512, G=9, k=33, m=15.0,
               label="seg-scale N=16/GPU C=256->256 HW=512 G=9 k=33 mu~U(-15,15) fp32 fwd+bwd"),
    "small": dict(N=8, S=32, F=32, H=56, W=56, G=4, k=9, m=3.0, label="smoke-size N=8 C=32->32 HW=56 G=4"),
}
FP32_PEAK_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md: fp32 vector == fp32 MFMA peak


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="ns", choices=sorted(WORKLOADS))
    ap.add_argument("--shape", default=None, help="ad-hoc workload N,S,F,H,W,G,k[,m] (overrides --workload)")
    ap.add_argument("--io", default="f32", choices=["f32", "bf16"],
                    help="storage type of x, y, dy, dx (BASELINE config 2 names bf16); arithmetic is fp32 either way")
    ap.add_argument("--algo", type=int, default=0, help="0 auto, 1 direct, 2 tiled")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (there is no CPU fallback for the operator)")
    # DAU_BENCH_BACKEND=gloo: rehearsal of the multi-rank path on a box with fewer GPUs than ranks (ranks share devices,
    # the all-reduce goes through gloo); the measured configuration is always nccl = RCCL, one rank per GPU
    backend = os.environ.get("DAU_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank %= max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # under torch.distributed.run (RANK set) the exchange path is used even for one rank, so that a 1-GPU box can
    # rehearse exactly what N>1 runs: RCCL init, split backward, async all-reduce
    use_dist = world > 1 or "RANK" in os.environ
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        # RCCL prints a version banner on stdout when the communicator is created; stdout is reserved for the one
        # JSON line, so the banner goes to stderr
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
        try:
            if backend == "nccl":
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
            else:
                dist.init_process_group(backend, rank=rank, world_size=world)
            dist.barrier()
            torch.cuda.synchronize()
        finally:
            os.dup2(saved_stdout, 1)
            os.close(saved_stdout)

    from dau_conv import _capi

    wl = WORKLOADS[args.workload]
    if args.shape:
        v = args.shape.split(",")
        wl = dict(N=int(v[0]), S=int(v[1]), F=int(v[2]), H=int(v[3]), W=int(v[4]), G=int(v[5]), k=int(v[6]),
                  m=float(v[7]) if len(v) > 7 else 3.0, label="ad-hoc " + args.shape)
    N, S, F, H, W, G, k, m = (wl[q] for q in ("N", "S", "F", "H", "W", "G", "k", "m"))
    gen = torch.Generator(device=dev)
    gen.manual_seed(1234 + rank)
    pgen = torch.Generator(device=dev)
    pgen.manual_seed(99)                       # parameters are replicated: same seed on every rank
    # distributions of the reference tests (dau_conv_test.py:342-368)
    x = torch.rand((N, S, H, W), device=dev, generator=gen)
    dy = torch.randn((N, F, H, W), device=dev, generator=gen)
    w = torch.randn((1, S, G, F), device=dev, generator=pgen) * 0.1
    lim = k // 2 - 0.01
    mu1 = ((torch.rand((1, S, G, F), device=dev, generator=pgen) * 2 - 1) * m).clamp_(-lim, lim)
    mu2 = ((torch.rand((1, S, G, F), device=dev, generator=pgen) * 2 - 1) * m).clamp_(-lim, lim)
    sigma = torch.full((1, S, G, F), 0.5, device=dev)

    if args.io == "bf16":
        x, dy = x.to(torch.bfloat16), dy.to(torch.bfloat16)
    plan = _capi.Plan(N, S, F, G, H, W, max_kernel_size=k, algo=args.algo,
                      flags=_capi.FLAG_USE_INTERPOLATION | (_capi.FLAG_IO_BF16 if args.io == "bf16" else 0),
                      sigma_hint=0.5, mu_learning_rate_factor=1.0)
    from dau_conv.distributed import OverlappedBackward
    exchange = OverlappedBackward((1, S, G, F), dev) if use_dist else None

    def step():
        y = plan.forward(x, w, mu1, mu2, sigma)
        if use_dist:
            # batch-sharded data parallelism: one all-reduce of [dw|dmu1|dmu2|dsigma] over RCCL/xGMI, issued after the
            # gather-dot pass and hidden under the dx pass
            dx = exchange.run(lambda need: plan.backward(x, dy, w, mu1, mu2, sigma, need_mask=need))
            exchange.wait()
        else:
            dx = plan.backward(x, dy, w, mu1, mu2, sigma)[0]
        return y, dx

    def fence():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    plan.check_status()
    fence()
    plan.profile_begin()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    prof = plan.profile_end()
    if use_dist:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    samples = float(N) * world * H * W * args.steps
    value = samples / elapsed / 1e9
    # algorithmic FLOPs per launch: 4 MAC per (n,px,s,g,f) for each gather-sum, 8 MAC for gather-dot
    unit_px = float(G) * N * H * W * S * F
    flops = {"gather_sum_fwd": 8.0 * unit_px, "gather_sum_dx": 8.0 * unit_px, "gather_dot": 16.0 * unit_px}
    kern = {}
    for name, (ms, launches) in prof.items():
        if launches:
            avg = ms / launches
            kern[name] = dict(avg_ms=round(avg, 4), launches=launches, tflops=round(flops[name] / (avg * 1e-3) / 1e12, 2))
    dominant = max(kern, key=lambda n: kern[n]["avg_ms"]) if kern else None
    # HBM traffic of the dominant kernel: PMC counters cannot be read from inside this process, so the number comes
    # from the committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this same command (profiles/*_pmc_traffic.json)
    traffic = None
    try:
        pmc = json.load(open(os.path.join(ROOT, "profiles", "r1_pmc_traffic.json")))["kernels"]
        key = {"gather_dot": "dau::gather_dot_kernel", "gather_sum_fwd": "dau::gather_mfma_kernel",
               "gather_sum_dx": "dau::gather_mfma_kernel"}.get(dominant, "")
        hits = [v["hbm_bytes"] for k, v in pmc.items() if k.startswith(key)] if key and args.workload == "ns" and not args.shape else []
        traffic = round(hits[0] / 1e9, 3) if hits else None
    except Exception:
        traffic = None
    roofline = None
    if dominant:
        ach = flops[dominant] / (kern[dominant]["avg_ms"] * 1e-3) / 1e12
        roofline = dict(bound="mfma", kernel=dominant, achieved=round(ach, 2), peak=FP32_PEAK_TFLOPS, unit="TFLOP/s",
                        frac=round(ach / FP32_PEAK_TFLOPS, 4), traffic=traffic, traffic_unit="GB per launch (2*FETCH_SIZE+WRITE_SIZE)", kernels=kern,
                        whole_step_tflops=round(32.0 * unit_px * args.steps / elapsed / 1e12, 2))
        # the BASELINE metric also asks for the HBM view: compulsory bytes of one fwd+bwd step (SURVEY.md 8d:
        # e*N*H*W*(3S+2F) + 7*4*S*G*F) over the step time, against the 8 TB/s roof -- ~1 %, the operator is compute bound
        e = 2 if args.io == "bf16" else 4
        step_bytes = float(e) * N * H * W * (3 * S + 2 * F) + 28.0 * S * G * F
        gbps = step_bytes * args.steps * world / elapsed / 1e9
        roofline["hbm_algorithmic"] = dict(bytes_per_step=step_bytes, achieved_GBps=round(gbps / world, 1), peak_GBps=8000.0,
                                           frac=round(gbps / world / 8000.0, 4))

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import dau_oracle as orc
        ncpu = min(N, 32)
        xs, dys = x[:ncpu].cpu().numpy(), dy[:ncpu].cpu().numpy()
        wn, m1, m2 = w.cpu().numpy(), mu1.cpu().numpy(), mu2.cpu().numpy()
        orc.forward(xs[:1, :4], wn[:, :4], m1[:, :4], m2[:, :4], 0.5)   # load + warm the library
        c0 = time.perf_counter()
        orc.forward(xs, wn, m1, m2, 0.5)
        orc.backward(xs, dys, wn, m1, m2, 0.5, unit_testing=False, mu_learning_rate_factor=1.0)
        ct = time.perf_counter() - c0
        cpu = dict(value=round(ncpu * H * W / ct / 1e9, 8), unit="GSamples/s", cores=orc.num_threads(), kind="port",
                   sample="oracle fwd+bwd on the first %d images of the same batch (%.1f s); double accumulation, OpenMP" % (ncpu, ct))

    if rank == 0:
        out = dict(metric="DAU fwd+bwd GSamples/s (N*H*W/s)", value=round(value, 6), unit="GSamples/s", n_gpus=world,
                   steps=args.steps, warmup=args.warmup, ms_per_step=round(elapsed / args.steps * 1e3, 3),
                   higher_is_better=True, scaling="weak", vs_baseline=None, dtype="f32", data="synthetic",
                   config=dict(workload=wl["label"] + (" [bf16 activations in HBM]" if args.io == "bf16" else "") +
                               ("" if backend == "nccl" else " [REHEARSAL: %s backend, ranks share GPUs]" % backend),
                               global_batch=N * world, parallelism="dp%d" % world,
                               algo_forward=plan.info["algo_forward"], algo_backward=plan.info["algo_backward"]),
                   roofline=roofline, cpu_baseline=cpu)
        print(json.dumps(out))
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
