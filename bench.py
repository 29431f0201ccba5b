#!/usr/bin/env python3
"""Benchmark of the DAU forward+backward hot path (BASELINE.json metric: GSamples/s = N*H*W / (t_fwd+t_bwd)).

    python bench.py --gpus N --steps K --warmup W

For N > 1 this starts N ranks itself (one per GPU, `python -m torch.distributed.run ... bench.py`), relays rank 0's one
JSON line and exits with the ranks' return code; started under torch.distributed.run (RANK set) it is one of those ranks.

One "step" = dau_conv_forward + dau_conv_backward (dx, dw, dmu1, dmu2, dsigma) over one synthetic batch that is
already resident in HBM.  With more than one rank the batch is sharded (weak scaling: the per-GPU batch is fixed), the
raw parameter-gradient sums [4,S,G,F] are all-reduced over RCCL/xGMI under the dx pass and finalized after the
exchange.  Rank 0 prints ONE JSON line.

Extra objects in the line:
  roofline      dominant kernel's ALGORITHMIC FLOPs per pass / its average time per pass, measured live with HIP
                events on the launch stream (dau_conv_profile_begin/_end; a pass over large offsets takes several
                window launches, whose times add).  The operator is compute bound (1640 FLOP/B at the north-star shape,
                SURVEY.md 8d), so the roof is the fp32 matrix/vector peak of 157.3 TFLOP/s.
  cpu_baseline  the CPU oracle (oracle/dau_oracle.c, OpenMP) timed on this box's host cores on a small slice of the
                same workload.  A reported baseline, not a target.
  layer         the same step through the drop-in layer (dau_conv.DAUConv2d + autograd, default check_offsets), so that
                the binding's host work is visible next to the ABI-level number.
"""
import argparse
import hashlib
import json
import math
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "dau-convnet_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

WORKLOADS = {
    # north star of BASELINE.json: N=128 C=256->256 HW=56, 4-unit DAU, fp32, max_kernel_size 9, mu ~ U(-3,3)
    "ns": dict(N=128, S=256, F=256, H=56, W=56, G=4, k=9, m=3.0,
               label="north-star N=128/GPU C=256->256 HW=56 G=4(2x2) k=9 mu~U(-3,3) sigma=0.5 fp32 fwd+bwd(dx,dw,dmu1,dmu2,dsigma)"),
    # AlexNet-DAU conv2 shape (configs[1])
    "c1": dict(N=64, S=96, F=256, H=27, W=27, G=4, k=9, m=3.0,
               label="AlexNet-DAU conv2 N=64/GPU C=96->256 HW=27 G=4 k=9 fp32 fwd+bwd"),
    # configs[2]: six units per channel; BASELINE names bf16 activations: run with --io bf16
    "c2": dict(N=128, S=256, F=256, H=56, W=56, G=6, k=9, m=3.0,
               label="ResNet-50-DAU layer N=128/GPU C=256->256 HW=56 G=6 k=9 fwd+bwd"),
    # configs[3]: per-GPU share of the N=1024 batch-sharded step
    "c3": dict(N=128, S=512, F=512, H=28, W=28, G=4, k=9, m=3.0,
               label="batch-sharded step N=128/GPU C=512->512 HW=28 G=4 k=9 fp32 fwd+bwd"),
    # configs[4]: segmentation-scale maps, nine live units (ten stored, one ignored, as the layer pads
    # odd unit counts), offsets up to +-17 under max_kernel_size 65
    "c4": dict(N=16, S=256, F=256, H=512, W=512, G=10, ignore=1, k=65, m=17.0,
               label="seg-scale N=16/GPU C=256->256 HW=512 G=9(+1 ignored) k=65 mu~U(-17,17) fp32 fwd+bwd"),
    # the same maps with offsets within +-16 (max_kernel_size 33)
    "c4k33": dict(N=16, S=256, F=256, H=512, W=512, G=10, ignore=1, k=33, m=15.0,
               label="seg-scale N=16/GPU C=256->256 HW=512 G=9(+1 ignored) k=33 mu~U(-15,15) fp32 fwd+bwd"),
    # the reference's "big kernel, small offsets" case at the north-star size (dau_conv_test.py:433,436)
    "nsk65": dict(N=128, S=256, F=256, H=56, W=56, G=4, k=65, m=3.0,
               label="north-star shape under max_kernel_size 65 with mu~U(-3,3) (per-call bucket selection)"),
    "small": dict(N=8, S=32, F=32, H=56, W=56, G=4, k=9, m=3.0, label="smoke-size N=8 C=32->32 HW=56 G=4"),
}
FP32_PEAK_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md: fp32 vector == fp32 MFMA peak
BF16_PEAK_TFLOPS = 2500.0  # dense bf16 MFMA peak (same guide; the 2:1-sparsity headline figure is not used)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="ns", choices=sorted(WORKLOADS))
    ap.add_argument("--shape", default=None, help="ad-hoc workload N,S,F,H,W,G,k[,m] (overrides --workload)")
    ap.add_argument("--io", default="f32", choices=["f32", "bf16"],
                    help="storage type of x, y, dy, dx (BASELINE config 2 names bf16); arithmetic is fp32 either way")
    ap.add_argument("--dense", action="store_true",
                    help="with --io bf16: DAU_FLAG_DENSE_BF16 (gather-sum passes of calls with |mu| <= 4 as a densified bf16 MFMA GEMM)")
    ap.add_argument("--split", action="store_true",
                    help="DAU_FLAG_DENSE_SPLIT_F16: the two-limb f16 dense gather-sum members of all three radii whatever the unit "
                         "count (default: the library's choice -- the radii that pay; fp32 accuracy, the parity gate keeps the fp32 bar)")
    ap.add_argument("--no-split", action="store_true",
                    help="DAU_FLAG_NO_DENSE_SPLIT: the exact fp32 gather (v_mfma_f32_4x4x1) for every call")
    ap.add_argument("--mu-range", type=float, default=None, metavar="M",
                    help="draw the offsets from U(-M, M) instead of the workload's own range (BASELINE: 3): which gather-sum member a "
                         "call takes depends on its largest offset (dense radius 2 / 3 / 4, else the exact gather)")
    ap.add_argument("--no-dsigma", action="store_true",
                    help="the step does not ask for dsigma (need mask of a layer whose sigma is not trained: the reference's default, "
                         "dau_sigma_trainable=False); a side line, never the headline: BASELINE's step has all five gradients")
    ap.add_argument("--graph", action="store_true",
                    help="capture one step into a HIP graph and time its replays (single rank; no per-kernel events, so the "
                         "line carries no roofline object: a launch-overhead probe for the small workloads)")
    ap.add_argument("--algo", type=int, default=0, help="0 auto, 1 direct, 2 tiled")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-layer", action="store_true", help="skip the layer-level (DAUConv2d + autograd) timing")
    ap.add_argument("--offsets", default="uniform", metavar="uniform|grid[:JITTER]",
                    help="offset distribution: uniform = mu ~ U(-m, m) per unit (the reference tests' and BASELINE's synthetic "
                         "data; the worst case for the binned large-offset kernels), grid = the units of every channel pair on "
                         "the regular grid the reference layer initialises them on (DAUGridMean) + U(-JITTER, JITTER), default 1")
    ap.add_argument("--check", type=int, default=None, metavar="IMAGES",
                    help="parity gate before timing: y and dx of the first IMAGES images against the CPU oracle (default 1; "
                         "--no-check: none)")
    ap.add_argument("--check-params", type=int, default=None, metavar="CHANNELS",
                    help="with --check: also dw, dmu1, dmu2, dsigma of the first CHANNELS output channels (sums over the WHOLE "
                         "batch: the oracle runs all N images on that slice of the output channels; default 4)")
    ap.add_argument("--no-check", action="store_true", help="no parity gate in front of the timing (the line then says parity_gate: null)")
    ap.add_argument("--steady-seconds", type=float, default=8.0,
                    help="after the timed steps, keep stepping for about this long (at most 200 steps) and report the mean step "
                         "time of that run as roofline.steady_state_ms: the kernels are power limited and the headline's few "
                         "steps are over before the chip has warmed up (0: skip)")
    args = ap.parse_args()
    # the gate is ON by default (about 1.5 s of oracle time at the north-star size, before the warm-up): the driver's fixed
    # command line must carry a parity verdict too
    if args.no_check:
        args.check, args.check_params = 0, 0
    else:
        args.check = 1 if args.check is None else args.check
        args.check_params = (4 if args.check > 0 else 0) if args.check_params is None else args.check_params
    return args


def launch_ranks(args):
    """--gpus N > 1 outside torch.distributed.run: start the N ranks (no GPU call has happened in this process), relay
    rank 0's JSON line, return the ranks' exit code."""
    import torch  # device_count() does not initialise the GPU
    backend = os.environ.get("DAU_BENCH_BACKEND", "nccl")
    have = torch.cuda.device_count()
    if backend == "nccl" and have < args.gpus:
        sys.stderr.write("bench.py: --gpus %d but only %d GPU(s) visible (one rank per GPU over RCCL); "
                         "DAU_BENCH_BACKEND=gloo rehearses the multi-rank path on fewer GPUs\n" % (args.gpus, have))
        return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True)
    line = None
    for out in proc.stdout:
        out = out.strip()
        if out.startswith("{") and '"metric"' in out:
            line = out
        elif out:
            sys.stderr.write(out + "\n")
    rc = proc.wait()
    if rc == 0 and line is None:
        sys.stderr.write("bench.py: the ranks finished without printing a result line\n")
        rc = 3
    if line is not None and rc == 0:
        print(line)
    return rc


def source_fingerprint():
    """sha-256 prefix over the kernel sources, computed as the Makefile does for the library's build id (file names and
    contents of csrc/*.hip, *.hpp, Makefile in sorted order, then include/dau_conv.h)."""
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "dau-convnet_amd", "csrc")
    for name in sorted(f for f in os.listdir(csrc) if f.endswith((".hip", ".hpp")) or f == "Makefile"):
        h.update(name.encode())
        with open(os.path.join(csrc, name), "rb") as f:
            h.update(f.read())
    with open(os.path.join(ROOT, "include", "dau_conv.h"), "rb") as f:
        h.update(f.read())
    return h.hexdigest()[:16]


def lib_fingerprint():
    """Build id of the LOADED library (dau_conv_build_id: the source fingerprint the Makefile compiled in).  Not a hash of
    the binary: hipcc derives symbol names from the source path, so the same sources built in another directory (the
    GPU box's copy, a fresh clone) give another binary."""
    from dau_conv import _capi
    return _capi.build_id()


def measured_traffic(workload_key, io, dominant, split_r=0):
    """HBM bytes per pass of the dominant kernel from the committed rocprofv3 --pmc passes of THIS command
    (tools/pmc_traffic.sh writes profiles/r4_pmc_traffic.json: counters cannot be read from inside the process).  Only
    used when that file was taken for this workload with this very build of the library; otherwise null + the reason."""
    path = os.path.join(ROOT, "profiles", "r4_pmc_traffic.json")
    try:
        pmc = json.load(open(path))
    except Exception:
        return None, "no PMC pass committed for this build (profiles/r4_pmc_traffic.json missing)"
    run = pmc.get("runs", {}).get("%s/%s" % (workload_key, io))
    if run is None:
        return None, "no PMC pass committed for workload %s/%s" % (workload_key, io)
    if run.get("src_sha256_16") != lib_fingerprint():
        return None, "committed PMC pass is from another build of the library (stale)"
    gather = "dau::s%d::split_gather_kernel" % split_r if split_r else "dau::gather_mfma_kernel"
    key = {"gather_dot": "dau::gather_dot_kernel", "gather_sum_fwd": gather, "gather_sum_dx": gather}.get(dominant, "")
    fam = run["kernels"].get(key)       # the family entry: all instantiations / window launches of one pass
    if not fam or "hbm_bytes_per_pass" not in fam:
        return None, "dominant kernel not in the committed PMC pass"
    return round(fam["hbm_bytes_per_pass"] / 1e9, 3), "profiles/r4_pmc_traffic.json (2*FETCH_SIZE + WRITE_SIZE, separate passes)"


def main():
    args = parse_args()
    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(launch_ranks(args))

    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE is %d (start it as `python bench.py --gpus N`, or under "
                         "torch.distributed.run with --nproc-per-node equal to --gpus)" % (args.gpus, world))
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

    wl = dict(WORKLOADS[args.workload])
    wl_key = args.workload
    if args.shape:
        v = args.shape.split(",")
        wl = dict(N=int(v[0]), S=int(v[1]), F=int(v[2]), H=int(v[3]), W=int(v[4]), G=int(v[5]), k=int(v[6]),
                  m=float(v[7]) if len(v) > 7 else 3.0, label="ad-hoc " + args.shape)
        wl_key = "adhoc:" + args.shape
    if args.mu_range is not None:
        wl["m"] = float(args.mu_range)
        wl["label"] += " [offsets ~ U(-%g, %g)]" % (wl["m"], wl["m"])
        wl_key += ":m%g" % wl["m"]
    N, S, F, H, W, G, k, m = (wl[q] for q in ("N", "S", "F", "H", "W", "G", "k", "m"))
    ignore = int(wl.get("ignore", 0))
    G_live = G - ignore
    gen = torch.Generator(device=dev)
    gen.manual_seed(1234 + rank)
    pgen = torch.Generator(device=dev)
    pgen.manual_seed(99)                       # parameters are replicated: same seed on every rank
    # distributions of the reference tests (dau_conv_test.py:342-368)
    x = torch.rand((N, S, H, W), device=dev, generator=gen)
    dy = torch.randn((N, F, H, W), device=dev, generator=gen)
    w = torch.randn((1, S, G, F), device=dev, generator=pgen) * 0.1
    if ignore:
        w[:, :, G - ignore:, :] = 0.0          # ZeroNLast, as the layer initialises its padded unit (dau_conv.py:317-329)
    lim = k // 2 - 0.01
    mu1 = ((torch.rand((1, S, G, F), device=dev, generator=pgen) * 2 - 1) * m).clamp_(-lim, lim)
    mu2 = ((torch.rand((1, S, G, F), device=dev, generator=pgen) * 2 - 1) * m).clamp_(-lim, lim)
    if args.offsets.startswith("grid"):
        # the live units on a near-square grid spanning +-(m - 1), as DAUGridMean places them (dau_conv.py:24-62 of the
        # reference), every (input, output) channel pair jittered on its own
        from dau_conv import DAUGridMean
        jitter = float(args.offsets.split(":")[1]) if ":" in args.offsets else 1.0
        gx = int(math.ceil(math.sqrt(G_live))); gy = (G_live + gx - 1) // gx
        ax = torch.tensor(DAUGridMean((gy, gx), max(m - 1.0, 0.0)).legacy_values(gx), dtype=torch.float32, device=dev)
        ay = torch.tensor(DAUGridMean((gy, gx), max(m - 1.0, 0.0)).legacy_values(gy), dtype=torch.float32, device=dev)
        g = torch.arange(G, device=dev)
        live = (g < G_live).float()
        base1 = (ax[(g % gx).clamp(max=gx - 1)] * live).view(1, 1, G, 1)
        base2 = (ay[(g // gx).clamp(max=gy - 1)] * live).view(1, 1, G, 1)
        mu1 = (base1 + (torch.rand((1, S, G, F), device=dev, generator=pgen) * 2 - 1) * jitter).clamp_(-lim, lim)
        mu2 = (base2 + (torch.rand((1, S, G, F), device=dev, generator=pgen) * 2 - 1) * jitter).clamp_(-lim, lim)
    sigma = torch.full((1, S, G, F), 0.5, device=dev)

    if args.io == "bf16":
        x, dy = x.to(torch.bfloat16), dy.to(torch.bfloat16)
    plan = _capi.Plan(N, S, F, G, H, W, max_kernel_size=k, number_units_ignore=ignore, algo=args.algo,
                      flags=_capi.FLAG_USE_INTERPOLATION | (_capi.FLAG_IO_BF16 if args.io == "bf16" else 0) |
                            (_capi.FLAG_DENSE_BF16 if args.dense else 0) | (_capi.FLAG_DENSE_SPLIT_F16 if args.split else 0) |
                            (_capi.FLAG_NO_DENSE_SPLIT if args.no_split else 0),
                      sigma_hint=0.5, mu_learning_rate_factor=1.0)
    from dau_conv.distributed import OverlappedBackward
    exchange = OverlappedBackward((1, S, G, F), dev) if use_dist else None

    last_grads = [None]
    need_mask = _capi.NEED_ALL & ~_capi.NEED_DSIGMA if args.no_dsigma else _capi.NEED_ALL

    def step():
        y = plan.forward(x, w, mu1, mu2, sigma)
        if use_dist:
            # batch-sharded data parallelism: one all-reduce of the raw sums [4,S,G,F] over RCCL/xGMI, issued after the
            # gather-dot pass and hidden under the dx pass; the elementwise tail runs on the reduced sums
            dx = exchange.run(lambda out: plan.backward_param_sums(x, dy, mu1, mu2, sigma, out=out),
                              lambda: plan.backward(x, dy, w, mu1, mu2, sigma, need_mask=_capi.NEED_DX)[0],
                              lambda sums: plan.finalize_param_grads(sums, w))
            exchange.wait()
        else:
            last_grads[0] = plan.backward(x, dy, w, mu1, mu2, sigma, need_mask=need_mask)
            dx = last_grads[0][0]
        return y, dx

    def fence():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    parity = None
    if args.check > 0:
        # EVERY rank takes the gate's step (with more than one rank a step holds a collective: the ranks must agree on how many
        # they run); rank 0 compares its shard with the oracle while the others wait at the fence below
        y, dx = step()
        torch.cuda.synchronize()
    if args.check > 0 and rank == 0:
        from oracle import dau_oracle as orc
        nchk = min(args.check, N)
        xs, dys = x[:nchk].float().cpu().numpy(), dy[:nchk].float().cpu().numpy()
        wn, m1n, m2n = w.cpu().numpy(), mu1.cpu().numpy(), mu2.cpu().numpy()
        want_y = orc.forward(xs, wn, m1n, m2n, 0.5, ignore=ignore)
        want_dx = orc.backward(xs, dys, wn, m1n, m2n, 0.5, ignore=ignore, need=("dx",))["dx"]
        rel, floor = (2e-2, 4e-3) if args.io == "bf16" else (1e-4, 1e-6)

        def viol(got, want):
            got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
            return float((np.abs(got - want) - (rel * np.abs(want) + floor * np.abs(want).max())).max())
        vy, vdx = viol(y[:nchk].float().cpu().numpy(), want_y), viol(dx[:nchk].float().cpu().numpy(), want_dx)
        parity = dict(images=nchk, y_violation=vy, dx_violation=vdx, ok=bool(vy <= 0 and vdx <= 0), rel=rel, floor=floor)
        if args.check_params > 0 and not use_dist:
            # the parameter gradients are sums over the batch: the oracle takes all N images on a slice of the output channels
            fs = min(F, args.check_params)
            dense_params = bool(args.dense and int(plan.info.get("gather_dense_bf16", 0)) == 2)
            prel, pfloor = (2e-2, 4e-3) if dense_params else (1e-4, 1e-6)
            want = orc.backward(x.float().cpu().numpy(), dy[:, :fs].float().cpu().numpy(), wn[..., :fs].copy(), m1n[..., :fs].copy(),
                                m2n[..., :fs].copy(), 0.5, ignore=ignore, need=("dw", "dmu1", "dmu2", "dsigma"))
            pv = {}
            for i, key in enumerate(("dw", "dmu1", "dmu2", "dsigma")):
                got = np.asarray(last_grads[0][i + 1][..., :fs].cpu().numpy(), np.float64)
                wv = np.asarray(want[key], np.float64)
                pv[key] = float((np.abs(got - wv) - (prel * np.abs(wv) + pfloor * np.abs(wv).max())).max())
            parity.update(param_channels=fs, param_violation=pv, param_rel=prel, param_floor=pfloor)
            parity["ok"] = bool(parity["ok"] and all(v <= 0 for v in pv.values()))
        # the oracle's OpenMP team keeps spinning on the host cores for a moment after its last parallel region; the launches of
        # a millisecond-sized step right behind it were seen to take 5x as long (C1 dense: 5.2 instead of 1.0 ms per step)
        time.sleep(1.0)
    if args.check > 0:
        ok = 1 if (parity is None or parity["ok"]) else 0
        if use_dist:           # every rank learns rank 0's verdict (a rank that left alone would hang the others' next collective)
            t = torch.tensor([ok], device=dev, dtype=torch.int32)
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            ok = int(t.item())
        if not ok:
            raise SystemExit("bench.py --check: the HIP path differs from the oracle: %s" % (parity,))

    for _ in range(args.warmup):
        step()
    plan.check_status()
    fence()
    if args.graph and not use_dist:
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):
            step()                                  # workspace and hint on the capture stream
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            step()
        for _ in range(2):
            graph.replay()
        fence()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            graph.replay()
        fence()
        elapsed = time.perf_counter() - t0
        prof = {}
    else:
        plan.profile_begin()
        if exchange is not None:
            exchange.measure_exposed(True)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        fence()
        elapsed = time.perf_counter() - t0
        prof = plan.profile_end()
        if exchange is not None:
            exposed_total, exposed_joins = exchange.exposed_ms()
            exchange.measure_exposed(False)
    exposed_ms = None
    if use_dist:
        t = torch.tensor([elapsed, (exposed_total / max(exposed_joins, 1)) if not args.graph else 0.0], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t[0].item())
        exposed_ms = float(t[1].item())
    # thermal steady state: NOT part of the headline (value / ms_per_step come from exactly --steps steps above); the same
    # step repeated for several seconds, timed as a whole
    steady = None
    if args.steady_seconds > 0 and not args.graph:
        per = elapsed / args.steps
        nsteady = int(max(4, min(200, math.ceil(args.steady_seconds / per))))
        fence()
        s0 = time.perf_counter()
        for _ in range(nsteady):
            step()
        fence()
        sel = time.perf_counter() - s0
        if use_dist:
            t = torch.tensor([sel], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            sel = float(t.item())
        steady = dict(ms=round(sel / nsteady * 1e3, 3), steps=nsteady)

    samples = float(N) * world * H * W * args.steps
    value = samples / elapsed / 1e9
    # algorithmic FLOPs per pass: 4 MAC per (n,px,s,g,f) for each gather-sum, 8 MAC for gather-dot,
    # over the live units
    unit_px = float(G_live) * N * H * W * S * F
    flops = {"gather_sum_fwd": 8.0 * unit_px, "gather_sum_dx": 8.0 * unit_px, "gather_dot": 16.0 * unit_px}
    dense = bool(args.dense and plan.info.get("gather_dense_bf16"))
    # the two-limb f16 dense member a call with these offsets takes for its gather-sum passes (0: the exact gather)
    split_bits = int(plan.info.get("gather_dense_split", 0))
    mu_max = float(max(mu1.abs().max().item(), mu2.abs().max().item()))
    split_r = next((r for r in (2, 3, 4) if (split_bits >> r) & 1 and mu_max <= r), 0)
    kern = {}
    for name, (ms, passes) in prof.items():
        if passes:
            avg = ms / passes
            kern[name] = dict(avg_ms=round(avg, 4), passes=passes, tflops=round(flops[name] / (avg * 1e-3) / 1e12, 2))
            if split_r and name != "gather_dot":
                taps = (2 * split_r + 1) ** 2
                ex = 2.0 * taps * 3 * N * H * W * S * F / (avg * 1e-3) / 1e12
                kern[name].update(form="two-limb f16 dense GEMM, radius %d (%d taps x 3 limb products)" % (split_r, taps),
                                  executed_tflops=round(ex, 1), executed_frac_of_f16_peak=round(ex / BF16_PEAK_TFLOPS, 4),
                                  note="tflops = ALGORITHMIC gather FLOPs / time (may exceed the fp32 roof: the pass runs on the f16 matrix cores)")
    dominant = max(kern, key=lambda n: kern[n]["avg_ms"]) if kern else None
    roofline = None
    if dominant:
        traffic, traffic_note = measured_traffic(wl_key, args.io, dominant, split_r)
        ach = flops[dominant] / (kern[dominant]["avg_ms"] * 1e-3) / 1e12
        peak, roof_note = FP32_PEAK_TFLOPS, None
        dense_level = int(plan.info.get("gather_dense_bf16", 0)) if dense else 0
        if (dense_level >= 1 and dominant != "gather_dot") or (dense_level == 2 and dominant == "gather_dot"):
            # the pass ran in its densified form on the bf16 matrix cores: price the FLOPs that form executes (9 x 9 = 81 taps per
            # (input, output) channel pair and pixel; 324 for the four parameter-gradient kinds) against the bf16 roof
            # (offsets within +-3, as every BASELINE workload draws them, take the radius-3 members: 7 x 7 taps / displacements)
            r3_level = int(plan.info.get("dense_bf16_radius3", 0))      # which passes have a 7 x 7 member in THIS plan
            has_r3 = r3_level >= (2 if dominant == "gather_dot" else 1)
            side = 7.0 if (mu_max <= 3.0 and has_r3) else 9.0
            taps = side * side * ((3.0 if args.no_dsigma else 4.0) if dominant == "gather_dot" else 1.0)
            ach = 2.0 * taps * N * H * W * S * F / (kern[dominant]["avg_ms"] * 1e-3) / 1e12
            peak = BF16_PEAK_TFLOPS
            roof_note = ("densified bf16 form: achieved = executed dense FLOPs (2*%d*N*H*W*S*F) / time against the dense bf16 "
                         "MFMA peak; kernels[*].tflops stay algorithmic (gather form)" % int(taps))
        if split_r and dominant != "gather_dot":
            ach = kern[dominant]["executed_tflops"]
            peak = BF16_PEAK_TFLOPS
            roof_note = ("two-limb f16 dense form: achieved = executed dense FLOPs / time against the dense f16 MFMA peak; "
                         "kernels[*].tflops stay algorithmic (gather form)")
        roofline = dict(bound="mfma", kernel=dominant, achieved=round(ach, 2), peak=peak, unit="TFLOP/s",
                        frac=round(ach / peak, 4), traffic=traffic, traffic_unit="GB per pass",
                        traffic_note=traffic_note, kernels=kern,
                        whole_step_tflops=round(32.0 * unit_px * args.steps / elapsed / 1e12, 2))
        if roof_note:
            roofline["note"] = roof_note
        if steady:
            roofline["steady_state_ms"] = steady["ms"]
            roofline["steady_state_steps"] = steady["steps"]
            roofline["steady_state_note"] = ("mean step time of %d further steps run right after the timed region (not part of "
                                             "value / ms_per_step): the kernels are power limited, this is the warmed-up chip" % steady["steps"])
        # the BASELINE metric also asks for the HBM view: compulsory bytes of one fwd+bwd step (SURVEY.md 8d:
        # e*N*H*W*(3S+2F) + 7*4*S*G*F) over the step time, against the 8 TB/s roof -- ~1 %, the operator is compute bound
        e = 2 if args.io == "bf16" else 4
        step_bytes = float(e) * N * H * W * (3 * S + 2 * F) + 28.0 * S * G * F
        gbps = step_bytes * args.steps * world / elapsed / 1e9
        roofline["hbm_algorithmic"] = dict(bytes_per_step=step_bytes, achieved_GBps=round(gbps / world, 1), peak_GBps=8000.0,
                                           frac=round(gbps / world / 8000.0, 4))

    # the same step through the drop-in layer: DAUConv2d + autograd with its default offset check ("async": the previous
    # call's status is read from pinned host memory, no sync) -- what a model built on the Python surface pays
    layer = None
    if rank == 0 and world == 1 and not args.no_layer and ignore == 0 and not args.dense:
        import dau_conv
        torch.cuda.empty_cache()
        lay = dau_conv.DAUConv2d(filters=F, dau_units=(1, G), max_kernel_size=k, use_bias=False, in_channels=S,
                                 mu_learning_rate_factor=1.0, dau_sigma_trainable=True).to(dev)
        with torch.no_grad():
            lay.weights.copy_(w); lay.mu1.copy_(mu1); lay.mu2.copy_(mu2)
        xl = x.detach().clone().requires_grad_(True)

        def layer_step():
            yl = lay(xl)
            yl.backward(dy)
            xl.grad = None
            for p_ in lay.parameters():
                p_.grad = None

        lsteps = max(2, min(args.steps, 10))
        for _ in range(2):
            layer_step()
        torch.cuda.synchronize()
        l0 = time.perf_counter()
        for _ in range(lsteps):
            layer_step()
        torch.cuda.synchronize()
        lms = (time.perf_counter() - l0) / lsteps * 1e3
        layer = dict(ms_per_step=round(lms, 3), steps=lsteps, check_offsets="async",
                     note="dau_conv.DAUConv2d forward + autograd backward (all five gradients, sigma trainable)")
        del lay, xl

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import dau_oracle as orc
        # bounded sample (about 8 s of CPU work at the ~6e10 FLOP/s the oracle reaches on the box's cores, so that the run
        # stays GPU-dominated): whole images of the same batch; where one image is already too much (512 x 512 maps), the
        # first output channels only -- the work is linear in the output channels, and the rate is scaled back to all F
        budget = 0.45e12
        per_image = 32.0 * G_live * H * W * S * F
        ncpu = int(max(1, min(N, 32, budget // per_image)))
        fcpu = F if per_image <= budget else int(max(8, min(F, budget // (per_image / F))))
        xs, dys = x[:ncpu].float().cpu().numpy(), dy[:ncpu, :fcpu].float().cpu().numpy()
        wn, m1, m2 = (t[..., :fcpu].contiguous().cpu().numpy() for t in (w, mu1, mu2))
        orc.forward(xs[:1, :4], wn[:, :4], m1[:, :4], m2[:, :4], 0.5)   # load + warm the library
        c0 = time.perf_counter()
        orc.forward(xs, wn, m1, m2, 0.5, ignore=ignore)
        orc.backward(xs, dys, wn, m1, m2, 0.5, ignore=ignore, unit_testing=False, mu_learning_rate_factor=1.0)
        ct = time.perf_counter() - c0
        cpu = dict(value=round(ncpu * H * W * (float(fcpu) / F) / ct / 1e9, 8), unit="GSamples/s", cores=orc.num_threads(),
                   kind="port",
                   sample="oracle fwd+bwd on the first %d image(s) of the same batch%s (%.1f s); double accumulation, OpenMP"
                          % (ncpu, "" if fcpu == F else ", output channels 0..%d of %d, rate scaled by %d/%d" % (fcpu - 1, F, fcpu, F), ct))

    if rank == 0:
        out = dict(metric="DAU fwd+bwd GSamples/s (N*H*W/s)", value=round(value, 6), unit="GSamples/s", n_gpus=world,
                   steps=args.steps, warmup=args.warmup, ms_per_step=round(elapsed / args.steps * 1e3, 3),
                   higher_is_better=True, scaling="weak", vs_baseline=None,
                   dtype=args.io, accumulate="f32", data="synthetic",
                   arithmetic=dict(gather_sum=("bf16 products (opt-in dense form), f32 sums" if dense else
                                               "two binary16 limbs per operand (22 significant bits; hi*hi + lo*hi + hi*lo on the f16 matrix cores), f32 sums: "
                                               "fp32 accuracy, gated at the fp32 bar" if split_r else "f32 (v_mfma_f32_4x4x1)"),
                                   gather_dot=("bf16 products (opt-in dense form), f32 sums" if dense and int(plan.info.get("gather_dense_bf16", 0)) == 2
                                               else "f32 (packed FMA + v_mfma_f32_4x4x1), partial sums float/double")),
                   ranks_seen=(dist.get_world_size() if use_dist else 1),
                   comm=(dict(backend=("rccl" if backend == "nccl" else backend), communicator_size=dist.get_world_size(),
                              exchange="all_reduce(sum) of raw param-grad sums [4,S,G,F] = %d floats per step, async under the dx pass; finalize after"
                              % (4 * S * G * F), bytes_per_step=16 * S * G * F,
                              exposed_ms=(None if exposed_ms is None else round(exposed_ms, 4)),
                              exposed_note="mean time per step the compute stream stands still at the join with the all-reduce (HIP events "
                                           "around the wait, after the dx pass was enqueued; max over ranks): what of the exchange the dx pass does not hide") if use_dist else None),
                   config=dict(workload=wl["label"] + (" [bf16 activations in HBM]" if args.io == "bf16" else "") +
                               (" [offsets: %s instead of U(-m,m)]" % args.offsets if args.offsets != "uniform" else "") +
                               (" [gather-sum passes%s as densified bf16 MFMA GEMM]" % (" and parameter gradients" if dense and int(plan.info.get("gather_dense_bf16", 0)) == 2 else "") if dense else "") +
                               (" [gather-sum passes: two-limb f16 dense GEMM of radius %d, fp32 accuracy; max|mu| = %.2f]" % (split_r, mu_max) if split_r else
                                ("" if dense else " [gather-sum passes: exact fp32 gather; max|mu| = %.2f]" % mu_max)) +
                               (" [SIDE LINE: the step does not ask for dsigma (dx, dw, dmu1, dmu2 only)]" if args.no_dsigma else "") +
                               (" [one step captured into a HIP graph, replays timed]" if args.graph and not use_dist else "") +
                               ("" if backend == "nccl" else " [REHEARSAL: %s backend, ranks share GPUs]" % backend),
                               global_batch=N * world, parallelism="dp%d" % world,
                               algo_forward=plan.info["algo_forward"], algo_backward=plan.info["algo_backward"],
                               static_offset_bucket=plan.info["offset_bucket"], bucket_sets=plan.info["bucket_sets"],
                               gather_dense_split_radii=[r for r in (2, 3, 4) if (split_bits >> r) & 1]),
                   roofline=roofline, cpu_baseline=cpu, layer=layer, parity_gate=parity, lib=lib_fingerprint())
        print(json.dumps(out))
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
