#!/bin/bash
# A/B for SURVEY.md 8(f) rank 1 (fused prefilter): the shipped gather-sum (planes staged by blur_pack_kernel, filled by
# DMA) against a diagnostic build in which every wave runs the irreducible instruction mix of an in-kernel prefilter fill
# instead of the plane DMA (-DDAU_DIAG_FUSED_BLUR, a LOWER bound of a fused kernel; its results are garbage).  Same
# device, interleaved rounds.  Run on the GPU box from the repo root; prints per-pass gather-sum times (forward, dx) and
# the step time; the staging kernels the fused form would save take 2 x ~0.37 ms per step at this shape.
set -e
cd "$GRAFT_REPO_ROOT/dau-convnet_amd/csrc"
make -s -j8 >/dev/null 2>&1
mkdir -p ../../build/diag_FUSED
/opt/rocm/bin/hipcc -O3 -std=c++20 -fPIC --offload-arch=gfx950 -I../../include -I. -fvisibility=hidden -DDAU_DIAG_FUSED_BLUR -c k_gather_mfma.hip -o /tmp/k_gm_fused.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../build/diag_FUSED/libdau_conv_hip.so dau_conv_api.o k_filters.o k_units.o k_direct.o /tmp/k_gm_fused.o k_gather_dot.o k_dense_bf16.o k_dense_wgrad.o
cd "$GRAFT_REPO_ROOT"
for r in 1 2; do
  for L in dau-convnet_amd/dau_conv build/diag_FUSED; do
    DAU_CONV_LIB=$GRAFT_REPO_ROOT/$L/libdau_conv_hip.so timeout -k 10 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-layer 2>/dev/null | \
      python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['roofline']['kernels']; print('$L', 'fwd_ms', k['gather_sum_fwd']['avg_ms'], 'dx_ms', k['gather_sum_dx']['avg_ms'], 'step_ms', d['ms_per_step'])"
  done
done
